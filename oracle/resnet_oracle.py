"""Plain-PyTorch ResNet-18/34/50 trunk (He et al. 2016; v1.5 stride placement like torchvision / timm ``resnet50``).  TEST INFRASTRUCTURE ONLY.

Neither ``timm`` nor ``torchvision`` exists in the build image, so the reference's encoder classes (mode/models/perceptual_encoders/*.py) cannot
be constructed as they are.  ``oracle/gen_golden_encoders.py`` registers this trunk under the two names their constructors call
(``timm.create_model``, ``torchvision.models.resnet18``) and then runs the REFERENCE classes on top of it: what fixture F15 pins is the
reference's FiLM wiring / forward / autograd; the trunk's equality with timm's is by construction of the textbook architecture and the
``state_dict`` key / shape contract only (parity of the trunk itself: unpinned - LABNOTES.md section 8).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=False)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.act1(self.bn1(self.conv1(x)))
        return self.act2(self.bn2(self.conv2(out)) + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes); self.act1 = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(planes); self.act2 = nn.ReLU(inplace=False)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False); self.bn3 = nn.BatchNorm2d(planes * 4); self.act3 = nn.ReLU(inplace=False)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.act2(self.bn2(self.conv2(self.act1(self.bn1(self.conv1(x))))))
        return self.act3(self.bn3(self.conv3(out)) + idt)


class ResNet(nn.Module):
    def __init__(self, block, depths, with_fc=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False); self.bn1 = nn.BatchNorm2d(64)
        self.act1 = self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for j in range(n):
                stride = (1 if i == 0 else 2) if j == 0 else 1
                ds = None
                if stride != 1 or inplanes != planes * block.expansion:
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * block.expansion))
                blocks.append(block(inplanes, planes, stride, ds))
                inplanes = planes * block.expansion
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.num_features = inplanes
        self.global_pool = self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(inplanes, 1000) if with_fc else nn.Identity()


def create_model(name, pretrained=False, num_classes=0):
    block, depths = {"resnet18": (BasicBlock, (2, 2, 2, 2)), "resnet34": (BasicBlock, (3, 4, 6, 3)), "resnet50": (Bottleneck, (3, 4, 6, 3))}[name]
    return ResNet(block, depths, with_fc=False)


def resnet18(pretrained=False):
    return ResNet(BasicBlock, (2, 2, 2, 2), with_fc=True)


def fill_encoder_state_dict(sd, seed):
    """Seeded, well-conditioned values for an encoder ``state_dict`` (in key order): He-scaled convolutions, non-trivial BatchNorm affines and
    running statistics, NON-zero FiLM weights (the reference zero-initialises FiLMLayer, which would make the modulation invisible)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    out = {}
    for k, v in sd.items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shp, dtype=v.dtype)
            continue
        n = rs.standard_normal(size=shp).astype(np.float32)
        if k.endswith("running_var"):
            a = 1.0 + 0.2 * np.abs(n)
        elif k.endswith("running_mean"):
            a = 0.1 * n
        elif ".bn" in k or "downsample.1" in k or k.startswith("bn") or ".bn1." in k:
            a = (1.0 + 0.1 * n) if k.endswith("weight") else 0.1 * n
        elif v.dim() == 4:
            a = n * np.float32((2.0 / (shp[1] * shp[2] * shp[3])) ** 0.5)
        elif v.dim() == 2:
            a = n * np.float32(0.5 * shp[1] ** -0.5)
        else:
            a = 0.1 * n
        out[k] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return out
