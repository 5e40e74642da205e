"""Golden fixture F15: the reference's FiLM-ResNet encoder classes (mode/models/perceptual_encoders/pretrained_resnets.py, resnets.py) run on
top of the stand-in trunk of oracle/resnet_oracle.py (timm / torchvision are absent from the build image; see that file's header) - eval
forward, and a training-mode forward + backward (batch statistics, FiLM and trunk gradients).  Build container only; imports /root/reference.

    python -m oracle.gen_golden_encoders      # writes tests/golden/F15_encoders.npz
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from .gen_golden import OUT
from . import resnet_oracle as R


def _import_reference_encoders():
    timm = types.ModuleType("timm"); timm.create_model = R.create_model
    tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models"); tvm.resnet18 = R.resnet18; tv.models = tvm
    sys.modules.update({"timm": timm, "torchvision": tv, "torchvision.models": tvm})
    sys.path.insert(0, "/root/reference")
    from mode.models.perceptual_encoders import pretrained_resnets as P, resnets as Q
    return P, Q


def main():
    torch.set_num_threads(8)
    P, Q = _import_reference_encoders()
    out = {}
    B, cond_dim, HW = 4, 32, 64
    rs = np.random.RandomState(7)
    img = torch.from_numpy(rs.standard_normal((B, 3, HW, HW)).astype(np.float32))
    cond = torch.from_numpy(rs.standard_normal((B, 1, cond_dim)).astype(np.float32))
    out["img"], out["cond"] = img.numpy(), cond.numpy()
    for tag, ctor, seed in (("r50", lambda: P.FiLMResNet50Policy(cond_dim), 500), ("r34", lambda: P.FiLMResNet34Policy(cond_dim), 501),
                            ("r18p", lambda: P.FiLMResNet18Policy(cond_dim), 502), ("r18f", lambda: Q.ResNetEncoderWithFiLM(cond_dim, latent_dim=96), 503)):
        m = ctor()
        sd = R.fill_encoder_state_dict(m.state_dict(), seed)
        m.load_state_dict(sd)
        c = cond if tag != "r18f" else cond.squeeze(1)
        m.eval()
        with torch.no_grad():
            y = m(img, c)
        out[f"{tag}_eval"] = y.numpy()
        # eval-mode gradients (BatchNorm on running statistics: well conditioned at any depth)
        xe = img.clone().requires_grad_(True); ce = c.clone().requires_grad_(True)
        ye = m(xe, ce)
        we = torch.from_numpy(np.random.RandomState(seed + 2).standard_normal(tuple(ye.shape)).astype(np.float32))
        (ye * we).sum().backward()
        out[f"{tag}_we"] = we.numpy(); out[f"{tag}_e_dimg"] = xe.grad.numpy(); out[f"{tag}_e_dcond"] = ce.grad.numpy()
        en = [k for k, p in m.named_parameters() if p.grad is not None]
        out[f"{tag}_e_gn_keys"] = np.array(en); out[f"{tag}_e_gn_vals"] = np.array([float(dict(m.named_parameters())[k].grad.norm()) for k in en], dtype=np.float64)
        m.zero_grad(set_to_none=True)
        m.train()
        xi = img.clone().requires_grad_(True); ci = c.clone().requires_grad_(True)
        yt = m(xi, ci)
        w = torch.from_numpy(np.random.RandomState(seed + 1).standard_normal(tuple(yt.shape)).astype(np.float32))
        (yt * w).sum().backward()
        out[f"{tag}_train"] = yt.detach().numpy(); out[f"{tag}_w"] = w.numpy()
        out[f"{tag}_dimg"] = xi.grad.numpy(); out[f"{tag}_dcond"] = ci.grad.numpy()
        names = [k for k, p in m.named_parameters() if p.grad is not None]
        pick = [names[0], names[len(names) // 3], names[2 * len(names) // 3]] + [k for k in names if "film" in k][:4] + [k for k in names if k.endswith("bn1.weight")][:2]
        for k in dict.fromkeys(pick):
            out[f"{tag}_g:{k}"] = dict(m.named_parameters())[k].grad.numpy()
        out[f"{tag}_gn_keys"] = np.array(names); out[f"{tag}_gn_vals"] = np.array([float(dict(m.named_parameters())[k].grad.norm()) for k in names], dtype=np.float64)
        out[f"{tag}_rm"] = m.state_dict()[[k for k in m.state_dict() if k.endswith("bn1.running_mean")][0]].numpy()      # running statistics moved by the training forward
        out[f"{tag}_keys"] = np.array(list(sd.keys()))
        print(tag, "out", tuple(y.shape), "params", sum(p.numel() for p in m.parameters()), "tensors with grad", len(names), flush=True)
    np.savez_compressed(os.path.join(OUT, "F15_encoders.npz"), B=B, cond_dim=cond_dim, **out)
    print("wrote F15_encoders.npz", os.path.getsize(os.path.join(OUT, "F15_encoders.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
