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
    deep_trunks_training_fixture(P, cond_dim)


DEEP_B, DEEP_HW, DEEP_SEED, DEEP_KEEP = 16, 128, 9, (0, 5, 10, 15)


def deep_inputs(cond_dim):
    """Inputs of F15b, regenerated from the seed by the test as well (3 MB of images that need not be stored)."""
    rs = np.random.RandomState(DEEP_SEED)
    img = torch.from_numpy(rs.standard_normal((DEEP_B, 3, DEEP_HW, DEEP_HW)).astype(np.float32))
    cond = torch.from_numpy(rs.standard_normal((DEEP_B, 1, cond_dim)).astype(np.float32))
    return img, cond


def deep_trunks_training_fixture(P, cond_dim):
    """F15b: TRAINING-mode forward + backward of the 34- / 50-layer encoders (the reference's classes) in FLOAT64 on 16 frames of 128 x 128.
    Why fp64: the gradient through 36 / 53 training-mode BatchNorms amplifies fp32 rounding so much that the reference's OWN fp32 run is 0.6-1.8e-2
    (ResNet-34) / 1.6-2.6e-2 (ResNet-50) away from this one (oracle/measure_fp32_encoder_grad_gap.py -> tests/golden/fp32_encoder_grad_gap.json) - an
    fp32 fixture would carry that much noise itself.  d img is stored for four of the sixteen frames + its norm over all; values are stored as fp32."""
    img, cond = deep_inputs(cond_dim)
    out = {}
    f32 = lambda t: t.detach().to(torch.float32).numpy()
    for tag, ctor, seed in (("r50", lambda: P.FiLMResNet50Policy(cond_dim), 500), ("r34", lambda: P.FiLMResNet34Policy(cond_dim), 501)):
        m = ctor()
        m.load_state_dict(R.fill_encoder_state_dict(m.state_dict(), seed))
        m = m.double().train()
        xi = img.double().requires_grad_(True); ci = cond.double().requires_grad_(True)
        yt = m(xi, ci)
        w = torch.from_numpy(np.random.RandomState(seed + 1).standard_normal(tuple(yt.shape)).astype(np.float32))
        (yt * w.double()).sum().backward()
        out[f"{tag}_train"] = f32(yt); out[f"{tag}_w"] = w.numpy()
        out[f"{tag}_dimg_keep"] = f32(xi.grad[list(DEEP_KEEP)]); out[f"{tag}_dimg_norm"] = np.float64(xi.grad.norm())
        out[f"{tag}_dcond"] = f32(ci.grad)
        params = dict(m.named_parameters())
        names = [k for k, p in params.items() if p.grad is not None]
        pick = [names[0], names[len(names) // 3], names[2 * len(names) // 3]] + [k for k in names if "film" in k][:4] + [k for k in names if k.endswith("bn1.weight")][:2]
        for k in dict.fromkeys(pick):
            if params[k].numel() <= 300_000:
                out[f"{tag}_g:{k}"] = f32(params[k].grad)
        out[f"{tag}_gn_keys"] = np.array(names); out[f"{tag}_gn_vals"] = np.array([float(params[k].grad.norm()) for k in names], dtype=np.float64)
        out[f"{tag}_rm"] = f32(m.state_dict()[[k for k in m.state_dict() if k.endswith("bn1.running_mean")][0]])
        print("F15b", tag, "train out", tuple(yt.shape), flush=True)
    np.savez_compressed(os.path.join(OUT, "F15b_encoders_train.npz"), B=DEEP_B, HW=DEEP_HW, seed=DEEP_SEED, keep=np.array(DEEP_KEEP), cond_dim=cond_dim, **out)
    print("wrote F15b_encoders_train.npz", os.path.getsize(os.path.join(OUT, "F15b_encoders_train.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
