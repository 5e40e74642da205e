"""How well conditioned are the TRAINING-mode gradients of the reference's deep FiLM-ResNet encoders in fp32?  (Grounds the tolerance of
tests/test_encoders.py::test_deep_trunk_training_gradients_vs_reference_fixture.)  The reference classes (stand-in trunk, see gen_golden_encoders.py) run
the F15b batch three ways on the CPU: fp64 (the yardstick), fp32, and fp32 with the convolutions computed in channels_last memory format (another
summation order - what a different convolution algorithm does to the last bits).  Build container only; imports /root/reference.

    python -m oracle.measure_fp32_encoder_grad_gap      # writes tests/golden/fp32_encoder_grad_gap.json
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import resnet_oracle as R
from .gen_golden import OUT
from .gen_golden_encoders import DEEP_KEEP, _import_reference_encoders, deep_inputs


def rel(a, b):
    a = a.double(); b = b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def run(m, img, cond, w, dtype, cl=False):
    m = m.to(dtype).train()
    for mod in m.modules():                                                  # fresh running statistics per run
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.reset_running_stats()
    x = img.to(dtype)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
        m = m.to(memory_format=torch.channels_last)
    xi = x.clone().requires_grad_(True); ci = cond.to(dtype).clone().requires_grad_(True)
    m.zero_grad(set_to_none=True)
    y = m(xi, ci)
    (y * w.to(dtype)).sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    return y.detach(), xi.grad.detach(), ci.grad.detach(), grads


def main():
    torch.set_num_threads(8)
    P, _ = _import_reference_encoders()
    cond_dim = 32
    img, cond = deep_inputs(cond_dim)
    rows = []
    for tag, ctor, seed in (("r50", lambda: P.FiLMResNet50Policy(cond_dim), 500), ("r34", lambda: P.FiLMResNet34Policy(cond_dim), 501)):
        m = ctor()
        m.load_state_dict(R.fill_encoder_state_dict(m.state_dict(), seed))
        w = torch.from_numpy(np.random.RandomState(seed + 1).standard_normal((img.shape[0], m.resnet.num_features)).astype(np.float32))
        ref = run(m, img, cond, w, torch.float64)
        row = {"model": tag, "batch": list(img.shape)}
        for name, kw in (("fp32", {}), ("fp32_channels_last", {"cl": True})):
            y, dimg, dcond, grads = run(m, img, cond, w, torch.float32, **kw)
            gn = {k: float(v.double().norm()) for k, v in ref[3].items()}
            big = 1e-3 * max(gn.values())
            per = {k: rel(grads[k], ref[3][k]) for k in grads if gn[k] > big}
            row[name] = {"out": rel(y, ref[0]), "d_img": rel(dimg, ref[1]), "d_img_kept_frames": rel(dimg[list(DEEP_KEEP)], ref[1][list(DEEP_KEEP)]),
                         "d_cond": rel(dcond, ref[2]), "worst_param_grad": max(per.values()), "worst_param": max(per, key=per.get),
                         "median_param_grad": float(np.median(list(per.values())))}
            print(tag, name, row[name], flush=True)
        rows.append(row)
    with open(os.path.join(OUT, "fp32_encoder_grad_gap.json"), "w") as f:
        json.dump({"what": "reference FiLM-ResNet classes (stand-in trunk), TRAINING mode, F15b batch: fp32 vs fp64 on the CPU, rel-L2", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
