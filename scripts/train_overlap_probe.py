"""A/B inside ONE process: training step with the optimizer after the backward (serial) vs. the per-block AdamW launches overlapped with
the rest of the backward on a second stream (FusedAdamW.step(overlap=True)), for several "adamw_blocks" caps.  Interleaved rounds."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mode_diffusion_policy_amd import _lib as L  # noqa: E402
from mode_diffusion_policy_amd.optim import FusedAdamW  # noqa: E402
from mode_diffusion_policy_amd.utils import rand_log_logistic  # noqa: E402

dev = torch.device("cuda:0")
M, den = bench.build_model(dev, "bf16"); m = den.inner_model; den.train()
B = 128
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 2, 2048, generator=g).to(dev); goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
opt = FusedAdamW(m, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
lib = L.load()


def run(n, overlap):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
        loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig)
        loss.backward()
        opt.step(overlap=overlap)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(5, False)
for rnd in range(4):
    out = []
    for name, ov, blocks in (("serial", False, 0), ("serial/1024", False, 1024), ("serial/512", False, 512), ("serial/384", False, 384), ("serial/256", False, 256),
                             ("serial/192", False, 192), ("overlap/384", True, 384), ("overlap/256", True, 256), ("overlap/192", True, 192)):
        lib.mode_set_option(b"adamw_blocks", blocks)
        run(2, ov)
        out.append(f"{name} {run(15, ov):6.2f}")
    print(" | ".join(out), flush=True)
