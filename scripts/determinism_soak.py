"""Race screen: the B=128 sampler graph replayed N times on the same inputs must reproduce the first result bit for bit (an LDS-DMA /
barrier ordering hazard in a kernel shows up as rare differing tiles), likewise a B=32 and a single-environment chunk and one training step's gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
M, den = bench.build_model(dev)
sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for B in (128, 32, 1):
    img, goal, x0 = bench.synthetic_inputs(dev, B)
    ref = M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True).clone()
    bad = 0
    for i in range(n):
        out = M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True)
        if not torch.equal(out, ref):
            bad += 1
    print(f"B={B}: {n} replays, {bad} mismatching, finite={bool(torch.isfinite(ref).all())}", flush=True)
    assert bad == 0
