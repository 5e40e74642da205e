"""Latency of one 10-step DDIM chunk (hipGraph replay) for small environment batches: ms per chunk call and effective weight-stream rate
(active bf16 weights per denoise step = 12 layers x (2 experts x 25.2 MB + 8.4 MB attention) = 705 MB)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
M, den = bench.build_model(dev)
sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128]:
    img, goal, x0 = bench.synthetic_inputs(dev, B)
    fn = lambda: M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:4d}: {ms:7.3f} ms/chunk  {ms / 120 * 1e3:6.1f} us/layer  weights {0.705 * 10 / ms:5.2f} TB/s  {B / ms * 1e3:8.1f} chunks/s", flush=True)
