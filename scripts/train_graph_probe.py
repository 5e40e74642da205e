"""Potential of a hipGraph-captured training step: capture forward + backward (fixed dropout seed — measurement only) and compare a replay
with the eager chain."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mode_diffusion_policy_amd.engine import capture_graph  # noqa: E402
from mode_diffusion_policy_amd.utils import rand_log_logistic  # noqa: E402

dev = torch.device("cuda:0")
M, den = bench.build_model(dev, "bf16"); m = den.inner_model; den.train()
B = 128
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 2, 2048, generator=g).to(dev); goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)


def fb():
    loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig)
    loss.backward()
    return loss


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"eager forward+backward: {timeit(fb):.2f} ms", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fb()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
try:
    gr = torch.cuda.CUDAGraph()
    with capture_graph(gr):
        fb()
    print(f"graph replay forward+backward: {timeit(gr.replay):.2f} ms", flush=True)
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:300], flush=True)
