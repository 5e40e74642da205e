"""Where the HOST time of a training step goes (the chain is eager: ~480 launches per step): per-phase host time without device syncs, then the
synchronised step time.  A step is GPU-bound when the host enqueues it faster than the GPU runs it."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mode_diffusion_policy_amd.optim import FusedAdamW  # noqa: E402
from mode_diffusion_policy_amd.utils import rand_log_logistic  # noqa: E402

dev = torch.device("cuda:0")
M, den = bench.build_model(dev)
m = den.inner_model
den.train()
B = 128
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 2, 2048, generator=g).to(dev); goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
FUSE = os.environ.get("MODE_FUSE_EXPERT_STEP", "1") == "1"
opt = FusedAdamW(m, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, fuse_expert_step=FUSE)
T = {"sigma": 0.0, "loss": 0.0, "backward": 0.0, "opt": 0.0}


def step(rec):
    t0 = time.perf_counter()
    sg = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
    t1 = time.perf_counter()
    loss, _ = den.loss({"state_images": img}, acts, goal, noise, sg)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step(overlap=not FUSE)
    t4 = time.perf_counter()
    if rec:
        T["sigma"] += t1 - t0; T["loss"] += t2 - t1; T["backward"] += t3 - t2; T["opt"] += t4 - t3


for _ in range(3):
    step(False)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step(True)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print(f"host enqueue {th / n * 1e3:.2f} ms/step, synchronised {tt / n * 1e3:.2f} ms/step; host phases (ms/step): " +
      ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in T.items()), flush=True)
print("cpu:", os.cpu_count(), "threads torch:", torch.get_num_threads(), "affinity:", len(os.sched_getaffinity(0)))
