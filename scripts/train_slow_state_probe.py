"""Root cause of the 'slow state' of the eager training chain (bench.py train_leg blocks of 12.6 / 24 / 50 ms per step in one process):
per window of 20 steps - host enqueue time (loop returns, no sync), wall time (after sync), GPU time (events around the window), CPU time of
this process, run-queue load, GPU clock / power.  A window is HOST-bound when enqueue ~ wall, GPU-bound when enqueue << wall ~ GPU time.
argv: windows [overlap|serial|noopt]"""
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
t_start = time.perf_counter()
M, den = bench.build_model(dev, "bf16"); m = den.inner_model; den.train()
B = 128
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 2, 2048, generator=g).to(dev); goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
mode = sys.argv[2] if len(sys.argv) > 2 else "overlap"
opt = FusedAdamW(m, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
n = 20
print(f"mode={mode} cpus={os.cpu_count()} affinity={len(os.sched_getaffinity(0))} torch threads={torch.get_num_threads()}", flush=True)
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with bench.PowerSampler(0) as ps:
        torch.cuda.synchronize(); c0 = time.process_time(); t0 = time.perf_counter(); e0.record()
        for _ in range(n):
            sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
            loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig)
            loss.backward()
            if mode != "noopt":
                opt.step(overlap=mode == "overlap")
        e1.record(); th = time.perf_counter() - t0
        torch.cuda.synchronize(); tw = time.perf_counter() - t0; cpu = time.process_time() - c0
    s = ps.summary()
    print(f"t={time.perf_counter() - t_start:6.1f}s w{w:2d}: wall {tw / n * 1e3:6.2f} host-enqueue {th / n * 1e3:6.2f} gpu {e0.elapsed_time(e1) / n:6.2f} ms/step  "
          f"cpu {cpu / tw * 100:4.0f}%  load {os.getloadavg()[0]:.1f}  sclk {s['sclk_mhz_avg']} W {s['socket_w_avg']}", flush=True)
