"""Per-step wall times of the training bench loop (diagnostic for first-process-after-another slowness)."""
import os, sys, time, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mode_diffusion_policy_amd.optim import FusedAdamW
from mode_diffusion_policy_amd.utils import rand_log_logistic
dev = torch.device("cuda:0")
M, den = bench.build_model(dev, "bf16"); m = den.inner_model; den.train()
B = 128
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 2, 2048, generator=g).to(dev); goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
opt = FusedAdamW(m, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
ts = []
for i in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
    loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    ts.append((t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for i, t in enumerate(ts):
    print(f"step {i:2d}: total {t[0]*1e3:7.2f} ms | host: fwd {t[1]*1e3:6.2f} bwd {t[2]*1e3:6.2f} opt {t[3]*1e3:5.2f} | drain {t[4]*1e3:6.2f}")
