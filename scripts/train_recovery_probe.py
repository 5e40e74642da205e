"""How long does the 'slow first training process after another GPU process' state last?  Prints ms/step of consecutive 25-step windows."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mode_diffusion_policy_amd.optim import FusedAdamW
from mode_diffusion_policy_amd.utils import rand_log_logistic
dev = torch.device("cuda:0")
t_start = time.perf_counter()
M, den = bench.build_model(dev, "bf16"); m = den.inner_model; den.train()
B = 128
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 2, 2048, generator=g).to(dev); goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
opt = FusedAdamW(m, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
import gc
if len(sys.argv) > 2 and sys.argv[2] == 'nogc':
    gc.collect(); gc.freeze(); gc.disable()
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(25):
        sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
        loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig)
        loss.backward()
        opt.step(overlap=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"t={t1 - t_start:6.1f}s window {w:2d}: {(t1 - t0) / 25 * 1e3:7.2f} ms/step", flush=True)
