"""Host-memory (RSS) and device-memory trace of a long eager training run: 600 steps, sampled every 100."""
import math, os, sys, torch, psutil
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
proc = psutil.Process()
for step in range(601):
    sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
    loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig)
    loss.backward(); opt.step(overlap=True)
    if step % 100 == 0:
        torch.cuda.synchronize()
        print(f"step {step:4d}: loss {float(loss):.4f}  host RSS {proc.memory_info().rss / 2**20:8.1f} MiB  device allocated {torch.cuda.memory_allocated() / 2**30:6.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB", flush=True)
