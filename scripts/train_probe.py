"""Probe: config-3 training step (B=128, full C2 model) timing through the HIP path."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import mode_diffusion_policy_amd as M
from mode_diffusion_policy_amd.ddp import optimizer_param_groups
dev = torch.device("cuda:0")
_, den = Bn.build_model(dev)
m = den.inner_model
den.train()
B = 128
img, goal, x0 = Bn.synthetic_inputs(dev, B)
acts = torch.randn(B, 10, 7, device=dev); noise = torch.randn(B, 10, 7, device=dev)
opt = torch.optim.AdamW(optimizer_param_groups(m, 0.05), lr=1e-4, betas=(0.9, 0.95), fused=True)
from mode_diffusion_policy_amd.utils import rand_log_logistic
import math
def step():
    sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
    opt.zero_grad(set_to_none=True)
    loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig)
    loss.backward()
    opt.step()
    return loss
for _ in range(3): l = step()
torch.cuda.synchronize(); print("loss", float(l), "mem GB", torch.cuda.max_memory_allocated() / 2**30)
def timeit(fn, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("full step ms", timeit(step))
sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
def fwd():
    with torch.no_grad():
        den.loss({"state_images": img}, acts, goal, noise, sig)
print("fwd only ms (incl. shadow refresh if stale)", timeit(fwd))
def fwdbwd():
    opt.zero_grad(set_to_none=True)
    loss, _ = den.loss({"state_images": img}, acts, goal, noise, sig); loss.backward()
print("fwd+bwd ms (weights unchanged)", timeit(fwdbwd))
