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

# training: the same step (same host seed -> same sigma draw, multinomial routing and dropout seed) must give bit-identical gradients
import math  # noqa: E402
from mode_diffusion_policy_amd.utils import rand_log_logistic  # noqa: E402

m = den.inner_model
den.train()
B = 128
g = torch.Generator().manual_seed(1)
img = torch.randn(B, 2, 2048, generator=g).to(dev); goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)


def grads():
    for p_ in m.parameters():                                # as after zero_grad(set_to_none=True): the next backward writes fresh gradients
        p_.grad = None
    if getattr(m.engine.arena, "grad", None) is not None:
        m.engine.arena.grad_pending = False                  # grad_mode="arena": the next backward overwrites the arena (a second one would accumulate)
    torch.manual_seed(123)
    sg = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
    loss, _ = den.loss({"state_images": img}, acts, goal, noise, sg)
    loss.backward()
    torch.cuda.synchronize()
    return torch.cat([p_.grad.reshape(-1).float() for p_ in m.parameters() if p_.grad is not None]).clone(), float(loss.detach())


ref_g, ref_l = grads()
bad = 0
nt = max(1, n // 20)
for i in range(nt):
    gi, li = grads()
    if not torch.equal(gi, ref_g) or li != ref_l:
        bad += 1
print(f"training step: {nt} repeats, {bad} with differing gradients, |g| = {float(ref_g.norm()):.4f}", flush=True)
assert bad == 0
