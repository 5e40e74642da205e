"""Which torch (non-library) launches are left in the agent training step, and which source lines issue them: torch.profiler with stacks over
3 steady-state steps of bench.py --mode agent's step (fused expert update + FlatAdamW for the encoders), grouped by kernel name and by the
innermost frame inside this repo."""
import collections, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mode_diffusion_policy_amd.optim import FusedAdamW, FlatAdamW
from mode_diffusion_policy_amd.perceptual_encoders import FiLMResNet50Policy, embed_visual_obs
from mode_diffusion_policy_amd.utils import rand_log_logistic
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
M, den = bench.build_model(dev, "bf16"); m = den.inner_model; den.train()
B = 64
es, eg = FiLMResNet50Policy(512).to(dev).train(), FiLMResNet50Policy(512).to(dev).train()
g = torch.Generator().manual_seed(1)
rs = torch.randn(B, 1, 3, 224, 224, generator=g).to(dev); rg = torch.randn(B, 1, 3, 224, 224, generator=g).to(dev)
goal = torch.randn(B, 1, 512, generator=g).to(dev); acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
opt = FusedAdamW(m, lr=1e-4, fuse_expert_step=True); opt_e = FlatAdamW(list(es.parameters()) + list(eg.parameters()), lr=1e-4)


def step():
    sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        emb = embed_visual_obs(es, eg, rs, rg, goal.squeeze(1))
        loss, _ = den.loss(emb, acts, goal, noise, sig)
    loss.backward(); opt.step(); opt_e.step(); opt_e.zero_grad(set_to_none=True)


for _ in range(4):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()

repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
by_op = collections.Counter(); by_site = collections.Counter(); t_op = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    kern = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
    if not kern:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if repo in fr and "scripts/" not in fr:
            site = fr.replace(repo + "/", ""); break
    if site == "?" and ev.stack:
        site = "autograd/" + ev.stack[0][-70:]
    by_op[ev.name] += len(kern); t_op[ev.name] += sum(k.duration for k in kern)
    by_site[(ev.name, site)] += len(kern)
print(f"torch launches per step by op (top-level aten ops that launch kernels; {N} steps averaged)")
for name, n in by_op.most_common(25):
    print(f"  {n / N:7.1f} x  {t_op[name] / N:8.1f} us  {name}")
print("by call site")
for (name, site), n in by_site.most_common(45):
    print(f"  {n / N:7.1f} x  {name:28s} {site}")
