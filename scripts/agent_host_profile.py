"""cProfile of the agent training step's HOST side (bench.py --mode agent's step, 5 steps after warm-up): where the ~35 ms of enqueue time go."""
import cProfile, math, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mode_diffusion_policy_amd.optim import FusedAdamW, FlatAdamW
from mode_diffusion_policy_amd.perceptual_encoders import FiLMResNet50Policy, embed_visual_obs
from mode_diffusion_policy_amd.utils import rand_log_logistic
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
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
