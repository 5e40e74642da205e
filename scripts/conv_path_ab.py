import torch, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from mode_diffusion_policy_amd import perceptual_encoders as E
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
torch.manual_seed(3)
enc = E.FiLMResNet50Policy(32).cuda().eval()
for n_, p_ in enc.named_parameters():
    if n_.startswith("film"): torch.nn.init.normal_(p_, std=0.05)
img = torch.randn(4, 3, 96, 96, device="cuda"); cond = torch.randn(4, 32, device="cuda")
sd = {k: v.clone() for k, v in enc.state_dict().items()}
grads = {}
for flag in (True, False, "again"):
    enc.load_state_dict(sd); enc.zero_grad(set_to_none=True)
    E.USE_HIP_CONV_WGRAD = bool(flag is True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = enc(img, cond)
    (out.float() ** 2).mean().backward()
    grads[flag] = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
r = sorted(((rel(grads[True][n], grads[False][n]), rel(grads["again"][n], grads[False][n]), float(grads[False][n].norm()), n) for n in grads[True]), reverse=True)
for x in r[:12]: print("%.3e  (plain vs plain again %.3e)  norm %.3e  %s" % x)
print("median", r[len(r)//2][0])
