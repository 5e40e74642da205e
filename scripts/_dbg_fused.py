import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import mode_diffusion_policy_amd as M
from mode_diffusion_policy_amd.optim import FusedAdamW
from test_gpu_train import build_train
from oracle.weights import make_inputs
cfg, sd, ma = build_train("c1e4", 41, "bf16")
_, _, mb = build_train("c1e4", 41, "bf16")
inp = {k: v.cuda() for k, v in make_inputs(cfg, 16, 5).items()}
dena, denb = M.GCDenoiser(ma, 0.5).train(), M.GCDenoiser(mb, 0.5).train()
sig = torch.full((16,), 0.9, device="cuda")
oa = FusedAdamW(ma, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
ob = FusedAdamW(mb, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, fuse_expert_step=True)
st = {"state_images": inp["state_images"]}
for step in range(2):
    torch.manual_seed(100 + step); torch.cuda.manual_seed(100 + step)
    la, _ = dena.loss(st, inp["actions"], inp["goals"], inp["noise"], sig); la.backward()
    torch.manual_seed(100 + step); torch.cuda.manual_seed(100 + step)
    lb, _ = denb.loss(st, inp["actions"], inp["goals"], inp["noise"], sig); lb.backward()
    print("loss", float(la), float(lb))
    ga = {n: p.grad.clone() for n, p in ma.named_parameters() if p.grad is not None}
    oa.step(); ob.step(); torch.cuda.synchronize()
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        if not torch.equal(pa, pb):
            d = (pa - pb).abs()
            nz = d.ne(0)
            idx = nz.nonzero()
            print(step, n, tuple(pa.shape), "mismatch", int(nz.sum()), "of", pa.numel(), "max", float(d.max()), "first", idx[0].tolist(), "last", idx[-1].tolist(),
                  "rows", int(nz.any(-1).sum()), "cols", int(nz.any(0).sum()) if pa.dim() == 2 else -1)
    print("exp_avg equal", torch.equal(oa.exp_avg, ob.exp_avg), "exp_avg_sq equal", torch.equal(oa.exp_avg_sq, ob.exp_avg_sq))
    if not torch.equal(oa.exp_avg, ob.exp_avg):
        d = (oa.exp_avg - ob.exp_avg).ne(0); print(" exp_avg diffs", int(d.sum()), "first", int(d.nonzero()[0]), "last", int(d.nonzero()[-1]))
