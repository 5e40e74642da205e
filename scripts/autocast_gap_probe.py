"""Yardstick for the bf16-autocast test of the encoders: the gap between an fp32 and a torch.autocast(bfloat16) run of (a) the plain torch ResNet-18 trunk
of oracle/resnet_oracle.py (torch's own BatchNorm) and (b) this package's FiLMResNet18Policy (fused HIP BatchNorm/FiLM pass), eval and train mode."""
import sys
import torch
sys.path.insert(0, ".")
import mode_diffusion_policy_amd as M
from oracle import resnet_oracle as R


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def run(net, call, train, B=4, hw=64):
    net.train(train)
    torch.manual_seed(0)
    x = torch.randn(B, 3, hw, hw, device="cuda")
    w = None
    res = {}
    for mode in ("fp32", "bf16"):
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16"):
            y = call(net, x)
        if w is None:
            w = torch.randn_like(y.float())
        (y.float() * w).sum().backward()
        first = [p for n, p in net.named_parameters() if n.endswith("conv1.weight")][0]
        res[mode] = (y.float().detach().clone(), first.grad.clone())
    return rel(res["bf16"][0], res["fp32"][0]), rel(res["bf16"][1], res["fp32"][1])


for B, hw in ((4, 64), (32, 112)):
    trunk = R.create_model("resnet18").cuda()
    trunk.load_state_dict({k: v.cuda() for k, v in R.fill_encoder_state_dict(trunk.state_dict(), 1).items()})
    def call_trunk(n, x):
        x = n.maxpool(n.act1(n.bn1(n.conv1(x)))) if hasattr(n, "act1") else n.maxpool(torch.relu(n.bn1(n.conv1(x))))
        for i in range(4):
            x = getattr(n, f"layer{i + 1}")(x)
        return n.global_pool(x).flatten(1)
    enc = M.FiLMResNet18Policy(32).cuda()
    enc.load_state_dict({k: v.cuda() for k, v in R.fill_encoder_state_dict(enc.state_dict(), 1).items()})
    cond = torch.randn(B, 1, 32, device="cuda")
    for train in (False, True):
        print(f"B={B} hw={hw} train={train}: torch trunk (out, d conv1) = {run(trunk, call_trunk, train, B, hw)}   HIP encoder = {run(enc, lambda n, x: n(x, cond), train, B, hw)}", flush=True)
