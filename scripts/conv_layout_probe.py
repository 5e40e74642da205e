"""Does MIOpen run the ResNet-50 convolutions faster from channels_last (NHWC) tensors on this stack?  Plain torch trunk (oracle/resnet_oracle.py,
nn.BatchNorm2d), forward + backward under autocast(bf16), B=64 @ 224: contiguous (NCHW) vs channels_last, with and without PYTORCH_MIOPEN_SUGGEST_NHWC."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import resnet_oracle as R
net = R.create_model("resnet50").cuda().train()
x = torch.randn(64, 3, 224, 224, device="cuda")
def run(fmt):
    n = net.to(memory_format=fmt); xi = x.contiguous(memory_format=fmt)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = n.maxpool(torch.relu(n.bn1(n.conv1(xi))))
            for i in range(4):
                h = getattr(n, f"layer{i + 1}")(h)
            y = n.global_pool(h).flatten(1)
        y.float().square().mean().backward()
        n.zero_grad(set_to_none=True)
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
print("PYTORCH_MIOPEN_SUGGEST_NHWC =", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"))
print(f"contiguous (NCHW): {run(torch.contiguous_format):.2f} ms / step")
print(f"channels_last    : {run(torch.channels_last):.2f} ms / step")
