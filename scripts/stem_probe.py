"""The encoders' stem on the library's kernels (csrc/conv_stem.hip) against MIOpen / aten at the agent's shapes: conv1 3 -> 64, 7 x 7 / 2 on 224 x 224 images
(forward and weight gradient) and the 3 x 3 / 2 max-pool on its 112 x 112 x 64 output (forward and backward).  us per call, torch events over 20 calls."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import perceptual_encoders as E


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for B in (1, 8, 64, 128):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 3, 224, 224, generator=g).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.08).cuda().contiguous(memory_format=torch.channels_last)
    w_lp = w.to(torch.bfloat16)
    xl = x.contiguous(memory_format=torch.channels_last).to(torch.bfloat16)
    y = E._stem_fwd(x, w_lp, (2, 2), (3, 3))
    dy = torch.randn_like(y)
    wp = w.clone().requires_grad_(True)

    def hip_wgrad():
        out = E._StemConvFn.apply(x, wp, w_lp, (2, 2), (3, 3))
        wp.grad = None
        out.backward(dy)

    def mio_cast_fwd():
        return F.conv2d(x.contiguous(memory_format=torch.channels_last).to(torch.bfloat16), w_lp, None, 2, 3)

    r = {"hip fwd (fp32 NCHW in)": t_us(lambda: E._stem_fwd(x, w_lp, (2, 2), (3, 3))),
         "hip fwd+wgrad": t_us(hip_wgrad),
         "MIOpen fwd (bf16 NHWC in)": t_us(lambda: F.conv2d(xl, w_lp, None, 2, 3)),
         "layout + cast + MIOpen fwd": t_us(mio_cast_fwd),
         "MIOpen wgrad": t_us(lambda: torch.ops.aten.convolution_backward(dy, xl, w_lp, None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (False, True, False)))}
    a = torch.relu(torch.randn(B, 64, 112, 112, generator=g)).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    ah = a.clone().requires_grad_(True); at = a.clone().requires_grad_(True)
    ph = E.max_pool(ah); pt = F.max_pool2d(at, 3, 2, 1)
    dp = torch.randn_like(pt)
    r["hip pool fwd"] = t_us(lambda: E.max_pool(ah))
    r["aten pool fwd"] = t_us(lambda: F.max_pool2d(at, 3, 2, 1))
    r["hip pool bwd"] = t_us(lambda: torch.autograd.grad(ph, ah, dp, retain_graph=True))
    r["aten pool bwd"] = t_us(lambda: torch.autograd.grad(pt, at, dp, retain_graph=True))
    print(f"B = {B}: " + ", ".join(f"{k} {v:.1f}" for k, v in r.items()), flush=True)
