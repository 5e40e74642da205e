"""The encoders' entry on the library's kernels (csrc/conv_stem.hip, ABI 12): the small-channel stem convolution (reference: torchvision / timm ResNet
`conv1`, mode/models/perceptual_encoders/resnets.py:96, pretrained_resnets.py:29) and the max-pool behind it, against plain torch fp32 references of
the same ops on the same bf16-rounded operands.  Tolerances: the forward output is one bf16 rounding of an fp32 sum (rel-L2 <= 4e-3 = 2^-8); the weight
gradient is an fp32 sum of bf16 products (rel-L2 <= 1e-4, differences come from the summation order only); the max-pool is exact."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _stem_case(seed, N, cin, cout, H, W_, k, s, p, layout):
    from mode_diffusion_policy_amd import perceptual_encoders as E
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, cin, H, W_, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5
    if layout == "nchw_f32":
        xd = x.cuda()
    elif layout == "nhwc_bf16":
        xd = x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    else:                                                                    # a strided window of a larger fp32 buffer
        big = torch.zeros(N, cin + 1, H + 3, W_ + 2).cuda()
        big[:, 1:, 2:-1, 1:-1] = x.cuda()
        xd = big[:, 1:, 2:-1, 1:-1]
        assert not xd.is_contiguous()
    w_lp = w.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xr = x.to(torch.bfloat16).double(); wr = w.to(torch.bfloat16).double()
    return E, xd, w_lp, xr, wr


GEOMS = [(2, 3, 64, 224, 224, 7, 2, 3), (3, 3, 64, 37, 45, 7, 2, 3), (1, 3, 64, 7, 7, 7, 2, 3), (2, 1, 16, 19, 23, 5, 1, 2), (2, 4, 48, 20, 11, 3, 3, 0),
         (1, 5, 32, 9, 30, 7, 1, 3), (3, 2, 64, 16, 16, 1, 1, 0), (2, 3, 64, 31, 18, 9, 2, 4)]


@pytest.mark.parametrize("layout", ["nchw_f32", "nhwc_bf16", "strided_f32"])
@pytest.mark.parametrize("N,cin,cout,H,W_,k,s,p", GEOMS)
def test_stem_conv_forward_and_weight_gradient_vs_torch(N, cin, cout, H, W_, k, s, p, layout):
    E, xd, w_lp, xr, wr = _stem_case(11 + N + H, N, cin, cout, H, W_, k, s, p, layout)
    y = E._stem_fwd(xd, w_lp, (s, s), (p, p))
    ref = F.conv2d(xr, wr, None, s, p)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    assert rel(y, ref) < 4e-3, rel(y, ref)
    # weight gradient for a random bf16 dy
    dy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16)
    w = torch.zeros(cout, cin, k, k, device="cuda", requires_grad=True)
    out = E._StemConvFn.apply(xd, w, w_lp, (s, s), (p, p))
    out.backward(dy.cuda().contiguous(memory_format=torch.channels_last))
    wref = wr.clone().requires_grad_(True)
    F.conv2d(xr, wref, None, s, p).backward(dy.double())
    assert w.grad.shape == wref.grad.shape and w.grad.dtype == torch.float32
    assert rel(w.grad, wref.grad) < 1e-4, rel(w.grad, wref.grad)
    # deterministic: the slabs are summed in a fixed order
    w2 = torch.zeros_like(w, requires_grad=True)
    E._StemConvFn.apply(xd, w2, w_lp, (s, s), (p, p)).backward(dy.cuda().contiguous(memory_format=torch.channels_last))
    assert torch.equal(w.grad, w2.grad)


def test_stem_conv_many_tiles_per_workgroup_and_empty_batch():
    """More 128-pixel tiles than workgroups (the persistent loops of both kernels), and N = 0."""
    from mode_diffusion_policy_amd import perceptual_encoders as E
    N, H = 20, 224                                                            # 20 * 98 = 1960 tiles > 512 workgroups
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 3, H, H, generator=g).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.08).cuda()
    w_lp = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = E._stem_fwd(x, w_lp, (2, 2), (3, 3))
    ref = F.conv2d(x.to(torch.bfloat16).float(), w_lp.float(), None, 2, 3)
    assert rel(y, ref) < 4e-3
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16).cuda()
    wp = w.clone().requires_grad_(True)
    E._StemConvFn.apply(x, wp, w_lp, (2, 2), (3, 3)).backward(dy.contiguous(memory_format=torch.channels_last))
    xr = x.to(torch.bfloat16).float(); wr = w_lp.float().requires_grad_(True)
    F.conv2d(xr, wr, None, 2, 3).backward(dy.float())
    assert rel(wp.grad, wr.grad) < 2e-3                                       # the fp32 torch reference itself carries ~1e-3 of summation error at 250 000 terms
    y0 = E._stem_fwd(x[:0], w_lp, (2, 2), (3, 3))
    assert y0.shape == (0, 64, 112, 112)
    w0 = w.clone().requires_grad_(True)
    E._StemConvFn.apply(x[:0], w0, w_lp, (2, 2), (3, 3)).sum().backward()
    assert float(w0.grad.abs().max()) == 0.0


@pytest.mark.parametrize("bias_affine", [True, False])
def test_stem_inference_epilogue_equals_conv_then_batchnorm(bias_affine):
    from mode_diffusion_policy_amd import perceptual_encoders as E
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 3, 64, 48, generator=g).cuda()
    conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(64, affine=bias_affine).cuda().eval()
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(64, generator=g) * 0.3); bn.running_var.copy_(torch.rand(64, generator=g) + 0.4)
        if bias_affine:
            bn.weight.copy_(torch.rand(64, generator=g) + 0.5); bn.bias.copy_(torch.randn(64, generator=g) * 0.2)
    E._store_channels_last(conv)
    w_lp = conv.weight.detach().to(torch.bfloat16)
    with torch.no_grad():
        fused = E._stem_fwd(x, w_lp, conv.stride, conv.padding, bn=bn, relu=True)
        ref = F.relu(F.batch_norm(F.conv2d(x.to(torch.bfloat16).float(), w_lp.float(), None, 2, 3), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
    assert rel(fused, ref) < 4e-3, rel(fused, ref)
    assert float(fused.float().min()) >= 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,C_,H,W_,k,s,p", [(2, 64, 112, 112, 3, 2, 1), (3, 8, 7, 9, 3, 2, 1), (1, 24, 5, 5, 2, 2, 0), (2, 16, 13, 6, 3, 1, 1), (2, 40, 9, 9, 5, 3, 2), (1, 8, 3, 3, 3, 2, 1)])
def test_max_pool_forward_and_backward_equal_aten(N, C_, H, W_, k, s, p, dtype):
    """Values after a ReLU (many exact ties at 0): the selection rule decides where the gradient goes, so the backward only matches with aten's rule."""
    from mode_diffusion_policy_amd import perceptual_encoders as E
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.relu(torch.randn(N, C_, H, W_, generator=g)).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    xa = x.clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    ya = E.max_pool(xa, k, s, p)
    yb = F.max_pool2d(xb, k, s, p)
    assert ya.shape == yb.shape and torch.equal(ya, yb)
    dy = torch.randn(yb.shape, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    ya.backward(dy); yb.backward(dy)
    assert torch.allclose(xa.grad.float(), xb.grad.float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-6 if dtype == torch.float32 else 1e-2)
    # no-grad forward (no window positions kept) gives the same values
    with torch.no_grad():
        assert torch.equal(E.max_pool(x, k, s, p), yb)


def test_max_pool_propagates_nan_like_aten():
    from mode_diffusion_policy_amd import perceptual_encoders as E
    x = torch.randn(1, 8, 6, 6).cuda().contiguous(memory_format=torch.channels_last)
    x[0, 3, 2, 2] = float("nan"); x[0, 5, 0, 0] = float("-inf")
    a = E.max_pool(x, 3, 2, 1); b = F.max_pool2d(x, 3, 2, 1)
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, 7.0), torch.nan_to_num(b, 7.0))


def test_encoder_with_and_without_the_hip_stem(monkeypatch):
    """The whole FiLM-ResNet-50 tower under bf16 autocast (eval-mode BatchNorm: batch statistics of a 4-sample batch make the gradients chaotic, see
    test_encoders.py), forward and backward, with the stem + max-pool on the library's kernels and on MIOpen / aten; and the no-grad inference path
    (stem convolution + BatchNorm + ReLU in one launch)."""
    from mode_diffusion_policy_amd import perceptual_encoders as E
    torch.manual_seed(4)
    m = E.FiLMResNet50Policy(32).cuda().eval()
    for n_, p_ in m.named_parameters():
        if n_.startswith("film"):
            torch.nn.init.normal_(p_, std=0.05)
    img = torch.randn(4, 3, 96, 96, device="cuda"); cond = torch.randn(4, 32, device="cuda")
    outs = {}
    for hip in (True, False):
        monkeypatch.setattr(E, "USE_HIP_STEM", hip)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(img, cond)
            with torch.no_grad():
                yi = m(img, cond)
        (y.float() ** 2).mean().backward()
        outs[hip] = (y.detach().float().clone(), yi.float().clone(), m.resnet.conv1.weight.grad.detach().clone(), m.resnet.layer1[0].conv1.weight.grad.detach().clone())
    assert rel(outs[True][0], outs[False][0]) < 2e-2 and rel(outs[True][1], outs[False][1]) < 2e-2
    assert rel(outs[True][0], outs[True][1]) < 2e-2                            # grad-mode and inference paths of the same weights
    # conv1's own weight gradient sums 4 x 48 x 48 signed pixel terms that largely cancel: the 1-ulp bf16 differences between the two stems' outputs show
    # up amplified (measured 0.13; the kernel itself is pinned to 1e-4 against fp64 above) - this line only guards sign / scale / layout
    assert rel(outs[True][2], outs[False][2]) < 0.3 and rel(outs[True][3], outs[False][3]) < 8e-2


@pytest.mark.parametrize("case", range(int(os.environ.get("MODE_FUZZ_STEM_CASES", "12"))))
def test_stem_conv_random_geometries(case):
    from mode_diffusion_policy_amd import perceptual_encoders as E
    g = torch.Generator().manual_seed(7000 + case)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    k = [1, 3, 5, 7][ri(0, 3)]
    cin = ri(1, min(5, 256 // (k * k)))
    cout = 16 * ri(1, 4)
    s = ri(1, 3); p = ri(0, k // 2)
    N = ri(1, 3); H = ri(k, 40); W_ = ri(k, 40)
    x = torch.randn(N, cin, H, W_, generator=g).cuda()
    w = (torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5).cuda()
    w_lp = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wp = w.clone().requires_grad_(True)
    y = E._StemConvFn.apply(x, wp, w_lp, (s, s), (p, p))
    xr = x.to(torch.bfloat16).double(); wr = w_lp.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, s, p)
    assert rel(y, ref) < 4e-3, (k, cin, cout, s, p, N, H, W_)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16).cuda()
    y.backward(dy.contiguous(memory_format=torch.channels_last)); ref.backward(dy.double())
    assert rel(wp.grad, wr.grad) < 1e-4, (k, cin, cout, s, p, N, H, W_)


def test_misaligned_views_are_handled():
    """The kernels move 16-byte chunks: an activation / gradient that starts at an odd offset of its storage takes aten (max-pool input) or a
    realigned copy (gradients) - same results."""
    from mode_diffusion_policy_amd import perceptual_encoders as E
    g = torch.Generator().manual_seed(2)
    buf = torch.relu(torch.randn(2 * 9 * 9 * 16 + 4, generator=g)).to(torch.bfloat16).cuda()
    x = buf[4:].view(2, 9, 9, 16).permute(0, 3, 1, 2)                          # channels_last view, 8 bytes into the storage
    assert x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 8
    assert torch.equal(E.max_pool(x), F.max_pool2d(x, 3, 2, 1))
    xa = x.contiguous(memory_format=torch.channels_last).clone().requires_grad_(True); xb = xa.detach().clone().requires_grad_(True)
    ya = E.max_pool(xa); yb = F.max_pool2d(xb, 3, 2, 1)
    dbuf = torch.randn(ya.numel() + 4, generator=g).to(torch.bfloat16).cuda()
    dy = dbuf[4:].view(2, ya.shape[2], ya.shape[3], 16).permute(0, 3, 1, 2)
    assert dy.data_ptr() % 16 == 8
    ya.backward(dy); yb.backward(dy)
    assert torch.allclose(xa.grad.float(), xb.grad.float(), rtol=1e-2, atol=1e-2)
    # stem weight gradient with a misaligned dy
    img = torch.randn(2, 3, 20, 20, generator=g).cuda()
    w = (torch.randn(16, 3, 7, 7, generator=g) * 0.1).cuda()
    w_lp = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wp = w.clone().requires_grad_(True)
    y = E._StemConvFn.apply(img, wp, w_lp, (2, 2), (3, 3))
    dbuf = torch.randn(y.numel() + 4, generator=g).to(torch.bfloat16).cuda()
    dys = dbuf[4:].view(2, y.shape[2], y.shape[3], 16).permute(0, 3, 1, 2)
    y.backward(dys)
    wr = w_lp.double().requires_grad_(True)
    F.conv2d(img.to(torch.bfloat16).double(), wr, None, 2, 3).backward(dys.double())
    assert rel(wp.grad, wr.grad) < 1e-4
