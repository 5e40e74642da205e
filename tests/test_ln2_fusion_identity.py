"""Host-side restatement of the fused ln_2 algebra (LABNOTES.md §4), CPU: moving RMSNorm's per-row division behind the GEMM and summing
per-64-column partial sums of squares gives the oracle's ln_2 -> expert up-projection result.  The HIP kernels implement exactly these
steps (MODE_EPI_RESIDUAL_NORM producer, MODE_EPI_SWIGLU row-scale consumer); their GPU parity is tests/test_gpu_kernels.py."""
import torch

from oracle import mode_oracle as O


def test_rmsnorm_commutes_with_the_up_projection():
    g = torch.Generator().manual_seed(0)
    N, D, H = 37, 256, 512
    x = torch.randn(N, D, generator=g) * 3.0
    gain = 1.0 + 0.2 * torch.randn(D, generator=g)
    W = torch.randn(2 * H, D, generator=g) * D ** -0.5
    b = torch.randn(2 * H, generator=g) * 0.1
    eps = 1e-6
    # oracle: ln_2 then Linear + SwishGLU
    h = O.rmsnorm(x, gain, eps)
    pre = h @ W.t() + b
    want = pre[:, :H] * torch.nn.functional.silu(pre[:, H:])
    # fused formulation: partial sums over 64-column groups (producer), row scale after the contraction (consumer)
    ss = x.view(N, D // 64, 64).pow(2).sum(-1)                         # [N, D/64] what the c_proj epilogue writes
    inv = 1.0 / torch.clamp(ss.sum(-1).sqrt() * D ** -0.5, min=eps)    # what the up-projection computes per row
    pre2 = ((x * gain) @ W.t()) * inv[:, None] + b
    got = pre2[:, :H] * torch.nn.functional.silu(pre2[:, H:])
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-6)
    # and the residual the combine kernel rebuilds from x, the partial sums and the gain is the oracle's normalised stream
    assert torch.allclose(x * inv[:, None] * gain, h, rtol=2e-6, atol=1e-7)
