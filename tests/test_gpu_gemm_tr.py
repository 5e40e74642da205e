"""GPU parity of the backward-pass GEMMs fed by the LDS transpose read (gemm_bf16_tr.hip) against plain fp32 torch matmuls of the same
bf16 operands: data gradient dX = dY @ W (W row-major [K, N] = nn.Linear's [out, in]) and weight gradient dW = dY^T @ X (row-major
activations), plain / grouped by expert / arbitrary K ranges / gathered rows.  Inputs are random (not symmetric), so an operand or
output transpose cannot pass."""
import ctypes as C

import pytest
import torch

from mode_diffusion_policy_amd import _lib as L
from hip_helpers import p, stream

pytestmark = pytest.mark.gpu


def _desc(**kw):
    return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, **kw)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(256, 128, 128), (1792, 3072, 1024), (300, 64, 64), (130, 256, 4096), (17, 192, 200)])
def test_dgrad_plain(M, K, N, out_dtype):
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = torch.randn(K, N, generator=g).to(torch.bfloat16).cuda()
    Cc = torch.full((M, N), float("nan"), dtype=out_dtype, device="cuda")
    d = _desc(out_dtype=L.MODE_BF16 if out_dtype == torch.bfloat16 else L.MODE_F32, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=N,
              C=p(Cc), ldc=N, flags=L.GEMM_W_KN)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "dgrad")
    ref = A.float() @ W.float()
    assert torch.isfinite(Cc.float()).all()
    assert rel(Cc.float(), ref) < (1e-5 if out_dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("counts", [[896, 896, 896, 896], [0, 1000, 3, 517], [129, 0, 0, 0]])
def test_dgrad_grouped_by_expert(counts):
    E, K, N = len(counts), 256, 384
    M = sum(counts)
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = torch.randn(E, K, N, generator=g).to(torch.bfloat16).cuda()
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    Cc = torch.full((M, N), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=N, w_expert_stride=K * N, C=p(Cc), ldc=N,
              expert_offsets=p(off), num_experts=E, flags=L.GEMM_W_KN)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "dgrad grouped")
    o = 0
    for e, c in enumerate(counts):
        if c:
            assert rel(Cc[o:o + c], A[o:o + c].float() @ W[e].float()) < 1e-5, e
        o += c


@pytest.mark.parametrize("split", [2, 4])
@pytest.mark.parametrize("counts,K,N", [([896, 896, 896, 896], 2048, 256), ([0, 1000, 3, 517], 512, 384), ([37], 256, 64)])
def test_dgrad_k_slices(counts, K, N, split):
    """Data gradient in K-slices (split_k: fp32 partial slabs split_stride apart, added by the consumer — the training backward's dU): the slabs
    sum to the unsplit product; weight-gradient layouts refuse the option."""
    E = len(counts); M = sum(counts)
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = torch.randn(E, K, N, generator=g).to(torch.bfloat16).cuda()
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    Cc = torch.full((split, M, N), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=N, w_expert_stride=K * N, C=p(Cc), ldc=N,
              expert_offsets=p(off), num_experts=E, flags=L.GEMM_W_KN, split_k=split, split_stride=M * N)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "dgrad slices")
    tot = Cc.sum(0)
    o = 0
    for e, c in enumerate(counts):
        if c:
            assert rel(tot[o:o + c], A[o:o + c].float() @ W[e].float()) < 1e-5, e
            ks = K // split
            assert rel(Cc[1, o:o + c], A[o:o + c, ks:2 * ks].float() @ W[e, ks:2 * ks].float()) < 1e-5      # slice 1 is exactly its K range
        o += c
    d2 = _desc(out_dtype=L.MODE_F32, M=N, N=N, K=M, A=p(A), lda=K, W=p(A), ldw=K, C=p(Cc), ldc=N, flags=L.GEMM_W_KN | L.GEMM_A_KM, split_k=2,
               split_stride=N * N)
    assert L.load().mode_gemm(C.byref(d2), stream()) != 0


@pytest.mark.parametrize("R,M,N", [(1792, 1024, 1024), (1792, 3072, 1024), (100, 64, 64), (77, 136, 200), (64, 128, 128)])
def test_wgrad_plain(R, M, N):
    g = torch.Generator().manual_seed(R + M)
    A = torch.randn(R, M, generator=g).to(torch.bfloat16).cuda()
    X = torch.randn(R, N, generator=g).to(torch.bfloat16).cuda()
    Cc = torch.full((M, N), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=M, N=N, K=R, A=p(A), lda=M, W=p(X), ldw=N, C=p(Cc), ldc=N, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "wgrad")
    assert rel(Cc, A.float().t() @ X.float()) < 1e-5


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("counts", [[896, 896, 896, 896], [0, 1001, 3, 517], [5, 0, 70, 0]])
def test_wgrad_expert_segments(counts, gather):
    """Per-expert weight gradients over the sorted dispatch order: arbitrary (unpadded, possibly empty) row ranges; X rows optionally gathered
    through the dispatch permutation; stale LDS / rows past a segment must contribute exactly nothing (poisoned neighbours)."""
    E, M, N = len(counts), 256, 128
    R = sum(counts)
    g = torch.Generator().manual_seed(9)
    A = torch.randn(R, M, generator=g).to(torch.bfloat16).cuda()
    ntok = R // 2 + 3
    X = torch.randn(ntok if gather else R, N, generator=g).to(torch.bfloat16).cuda()
    perm = torch.randint(0, ntok, (R,), generator=g, dtype=torch.int32).cuda() if gather else None
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    Cc = torch.full((E, M, N), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=M, N=N, K=R, A=p(A), lda=M, W=p(X), ldw=N, C=p(Cc), ldc=N, k_group_offsets=p(off), num_k_groups=E,
              c_group_stride=M * N, w_rows=p(perm), flags=L.GEMM_W_KN | L.GEMM_A_KM)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "wgrad groups")
    Xs = X[perm.long()] if gather else X
    o = 0
    for e, c in enumerate(counts):
        ref = A[o:o + c].float().t() @ Xs[o:o + c].float()
        if c == 0:
            assert float(Cc[e].abs().max()) == 0.0
        else:
            assert rel(Cc[e], ref) < 1e-5, e
        o += c


@pytest.mark.parametrize("M,K,N", [(128, 2048, 1024), (70, 36, 52), (128, 24576, 256)])
def test_f32_dgrad_layout(M, K, N):
    """fp32 (router) data gradient straight from the [out, in] weight: C = A @ W with W row-major [K, N]."""
    g = torch.Generator().manual_seed(K)
    A = torch.randn(M, K, generator=g).cuda(); W = torch.randn(K, N, generator=g).cuda()
    Cc = torch.full((M, N), float("nan"), device="cuda")
    d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=N, C=p(Cc), ldc=N,
                       flags=L.GEMM_W_KN)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "f32 dgrad")
    assert rel(Cc, A.double() @ W.double()) < (2e-6 if K <= 4096 else 3e-5)      # one k-ordered fp32 chain: error ~ sqrt(K) * 2^-24


@pytest.mark.parametrize("R,M,N,groups", [(128, 2048, 1024, None), (77, 36, 52, None), (300, 64, 128, [0, 100, 100, 171, 300])])
def test_f32_wgrad_layout(R, M, N, groups):
    g = torch.Generator().manual_seed(R)
    A = torch.randn(R, M, generator=g).cuda(); X = torch.randn(R, N, generator=g).cuda()
    ng = len(groups) - 1 if groups else 1
    Cc = torch.full((ng, M, N), float("nan"), device="cuda")
    off = torch.tensor(groups, dtype=torch.int32).cuda() if groups else None
    d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=M, N=N, K=R, A=p(A), lda=M, W=p(X), ldw=N, C=p(Cc), ldc=N,
                       k_group_offsets=p(off), num_k_groups=ng if groups else 0, c_group_stride=M * N, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "f32 wgrad")
    gr = groups or [0, R]
    for z in range(ng):
        ref = A[gr[z]:gr[z + 1]].double().t() @ X[gr[z]:gr[z + 1]].double()
        if gr[z] == gr[z + 1]:
            assert float(Cc[z].abs().max()) == 0.0
        else:
            assert rel(Cc[z], ref) < 2e-6


# ---- the persistent ping-pong form of the same products (gemm_bf16_pptr.hip): forced with "gemm_tr_cfg" 6, compared with fp32 torch AND with the ring kernels
@pytest.fixture
def force_pptr():
    lib = L.load()
    lib.mode_set_option(b"gemm_tr_cfg", 6)
    yield lib
    lib.mode_set_option(b"gemm_tr_cfg", 0)


def _run(d, what):
    L.check(L.load().mode_gemm(C.byref(d), stream()), what)
    torch.cuda.synchronize()


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("counts,K,N,split", [([896, 896, 896, 896], 1024, 512, 1), ([871, 925, 903, 885], 256, 256, 1), ([0, 1000, 3, 517], 512, 768, 1),
                                              ([129, 0, 0, 0], 128, 256, 1), ([300, 1100, 20, 257], 2048, 256, 4), ([896, 896, 896, 896], 1024, 256, 2)])
def test_pptr_dgrad_grouped(force_pptr, counts, K, N, split, out_dtype):
    """Ragged expert segments (incl. empty ones, more than four 256-row tiles, a 3-row segment), K-slices; rows past a segment must never be stored
    (NaN canaries stay) and every slice is exactly its K range."""
    if split > 1 and out_dtype == torch.bfloat16:
        pytest.skip("K-slices write fp32 slabs")
    E = len(counts); M = sum(counts)
    g = torch.Generator().manual_seed(sum(counts) + K + N)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = torch.randn(E, K, N, generator=g).to(torch.bfloat16).cuda()
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    Cc = torch.full((split, M + 7, N), float("nan"), dtype=out_dtype, device="cuda")
    d = _desc(out_dtype=L.MODE_BF16 if out_dtype == torch.bfloat16 else L.MODE_F32, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=N, w_expert_stride=K * N,
              C=p(Cc), ldc=N, expert_offsets=p(off), num_experts=E, flags=L.GEMM_W_KN, split_k=split, split_stride=(M + 7) * N)
    _run(d, "pptr dgrad")
    assert torch.isnan(Cc[:, M:].float()).all()                       # canary rows behind the last segment
    tot = Cc[:, :M].float().sum(0)
    assert torch.isfinite(tot).all()
    tol = 1e-5 if out_dtype == torch.float32 else 4e-3
    o = 0
    for e, c in enumerate(counts):
        if c:
            assert rel(tot[o:o + c], A[o:o + c].float() @ W[e].float()) < tol, e
            if split > 1:
                ks = K // split
                assert rel(Cc[1, o:o + c], A[o:o + c, ks:2 * ks].float() @ W[e, ks:2 * ks].float()) < 1e-5
        o += c
    # same k-ordered fp32 MFMA chain as the ring kernels: bit-identical
    force_pptr.mode_set_option(b"gemm_tr_cfg", 7)
    C2 = torch.full_like(Cc, float("nan"))
    d.C = p(C2)
    _run(d, "ring dgrad")
    assert torch.equal(C2[:, :M].view(torch.int16 if out_dtype == torch.bfloat16 else torch.int32), Cc[:, :M].view(torch.int16 if out_dtype == torch.bfloat16 else torch.int32))


@pytest.mark.parametrize("M,K,N", [(256, 128, 256), (1792, 3072, 1024), (300, 256, 512), (3584, 1024, 4096)])
def test_pptr_dgrad_plain(force_pptr, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = torch.randn(K, N, generator=g).to(torch.bfloat16).cuda()
    Cc = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    d = _desc(out_dtype=L.MODE_BF16, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=N, C=p(Cc), ldc=N, flags=L.GEMM_W_KN)
    _run(d, "pptr dgrad plain")
    assert rel(Cc.float(), A.float() @ W.float()) < 4e-3


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("counts,M,N", [([896, 896, 896, 896], 256, 512), ([871, 925, 903, 885], 512, 256), ([0, 1001, 3, 517], 256, 256), ([5, 0, 70, 0], 256, 256),
                                        ([64, 128, 129, 1], 256, 256)])
def test_pptr_wgrad_expert_segments(force_pptr, counts, M, N, out_dtype):
    """Per-expert weight gradients over arbitrary (unpadded, empty, shorter than one K-step pair) row ranges: rows past a range contribute exactly
    nothing even when the neighbouring rows are poisoned."""
    E = len(counts); R = sum(counts)
    g = torch.Generator().manual_seed(R + M)
    A = torch.randn(R, M, generator=g).to(torch.bfloat16).cuda()
    X = torch.randn(R, N, generator=g).to(torch.bfloat16).cuda()
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    Cc = torch.full((E, M, N), float("nan"), dtype=out_dtype, device="cuda")
    d = _desc(out_dtype=L.MODE_BF16 if out_dtype == torch.bfloat16 else L.MODE_F32, M=M, N=N, K=R, A=p(A), lda=M, W=p(X), ldw=N, C=p(Cc), ldc=N,
              k_group_offsets=p(off), num_k_groups=E, c_group_stride=M * N, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    _run(d, "pptr wgrad")
    tol = 1e-5 if out_dtype == torch.float32 else 4e-3
    o = 0
    for e, c in enumerate(counts):
        if c == 0:
            assert float(Cc[e].float().abs().max()) == 0.0
        else:
            assert rel(Cc[e].float(), A[o:o + c].float().t() @ X[o:o + c].float()) < tol, e
        o += c
    if out_dtype == torch.float32:
        force_pptr.mode_set_option(b"gemm_tr_cfg", 7)
        C2 = torch.full_like(Cc, float("nan"))
        d.C = p(C2)
        _run(d, "ring wgrad")
        assert torch.equal(C2.view(torch.int32), Cc.view(torch.int32))


@pytest.mark.parametrize("R,M,N", [(1792, 1024, 1024), (100, 256, 256), (1793, 512, 768)])
def test_pptr_wgrad_plain(force_pptr, R, M, N):
    g = torch.Generator().manual_seed(R + M)
    A = torch.randn(R, M, generator=g).to(torch.bfloat16).cuda()
    X = torch.randn(R, N, generator=g).to(torch.bfloat16).cuda()
    Cc = torch.full((M, N), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=M, N=N, K=R, A=p(A), lda=M, W=p(X), ldw=N, C=p(Cc), ldc=N, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    _run(d, "pptr wgrad plain")
    assert rel(Cc, A.float().t() @ X.float()) < 1e-5


def test_pptr_training_shapes_full_size(force_pptr):
    """The four expert GEMMs of one C2 / B = 128 backward layer at their real sizes (ragged multinomial segments), against fp32 torch."""
    D, E = 1024, 4
    counts = [871, 925, 903, 885]; NK = sum(counts)
    g = torch.Generator().manual_seed(3)
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    rn = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).cuda()
    dY, W2, H = rn(NK, D), rn(E, D, 4 * D), rn(NK, 4 * D)
    dH = torch.full((NK, 4 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    d = _desc(out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=p(dY), lda=D, W=p(W2), ldw=4 * D, w_expert_stride=4 * D * D, C=p(dH), ldc=4 * D,
              expert_offsets=p(off), num_experts=E, flags=L.GEMM_W_KN)
    _run(d, "dH")
    dW2 = torch.full((E, D, 4 * D), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=D, N=4 * D, K=NK, A=p(dY), lda=D, W=p(H), ldw=4 * D, C=p(dW2), ldc=4 * D, k_group_offsets=p(off), num_k_groups=E,
              c_group_stride=4 * D * D, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    _run(d, "dW2")
    dP, W1, U = rn(NK, 8 * D), rn(E, 8 * D, D), rn(NK, D)
    dU = torch.full((4, NK, D), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=NK, N=D, K=8 * D, A=p(dP), lda=8 * D, W=p(W1), ldw=D, w_expert_stride=8 * D * D, C=p(dU), ldc=D,
              expert_offsets=p(off), num_experts=E, flags=L.GEMM_W_KN, split_k=4, split_stride=NK * D)
    _run(d, "dU")
    dW1 = torch.full((E, 8 * D, D), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=8 * D, N=D, K=NK, A=p(dP), lda=8 * D, W=p(U), ldw=D, C=p(dW1), ldc=D, k_group_offsets=p(off), num_k_groups=E,
              c_group_stride=8 * D * D, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    _run(d, "dW1")
    o = 0
    for e, c in enumerate(counts):
        s = slice(o, o + c)
        assert rel(dH[s].float(), dY[s].float() @ W2[e].float()) < 4e-3
        assert rel(dW2[e], dY[s].float().t() @ H[s].float()) < 1e-5
        assert rel(dU[:, s].sum(0), dP[s].float() @ W1[e].float()) < 1e-5
        assert rel(dW1[e], dP[s].float().t() @ U[s].float()) < 1e-5
        o += c


@pytest.mark.parametrize("R,Rw,M,cin,taps,groups", [(1000, 1300, 64, 64, 9, 7), (777, 900, 256, 128, 4, 3), (300, 300, 136, 192, 9, 1), (4096, 5000, 128, 64, 1, 16)])
def test_wgrad_w_rows_in_taps(R, Rw, M, cin, taps, groups):
    """ABI 10: one weight-gradient product over `taps` column blocks, tap t reading W's rows through its own index table (a k x k convolution's dW as
    one launch: C[M][t * cin + c] = sum_r A[r][M] * W[idx[t][r]][c]), K cut into groups of arbitrary length."""
    g = torch.Generator().manual_seed(R + cin)
    A = torch.randn(R, M, generator=g).to(torch.bfloat16).cuda()
    X = torch.randn(Rw, cin, generator=g).to(torch.bfloat16).cuda()
    idx = torch.randint(0, Rw, (taps, R), generator=g, dtype=torch.int32)
    idx[torch.rand(taps, R, generator=g) < 0.15] = -1                          # a negative index reads a zero row (out-of-image filter taps)
    idx = idx.cuda()
    cuts = sorted(torch.randint(0, R + 1, (groups - 1,), generator=g).tolist())
    off = torch.tensor([0] + cuts + [R], dtype=torch.int32).cuda()
    Cc = torch.full((groups, M, taps * cin), float("nan"), device="cuda")
    d = _desc(out_dtype=L.MODE_F32, M=M, N=taps * cin, K=R, A=p(A), lda=M, W=p(X), ldw=cin, C=p(Cc), ldc=taps * cin, k_group_offsets=p(off), num_k_groups=groups,
              c_group_stride=M * taps * cin, w_rows=p(idx), w_tap_cols=cin, w_rows_tap_stride=R, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    L.check(L.load().mode_gemm(C.byref(d), stream()), "wgrad taps")
    tot = Cc.sum(0)
    for t in range(taps):
        Xt = X[idx[t].clamp_min(0).long()].float() * (idx[t] >= 0).float()[:, None]
        assert rel(tot[:, t * cin:(t + 1) * cin], A.float().t() @ Xt) < 1e-5, t
    bad = _desc(out_dtype=L.MODE_F32, M=M, N=taps * cin, K=R, A=p(A), lda=M, W=p(X), ldw=cin, C=p(Cc), ldc=taps * cin, w_rows=p(idx), w_tap_cols=40,
                w_rows_tap_stride=R, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    assert L.load().mode_gemm(C.byref(bad), stream()) != 0                      # taps must be multiples of 64 columns


@pytest.mark.parametrize("n,cin,cout,H,W_,k,stride,pad", [(3, 64, 64, 12, 9, 3, 1, 1), (2, 128, 256, 15, 15, 3, 2, 1), (4, 256, 128, 8, 8, 1, 2, 0), (1, 64, 192, 5, 7, 3, 1, 1),
                                                           (2, 512, 512, 7, 7, 3, 1, 1), (65, 64, 64, 16, 16, 3, 1, 1)])
def test_conv_forward_and_data_gradient_as_gemm_with_a_rows_in_taps(n, cin, cout, H, W_, k, stride, pad):
    """ABI 10 `a_tap_cols`: a k x k convolution on channels_last bf16 data as ONE mode_gemm call - forward (K-contiguous channels_last weight) and data gradient
    (MODE_GEMM_W_KN on the same weight memory) - against torch's conv2d / its autograd in fp32 on the same bf16 values.  Index tables: -1 = outside the image
    (forward) / no output pixel read this input through this tap (data gradient, strides)."""
    from mode_diffusion_policy_amd import perceptual_encoders as E
    torch.manual_seed(n * 7 + cin + cout + k)
    x = torch.randn(n, cin, H, W_, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * (k * k * cin) ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = E._conv_fwd_taps(x, w, (stride, stride), (pad, pad))
    xr = x.float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, w.float(), None, stride, pad)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert rel(y.float(), yr) < 4e-3
    dy = torch.randn_like(y)
    dx = E._conv_dgrad_taps(dy, w, x.shape, (stride, stride), (pad, pad))
    yr.backward(dy.float())
    assert dx.shape == x.shape and rel(dx.float(), xr.grad) < 4e-3
    torch.cuda.synchronize()
