"""GPU parity tests of the individual HIP kernels, called through the C-ABI, against the oracle / plain fp32 torch math
on the same seeded inputs.  Integer outputs bit-exact; bf16 kernels within bf16 rounding of an fp32 reference fed the same
bf16-rounded inputs; fp32 kernels <= 1e-5."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from mode_diffusion_policy_amd import _lib as L  # noqa: E402
from oracle import mode_oracle as O  # noqa: E402

import hip_helpers as H  # noqa: E402


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


# ----------------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1792, 3072, 1024), (100, 192, 64), (257, 1024, 256), (14, 64, 128), (1, 128, 64)])
@pytest.mark.parametrize("cfg", [0, 1, 4, 6, 8, 13, 14, 17, 18])
def test_gemm_bf16_plain_bias(M, N, K, cfg):
    # asymmetric operands: a transposed/permuted C-write cannot pass
    A = rnd(M, K, seed=1).to(torch.bfloat16); W = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
    b = rnd(N, seed=3)
    ref = A.float() @ W.float().t() + b
    L.load().mode_set_option(b"gemm_cfg", cfg)       # every tile geometry / ring depth (0 = auto heuristic) must agree
    try:
        out = H.gemm(A.to(dev()), W.to(dev()), L.EPI_BIAS, bias=b.to(dev()), out_dtype=torch.float32)
        assert rel(out, ref) < 2e-3
        out16 = H.gemm(A.to(dev()), W.to(dev()), L.EPI_BIAS, bias=b.to(dev()), out_dtype=torch.bfloat16)
        assert rel(out16.float(), ref) < 6e-3
    finally:
        L.load().mode_set_option(b"gemm_cfg", 0)


def test_gemm_bf16_identity_asymmetric():
    """A = I with an asymmetric W catches row<->col swaps in the MFMA C layout (guide: 'A=I-check with ASYMMETRIC B')."""
    K = 128
    A = torch.eye(K).to(torch.bfloat16)
    W = (torch.arange(256 * K).reshape(256, K) % 251).float().to(torch.bfloat16)
    out = H.gemm(A.to(dev()), W.to(dev()), out_dtype=torch.float32)
    assert torch.equal(out.cpu(), W.float().t().contiguous())


@pytest.mark.parametrize("cfg", [0, 1, 4, 6, 8, 13, 17, 18])
@pytest.mark.parametrize("M,D", [(300, 256), (3584, 1024)])
def test_gemm_bf16_swiglu_residual(M, D, cfg):
    L.load().mode_set_option(b"gemm_cfg", cfg)
    try:
        _swiglu_residual(M, D)
    finally:
        L.load().mode_set_option(b"gemm_cfg", 0)


def _swiglu_residual(M, D):
    A = rnd(M, D, seed=4).to(torch.bfloat16); W1 = rnd(8 * D, D, seed=5, scale=D ** -0.5).to(torch.bfloat16); b1 = rnd(8 * D, seed=6, scale=0.1)
    h = A.float() @ W1.float().t() + b1
    ref = h[:, : 4 * D] * torch.nn.functional.silu(h[:, 4 * D:])
    out = H.gemm(A.to(dev()), W1.to(dev()), L.EPI_SWIGLU, bias=b1.to(dev()), out_dtype=torch.float32)
    assert out.shape == (M, 4 * D) and rel(out, ref) < 3e-3
    Wo = rnd(D, D, seed=7, scale=D ** -0.5).to(torch.bfloat16); r = rnd(M, D, seed=8)
    out2 = H.gemm(A.to(dev()), Wo.to(dev()), L.EPI_RESIDUAL, resid=r.to(dev()), out_dtype=torch.float32)
    assert rel(out2, A.float() @ Wo.float().t() + r) < 2e-3


@pytest.mark.parametrize("cfg", [0, 1, 4, 6, 8, 13, 17, 18])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N_tok,E,k,D", [(70, 4, 2, 64), (1792, 4, 2, 256), (112, 2, 1, 256), (5, 4, 2, 64), (900, 4, 2, 128)])
def test_grouped_gather_gemm(dtype, N_tok, E, k, D, cfg):
    L.load().mode_set_option(b"gemm_cfg", cfg)
    try:
        _grouped_gather_gemm(dtype, N_tok, E, k, D)
    finally:
        L.load().mode_set_option(b"gemm_cfg", 0)


def _grouped_gather_gemm(dtype, N_tok, E, k, D):
    """dispatch meta -> gathered grouped SwiGLU GEMM -> grouped GEMM, vs the oracle's per-expert loop (modedit.py:561-566)."""
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    probs = torch.rand(N_tok, E, generator=g)
    idx = torch.sort(probs, dim=-1, descending=True, stable=True).indices[:, :k].contiguous()
    w = probs.gather(1, idx); w = w / w.sum(-1, keepdim=True)
    meta = H.dispatch_meta(idx.int().to(dev()), w.to(dev()), 1, N_tok, E)
    counts, perm, slot = O.dispatch_permutation(idx, E)
    assert torch.equal(meta["counts"].cpu().long(), counts)
    assert torch.equal(meta["perm"].cpu().long(), perm)                       # bit-exact permutation
    u = rnd(N_tok, D, seed=12).to(dtype)
    W1 = rnd(E, 8 * D, D, seed=13, scale=D ** -0.5).to(dtype); b1 = rnd(E, 8 * D, seed=14, scale=0.1)
    W2 = rnd(E, D, 4 * D, seed=15, scale=(4 * D) ** -0.5).to(dtype)
    Hs = H.gemm(u.to(dev()), W1.to(dev()), L.EPI_SWIGLU, bias=b1.to(dev()), out_dtype=dtype, a_rows=meta["perm"], offsets=meta["offsets"],
                num_experts=E, M=N_tok * k, w_estride=8 * D * D, b_estride=8 * D)
    Y = H.gemm(Hs, W2.to(dev()), L.EPI_NONE, out_dtype=torch.float32, offsets=meta["offsets"], num_experts=E, M=N_tok * k,
               w_estride=4 * D * D)
    # reference in sorted-row order
    off = 0; Href = torch.zeros(N_tok * k, 4 * D); Yref = torch.zeros(N_tok * k, D)
    for e in range(E):
        n = int(counts[e]); rows = perm[off: off + n]
        h = u[rows].float() @ W1[e].float().t() + b1[e]
        hh = h[:, : 4 * D] * torch.nn.functional.silu(h[:, 4 * D:])
        Href[off: off + n] = hh
        hq = hh.to(dtype).float()                                            # the kernel chain rounds H to the compute dtype
        Yref[off: off + n] = hq @ W2[e].float().t()
        off += n
    tol = 4e-3 if dtype == torch.bfloat16 else 1e-5
    assert rel(Hs.float(), Href) < tol * 2
    assert rel(Y, Yref) < tol * 2
    # combine + norm (ascending expert order, residual from u)
    gain = 1 + 0.1 * rnd(D, seed=16); cond = rnd(3, D, seed=17)
    rpc = (N_tok + 2) // 3
    u32 = u.float()
    xn, hh = H.combine_norm(u32.to(dev()), Y, meta["pos"], meta["posw"], k, gain.to(dev()), cond.to(dev()), rpc, h_dtype=dtype)
    nxt = torch.zeros(N_tok, D); off = 0
    Yc = Y.cpu()
    for e in range(E):
        n = int(counts[e]); rows = perm[off: off + n]; sl = slot[off: off + n]
        nxt[rows] += w[rows, sl].unsqueeze(-1) * Yc[off: off + n]
        off += n
    xref = u32 + nxt
    assert rel(xn, xref) < 1e-6
    href = O.rmsnorm(xref, gain) + cond[torch.arange(N_tok) // rpc]
    assert rel(hh.float(), href) < (5e-3 if dtype == torch.bfloat16 else 1e-6)


@pytest.mark.parametrize("rows_per_expert", [448, 336, 500])
def test_one_round_down_projection_geometry_is_bit_identical(rows_per_expert):
    """The K-sliced, grouped expert down-projection of 17 .. 36 environments (two active experts, 4 slices of K = 4096) fills ONE round of the part with
    128 x 128 tiles and takes them on a 3-slot ring (geometry 20, "gemm_dn_ring3"): same k-ordered fp32 chain as the 128 x 64 tiles it replaces -
    bit-identical slabs (a sample's result must not depend on the batch it shares), correct against fp64."""
    import ctypes as C
    E, D, S = 2, 1024, 4
    NK = E * rows_per_expert
    hid = rnd(NK, 4 * D, seed=31).to(torch.bfloat16).to(dev()); W2 = rnd(E, D, 4 * D, seed=32, scale=(4 * D) ** -0.5).to(torch.bfloat16).to(dev())
    offs = torch.tensor([0, rows_per_expert, NK], dtype=torch.int32, device=dev())
    lib = L.load(); out = {}
    for tag, opt in (("ring3", 1), ("tiles64", 0)):
        lib.mode_set_option(b"gemm_dn_ring3", opt)
        try:
            Y = torch.full((S, NK, D), float("nan"), dtype=torch.bfloat16, device=dev())
            d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=NK, N=D, K=4 * D, A=hid.data_ptr(), lda=4 * D, W=W2.data_ptr(),
                               ldw=4 * D, w_expert_stride=4 * D * D, C=Y.data_ptr(), ldc=D, expert_offsets=offs.data_ptr(), num_experts=E, split_k=S,
                               split_stride=NK * D)
            L.check(lib.mode_gemm(C.byref(d), H.stream()))
            out[tag] = Y
        finally:
            lib.mode_set_option(b"gemm_dn_ring3", 1)
    assert torch.equal(out["ring3"], out["tiles64"])
    want = torch.cat([hid[e * rows_per_expert:(e + 1) * rows_per_expert].double().cpu() @ W2[e].double().cpu().t() for e in range(E)])
    assert rel(out["ring3"].float().sum(0), want.float()) < 6e-3


@pytest.mark.parametrize("cfg", [0, 4, 17, 18, 20])
@pytest.mark.parametrize("S", [2, 4])
def test_gemm_bf16_split_k(cfg, S):
    """split-K: slice z writes its partial sums to slab z; the slabs add up to the un-split product."""
    import ctypes as C
    M, N, K = 515, 256, 1024
    A = rnd(M, K, seed=71).to(torch.bfloat16).to(dev()); W = rnd(N, K, seed=72, scale=K ** -0.5).to(torch.bfloat16).to(dev())
    lib = L.load(); lib.mode_set_option(b"gemm_cfg", cfg)
    try:
        slabs = torch.full((S, M, N), float("nan"), device=dev())
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=M, N=N, K=K, A=A.data_ptr(), lda=K, W=W.data_ptr(),
                           ldw=K, C=slabs.data_ptr(), ldc=N, split_k=S, split_stride=M * N)
        L.check(lib.mode_gemm(C.byref(d), H.stream()))
        ref = A.float().cpu() @ W.float().cpu().t()
        assert rel(slabs.sum(0), ref) < 2e-3
        part = A.float().cpu()[:, : K // S] @ W.float().cpu()[:, : K // S].t()
        assert rel(slabs[0], part) < 2e-3
    finally:
        lib.mode_set_option(b"gemm_cfg", 0)


@pytest.mark.parametrize("M,N,K,epi", [(37, 4, 512, L.EPI_BIAS), (128, 512, 256, L.EPI_BIAS_GELU), (10, 2, 512, L.EPI_BIAS),
                                       (256, 1024, 2048, L.EPI_NONE), (65, 130, 30, L.EPI_NONE), (16, 64, 7, L.EPI_NONE),
                                       (10, 2048, 1024, L.EPI_BIAS_GELU), (10, 4, 2048, L.EPI_BIAS), (1, 1024, 1024, L.EPI_NONE), (16, 6, 100, L.EPI_BIAS)])
def test_gemm_f32(M, N, K, epi):
    A = rnd(M, K, seed=21); W = rnd(N, K, seed=22, scale=K ** -0.5); b = rnd(N, seed=23)
    ref = A @ W.t()
    if epi != L.EPI_NONE:
        ref = ref + b
    if epi == L.EPI_BIAS_GELU:
        ref = torch.nn.functional.gelu(ref)
    for flags in (0, L.GEMM_SKINNY_OK):          # MFMA k-ordered chain, and the skinny weight-streaming GEMV (taken when M <= 16)
        out = H.gemm(A.to(dev()), W.to(dev()), epi, bias=b.to(dev()) if epi != L.EPI_NONE else None, flags=flags)
        assert rel(out, ref) < 1e-5


@pytest.mark.parametrize("K", [64, 1000, 1024, 2112])
def test_gemm_f32_result_does_not_depend_on_tile_or_operand_layout(K):
    """gemm_f32.hip picks 32 x 32 tiles (128-k LDS fills) for products with few tiles and 64 x 64 (64-k fills) otherwise, and takes either operand as [rows][K] or
    [K][cols]: an output element is the same k-ascending v_mfma_f32_16x16x4_f32 chain in every case - the same rows of A inside a larger product, and the same
    product through the transposed-operand layouts (K-grouped: partial slabs), must give the same BITS.  Also K % 16 != 0 tails and a K that ends inside a fill."""
    import ctypes as C
    from hip_helpers import p, stream
    lib = L.load()
    N = 256
    A = rnd(600, K, seed=5).to(dev()); W = rnd(N, K, seed=6, scale=K ** -0.5).to(dev())

    def run(M, a=None, w=None, flags=0, **kw):
        a = A if a is None else a; w = W if w is None else w
        out = torch.full((kw.get("num_k_groups", 1), M, N), float("nan"), device=dev())
        d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=M, N=N, K=K, A=p(a), lda=a.stride(0), W=p(w), ldw=w.stride(0), C=p(out), ldc=N,
                           flags=flags, **kw)
        L.check(lib.mode_gemm(C.byref(d), stream()), "gemm")
        torch.cuda.synchronize()
        return out
    small, large = run(64)[0], run(600)[0]                       # 8 x 8 tiles of 32 x 32 against 10 x 4 tiles of 64 x 64
    ref = A.double() @ W.double().t()
    assert rel(large, ref.float()) < 1e-5
    assert torch.equal(small.view(torch.int32), large[:64].view(torch.int32))
    if K % 4 == 0:
        At = A[:64].t().contiguous(); Wt = W.t().contiguous()     # [K][M], [K][N]
        for flags, a, w in ((L.GEMM_W_KN, A[:64].contiguous(), Wt), (L.GEMM_A_KM | L.GEMM_W_KN, At, Wt), (L.GEMM_A_KM, At, W)):
            assert torch.equal(run(64, a, w, flags)[0].view(torch.int32), small.view(torch.int32)), flags
        # K-groups: the slab of group g is the chain over its own k range - equal to the plain product of that column range
        ng = 4 if K % 16 == 0 else 0
        if ng:
            off = torch.arange(0, K + 1, K // ng, dtype=torch.int32, device=dev())
            slabs = run(64, At, Wt, L.GEMM_A_KM | L.GEMM_W_KN, k_group_offsets=p(off), num_k_groups=ng, c_group_stride=64 * N)
            for g in range(ng):
                k0, k1 = g * (K // ng), (g + 1) * (K // ng)
                a_g = A[:64, k0:k1].contiguous(); w_g = W[:, k0:k1].contiguous()
                out = torch.full((64, N), float("nan"), device=dev())
                d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=64, N=N, K=k1 - k0, A=p(a_g), lda=k1 - k0, W=p(w_g), ldw=k1 - k0,
                                   C=p(out), ldc=N)
                L.check(lib.mode_gemm(C.byref(d), stream()), "gemm")
                torch.cuda.synchronize()
                assert torch.equal(slabs[g].view(torch.int32), out.view(torch.int32)), g


# ------------------------------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("rows,D", [(7, 64), (1792, 1024), (112, 256)])
def test_rmsnorm_cond(rows, D):
    x = rnd(rows, D, seed=31, scale=3.0); g = 1 + 0.1 * rnd(D, seed=32); c = rnd((rows + 13) // 14, D, seed=33)
    ref = O.rmsnorm(x, g) + c[torch.arange(rows) // 14]
    y32, ylp = H.rmsnorm(x.to(dev()), g.to(dev()), c.to(dev()), 14)
    assert rel(y32, ref) < 1e-6 and rel(ylp.float(), ref) < 4e-3
    y32b, _ = H.rmsnorm(x.to(dev()), g.to(dev()), None, 1)
    assert rel(y32b, O.rmsnorm(x, g)) < 1e-6
    z = torch.zeros(4, D)                                         # eps clamp: all-zero row stays finite (norm.clamp(min=eps))
    y0, _ = H.rmsnorm(z.to(dev()), g.to(dev()), None, 1)
    assert torch.isfinite(y0).all() and float(y0.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,Hh,hd", [(3, 14, 4, 32), (8, 14, 8, 128), (128, 14, 8, 128), (2, 13, 2, 64), (1, 16, 1, 32), (6, 14, 4, 16), (2, 14, 2, 48)])
def test_attention(dtype, B, T, Hh, hd):
    D = Hh * hd
    qkv = rnd(B * T, 3 * D, seed=41).to(dtype)
    qg = 1 + 0.1 * rnd(hd, seed=42); kg = 1 + 0.1 * rnd(hd, seed=43)
    y = H.attn(qkv.to(dev()), qg.to(dev()), kg.to(dev()), B, T, Hh, hd)
    q, k, v = (t.float().view(B, T, Hh, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    q = O.rmsnorm(q, qg); k = O.rmsnorm(k, kg)
    att = (q @ k.transpose(-2, -1)) / math.sqrt(hd)
    att = att.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float("-inf")).softmax(-1)
    ref = (att @ v).transpose(1, 2).reshape(B * T, D)
    assert not torch.isnan(y.float()).any()
    assert rel(y.float(), ref) < (1.2e-2 if dtype == torch.bfloat16 else 1e-5)
    # causality property: token 0 attends only to itself -> y[:,0] == v[:,0]
    y0 = y.float().cpu().view(B, T, D)[:, 0]
    v0 = qkv.float().view(B, T, 3 * D)[:, 0, 2 * D:]
    assert rel(y0, v0) < (8e-3 if dtype == torch.bfloat16 else 1e-6)


@pytest.mark.parametrize("B,T,Hh", [(128, 14, 8), (1, 14, 8), (5, 14, 2), (37, 16, 4), (9, 5, 1), (64, 11, 8), (130, 14, 8)])
def test_fused_qkv_attention_is_bit_identical_to_gemm_plus_attention(B, T, Hh):
    """mode_qkv_attn_fwd (one launch: QKV projection + qk-RMSNorm + causal attention, q | k | v never in HBM) against the two kernels it replaces:
    every output bit equal (same k-ordered MFMA chain, same rounding points, one shared attention body) - whole and partial sample groups, ragged
    last group, 1-16 tokens per sample, and against the fp32 torch reference at the attention tolerance."""
    hd = 128
    D = Hh * hd
    h = rnd(B * T, D, seed=61).to(torch.bfloat16).to(dev())
    w = rnd(3 * D, D, seed=62, scale=D ** -0.5).to(torch.bfloat16).to(dev())
    b = (0.1 * rnd(3 * D, seed=63)).to(dev())
    qg = (1 + 0.1 * rnd(hd, seed=64)).to(dev()); kg = (1 + 0.1 * rnd(hd, seed=65)).to(dev())
    rc, y = H.qkv_attn(h, w, b, qg, kg, B, T, Hh)
    assert rc == 0
    lib = L.load()
    try:                                                                        # the other three geometries of the kernel: same bits
        for waves, w3 in ((8, 0), (4, 1), (4, 0)):
            assert lib.mode_set_option(b"qkv_attn_waves", waves) == 0 and lib.mode_set_option(b"qkv_attn_w3", w3) == 0
            rc2, y_two = H.qkv_attn(h, w, b, qg, kg, B, T, Hh)
            assert rc2 == 0 and torch.equal(y, y_two), (waves, w3)
    finally:
        lib.mode_set_option(b"qkv_attn_waves", 8); lib.mode_set_option(b"qkv_attn_w3", 1)
    qkv = H.gemm(h, w, epilogue=L.EPI_BIAS, bias=b)
    y2 = H.attn(qkv, qg, kg, B, T, Hh, hd)
    assert not torch.isnan(y.float()).any()
    assert torch.equal(y, y2)
    q, k, v = (t.float().cpu().view(B, T, Hh, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    q = O.rmsnorm(q, qg.cpu()); k = O.rmsnorm(k, kg.cpu())
    att = (q @ k.transpose(-2, -1)) / math.sqrt(hd)
    att = att.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float("-inf")).softmax(-1)
    assert rel(y.float(), (att @ v).transpose(1, 2).reshape(B * T, D)) < 1.2e-2


def test_fused_qkv_attention_refuses_what_it_does_not_take():
    h = rnd(28, 256, seed=66).to(torch.bfloat16).to(dev()); w = rnd(768, 256, seed=67).to(torch.bfloat16).to(dev())
    b = rnd(768, seed=68).to(dev()); g = torch.ones(64, device=dev())
    assert H.qkv_attn(h, w, b, g, g, 2, 14, 4)[0] == -2                          # head_dim 64: the two kernels
    assert H.qkv_attn(h.float(), w.float(), b, g, g, 2, 14, 4)[0] == -2          # fp32 parity mode


# ---------------------------------------------------------------------------------------------------------------- routing
@pytest.mark.parametrize("R,E,k", [(1, 4, 2), (128, 4, 2), (8, 2, 1), (77, 8, 3), (5, 3, 2)])
def test_route_topk_bit_exact(R, E, k):
    logits = rnd(R, E, seed=51, scale=2.0)
    sh, pr, idx, w = H.route_topk(logits.to(dev()), k, True)
    lg = logits - logits.max(-1, keepdim=True).values
    probs = torch.softmax(lg, -1).clamp(1e-9, 1 - 1e-9)
    ridx, rw = O.topk_route(probs, k, True)
    assert torch.equal(sh.cpu(), lg)
    assert torch.equal(idx.cpu().long(), ridx)                    # bit-exact integers
    assert rel(pr, probs) < 1e-6 and rel(w, rw) < 1e-6
    # exact tie -> lower expert id first (documented tie rule)
    t = torch.zeros(2, E); sh2, pr2, idx2, _ = H.route_topk(t.to(dev()), k, True)
    assert idx2.cpu().tolist() == [list(range(k))] * 2


@pytest.mark.parametrize("R,tpr,E,k", [(8, 14, 4, 2), (1, 1792, 4, 2), (112, 1, 2, 1), (1792, 1, 4, 2), (3000, 1, 4, 2)])
def test_dispatch_meta_bit_exact(R, tpr, E, k):
    lib = L.load()
    g = torch.Generator().manual_seed(61)
    probs = torch.rand(R, E, generator=g)
    if R == 1:
        probs = torch.tensor([[0.1, 0.5, 0.05, 0.35]])[:, :E]
    idx = torch.multinomial(probs, k, replacement=False, generator=g)          # unsorted slots, like training (modedit.py:390)
    w = probs.gather(1, idx)
    N = R * tpr
    meta = H.dispatch_meta(idx.int().to(dev()), w.to(dev()), tpr, N, E)
    tok_idx = idx.repeat_interleave(tpr, 0)
    counts, perm, slot = O.dispatch_permutation(tok_idx, E)
    assert torch.equal(meta["counts"].cpu().long(), counts)
    assert torch.equal(meta["offsets"].cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]))
    assert torch.equal(meta["perm"].cpu().long(), perm)
    pos = meta["pos"].cpu().long().view(N, k); posw = meta["posw"].cpu().view(N, k)
    # pos[t, j] is the sorted row of token t's j-th expert in ASCENDING expert order
    srt = torch.sort(tok_idx, dim=-1)
    wtok = w.repeat_interleave(tpr, 0).gather(1, srt.indices)
    assert torch.equal(perm[pos.reshape(-1)], torch.arange(N).repeat_interleave(k))
    assert torch.equal(posw, wtok)


# ------------------------------------------------------------------------------------------------- fused ln_2 (c_proj -> experts -> combine)
@pytest.mark.parametrize("cfg", [0, 1, 4, 6, 13, 17, 18])
@pytest.mark.parametrize("N_tok,D,E,k", [(70, 128, 4, 2), (1792, 256, 4, 2), (1792, 1024, 4, 2), (37, 64, 2, 1)])
def test_fused_ln2_chain_matches_separate_kernels(N_tok, D, E, k, cfg):
    """MODE_EPI_RESIDUAL_NORM producer + MODE_EPI_SWIGLU(row_ss) consumer + combine(u_ss) against the three-kernel formulation
    (c_proj+residual -> rmsnorm -> up-projection -> combine): the same mathematics with ln_2's division moved behind the GEMM."""
    import ctypes as C
    lib = L.load()
    p, st = H.p, H.stream()
    bf = torch.bfloat16
    ya = rnd(N_tok, D, seed=1).to(bf).to(dev()); wo = rnd(D, D, seed=2, scale=D ** -0.5).to(bf).to(dev())
    x0 = rnd(N_tok, D, seed=3).to(dev()); g2 = (1.0 + 0.2 * rnd(D, seed=4)).to(dev())
    W1 = rnd(E, 8 * D, D, seed=5, scale=D ** -0.5).to(bf).to(dev()); b1 = rnd(E, 8 * D, seed=6, scale=0.1).to(dev())
    logits = rnd(N_tok, E, seed=7).to(dev())
    _, _, idx, w = H.route_topk(logits, k)
    meta = H.dispatch_meta(idx, w, 1, N_tok, E)
    NK = N_tok * k
    eps = 1e-6
    lib.mode_set_option(b"gemm_cfg", cfg)
    lib.mode_set_option(b"gemm_skinny_rows", 0)          # both formulations on the tiled kernel: the residual stream must then be bit-identical
    try:
        # --- separate kernels
        x_ref = H.gemm(ya, wo, L.EPI_RESIDUAL, resid=x0, out_dtype=torch.float32)
        xn_ref, h_ref = H.rmsnorm(x_ref, g2, eps=eps)
        hid_ref = H.gemm(h_ref, W1, L.EPI_SWIGLU, bias=b1, out_dtype=bf, a_rows=meta["perm"], offsets=meta["offsets"], num_experts=E, M=NK,
                         w_estride=8 * D * D, b_estride=8 * D)
        # --- fused
        x = torch.full((N_tok, D), float("nan"), device=dev()); xg = torch.zeros(N_tok, D, dtype=bf, device=dev())
        ss = torch.full((N_tok, D // 64), float("nan"), device=dev())
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_RESIDUAL_NORM, out_dtype=L.MODE_F32, M=N_tok, N=D, K=D, A=p(ya), lda=D, W=p(wo), ldw=D,
                           resid=p(x0), ldr=D, C=p(x), ldc=D, C2=p(xg), ldc2=D, gain=p(g2), row_ss_out=p(ss))
        L.check(lib.mode_gemm(C.byref(d), st), "c_proj fused")
        assert torch.equal(x, x_ref)                                                   # the residual stream itself is unchanged
        assert rel(ss.sum(1), x_ref.double().pow(2).sum(1).float()) < 1e-6
        assert rel(xg.float(), x_ref * g2) < 4e-3
        hid = torch.full((NK, 4 * D), float("nan"), dtype=bf, device=dev())
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_SWIGLU, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=p(xg), lda=D, W=p(W1), ldw=D,
                           w_expert_stride=8 * D * D, bias=p(b1), bias_expert_stride=8 * D, C=p(hid), ldc=4 * D, a_rows=p(meta["perm"]),
                           expert_offsets=p(meta["offsets"]), num_experts=E, row_ss=p(ss), row_ss_n=D // 64, row_eps=eps)
        L.check(lib.mode_gemm(C.byref(d), st), "up-projection fused")
        assert rel(hid.float(), hid_ref.float()) < 8e-3                                 # bf16 rounding of x*g instead of x*g/n
        # exact check of the consumer's arithmetic against fp32 torch fed the same bf16 operand
        nrm = (x_ref.double().pow(2).sum(1).sqrt() * D ** -0.5).clamp_min(eps)
        pre = torch.zeros(NK, 8 * D, dtype=torch.float64)
        perm = meta["perm"].cpu().long(); offs = meta["offsets"].cpu().long()
        for e in range(E):
            rows = perm[offs[e]:offs[e + 1]]
            pre[offs[e]:offs[e + 1]] = (xg[rows].double().cpu() @ W1[e].double().cpu().t()) / nrm.cpu()[rows, None] + b1[e].double().cpu()
        want = pre[:, :4 * D] * torch.nn.functional.silu(pre[:, 4 * D:])
        assert rel(hid.float(), want.float()) < 4e-3
    finally:
        lib.mode_set_option(b"gemm_cfg", 0)
        lib.mode_set_option(b"gemm_skinny_rows", 32)
    # --- combine: un-normalised u + partial sums + gain == normalised u
    Y = rnd(NK, D, seed=8).to(bf).to(dev()); g1 = (1.0 + 0.1 * rnd(D, seed=9)).to(dev()); cond = rnd(N_tok, D, seed=10).to(dev())
    xn_a, h_a = H.combine_norm(xn_ref, Y, meta["pos"], meta["posw"], k, g1, cond, 1, eps=eps)
    xn_b = torch.empty_like(x); h_b = torch.empty(N_tok, D, dtype=bf, device=dev())
    L.check(lib.mode_moe_combine_norm_fused_fwd(p(x), p(ss), D // 64, p(g2), p(Y), L.MODE_BF16, 1, 0, p(meta["pos"]), p(meta["posw"]), N_tok, D, k,
                                                p(g1), p(cond), 1, eps, p(xn_b), p(h_b), L.MODE_BF16, st), "combine fused")
    assert rel(xn_b, xn_a) < 1e-6 and rel(h_b.float(), h_a.float()) < 4e-3



@pytest.mark.parametrize("N_tok,D,E,k,S", [(14, 1024, 4, 2, 4), (28, 1024, 4, 2, 2), (14, 256, 4, 2, 4), (28, 256, 2, 1, 1), (9, 128, 4, 2, 2)])
def test_small_rows_fused_ln2_chain(N_tok, D, E, k, S):
    """The small-batch chain (B <= 2 environments, MODE_GEMM_SMALL_ROWS): every GEMM is the weight streamer — also the grouped ones, whose
    segments have at most N_tok rows although M = N_tok*k exceeds "gemm_skinny_rows" — and the fused ln_2 works on 16-column partials:
    c_proj (RESIDUAL_NORM, [M, D/16] sums of squares) -> up-projection (row scale from D/16 partials) -> K-sliced down-projection ->
    one-workgroup-per-row combine (u_ss with D/16 partials, S slabs).  Checked against fp64 torch fed the same bf16 operands."""
    import ctypes as C
    lib = L.load()
    p, st = H.p, H.stream()
    bf = torch.bfloat16
    ya = rnd(N_tok, D, seed=1).to(bf).to(dev()); wo = rnd(D, D, seed=2, scale=D ** -0.5).to(bf).to(dev())
    x0 = rnd(N_tok, D, seed=3).to(dev()); g2 = (1.0 + 0.2 * rnd(D, seed=4)).to(dev())
    W1 = rnd(E, 8 * D, D, seed=5, scale=D ** -0.5).to(bf).to(dev()); b1 = rnd(E, 8 * D, seed=6, scale=0.1).to(dev())
    W2 = rnd(E, D, 4 * D, seed=11, scale=(4 * D) ** -0.5).to(bf).to(dev())
    logits = rnd(N_tok, E, seed=7).to(dev())
    _, _, idx, w = H.route_topk(logits, k)
    meta = H.dispatch_meta(idx, w, 1, N_tok, E)
    NK, eps, n16 = N_tok * k, 1e-6, D // 16
    FL = L.GEMM_SMALL_ROWS
    # c_proj + residual + ln_2 producer: the residual stream is bit-identical to the plain RESIDUAL epilogue of the same kernel
    x_ref = H.gemm(ya, wo, L.EPI_RESIDUAL, resid=x0, out_dtype=torch.float32)
    x = torch.full((N_tok, D), float("nan"), device=dev()); xg = torch.zeros(N_tok, D, dtype=bf, device=dev())
    ss = torch.full((N_tok, n16), float("nan"), device=dev())
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_RESIDUAL_NORM, out_dtype=L.MODE_F32, M=N_tok, N=D, K=D, A=p(ya), lda=D, W=p(wo), ldw=D,
                       resid=p(x0), ldr=D, C=p(x), ldc=D, C2=p(xg), ldc2=D, gain=p(g2), row_ss_out=p(ss), flags=FL)
    L.check(lib.mode_gemm(C.byref(d), st), "c_proj small")
    assert torch.equal(x, x_ref)
    assert rel(ss, x_ref.double().pow(2).view(N_tok, n16, 16).sum(-1).float()) < 1e-6
    assert rel(xg.float(), x_ref * g2) < 4e-3
    # up-projection: M = NK rows may exceed gemm_skinny_rows, every expert segment has <= N_tok rows
    hid = torch.full((NK, 4 * D), float("nan"), dtype=bf, device=dev())
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_SWIGLU, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=p(xg), lda=D, W=p(W1), ldw=D,
                       w_expert_stride=8 * D * D, bias=p(b1), bias_expert_stride=8 * D, C=p(hid), ldc=4 * D, a_rows=p(meta["perm"]),
                       expert_offsets=p(meta["offsets"]), num_experts=E, row_ss=p(ss), row_ss_n=n16, row_eps=eps, flags=FL)
    L.check(lib.mode_gemm(C.byref(d), st), "up-projection small")
    nrm = (x_ref.double().pow(2).sum(1).sqrt() * D ** -0.5).clamp_min(eps).cpu()
    perm = meta["perm"].cpu().long(); offs = meta["offsets"].cpu().long()
    pre = torch.zeros(NK, 8 * D, dtype=torch.float64)
    for e in range(E):
        rows = perm[offs[e]:offs[e + 1]]
        pre[offs[e]:offs[e + 1]] = (xg[rows].double().cpu() @ W1[e].double().cpu().t()) / nrm[rows, None] + b1[e].double().cpu()
    want = pre[:, :4 * D] * torch.nn.functional.silu(pre[:, 4 * D:])
    assert rel(hid.float(), want.float()) < 4e-3
    # K-sliced down-projection through the streamer
    Y = torch.full((S, NK, D), float("nan"), dtype=bf, device=dev())
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=NK, N=D, K=4 * D, A=p(hid), lda=4 * D, W=p(W2), ldw=4 * D,
                       w_expert_stride=4 * D * D, C=p(Y), ldc=D, expert_offsets=p(meta["offsets"]), num_experts=E, split_k=S, split_stride=NK * D, flags=FL)
    L.check(lib.mode_gemm(C.byref(d), st), "down-projection small")
    ywant = torch.zeros(NK, D, dtype=torch.float64)
    for e in range(E):
        ywant[offs[e]:offs[e + 1]] = hid[offs[e]:offs[e + 1]].double().cpu() @ W2[e].double().cpu().t()
    assert rel(Y.float().sum(0), ywant.float()) < 6e-3
    # combine (one workgroup per row): normalised residual rebuilt from x, the 16-column partials and the gain; slabs added in slice order
    g1 = (1.0 + 0.1 * rnd(D, seed=9)).to(dev()); cond = rnd(N_tok, D, seed=10).to(dev())
    xn_b = torch.empty_like(x); h_b = torch.empty(N_tok, D, dtype=bf, device=dev())
    L.check(lib.mode_moe_combine_norm_fused_fwd(p(x), p(ss), n16, p(g2), p(Y), L.MODE_BF16, S, NK * D, p(meta["pos"]), p(meta["posw"]), N_tok, D, k,
                                                p(g1), p(cond), 1, eps, p(xn_b), p(h_b), L.MODE_BF16, st), "combine small")
    pos = meta["pos"].cpu().long().view(N_tok, k); posw = meta["posw"].cpu().view(N_tok, k).double()
    ysum = Y.double().cpu().sum(0)
    xn_want = x_ref.double().cpu() / nrm[:, None] * g2.double().cpu() + (posw[:, :, None] * ysum[pos]).sum(1)
    n1 = (xn_want.pow(2).sum(1).sqrt() * D ** -0.5).clamp_min(eps)
    h_want = xn_want / n1[:, None] * g1.double().cpu() + cond.double().cpu()
    assert rel(xn_b, xn_want.float()) < 1e-5 and rel(h_b.float(), h_want.float()) < 4e-3
    # without the streamer the flag is refused instead of silently changing the partials' granularity
    lib.mode_set_option(b"gemm_skinny_rows", 0)
    try:
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_RESIDUAL_NORM, out_dtype=L.MODE_F32, M=N_tok, N=D, K=D, A=p(ya), lda=D, W=p(wo), ldw=D,
                           resid=p(x0), ldr=D, C=p(x), ldc=D, C2=p(xg), ldc2=D, gain=p(g2), row_ss_out=p(ss), flags=FL)
        assert lib.mode_gemm(C.byref(d), st) == -2          # MODE_ERR_UNSUPPORTED
    finally:
        lib.mode_set_option(b"gemm_skinny_rows", 32)



@pytest.mark.parametrize("B,D", [(128, 1024), (1, 1024), (2, 256)])
def test_identity_rows_promise_matches_loaded_gather(B, D):
    """MODE_GEMM_IDENTITY_ROWS: under a uniform-sigma step every sample routes to the same experts, so inside each expert segment the dispatch
    permutation is the ascending identity (checked here on the real dispatch kernel) and the up-projection may compute its gather instead of
    loading it: bit-identical output with and without the flag (persistent ping-pong kernel at B = 128, streaming kernel at B <= 2)."""
    import ctypes as C
    lib = L.load()
    p, st = H.p, H.stream()
    bf = torch.bfloat16
    T, E, k = 14, 4, 2
    N, NK = B * T, B * T * k
    idx = torch.tensor([[3, 1]] * 1, dtype=torch.int32, device=dev()); w = torch.tensor([[0.7, 0.3]], device=dev())
    ml = L.ModeMetaLayout(); lib.mode_moe_meta_layout(N, E, k, C.byref(ml))
    meta = torch.empty(ml.total_words, dtype=torch.int32, device=dev())
    L.check(lib.mode_dit_dispatch(p(idx), p(w), 1, k, 1, N, N, E, k, p(meta), st))       # one routing row shared by all N tokens
    offs = meta[ml.offsets: ml.offsets + E + 1].cpu().long(); perm = meta[ml.perm: ml.perm + NK].cpu().long()
    for e in range(E):
        seg = perm[offs[e]: offs[e + 1]]
        assert seg.numel() in (0, N) and torch.equal(seg, torch.arange(seg.numel()))
    x = rnd(N, D, seed=1).to(bf).to(dev()); W1 = rnd(E, 8 * D, D, seed=2, scale=D ** -0.5).to(bf).to(dev()); b1 = rnd(E, 8 * D, seed=3, scale=0.1).to(dev())
    ss = (torch.rand(N, D // 16, device=dev()) + 0.5) if N <= 32 else (torch.rand(N, D // 64, device=dev()) + 0.5)
    outs = []
    for extra in (0, L.GEMM_IDENTITY_ROWS):
        out = torch.full((NK, 4 * D), float("nan"), dtype=bf, device=dev())
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_SWIGLU, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=p(x), lda=D, W=p(W1), ldw=D,
                           w_expert_stride=8 * D * D, bias=p(b1), bias_expert_stride=8 * D, C=p(out), ldc=4 * D, a_rows=p(meta) + 4 * ml.perm,
                           expert_offsets=p(meta) + 4 * ml.offsets, num_experts=E, row_ss=p(ss), row_ss_n=ss.shape[1], row_eps=1e-6,
                           flags=L.GEMM_UNIFORM_GROUPS | (L.GEMM_SMALL_ROWS if N <= 32 else 0) | extra)
        L.check(lib.mode_gemm(C.byref(d), st), "up-projection")
        outs.append(out)
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


# ------------------------------------------------------------------------------------------------- weight-streaming GEMM for a handful of rows
def _both_paths(fn):
    """fn() with the weight-streaming kernel (default only for M <= 32; forced here up to 128 rows) and with the tiled kernel; returns (skinny, tiled)."""
    lib = L.load()
    lib.mode_set_option(b"gemm_skinny_rows", 128)
    a = fn()
    lib.mode_set_option(b"gemm_skinny_rows", 0)
    try:
        b = fn()
    finally:
        lib.mode_set_option(b"gemm_skinny_rows", 32)
    return a, b


@pytest.mark.parametrize("M,N,K", [(1, 128, 128), (14, 3072, 1024), (28, 64, 256), (56, 1024, 1024), (100, 192, 384), (128, 1024, 4096), (17, 20, 128)])
def test_skinny_gemm_epilogues(M, N, K):
    A = rnd(M, K, seed=1).to(torch.bfloat16).to(dev()); W = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16).to(dev())
    b = rnd(N, seed=3).to(dev()); r = rnd(M, N, seed=4).to(dev())
    ref = A.float() @ W.float().t()
    for epi, kw, want in ((L.EPI_NONE, {}, ref), (L.EPI_BIAS, dict(bias=b), ref + b), (L.EPI_RESIDUAL, dict(resid=r), ref + r),
                          (L.EPI_BIAS_GELU, dict(bias=b), torch.nn.functional.gelu(ref + b))):
        s32, t32 = _both_paths(lambda: H.gemm(A, W, epi, out_dtype=torch.float32, **kw))
        assert rel(s32, want) < 2e-3 and rel(s32, t32) < 1e-5, epi       # same products, another fp32 summation order
        if epi != L.EPI_RESIDUAL:
            s16, _ = _both_paths(lambda: H.gemm(A, W, epi, out_dtype=torch.bfloat16, **kw))
            assert rel(s16.float(), want) < 6e-3, epi
    if N % 8 == 0:                                                       # SwiGLU: W is [2*(N/2), K]
        bb = rnd(N, seed=5, scale=0.1).to(dev())
        s, t = _both_paths(lambda: H.gemm(A, W, L.EPI_SWIGLU, bias=bb, out_dtype=torch.float32))
        h = ref + bb
        assert rel(s, h[:, : N // 2] * torch.nn.functional.silu(h[:, N // 2:])) < 3e-3 and rel(s, t) < 1e-5


@pytest.mark.parametrize("N_tok,E,k,D", [(14, 4, 2, 128), (56, 4, 2, 256), (5, 4, 2, 128), (64, 2, 1, 128), (28, 4, 2, 1024)])
def test_skinny_grouped_gather_and_split_k(N_tok, E, k, D):
    """Grouped + gathered up-projection (SwiGLU) and the K-sliced down-projection through the weight-streaming kernel: against the tiled kernel
    and fp32 torch, with ragged expert segments (some experts empty)."""
    import ctypes as C
    lib = L.load(); p, st = H.p, H.stream()
    bf = torch.bfloat16
    x = rnd(N_tok, D, seed=1).to(bf).to(dev())
    W1 = rnd(E, 8 * D, D, seed=2, scale=D ** -0.5).to(bf).to(dev()); b1 = rnd(E, 8 * D, seed=3, scale=0.1).to(dev())
    W2 = rnd(E, D, 4 * D, seed=4, scale=(4 * D) ** -0.5).to(bf).to(dev())
    logits = rnd(N_tok, E, seed=5).to(dev()); logits[:, 0] -= 100.0          # expert 0 stays empty
    _, _, idx, w = H.route_topk(logits, k)
    meta = H.dispatch_meta(idx, w, 1, N_tok, E)
    NK = N_tok * k
    hid_s, hid_t = _both_paths(lambda: H.gemm(x, W1, L.EPI_SWIGLU, bias=b1, out_dtype=bf, a_rows=meta["perm"], offsets=meta["offsets"], num_experts=E,
                                              M=NK, w_estride=8 * D * D, b_estride=8 * D))
    assert rel(hid_s.float(), hid_t.float()) < 4e-3
    perm = meta["perm"].cpu().long(); offs = meta["offsets"].cpu().long()
    pre = torch.zeros(NK, 8 * D)
    for e in range(E):
        rows = perm[offs[e]:offs[e + 1]]
        pre[offs[e]:offs[e + 1]] = x[rows].float().cpu() @ W1[e].float().cpu().t() + b1[e].cpu()
    assert rel(hid_s.float(), pre[:, :4 * D] * torch.nn.functional.silu(pre[:, 4 * D:])) < 4e-3

    def down(split):
        Y = torch.full((split, NK, D), float("nan"), dtype=bf, device=dev())
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=NK, N=D, K=4 * D, A=p(hid_t), lda=4 * D, W=p(W2), ldw=4 * D,
                           w_expert_stride=4 * D * D, C=p(Y), ldc=D, expert_offsets=p(meta["offsets"]), num_experts=E, split_k=split, split_stride=NK * D)
        L.check(lib.mode_gemm(C.byref(d), st), "down")
        return Y.float().sum(0)
    want = torch.zeros(NK, D)
    for e in range(E):
        want[offs[e]:offs[e + 1]] = hid_t[offs[e]:offs[e + 1]].float().cpu() @ W2[e].float().cpu().t()
    for split in (1, 2):
        s, t = _both_paths(lambda: down(split))
        assert rel(s, want) < 6e-3 and rel(s, t) < 6e-3, split


def test_probe_mfma_burn_runs_and_reports_flops():
    """The measurement entry point bench.py uses for the sustained MFMA rate (csrc/probe.hip): launches, writes finite sums, reports its FLOPs."""
    import ctypes as C
    lib = L.load()
    seed = torch.tensor([7], dtype=torch.int32, device=dev())
    out = torch.full((4 * 512,), float("nan"), device=dev())
    fl = C.c_double(0.0)
    L.check(lib.mode_probe_mfma_burn(seed.data_ptr(), out.data_ptr(), 4, 130, C.byref(fl), H.stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out.abs().max()) > 0
    assert fl.value == 4 * 8 * 130 * 16 * 2.0 * 16 * 16 * 32
    assert lib.mode_probe_mfma_burn(None, None, 4, 10, None, H.stream()) == -1          # MODE_ERR_BAD_ARG


@pytest.mark.parametrize("counts", [(920, 847, 902, 915), (896, 896, 896, 896), (2000, 0, 1500, 84), (1030, 1024, 770, 760), (3584, 0, 0, 0)])
@pytest.mark.parametrize("epi", [L.EPI_BIAS, L.EPI_NONE])
def test_ragged_grouped_gemm_256_row_pingpong_tile(counts, epi):
    """Ragged expert segments (device-side offsets, no uniformity promise) at the training forward's up-projection geometry: the heuristic takes the
    256-row ping-pong tile (gemm_cfg 18).  Bit-identical to the 128x128 ring kernel and to the 224-row tile, correct against fp32 torch - for balanced
    counts, counts above 1024 (a fifth tile), empty experts and everything on one expert (the output is NaN-prefilled: an unwritten element fails the comparison)."""
    E, K, N = 4, 256, 4096
    M = sum(counts)
    torch.manual_seed(M + N)
    A = (torch.randn(M, K) * 0.5).to(torch.bfloat16).to(dev())
    W = (torch.randn(E, N, K) * K ** -0.5).to(torch.bfloat16).to(dev())
    b = torch.randn(E, N).to(dev()) if epi == L.EPI_BIAS else None
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=dev())
    outs = {}
    for cfg in (0, 1, 17, 18):
        L.load().mode_set_option(b"gemm_cfg", cfg)
        try:
            outs[cfg] = H.gemm(A, W, epi, bias=b, out_dtype=torch.bfloat16, offsets=off, num_experts=E, w_estride=N * K, b_estride=N if b is not None else 0)
        finally:
            L.load().mode_set_option(b"gemm_cfg", 0)
    ref = torch.empty(M, N)
    lo = 0
    for e, c in enumerate(counts):
        ref[lo:lo + c] = A[lo:lo + c].float().cpu() @ W[e].float().cpu().t() + (b[e].cpu() if b is not None else 0)
        lo += c
    assert rel(outs[0].float(), ref) < 6e-3
    for cfg in (1, 17, 18):
        assert torch.equal(outs[cfg].view(torch.int16), outs[0].view(torch.int16)), cfg


@pytest.mark.gpu
@pytest.mark.parametrize("counts", [(871, 925, 903, 885), (896, 896, 896, 896), (2000, 0, 1500, 84), (1030, 1024, 770, 760)])
def test_ragged_down_projection_in_k_slices_on_the_256_row_pingpong_tile(counts):
    """The training forward's expert down-projection [NK, 4D] x [D, 4D]^T with ragged segments in FOUR K-slices (bf16 slabs, as the inference chain cuts it):
    the heuristic takes the 256-row ping-pong tile (4 experts x 4 row tiles x 4 column tiles x 4 slices = one tile per CU).  Every slab bit-identical to the
    ring kernels' and to the forced 256-row tile; the slab sum correct against fp32 torch; slabs NaN-prefilled."""
    import ctypes as C
    from hip_helpers import p, stream
    lib = L.load()
    E, K, N, S = 4, 4096, 1024, 4
    M = sum(counts)
    torch.manual_seed(M)
    A = (torch.randn(M, K) * 0.5).to(torch.bfloat16).to(dev())
    W = (torch.randn(E, N, K) * K ** -0.5).to(torch.bfloat16).to(dev())
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=dev())
    outs = {}
    for cfg in (0, 1, 18):
        Y = torch.full((S, M, N), float("nan"), dtype=torch.bfloat16, device=dev())
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=K, w_expert_stride=N * K,
                           C=p(Y), ldc=N, expert_offsets=p(off), num_experts=E, split_k=S, split_stride=M * N)
        lib.mode_set_option(b"gemm_cfg", cfg)
        try:
            L.check(lib.mode_gemm(C.byref(d), stream()), "gemm")
        finally:
            lib.mode_set_option(b"gemm_cfg", 0)
        torch.cuda.synchronize()
        outs[cfg] = Y
    ref = torch.empty(M, N)
    lo = 0
    for e, c in enumerate(counts):
        ref[lo:lo + c] = A[lo:lo + c].float().cpu() @ W[e].float().cpu().t()
        lo += c
    assert rel(outs[0].float().sum(0), ref) < 6e-3
    for cfg in (1, 18):
        assert torch.equal(outs[cfg].view(torch.int16), outs[0].view(torch.int16)), cfg


@pytest.mark.gpu
@pytest.mark.parametrize("M", [33, 100, 448, 512, 513])
def test_residual_norm_epilogue_no_k_loop_kernel_is_bit_identical_to_the_ring(M):
    """MODE_EPI_RESIDUAL_NORM (c_proj + residual + first half of ln_2) on the register-resident-weights kernel (rollout batch sizes, "gemm_mid_rows_rn") against
    the ring kernel: the fp32 residual stream, the bf16 gain-scaled copy AND the per-64-column sums of squares must be the same bits - a sample's result must
    not depend on the batch size it is computed in (batch-slice consistency)."""
    import ctypes as C
    from hip_helpers import p, stream
    lib = L.load()
    D = 1024
    g = torch.Generator().manual_seed(M)
    ya = (torch.randn(M, D, generator=g) * 0.5).to(torch.bfloat16).cuda(); wo = (torch.randn(D, D, generator=g) * D ** -0.5).to(torch.bfloat16).cuda()
    x0 = torch.randn(M, D, generator=g).cuda(); g2 = (1 + 0.1 * torch.randn(D, generator=g)).cuda()
    outs = {}
    for rn in (512, 0):
        x = torch.full((M, D), float("nan"), device="cuda"); xg = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda"); ss = torch.full((M, D // 64), float("nan"), device="cuda")
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_RESIDUAL_NORM, out_dtype=L.MODE_F32, M=M, N=D, K=D, A=p(ya), lda=D, W=p(wo), ldw=D, resid=p(x0), ldr=D,
                           C=p(x), ldc=D, C2=p(xg), ldc2=D, gain=p(g2), row_ss_out=p(ss))
        lib.mode_set_option(b"gemm_mid_rows_rn", rn)
        try:
            L.check(lib.mode_gemm(C.byref(d), stream()), "gemm")
        finally:
            lib.mode_set_option(b"gemm_mid_rows_rn", 512)
        torch.cuda.synchronize()
        outs[rn] = (x, xg, ss)
    for a, b in zip(outs[512], outs[0]):
        assert torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32), b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32))
    ref = x0 + ya.float() @ wo.float().t()
    assert float((outs[512][0] - ref).norm() / ref.norm()) < 1e-3
    assert float((outs[512][2].sum(1) - (outs[512][0] ** 2).sum(1)).abs().max() / (outs[512][0] ** 2).sum(1).max()) < 1e-5


def test_large_lds_kernels_keep_their_launch_attribute_per_device():
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: every launcher of a > 64-KiB-LDS kernel keeps a per-device "done" bit
    (mode_common.h: LdsLimitOnce) instead of one process-wide flag.  With one GPU in the box the table is exercised by device round trips: the
    154-KiB ping-pong GEMM, the 160-KiB fused QKV + attention and a ring kernel launch before and after `set_device` hops (and on every visible
    device, if there are several) and keep returning the same bits."""
    M_, N_, K_ = 1792, 2048, 1024
    A = rnd(M_, K_, seed=71).to(torch.bfloat16); W = rnd(N_, K_, seed=72, scale=K_ ** -0.5).to(torch.bfloat16)
    hh = rnd(8 * 14, 1024, seed=73).to(torch.bfloat16); wq = rnd(3072, 1024, seed=74, scale=1 / 32).to(torch.bfloat16); bq = rnd(3072, seed=75)
    lib = L.load()
    outs = []
    ndev = torch.cuda.device_count()
    try:
        for rnd_trip in range(2):
            for d_ in range(ndev):
                torch.cuda.set_device(d_)
                dv = torch.device("cuda", d_)
                lib.mode_set_option(b"gemm_cfg", 17)                                # the persistent ping-pong kernel (154 KiB of LDS)
                o1 = H.gemm(A.to(dv), W.to(dv), out_dtype=torch.bfloat16)
                lib.mode_set_option(b"gemm_cfg", 4)                                 # the three-slot 128 x 64 ring (72 KiB)
                o2 = H.gemm(A.to(dv), W.to(dv), out_dtype=torch.bfloat16)
                lib.mode_set_option(b"gemm_cfg", 0)
                g1 = torch.ones(128, device=dv)
                rc, o3 = H.qkv_attn(hh.to(dv), wq.to(dv), bq.to(dv), g1, g1, 8, 14, 8)
                assert rc == 0
                torch.cuda.synchronize(dv)
                outs.append((o1.cpu(), o2.cpu(), o3.cpu()))
    finally:
        lib.mode_set_option(b"gemm_cfg", 0)
        torch.cuda.set_device(0)
    ref = A.float() @ W.float().t()
    assert rel(outs[0][0].float(), ref) < 6e-3
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
