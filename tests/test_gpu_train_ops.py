"""GPU parity of the training-side HIP kernels (through the C-ABI) against plain fp32 PyTorch autograd of the same op."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from mode_diffusion_policy_amd import _lib as L  # noqa: E402
from oracle import mode_oracle as O  # noqa: E402

import hip_helpers as H  # noqa: E402

dev = lambda: torch.device("cuda:0")
DT = {torch.bfloat16: L.MODE_BF16, torch.float32: L.MODE_F32}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("R,Ccols", [(70, 33), (1792, 1024), (200, 64)])
def test_transpose_gather_pad(dtype, R, Ccols):
    lib = L.load()
    src = rnd(R + 5, Ccols, seed=1).to(dtype).to(dev())
    g = torch.Generator().manual_seed(2)
    rows = torch.randperm(R + 5, generator=g)[:R].int()
    dcols = (torch.arange(R) + (torch.arange(R) >= R // 2).int() * 7).int()          # a gap of 7 padded columns in the middle
    ld = R + 7
    dst = torch.zeros(Ccols, ld, dtype=dtype, device=dev())
    rows_d, dcols_d = rows.to(dev()), dcols.to(dev())                                 # keep device copies alive across the launch
    L.check(lib.mode_transpose(src.data_ptr(), src.stride(0), R, Ccols, dst.data_ptr(), ld, rows_d.data_ptr(),
                               dcols_d.data_ptr(), DT[dtype], H.stream()))
    ref = torch.zeros(Ccols, ld, dtype=dtype)
    ref[:, dcols.long()] = src.cpu()[rows.long()].t()
    assert torch.equal(dst.cpu(), ref)
    dst2 = torch.empty(Ccols, R, dtype=dtype, device=dev())
    L.check(lib.mode_transpose(src.data_ptr(), src.stride(0), R, Ccols, dst2.data_ptr(), R, None, None, DT[dtype], H.stream()))
    assert torch.equal(dst2.cpu(), src.cpu()[:R].t())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_colsum_segments(dtype):
    lib = L.load()
    R, Cc = 3584, 8192 if dtype == torch.bfloat16 else 1024
    X = rnd(R, Cc, seed=3).to(dtype).to(dev())
    offs = torch.tensor([0, 1000, 1000, 2500, R], dtype=torch.int32, device=dev())
    wsb = lib.mode_colsum_workspace_bytes(R, Cc, 4)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    out = torch.empty(4, Cc, device=dev())
    L.check(lib.mode_colsum(X.data_ptr(), Cc, R, Cc, DT[dtype], offs.data_ptr(), 0, 4, out.data_ptr(), 0, ws.data_ptr(), wsb, H.stream()))
    Xc = X.float().cpu()
    ref = torch.stack([Xc[0:1000].sum(0), Xc[1000:1000].sum(0), Xc[1000:2500].sum(0), Xc[2500:].sum(0)])
    assert rel(out, ref) < 1e-5
    # uniform segments (per-sample sums over T=14 token rows) + accumulate
    nseg = R // 14
    wsb = lib.mode_colsum_workspace_bytes(R, Cc, nseg); ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    out2 = torch.ones(nseg, Cc, device=dev())
    L.check(lib.mode_colsum(X.data_ptr(), Cc, R, Cc, DT[dtype], None, 14, nseg, out2.data_ptr(), 1, ws.data_ptr(), wsb, H.stream()))
    assert rel(out2, 1 + Xc.view(nseg, 14, Cc).sum(1)) < 1e-5
    # determinism
    out3 = torch.ones(nseg, Cc, device=dev())
    L.check(lib.mode_colsum(X.data_ptr(), Cc, R, Cc, DT[dtype], None, 14, nseg, out3.data_ptr(), 1, ws.data_ptr(), wsb, H.stream()))
    assert torch.equal(out2, out3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_swiglu_fwd_bwd(dtype, p):
    lib = L.load()
    R, Hd = 300, 256
    P = rnd(R, 2 * Hd, seed=4).to(dtype)
    dH = rnd(R, Hd, seed=5).to(dtype)
    out = torch.empty(R, Hd, dtype=dtype, device=dev()); dP = torch.empty(R, 2 * Hd, dtype=dtype, device=dev())
    Pd, dHd = P.to(dev()), dH.to(dev())
    L.check(lib.mode_swiglu_fwd(Pd.data_ptr(), out.data_ptr(), R, Hd, DT[dtype], 1234, p, H.stream()))
    L.check(lib.mode_swiglu_bwd(Pd.data_ptr(), dHd.data_ptr(), dP.data_ptr(), R, Hd, DT[dtype], 1234, p, H.stream()))
    Pf = P.float().requires_grad_(True)
    h = Pf[:, :Hd] * torch.nn.functional.silu(Pf[:, Hd:])
    o = out.float().cpu()
    if p == 0:
        mask = torch.ones_like(h)
    else:
        mask = (o != 0).float() / (1 - p)                                         # mask recovered from the forward output
        keep = float((o != 0).float().mean())
        assert abs(keep - (1 - p)) < 0.02                                         # Bernoulli(1-p) keep rate
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel(o, (h * mask).detach()) < tol
    (h * mask).backward(dH.float())
    assert rel(dP.float(), Pf.grad) < tol                                         # same mask regenerated in the backward


@pytest.mark.parametrize("rows,D,k", [(37, 64, 0), (1792, 1024, 2), (112, 256, 1)])
def test_rmsnorm_bwd_with_gather(rows, D, k):
    lib = L.load()
    x = rnd(rows, D, seed=6, scale=2.0); g = 1 + 0.1 * rnd(D, seed=7)
    x[3] = 0.0                                                                     # eps-clamped row
    da = rnd(rows, D, seed=8); db = rnd(rows, D, seed=9)
    NK = rows * max(k, 1)
    Gm = rnd(NK, D, seed=10); pos = torch.randperm(NK, generator=torch.Generator().manual_seed(11)).int()
    dx0 = rnd(rows, D, seed=12)
    dx = dx0.clone().to(dev()); nb = (rows + 3) // 4
    dgp = torch.empty(nb, D, device=dev()); dyo = torch.empty(rows, D, device=dev()); dxlp = torch.empty(rows, D, dtype=torch.bfloat16, device=dev())
    xd, gd, dad, dbd, Gd, pd = (t.to(dev()) for t in (x, g, da, db, Gm, pos))
    L.check(lib.mode_rmsnorm_bwd(xd.data_ptr(), gd.data_ptr(), dad.data_ptr(), dbd.data_ptr(), Gd.data_ptr() if k else None,
                                 pd.data_ptr() if k else None, k, rows, D, 1e-6, dx.data_ptr(), 1, dgp.data_ptr(), dyo.data_ptr(), dxlp.data_ptr(),
                                 L.MODE_BF16, H.stream()))
    xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True)
    dy = da + db
    if k:
        dy = dy + Gm[pos.long()].view(rows, k, D).sum(1)
    O.rmsnorm(xr, gr).backward(dy)
    assert rel(dyo, dy) < 1e-6
    assert rel(dx, dx0 + xr.grad) < 1e-5
    assert rel(dxlp.float(), dx0 + xr.grad) < 5e-3
    assert rel(dgp.sum(0), gr.grad) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,D,k", [(200, 256, 2), (203, 1024, 2), (50, 512, 1), (64, 1024, 3), (30, 768, 2)])     # (bf16, k <= 2, D = 256 / 512 / 1024: the all-loads-first kernel)
def test_combine_bwd(dtype, N, D, k):
    lib = L.load()
    dy = rnd(N, D, seed=13); Y = rnd(N * k, D, seed=14).to(dtype)
    pos = torch.randperm(N * k, generator=torch.Generator().manual_seed(15)).int(); posw = torch.rand(N * k, generator=torch.Generator().manual_seed(16))
    dYs = torch.zeros(N * k, D, dtype=dtype, device=dev()); dw = torch.empty(N * k, device=dev())
    dyd, Yd, pd, pwd = dy.to(dev()), Y.to(dev()), pos.to(dev()), posw.to(dev())
    L.check(lib.mode_moe_combine_bwd(dyd.data_ptr(), Yd.data_ptr(), DT[dtype], pd.data_ptr(), pwd.data_ptr(), N, D, k, dYs.data_ptr(), dw.data_ptr(),
                                     H.stream()))
    Yf = Y.float().requires_grad_(True); w = posw.clone().requires_grad_(True)
    nxt = (w.view(N, k, 1) * Yf[pos.long()].view(N, k, D)).sum(1)
    nxt.backward(dy)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel(dYs.float(), Yf.grad) < tol
    assert rel(dw, w.grad) < 1e-5


def _attn_ref(qkv, qg, kg, B, T, Hh, hd, mask=None):
    D = Hh * hd
    q, k, v = (t.view(B, T, Hh, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    q = O.rmsnorm(q, qg); k = O.rmsnorm(k, kg)
    att = (q @ k.transpose(-2, -1)) / (hd ** 0.5)
    att = att.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float("-inf")).softmax(-1)
    if mask is not None:
        att = att * mask
    return (att @ v).transpose(1, 2).reshape(B * T, D)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,Hh,hd,p", [(3, 14, 4, 32, 0.0), (8, 14, 8, 128, 0.0), (4, 14, 2, 32, 0.3), (2, 13, 2, 64, 0.0)])
def test_attention_backward(dtype, B, T, Hh, hd, p):
    lib = L.load()
    D = Hh * hd
    qkv = rnd(B * T, 3 * D, seed=21).to(dtype)
    qg = 1 + 0.1 * rnd(hd, seed=22); kg = 1 + 0.1 * rnd(hd, seed=23)
    dy = rnd(B * T, D, seed=24).to(dtype)
    seed = 777
    mask = None
    if p > 0:                                                                 # recover the hash mask: V := one-hot(token) -> O[:, :T] = Pd
        probe = qkv.clone().float().view(B, T, 3, Hh, hd)
        probe[:, :, 2] = 0
        for t in range(T):
            probe[:, t, 2, :, t] = 1.0
        yp = H.attn(probe.view(B * T, 3 * D).to(dtype).to(dev()), qg.to(dev()), kg.to(dev()), B, T, Hh, hd, seed=seed, p_drop=p)
        Pd = yp.float().cpu().view(B, T, Hh, hd)[..., :T].permute(0, 2, 1, 3)                # [B,H,q,k]
        tril = torch.ones(T, T).tril().bool()
        mask = ((Pd != 0) & tril).float() / (1 - p)
        keep = float(((Pd != 0) & tril).float().sum() / (B * Hh * tril.sum()))
        assert abs(keep - (1 - p)) < 0.06
    qd, qgd, kgd, dyd = qkv.to(dev()), qg.to(dev()), kg.to(dev()), dy.to(dev())
    dqkv = torch.full((B * T, 3 * D), float("nan"), dtype=dtype, device=dev())
    pq = torch.empty(B * Hh, hd, device=dev()); pk = torch.empty(B * Hh, hd, device=dev())
    L.check(lib.mode_attn_block_bwd(qd.data_ptr(), qgd.data_ptr(), kgd.data_ptr(), dyd.data_ptr(), dqkv.data_ptr(), pq.data_ptr(), pk.data_ptr(),
                                    DT[dtype], B, T, Hh, hd, 1e-6, seed, p, H.stream()))
    x = qkv.float().requires_grad_(True); a = qg.clone().requires_grad_(True); c = kg.clone().requires_grad_(True)
    y = _attn_ref(x, a, c, B, T, Hh, hd, mask)
    y.backward(dy.float())
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert rel(dqkv.float(), x.grad) < tol
    assert rel(pq.sum(0), a.grad) < tol and rel(pk.sum(0), c.grad) < tol
    if p > 0:                                                                 # forward with the same mask
        yf = H.attn(qd, qgd, kgd, B, T, Hh, hd, seed=seed, p_drop=p)
        assert rel(yf.float(), y.detach()) < (2e-2 if dtype == torch.bfloat16 else 1e-5)
    # the VALU form of the five matrix products ("attn_bwd_mfma" 0; the only form for head_dim % 16 != 0): same results to fp32 summation order
    dq2 = torch.full_like(dqkv, float("nan")); pq2 = torch.empty_like(pq); pk2 = torch.empty_like(pk)
    lib.mode_set_option(b"attn_bwd_mfma", 0)
    try:
        L.check(lib.mode_attn_block_bwd(qd.data_ptr(), qgd.data_ptr(), kgd.data_ptr(), dyd.data_ptr(), dq2.data_ptr(), pq2.data_ptr(), pk2.data_ptr(),
                                        DT[dtype], B, T, Hh, hd, 1e-6, seed, p, H.stream()))
    finally:
        lib.mode_set_option(b"attn_bwd_mfma", 1)
    assert rel(dq2.float(), dqkv.float()) < (4e-3 if dtype == torch.bfloat16 else 2e-6)     # bf16: an occasional last-bit difference of a stored value
    assert rel(pq2, pq) < 1e-5 and rel(pk2, pk) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_expert_weight_gradient_gemm(dtype):
    """dW2_e = dY_e^T H_e for all experts in ONE launch: transposes into the 64-padded per-expert layout + K-group GEMM."""
    lib = L.load()
    N, E, k, D, Hd = 300, 4, 2, 64, 256
    g = torch.Generator().manual_seed(31)
    probs = torch.rand(N, E, generator=g); probs[:, 3] = 0                       # expert 3 gets no token -> its gradient must be exactly 0
    idx = torch.sort(probs, dim=-1, descending=True, stable=True).indices[:, :k].contiguous()
    w = probs.gather(1, idx)
    meta = H.dispatch_meta(idx.int().to(dev()), w.to(dev()), 1, N, E)
    counts = meta["counts"].cpu().long(); poff = meta["poffsets"].cpu().long()
    assert torch.equal(poff[1:] - poff[:-1], (counts + 63) // 64 * 64)
    NK = N * k; NKp = (NK + 63) // 64 * 64 + 64 * E
    dY = rnd(NK, D, seed=32).to(dtype).to(dev()); Hs = rnd(NK, Hd, seed=33).to(dtype).to(dev())
    dYT = torch.zeros(D, NKp, dtype=dtype, device=dev()); HT = torch.zeros(Hd, NKp, dtype=dtype, device=dev())
    L.check(lib.mode_transpose(dY.data_ptr(), D, NK, D, dYT.data_ptr(), NKp, None, meta["prow"].data_ptr(), DT[dtype], H.stream()))
    L.check(lib.mode_transpose(Hs.data_ptr(), Hd, NK, Hd, HT.data_ptr(), NKp, None, meta["prow"].data_ptr(), DT[dtype], H.stream()))
    dW = torch.full((E, D, Hd), float("nan"), device=dev())
    d = L.ModeGemmDesc(dtype=DT[dtype], epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=D, N=Hd, K=NKp, A=dYT.data_ptr(), lda=NKp, W=HT.data_ptr(),
                       ldw=NKp, C=dW.data_ptr(), ldc=Hd, k_group_offsets=meta["poffsets"].data_ptr(), num_k_groups=E, c_group_stride=D * Hd)
    L.check(lib.mode_gemm(C.byref(d), H.stream()))
    offs = meta["offsets"].cpu().long()
    for e in range(E):
        ref = dY[offs[e]: offs[e + 1]].float().cpu().t() @ Hs[offs[e]: offs[e + 1]].float().cpu()
        if counts[e] == 0:
            assert float(dW[e].abs().max()) == 0.0
        else:
            assert rel(dW[e], ref) < (3e-3 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("Ly,R,E,D", [(12, 128, 4, 256), (2, 10, 2, 64), (3, 1, 8, 128)])
def test_router_mlp_all_layers_fwd_bwd(Ly, R, E, D):
    """Layer-batched router MLP (RouterCond.router, modedit.py:190-200): logits of every layer in one launch and the batched backward
    (dpre, dW3) against fp32 autograd of the per-layer Linear -> GELU -> Linear."""
    lib = L.load()
    H2 = 2 * D
    pre = rnd(R, Ly, H2, seed=1).to(dev()).requires_grad_(True)                     # [R][L][2D] pre-GELU activations
    w3 = (rnd(Ly, E, H2, seed=2) * H2 ** -0.5).to(dev()).requires_grad_(True)
    b3 = rnd(Ly, E, seed=3).to(dev())
    hid = torch.nn.functional.gelu(pre)
    ref_logits = torch.einsum("rln,len->lre", hid, w3) + b3[:, None, :]
    hid_c = hid.detach().contiguous()
    logits = torch.full((Ly, R, E), float("nan"), device=dev())
    L.check(lib.mode_router_logits(hid_c.data_ptr(), Ly * H2, w3.data_ptr(), E * H2, b3.data_ptr(), E, Ly, R, E, H2, logits.data_ptr(), H.stream()))
    assert rel(logits, ref_logits.detach()) < 2e-6
    dlog = rnd(Ly, R, E, seed=4).to(dev())
    ref_logits.backward(dlog)
    dpre = torch.full((R, Ly, H2), float("nan"), device=dev()); dw3 = torch.full((Ly, E, H2), float("nan"), device=dev())
    L.check(lib.mode_router_mlp_bwd(dlog.data_ptr(), pre.detach().data_ptr(), w3.detach().data_ptr(), Ly, R, E, H2, dpre.data_ptr(), dw3.data_ptr(),
                                    H.stream()))
    assert rel(dpre, pre.grad) < 2e-6 and rel(dw3, w3.grad) < 2e-6


@pytest.mark.parametrize("with_lp", [False, True])
def test_adamw_kernel_matches_torch(with_lp):
    """mode_adamw_step on a flat slice == torch.optim.AdamW (decoupled decay, bias correction) over several steps; n not a multiple of the
    kernel's unroll width; optional bf16 shadow output; grad_scale folds the data-parallel mean."""
    lib = L.load()
    n = 4 * 100003
    p0 = rnd(n, seed=1); g = [rnd(n, seed=10 + i) for i in range(3)]
    p = p0.clone().to(dev()); m = torch.zeros(n, device=dev()); v = torch.zeros(n, device=dev())
    lp = torch.zeros(n, dtype=torch.bfloat16, device=dev()) if with_lp else None
    pt = p0.clone().to(dev()).requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    for step in range(1, 4):
        gd = (g[step - 1] * 2.0).to(dev())                                          # "summed over 2 ranks"
        L.check(lib.mode_adamw_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 3e-3, 0.9, 0.95, 1e-8, 0.05, step, 0.5,
                                    None if lp is None else lp.data_ptr(), None, 0.0, H.stream()))
        pt.grad = g[step - 1].to(dev())
        opt.step()
        assert rel(p, pt.detach()) < 1e-6, step
    if with_lp:
        assert torch.equal(lp, p.to(torch.bfloat16))


def test_iota_offsets():
    out = torch.full((13,), -1, dtype=torch.int32, device=dev())
    L.check(L.load().mode_iota_i32(out.data_ptr(), 13, 2048, H.stream()))
    assert torch.equal(out.cpu(), torch.arange(13, dtype=torch.int32) * 2048)


@pytest.mark.parametrize("N_tok,E,k,D,p_drop", [(448, 4, 2, 256, 0.0), (70, 4, 2, 64, 0.0), (1792, 4, 2, 128, 0.0), (112, 2, 1, 256, 0.0)])
def test_moe_grouped_mlp_fwd_bwd(N_tok, E, k, D, p_drop):
    """SURVEY §8b export `moe_grouped_mlp_fwd/_bwd`: the expert MLP of one block on the sorted dispatch order vs fp32 autograd of the
    reference's per-expert loop (modedit.py:557-566 with FusedMLPV2 / SwishGLU) on the same bf16-rounded operands."""
    lib = L.load()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(N_tok + D)
    x = rnd(N_tok, D, seed=1).to(bf)
    w1 = (rnd(E, 8 * D, D, seed=2) * D ** -0.5).to(bf); b1 = rnd(E, 8 * D, seed=3) * 0.1
    w2 = (rnd(E, D, 4 * D, seed=4) * (4 * D) ** -0.5).to(bf)
    logits = torch.randn(N_tok, E, generator=g)
    _, _, idx, w = H.route_topk(logits.to(dev()), k)
    meta = H.dispatch_meta(idx, w, 1, N_tok, E)
    perm, offsets = meta["perm"], meta["offsets"]
    NK = N_tok * k
    xd, w1d, b1d, w2d = x.to(dev()), w1.to(dev()), b1.to(dev()), w2.to(dev())
    pbuf = torch.empty(NK, 8 * D, dtype=bf, device=dev()); hbuf = torch.empty(NK, 4 * D, dtype=bf, device=dev())
    y = torch.full((NK, D), float("nan"), device=dev())
    dy = rnd(NK, D, seed=5).to(bf).to(dev())
    dxs = torch.full((NK, D), float("nan"), device=dev()); dw1 = torch.full((E, 8 * D, D), float("nan"), device=dev())
    db1 = torch.full((E, 8 * D), float("nan"), device=dev()); dw2 = torch.full((E, D, 4 * D), float("nan"), device=dev())
    d = L.ModeGroupedMlpDesc(dtype=L.MODE_BF16, N=N_tok, D=D, E=E, k=k, x=xd.data_ptr(), perm=perm.data_ptr(), offsets=offsets.data_ptr(),
                             w1=w1d.data_ptr(), b1=b1d.data_ptr(), w2=w2d.data_ptr(), p=pbuf.data_ptr(), h=hbuf.data_ptr(), y=y.data_ptr(),
                             y_dtype=L.MODE_F32, seed=0, p_drop=p_drop, dy=dy.data_ptr(), dxs=dxs.data_ptr(), dw1=dw1.data_ptr(),
                             db1=db1.data_ptr(), dw2=dw2.data_ptr())
    L.check(lib.mode_moe_grouped_mlp_fwd(C.byref(d), H.stream()), "mlp fwd")
    # inference form (fused SwiGLU epilogue, no pre-activation kept) must agree with the training form
    y2 = torch.full((NK, D), float("nan"), device=dev()); h2 = torch.empty_like(hbuf)
    d2 = L.ModeGroupedMlpDesc(dtype=L.MODE_BF16, N=N_tok, D=D, E=E, k=k, x=xd.data_ptr(), perm=perm.data_ptr(), offsets=offsets.data_ptr(),
                              w1=w1d.data_ptr(), b1=b1d.data_ptr(), w2=w2d.data_ptr(), p=None, h=h2.data_ptr(), y=y2.data_ptr(), y_dtype=L.MODE_F32)
    L.check(lib.mode_moe_grouped_mlp_fwd(C.byref(d2), H.stream()), "mlp fwd (fused)")
    wsb = lib.mode_moe_grouped_mlp_workspace_bytes(N_tok, D, E, k, L.MODE_BF16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    L.check(lib.mode_moe_grouped_mlp_bwd(C.byref(d), ws.data_ptr(), wsb, H.stream()), "mlp bwd")
    # fp32 autograd reference on the sorted rows
    xs = x.float()[perm.cpu().long()].requires_grad_(True)
    W1 = w1.float().requires_grad_(True); B1 = b1.clone().requires_grad_(True); W2 = w2.float().requires_grad_(True)
    off = offsets.cpu().tolist()
    outs = []
    for e in range(E):
        seg = xs[off[e]:off[e + 1]]
        pre = seg @ W1[e].t() + B1[e]
        val, gate = pre.tensor_split(2, dim=-1)
        outs.append((val * torch.nn.functional.silu(gate)) @ W2[e].t())
    ref = torch.cat(outs)
    ref.backward(dy.float().cpu())
    assert rel(y, ref.detach()) < 1e-2 and rel(y2, ref.detach()) < 1e-2 and rel(y2, y) < 6e-3
    assert rel(dxs, xs.grad) < 1.5e-2
    for got, want, name in ((dw1, W1.grad, "dw1"), (db1, B1.grad, "db1"), (dw2, W2.grad, "dw2")):
        for e in range(E):
            if off[e + 1] > off[e]:
                assert rel(got[e], want[e]) < 1.5e-2, (name, e)
            else:
                assert float(got[e].abs().max()) == 0.0, (name, e)            # un-routed expert: exact zeros, not garbage


@pytest.mark.parametrize("rows,D,rpc", [(1792, 1024, 14), (70, 64, 14), (33, 128, 0)])
def test_rmsnorm_cond_bwd_composite(rows, D, rpc):
    lib = L.load()
    x = rnd(rows, D, seed=1, scale=2.0).requires_grad_(True); g = (1 + 0.1 * rnd(D, seed=2)).requires_grad_(True)
    nc = (rows + rpc - 1) // rpc if rpc else 0
    cond = rnd(nc, D, seed=3).requires_grad_(True) if rpc else None
    y = O.rmsnorm(x, g)
    if rpc:
        y = y + cond[torch.arange(rows) // rpc]
    dy = rnd(rows, D, seed=4)
    y.backward(dy)
    xd, gd, dyd = x.detach().to(dev()), g.detach().to(dev()), dy.to(dev())
    dx = torch.full((rows, D), float("nan"), device=dev()); dg = torch.full((D,), float("nan"), device=dev())
    dc = torch.full((nc, D), float("nan"), device=dev()) if rpc else None
    wsb = lib.mode_rmsnorm_cond_bwd_workspace_bytes(rows, D, rpc)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    L.check(lib.mode_rmsnorm_cond_bwd(xd.data_ptr(), gd.data_ptr(), dyd.data_ptr(), rows, D, rpc, 1e-6, dx.data_ptr(), dg.data_ptr(),
                                      None if dc is None else dc.data_ptr(), ws.data_ptr(), wsb, H.stream()), "rmsnorm_cond_bwd")
    assert rel(dx, x.grad) < 1e-5 and rel(dg, g.grad) < 1e-5
    if rpc:
        assert rel(dc, cond.grad) < 1e-5


@pytest.mark.parametrize("rows,Hdim,E,p_drop", [(3584, 4096, 4, 0.1), (70, 64, 4, 0.0), (900, 256, 2, 0.0), (33, 128, 8, 0.25)])
def test_swiglu_bwd_bias_fused_matches_separate_kernels(rows, Hdim, E, p_drop):
    """mode_swiglu_bwd_bias == mode_swiglu_bwd followed by the segmented column sum (dP to one bf16 ulp, bias sums to fp32 reduction-order noise),
    incl. empty experts and segment boundaries that fall inside a row block."""
    lib = L.load()
    bf = torch.bfloat16
    P = rnd(rows, 2 * Hdim, seed=1).to(bf).to(dev()); dH = rnd(rows, Hdim, seed=2).to(bf).to(dev())
    g = torch.Generator().manual_seed(rows)
    cuts = sorted(torch.randint(0, rows + 1, (E - 1,), generator=g).tolist())
    if E >= 4:
        cuts[1] = cuts[0]                                                    # an empty expert
    off = torch.tensor([0] + cuts + [rows], dtype=torch.int32, device=dev())
    dP0 = torch.empty(rows, 2 * Hdim, dtype=bf, device=dev())
    L.check(lib.mode_swiglu_bwd(P.data_ptr(), dH.data_ptr(), dP0.data_ptr(), rows, Hdim, L.MODE_BF16, 77, p_drop, H.stream()))
    ref_db = torch.zeros(E, 2 * Hdim)
    o = off.cpu().tolist()
    for e in range(E):
        ref_db[e] = dP0[o[e]:o[e + 1]].float().sum(0).cpu()
    dP1 = torch.full((rows, 2 * Hdim), float("nan"), dtype=bf, device=dev()); db = torch.full((E, 2 * Hdim), float("nan"), device=dev())
    wsb = lib.mode_swiglu_bwd_bias_workspace_bytes(rows, Hdim, E)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    L.check(lib.mode_swiglu_bwd_bias(P.data_ptr(), dH.data_ptr(), dP1.data_ptr(), rows, Hdim, L.MODE_BF16, 77, p_drop, off.data_ptr(), E, db.data_ptr(),
                                     ws.data_ptr(), wsb, H.stream()))
    # same arithmetic, different instruction selection (fma contraction): equal up to one bf16 ulp on isolated elements
    assert rel(dP1.float(), dP0.float()) < 1e-3 and float((dP1 != dP0).float().mean()) < 0.02
    for e in range(E):                                                       # the fused sums are the column sums of the dP it wrote
        ref_db[e] = dP1[o[e]:o[e + 1]].float().sum(0).cpu()
    assert torch.isfinite(db).all() and float((db.cpu() - ref_db).abs().max()) <= 1e-4 * max(1.0, float(ref_db.abs().max()))
    for e in range(E):
        if o[e] == o[e + 1]:
            assert float(db[e].abs().max()) == 0.0
    # the vectorised forward agrees with the element kernel's arithmetic (same hash mask)
    Hd = torch.empty(rows, Hdim, dtype=bf, device=dev())
    L.check(lib.mode_swiglu_fwd(P.data_ptr(), Hd.data_ptr(), rows, Hdim, L.MODE_BF16, 77, p_drop, H.stream()))
    v, gt = P.float().tensor_split(2, dim=-1)
    ref_h = v * torch.nn.functional.silu(gt)
    keep = Hd.float() != 0
    if p_drop > 0:
        frac = 1.0 - float(keep.float().mean())
        assert abs(frac - p_drop) < 0.02 + 2.0 / (rows * Hdim) ** 0.5
        assert rel(Hd.float()[keep], (ref_h / (1 - p_drop))[keep]) < 6e-3
    else:
        assert rel(Hd.float(), ref_h) < 6e-3


@pytest.mark.parametrize("N,T,E,k,norm", [(1792, 14, 4, 2, 1), (300, 1, 8, 3, 0), (64, 4, 16, 16, 1), (1000, 1, 5, 1, 1)])
def test_sample_experts_is_the_exponential_race_of_torch_multinomial(N, T, E, k, norm):
    """mode_moe_sample_experts against the same race written in torch ((p / q).topk(k) - torch.multinomial's own algorithm without replacement):
    identical ids in identical order, combine weights of mode_moe_weights_from_idx; plus a distribution check of the first pick against p."""
    lib = L.load()
    torch.manual_seed(N + E)
    R = N // T
    probs = torch.softmax(torch.randn(R, E, device="cuda") * 1.5, -1).clamp(1e-9, 1 - 1e-9)
    expo = torch.empty(N, E, device="cuda").exponential_()
    idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); w = torch.empty(N, k, device="cuda")
    L.check(lib.mode_moe_sample_experts(probs.data_ptr(), expo.data_ptr(), N, T, E, k, norm, idx.data_ptr(), w.data_ptr(), None), "sample")
    ptok = probs.repeat_interleave(T, 0)
    ref_idx = (ptok / expo).topk(k, dim=-1).indices
    assert torch.equal(idx.long(), ref_idx)
    assert all(len(set(r)) == k for r in idx[:50].tolist())                                    # without replacement
    pw = ptok.gather(1, ref_idx)
    ref_w = pw / pw.sum(-1, keepdim=True) if norm else pw
    assert torch.equal(w, ref_w) or float((w - ref_w).abs().max()) < 1e-7
    w2 = torch.empty_like(w)
    L.check(lib.mode_moe_weights_from_idx(probs.data_ptr(), idx.data_ptr(), N, T, E, k, norm, w2.data_ptr(), None), "w_from_idx")
    assert torch.equal(w, w2)
    # same generator state -> the SAME ids as torch.multinomial itself (its no-replacement path draws q = empty_like(p).exponential_() and takes
    # topk(p / q); aten/src/ATen/native/Distributions.cpp): the replacement changes launches, not the random stream
    torch.manual_seed(77)
    ref_m = torch.multinomial(ptok, k, replacement=False)
    torch.manual_seed(77)
    expo2 = torch.empty(N, E, device="cuda").exponential_()
    L.check(lib.mode_moe_sample_experts(probs.data_ptr(), expo2.data_ptr(), N, T, E, k, norm, idx.data_ptr(), w.data_ptr(), None), "sample")
    assert torch.equal(idx.long(), ref_m)
    # distribution: many draws of the FIRST pick from one row follow p (the race is exact, not an approximation)
    M = 200_000
    p1 = probs[:1].contiguous()
    ex = torch.empty(M, E, device="cuda").exponential_()
    i1 = torch.empty(M, 1, dtype=torch.int32, device="cuda"); w1 = torch.empty(M, 1, device="cuda")
    L.check(lib.mode_moe_sample_experts(p1.data_ptr(), ex.data_ptr(), M, M, E, 1, 0, i1.data_ptr(), w1.data_ptr(), None), "sample")
    freq = torch.bincount(i1.view(-1).long(), minlength=E).float() / M
    sd = (p1[0] * (1 - p1[0]) / M).sqrt()
    assert bool(((freq - p1[0]).abs() < 5 * sd + 1e-4).all()), (freq, p1)


@pytest.mark.parametrize("Ly,R,T,E,k,tok", [(12, 128, 14, 4, 2, False), (3, 1792, 1, 4, 2, True), (2, 40, 5, 16, 3, False), (1, 2500, 1, 7, 2, True)])
def test_moe_aux_stats_vs_torch_expressions(Ly, R, T, E, k, tok):
    """One launch for the router side channels of a training forward (load-balancing term, z-loss, usage, one-hot-of-k mask) against the torch
    expressions it replaced (= the reference's, modedit.py:584-593, 816-820, 930-969)."""
    lib = L.load()
    torch.manual_seed(Ly * 100 + R)
    N = R * T
    Rs = N if tok else R
    idx = torch.stack([torch.stack([torch.randperm(E, device="cuda")[:k] for _ in range(R)]) for _ in range(Ly)]).to(torch.int32)
    w = torch.rand(Ly, R, k, device="cuda")
    shifted = torch.randn(Ly, Rs, E, device="cuda")
    shifted = shifted - shifted.max(-1, keepdim=True).values
    st = torch.empty(Ly * E + 2 * Ly + 2, device="cuda")
    frac, lb, zl = st[:Ly * E].view(Ly, E), st[Ly * E: Ly * E + Ly], st[Ly * E + Ly: Ly * E + 2 * Ly]
    mask = torch.full((Ly, N, E), 7.0, device="cuda")
    usage = torch.full((Ly, E), 5, dtype=torch.int64, device="cuda")
    L.check(lib.mode_moe_aux_stats(idx.data_ptr(), w.data_ptr(), Ly, R, T, E, k, shifted.data_ptr(), Rs, frac.data_ptr(), lb.data_ptr(), zl.data_ptr(),
                                   st[Ly * E + 2 * Ly:].data_ptr(), st[Ly * E + 2 * Ly + 1:].data_ptr(), mask.data_ptr(), usage.data_ptr(), None), "aux")
    idx64 = idx.unsqueeze(2).expand(Ly, R, T, k).reshape(Ly, N, k).long()
    wtok = w.unsqueeze(2).expand(Ly, R, T, k).reshape(Ly, N, k)
    mref = torch.zeros(Ly, N, E, device="cuda").scatter_(2, idx64, 1.0)
    rp = torch.zeros(Ly, N, E, device="cuda").scatter_(2, idx64, wtok)
    fref = mref.sum(1) / N
    lbref = E * (rp.mean(1) * fref).sum(-1)
    zref = torch.log(torch.exp(shifted).sum(-1) + 1e-6).pow(2).mean(-1)
    assert torch.equal(mask, mref)
    assert torch.equal(usage, 5 + mref.sum(1).long())
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(frac, fref) < 1e-6 and rel(lb, lbref) < 1e-5 and rel(zl, zref) < 1e-5
    assert abs(float(st[Ly * E + 2 * Ly]) - float(lbref.mean())) < 1e-5 * float(lbref.mean()) and abs(float(st[-1]) - float(zref.mean())) < 1e-5 * float(zref.mean())
    # without the optional outputs
    L.check(lib.mode_moe_aux_stats(idx.data_ptr(), w.data_ptr(), Ly, R, T, E, k, shifted.data_ptr(), Rs, frac.data_ptr(), lb.data_ptr(), zl.data_ptr(),
                                   st[Ly * E + 2 * Ly:].data_ptr(), st[Ly * E + 2 * Ly + 1:].data_ptr(), None, None, None), "aux")
    assert rel(lb, lbref) < 1e-5
