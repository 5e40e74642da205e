"""Thin test-side wrappers that call the C-ABI (ctypes) with torch device tensors."""
from __future__ import annotations

import ctypes as C

import torch

from mode_diffusion_policy_amd import _lib as L


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def dt_of(t):
    return L.MODE_BF16 if t.dtype == torch.bfloat16 else L.MODE_F32


def gemm(A, W, epilogue=L.EPI_NONE, bias=None, resid=None, out_dtype=None, n_out=None, a_rows=None, offsets=None, num_experts=0,
         M=None, w_estride=0, b_estride=0, flags=0):
    lib = L.load()
    out_dtype = out_dtype or A.dtype
    K = A.shape[-1]
    Wm = W if W.dim() == 2 else W[0]
    N = n_out if n_out is not None else (Wm.shape[0] // 2 if epilogue == L.EPI_SWIGLU else Wm.shape[0])
    M = M if M is not None else A.shape[0]
    Cc = torch.full((M, N), float("nan"), dtype=out_dtype, device=A.device)
    d = L.ModeGemmDesc(dtype=dt_of(A), epilogue=epilogue, out_dtype=L.MODE_BF16 if out_dtype == torch.bfloat16 else L.MODE_F32,
                       M=M, N=N, K=K, A=p(A), lda=A.stride(0), W=p(W), ldw=Wm.stride(0), w_expert_stride=w_estride,
                       bias=p(bias), bias_expert_stride=b_estride, resid=p(resid), ldr=(resid.stride(0) if resid is not None else 0),
                       C=p(Cc), ldc=Cc.stride(0), a_rows=p(a_rows), expert_offsets=p(offsets), num_experts=num_experts, flags=flags)
    L.check(lib.mode_gemm(C.byref(d), stream()), "gemm")
    return Cc


def route_topk(logits, k, normalize=True):
    lib = L.load()
    R, E = logits.shape
    sh = torch.empty_like(logits); pr = torch.empty_like(logits)
    idx = torch.full((R, k), -1, dtype=torch.int32, device=logits.device)
    w = torch.empty(R, k, dtype=torch.float32, device=logits.device)
    L.check(lib.mode_moe_route_topk_f32(p(logits), R, E, k, int(normalize), p(sh), p(pr), p(idx), p(w), stream()), "route")
    return sh, pr, idx, w


def dispatch_meta(idx, w, tokens_per_row, N, E):
    lib = L.load()
    R, k = idx.shape
    dev = idx.device
    i32 = lambda *s: torch.full(s, -1, dtype=torch.int32, device=dev)
    counts, offsets, perm, pos = i32(E), i32(E + 1), i32(N * k), i32(N * k)
    posw = torch.empty(N * k, dtype=torch.float32, device=dev)
    poffsets, prow = i32(E + 1), i32(N * k)
    L.check(lib.mode_moe_dispatch_meta(p(idx), p(w), R, tokens_per_row, N, E, k, p(counts), p(offsets), p(perm), p(pos), p(posw),
                                       p(poffsets), p(prow), stream()), "dispatch_meta")
    return dict(counts=counts, offsets=offsets, perm=perm, pos=pos, posw=posw, poffsets=poffsets, prow=prow)


def rmsnorm(x, g, cond=None, rows_per_cond=1, eps=1e-6, lp_dtype=torch.bfloat16, want_f32=True):
    lib = L.load()
    rows, D = x.shape
    y32 = torch.empty_like(x) if want_f32 else None
    ylp = torch.empty(rows, D, dtype=lp_dtype, device=x.device)
    L.check(lib.mode_rmsnorm_cond_fwd(p(x), p(g), p(cond), rows, D, rows_per_cond, eps, p(y32), p(ylp),
                                      L.MODE_BF16 if lp_dtype == torch.bfloat16 else L.MODE_F32, stream()), "rmsnorm")
    return y32, ylp


def attn(qkv, qg, kg, B, T, H, hd, eps=1e-6, seed=0, p_drop=0.0):
    lib = L.load()
    y = torch.full((B * T, H * hd), float("nan"), dtype=qkv.dtype, device=qkv.device)
    L.check(lib.mode_attn_block_fwd(p(qkv), p(qg), p(kg), p(y), dt_of(qkv), B, T, H, hd, eps, seed, p_drop, stream()), "attn")
    return y


def qkv_attn(h, wqkv, bqkv, qg, kg, B, T, H, eps=1e-6):
    """mode_qkv_attn_fwd: returns (status, y) - the caller decides what an UNSUPPORTED (-2) shape means."""
    lib = L.load()
    D = h.shape[1]
    y = torch.full((B * T, D), float("nan"), dtype=h.dtype, device=h.device)
    d = L.ModeQkvAttnDesc(dtype=dt_of(h), B=B, T=T, H=H, D=D, h=p(h), ldh=h.stride(0), wqkv=p(wqkv), ldw=wqkv.stride(0), bqkv=p(bqkv), q_gain=p(qg),
                          k_gain=p(kg), eps=eps, y=p(y), ldy=y.stride(0))
    return lib.mode_qkv_attn_fwd(C.byref(d), stream()), y


def combine_norm(u, Y, pos, posw, k, g, cond, rows_per_cond, eps=1e-6, h_dtype=torch.bfloat16):
    lib = L.load()
    N, D = u.shape
    xn = torch.empty_like(u)
    h = torch.empty(N, D, dtype=h_dtype, device=u.device)
    L.check(lib.mode_moe_combine_norm_fwd(p(u), p(Y), dt_of(Y), 1, 0, p(pos), p(posw), N, D, k, p(g), p(cond), rows_per_cond, eps, p(xn), p(h),
                                          L.MODE_BF16 if h_dtype == torch.bfloat16 else L.MODE_F32, stream()), "combine_norm")
    return xn, h
