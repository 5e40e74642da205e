"""GPU micro-benchmark of the bf16 MFMA GEMM at the four per-layer shapes of BASELINE config 2 (B=128 -> N=1792 tokens).
Usage (GPU box): python tools/gemm_bench.py [--reps 50] [--opt key=value ...]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--only", default="")
    ap.add_argument("--splitk", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="samples (tokens = 14 x batch)")
    ap.add_argument("--experts", type=int, default=2, help="active experts: 2 = uniform-sigma inference (all samples route alike), 4 = training-like")
    ap.add_argument("--y-bf16", action="store_true", help="gemm2 writes bf16 (the product path) instead of fp32")
    ap.add_argument("--rowss", action="store_true", help="gemm1 consumes per-row partial sums of squares (fused ln_2)")
    ap.add_argument("--nogather", action="store_true", help="gemm1 reads pre-sorted rows (no a_rows gather)")
    a = ap.parse_args()
    lib = L.load()
    for o in a.opt:
        k, v = o.split("=")
        assert lib.mode_set_option(k.encode(), int(v)) == 0, o
    dev = torch.device("cuda:0")
    D, N, E, k = 1024, 14 * a.batch, 4, 2
    NK = N * k
    bf = torch.bfloat16
    x = torch.randn(N, D, device=dev).to(bf)
    pairs = [[1, 2], [0, 3]] if a.experts == 4 else [[1, 2], [1, 2]]
    idx = torch.tensor(pairs * (a.batch // 2), dtype=torch.int32, device=dev); w = torch.tensor([[0.6, 0.4]] * a.batch, device=dev)
    ml = L.ModeMetaLayout(); lib.mode_moe_meta_layout(N, E, k, C.byref(ml))
    meta = torch.empty(ml.total_words, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.mode_dit_dispatch(idx.data_ptr(), w.data_ptr(), 1, a.batch * k, a.batch, 14, N, E, k, meta.data_ptr(), st))
    mp = meta.data_ptr()
    nl = 6   # cycle through several weight sets so weights are not L2-hot between launches (as in the real layer loop)
    wqkv = [torch.randn(3 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]; bqkv = torch.randn(3 * D, device=dev)
    wo = [torch.randn(D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]
    w1 = [torch.randn(E, 8 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]; b1 = torch.randn(E, 8 * D, device=dev)
    w2 = [torch.randn(E, D, 4 * D, device=dev).to(bf) * 0.015 for _ in range(nl)]
    qkv = torch.empty(N, 3 * D, dtype=bf, device=dev); xr = torch.randn(N, D, device=dev); xo = torch.empty(N, D, device=dev)
    Hb = torch.empty(NK, 4 * D, dtype=bf, device=dev); Y = torch.empty(max(a.splitk, 1), NK, D, device=dev)
    hin = torch.randn(NK, 4 * D, device=dev).to(bf)
    xs = torch.randn(NK, D, device=dev).to(bf)           # pre-sorted (duplicated) expert inputs

    ssb = torch.rand(N, D // 64, device=dev) + 0.5
    ss_kw = dict(row_ss=ssb.data_ptr(), row_ss_n=D // 64, row_eps=1e-6) if a.rowss else {}

    def desc(**kw):
        base = dict(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=N, N=D, K=D, A=x.data_ptr(), lda=D, W=None, ldw=D,
                    w_expert_stride=0, bias=None, bias_expert_stride=0, resid=None, ldr=0, C=None, ldc=D, a_rows=None, expert_offsets=None,
                    num_experts=0, split_k=1, split_stride=0, flags=0)
        if kw.get("epilogue") == L.EPI_SWIGLU:
            base.update(ss_kw)
        base.update(kw)
        return L.ModeGemmDesc(**base)
    shapes = {
        "qkv   [1792x1024]x[3072x1024]": ([desc(epilogue=L.EPI_BIAS, N=3 * D, W=wqkv[i].data_ptr(), bias=bqkv.data_ptr(), C=qkv.data_ptr(), ldc=3 * D)
                                           for i in range(nl)], 2.0 * N * D * 3 * D),
        "cproj [1792x1024]x[1024x1024]": ([desc(epilogue=L.EPI_RESIDUAL, out_dtype=L.MODE_F32, W=wo[i].data_ptr(), resid=xr.data_ptr(), ldr=D,
                                                 C=xo.data_ptr()) for i in range(nl)], 2.0 * N * D * D),
        "gemm1 grouped swiglu [3584x1024]x[2x8192x1024]": ([desc(epilogue=L.EPI_SWIGLU, M=NK, N=4 * D, W=w1[i].data_ptr(), w_expert_stride=8 * D * D,
                                                                   bias=b1.data_ptr(), bias_expert_stride=8 * D, C=Hb.data_ptr(), ldc=4 * D,
                                                                   a_rows=(None if a.nogather else mp + 4 * ml.perm), expert_offsets=mp + 4 * ml.offsets, num_experts=E,
                                                                   **(dict(A=xs.data_ptr()) if a.nogather else {})) for i in range(nl)], 2.0 * NK * D * 8 * D),
        "gemm2 grouped [3584x4096]x[2x1024x4096]": ([desc(out_dtype=(L.MODE_BF16 if a.y_bf16 else L.MODE_F32), M=NK, N=D, K=4 * D, A=hin.data_ptr(), lda=4 * D, W=w2[i].data_ptr(), ldw=4 * D,
                                                            w_expert_stride=4 * D * D, C=Y.data_ptr(), expert_offsets=mp + 4 * ml.offsets, num_experts=E, split_k=a.splitk,
                                                            split_stride=NK * D) for i in range(nl)], 2.0 * NK * 4 * D * D),
    }
    tot_us = 0.0
    for name, (ds, fl) in shapes.items():
        if a.only and a.only not in name:
            continue
        for d in ds:
            L.check(lib.mode_gemm(C.byref(d), st))
        torch.cuda.synchronize()
        if True:
            evs = []
            for i in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.mode_gemm(C.byref(ds[i % nl]), st); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            us = sum(x.elapsed_time(y) for x, y in evs) * 1e3 / a.reps
        tot_us += us
        print(f"{name:52s} {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s")
    print(f"sum {tot_us:.1f} us/layer  opts={a.opt}")


if __name__ == "__main__":
    main()
