"""GPU probe of the persistent ping-pong GEMM (gemm_bf16_pp.hip) at the config-2 expert shapes: bit-identity against the 128x128 kernel
(repeated launches = race screen), then interleaved timing rounds of the tile configurations.
Usage (GPU box): python scripts/pp_probe.py [--reps 30] [--rounds 5] [--screen 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--screen", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--cfgs", default="0,17", help="gemm_cfg values to time: 17 = persistent ping-pong kernel, 0 = the 128x128 family (gemm_pp off), 1 / 13 ... = one ring geometry")
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda:0")
    D, N, E, k = 1024, 14 * a.batch, 4, 2
    NK = N * k
    bf = torch.bfloat16
    torch.manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(N, D, device=dev).to(bf)
    ssb = (x.float() ** 2).view(N, D // 64, 64).sum(-1).contiguous()
    nl = 6
    w1 = [torch.randn(E, 8 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]; b1 = torch.randn(E, 8 * D, device=dev)
    w2 = [torch.randn(E, D, 4 * D, device=dev).to(bf) * 0.015 for _ in range(nl)]
    hin = torch.randn(NK, 4 * D, device=dev).to(bf)

    def meta_for(pairs):
        idx = torch.tensor(pairs, dtype=torch.int32, device=dev); w = torch.full((a.batch, k), 0.5, device=dev)
        ml = L.ModeMetaLayout(); lib.mode_moe_meta_layout(N, E, k, C.byref(ml))
        meta = torch.empty(ml.total_words, dtype=torch.int32, device=dev)
        L.check(lib.mode_dit_dispatch(idx.data_ptr(), w.data_ptr(), 1, a.batch * k, a.batch, 14, N, E, k, meta.data_ptr(), st))
        return meta, ml

    g = torch.Generator().manual_seed(5)
    ragged = [sorted(torch.randperm(E, generator=g)[:k].tolist()) for _ in range(a.batch)]
    routings = {"uniform(2 experts)": [[1, 2]] * a.batch, "ragged(4 experts)": ragged}

    def gemm1_desc(meta, ml, wi, out):
        mp = meta.data_ptr()
        return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_SWIGLU, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=x.data_ptr(), lda=D, W=w1[wi].data_ptr(), ldw=D,
                              w_expert_stride=8 * D * D, bias=b1.data_ptr(), bias_expert_stride=8 * D, C=out.data_ptr(), ldc=4 * D, a_rows=mp + 4 * ml.perm,
                              expert_offsets=mp + 4 * ml.offsets, num_experts=E, row_ss=ssb.data_ptr(), row_ss_n=D // 64, row_eps=1e-6,
                              flags=int(os.environ.get("PP_DESC_FLAGS", "0")))

    def gemm2_desc(meta, ml, wi, out, S):
        mp = meta.data_ptr()
        return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=NK, N=D, K=4 * D, A=hin.data_ptr(), lda=4 * D, W=w2[wi].data_ptr(), ldw=4 * D,
                              w_expert_stride=4 * D * D, C=out.data_ptr(), ldc=D, expert_offsets=mp + 4 * ml.offsets, num_experts=E, split_k=S, split_stride=NK * D)

    def setv(v):                                                # (round 2's "NNfK" ablation variants no longer exist: the product ships no pp_flags)
        lib.mode_set_option(b"gemm_cfg", int(v))
        lib.mode_set_option(b"gemm_pp", 0 if int(v) == 0 else 1)

    def run(cfg, d):
        setv(cfg)
        L.check(lib.mode_gemm(C.byref(d), st))
        setv(0)
        lib.mode_set_option(b"gemm_pp", 1)

    ok = True
    for name, pairs in routings.items():
        meta, ml = meta_for(pairs)
        ref1 = torch.zeros(NK, 4 * D, dtype=bf, device=dev); run(1, gemm1_desc(meta, ml, 0, ref1))
        for cfg in ("17",):
            bad = 0
            for it in range(a.screen):
                out = torch.full((NK, 4 * D), float("nan"), dtype=bf, device=dev)
                run(cfg, gemm1_desc(meta, ml, 0, out))
                if not torch.equal(out.view(torch.int16), ref1.view(torch.int16)):
                    bad += 1
                    if bad == 1:
                        diff = (out.float() - ref1.float()); nz = diff.ne(0) | out.float().isnan()
                        rows = nz.any(1).nonzero().flatten(); cols = nz.any(0).nonzero().flatten()
                        print(f"  MISMATCH gemm1 cfg {cfg} {name}: {int(nz.sum())} elements, rows {rows[:8].tolist()}..{rows[-3:].tolist()} ({rows.numel()}), "
                              f"cols {cols[:8].tolist()}..{cols[-3:].tolist()} ({cols.numel()}), max |d| {float(diff.nan_to_num(1e9).abs().max()):.3g}")
            print(f"gemm1 cfg {cfg} {name}: {a.screen - bad}/{a.screen} launches bit-identical to the 128x128 kernel")
            ok &= bad == 0
        for S in (2, 4):
            ref2 = torch.zeros(S, NK, D, dtype=bf, device=dev); run(1, gemm2_desc(meta, ml, 0, ref2, S))
            for cfg in ("17",):
                bad = 0
                for it in range(a.screen):
                    out = torch.full((S, NK, D), float("nan"), dtype=bf, device=dev)
                    run(cfg, gemm2_desc(meta, ml, 0, out, S))
                    if not torch.equal(out.view(torch.int16), ref2.view(torch.int16)):
                        bad += 1
                        if bad == 1:
                            nz = (out.float() - ref2.float()).ne(0) | out.float().isnan()
                            print(f"  MISMATCH gemm2 S={S} cfg {cfg} {name}: {int(nz.sum())} elements")
                print(f"gemm2 S={S} cfg {cfg} {name}: {a.screen - bad}/{a.screen} bit-identical")
                ok &= bad == 0
    print("BIT-IDENTITY", "OK" if ok else "FAILED")

    # ---- timing: interleaved rounds (variants x rounds in one process), weights cycled so they are not L2-hot between launches
    cfgs = a.cfgs.split(",")
    for name, pairs in routings.items():
        meta, ml = meta_for(pairs)
        out1 = torch.empty(NK, 4 * D, dtype=bf, device=dev)
        cases = {"gemm1 swiglu+ln2 [3584x1024]x[8192x1024]": (lambda wi: gemm1_desc(meta, ml, wi, out1), 2.0 * NK * D * 8 * D)}
        for S in (2, 4):
            o2 = torch.empty(S, NK, D, dtype=bf, device=dev)
            cases[f"gemm2 S={S} [3584x4096]x[1024x4096]"] = (lambda wi, o2=o2, S=S: gemm2_desc(meta, ml, wi, o2, S), 2.0 * NK * 4 * D * D)
        for cname, (mk, fl) in cases.items():
            ds = [mk(i) for i in range(nl)]
            res = {c: [] for c in cfgs}
            for r in range(a.rounds):
                for c in cfgs:
                    setv(c)
                    for d in ds:
                        lib.mode_gemm(C.byref(d), st)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(a.reps):
                        lib.mode_gemm(C.byref(ds[i % nl]), st)
                    e1.record(); torch.cuda.synchronize()
                    res[c].append(e0.elapsed_time(e1) * 1e3 / a.reps)
            setv(0); lib.mode_set_option(b"gemm_pp", 1)
            line = "  ".join(f"{c}: med {sorted(v)[len(v) // 2]:6.1f} min {min(v):6.1f} us ({fl / min(v) / 1e6:6.0f} TF/s)" for c, v in res.items())
            print(f"{name:20s} {cname:44s} {line}")

    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
