"""Round-5 experiment: the L2 run-ahead touch of the persistent ping-pong GEMM (scripts/probe/gemm_bf16_pp_l2touch.hip, -DPP_L2_TOUCH=d [-DPP_L2_TOUCH_SPLIT=1]) against the
shipped kernel, in ONE process (every library is its own ctypes handle), interleaved rounds, the config-2 expert shapes at B = 128:
  * up-projection   [3584 x 1024] x [2 x 8192 x 1024]  (SwiGLU + fused ln_2, gathered identity rows, uniform routing)
  * down-projection [3584 x 4096] x [1024 x 4096] in 4 K-slices
each with COLD weights (12 layers' matrices cycled: 400 / 200 MB, beyond the 256-MB Infinity Cache - what the chain presents) and WARM weights (one
layer's matrix every launch: Infinity-Cache resident after the first pass).  Outputs are compared bit for bit with the shipped kernel first.
Usage (GPU box): python scripts/pp_l2touch_probe.py t2 t4 s2 s4     (tags of scripts/build_pp_variant.sh)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mode_diffusion_policy_amd")


def open_lib(path):
    lib = C.CDLL(path)
    for name, (res, args) in L.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    tags = sys.argv[1:] or ["t2", "t4", "s2", "s4"]
    libs = {"base": L.load()}
    for t in tags:
        libs[t] = open_lib(os.path.join(HERE, f"libmode_hip_{t}.so"))
    dev = torch.device("cuda:0")
    B, D, E, k = 128, 1024, 4, 2
    N = 14 * B; NK = N * k
    bf = torch.bfloat16
    torch.manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(N, D, device=dev).to(bf)
    ssb = (x.float() ** 2).view(N, D // 64, 64).sum(-1).contiguous()
    nl = 12
    w1 = [torch.randn(E, 8 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]; b1 = torch.randn(E, 8 * D, device=dev)
    w2 = [torch.randn(E, D, 4 * D, device=dev).to(bf) * 0.015 for _ in range(nl)]
    hin = torch.randn(NK, 4 * D, device=dev).to(bf)
    base = libs["base"]
    idx = torch.tensor([[1, 2]] * B, dtype=torch.int32, device=dev); w = torch.full((B, k), 0.5, device=dev)
    ml = L.ModeMetaLayout(); base.mode_moe_meta_layout(N, E, k, C.byref(ml))
    meta = torch.empty(ml.total_words, dtype=torch.int32, device=dev)
    L.check(base.mode_dit_dispatch(idx.data_ptr(), w.data_ptr(), 1, B * k, B, 14, N, E, k, meta.data_ptr(), st))
    mp = meta.data_ptr()

    def d1(wi, out):
        return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_SWIGLU, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=x.data_ptr(), lda=D, W=w1[wi].data_ptr(), ldw=D,
                              w_expert_stride=8 * D * D, bias=b1.data_ptr(), bias_expert_stride=8 * D, C=out.data_ptr(), ldc=4 * D, a_rows=mp + 4 * ml.perm,
                              expert_offsets=mp + 4 * ml.offsets, num_experts=E, row_ss=ssb.data_ptr(), row_ss_n=D // 64, row_eps=1e-6,
                              flags=L.GEMM_UNIFORM_GROUPS | L.GEMM_IDENTITY_ROWS)

    def d2(wi, out):
        return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=NK, N=D, K=4 * D, A=hin.data_ptr(), lda=4 * D, W=w2[wi].data_ptr(), ldw=4 * D,
                              w_expert_stride=4 * D * D, C=out.data_ptr(), ldc=D, expert_offsets=mp + 4 * ml.offsets, num_experts=E, split_k=4, split_stride=NK * D,
                              flags=L.GEMM_UNIFORM_GROUPS)

    o1 = torch.empty(NK, 4 * D, dtype=bf, device=dev); o2 = torch.empty(4, NK, D, dtype=bf, device=dev)
    ref = {}
    for name, lib in libs.items():
        for it in range(6):
            a = torch.full_like(o1, float("nan")); b = torch.full_like(o2, float("nan"))
            L.check(lib.mode_gemm(C.byref(d1(it % nl, a)), st)); L.check(lib.mode_gemm(C.byref(d2(it % nl, b)), st))
            torch.cuda.synchronize()
            key = it % nl
            if name == "base":
                ref[key] = (a.clone(), b.clone())
            else:
                assert torch.equal(a.view(torch.int16), ref[key][0].view(torch.int16)), (name, "up", it)
                assert torch.equal(b.view(torch.int16), ref[key][1].view(torch.int16)), (name, "down", it)
    print("bit-identical to the shipped kernel:", ", ".join(t for t in libs if t != "base"))

    reps, rounds = 48, 7
    cases = {"up   cold": (lambda i: d1(i % nl, o1), 2.0 * NK * D * 8 * D), "up   warm": (lambda i: d1(0, o1), 2.0 * NK * D * 8 * D),
             "down cold": (lambda i: d2(i % nl, o2), 2.0 * NK * 4 * D * D), "down warm": (lambda i: d2(0, o2), 2.0 * NK * 4 * D * D)}
    for cname, (mk, fl) in cases.items():
        ds = [mk(i) for i in range(nl)]
        res = {t: [] for t in libs}
        for r in range(rounds):
            for t, lib in libs.items():
                for d in ds:
                    lib.mode_gemm(C.byref(d), st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    lib.mode_gemm(C.byref(ds[i % nl]), st)
                e1.record(); torch.cuda.synchronize()
                res[t].append(e0.elapsed_time(e1) * 1e3 / reps)
        print(f"{cname}: " + "   ".join(f"{t}: med {sorted(v)[len(v) // 2]:6.2f} min {min(v):6.2f} us ({fl / sorted(v)[len(v) // 2] / 1e6:5.0f} TF/s)" for t, v in res.items()))


if __name__ == "__main__":
    main()
