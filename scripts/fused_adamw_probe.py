"""Where the time of the weight-gradient GEMM with the AdamW epilogue (ModeAdamWFuse) goes, at the config-2 expert shapes (B = 128: 3584 sorted rows, four
ragged expert segments): the fused launch, the same launch with EMPTY K ranges (= its epilogue alone: pure p / m / v / shadow streaming in 128 x 128 tiles),
the plain weight-gradient GEMM, and mode_adamw_step over the same number of elements (the streaming pass it replaces).  Usage: python scripts/fused_adamw_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
D, E, NK = 1024, 4, 3584
bf = torch.bfloat16
torch.manual_seed(0)
offs = torch.tensor([0, 871, 1796, 2699, 3584], dtype=torch.int32, device=dev)
zero = torch.zeros(E + 1, dtype=torch.int32, device=dev)
nl = 6                                                              # cycle distinct parameter slices: 6 x (134 + 67 M) x 4 arenas >> Infinity Cache
shapes = {"dW1 [4 x 8192 x 1024]": (8 * D, D), "dW2 [4 x 1024 x 4096]": (D, 4 * D)}
for name, (M_, N_) in shapes.items():
    n = E * M_ * N_
    dY = torch.randn(NK, M_, device=dev).to(bf); X = torch.randn(NK, N_, device=dev).to(bf)
    P = [torch.randn(n, device=dev) * 0.02 for _ in range(nl)]; Mo = [torch.zeros(n, device=dev) for _ in range(nl)]; V = [torch.zeros(n, device=dev) for _ in range(nl)]
    LP = [torch.zeros(n, dtype=bf, device=dev) for _ in range(nl)]; G = [torch.empty(n, device=dev) for _ in range(nl)]

    gsq = torch.zeros(4096, device=dev)

    def fz(i):
        return L.ModeAdamWFuse(gsq=gsq.data_ptr(), gsq_capacity=4096, grad_base=G[i].data_ptr(), param_base=P[i].data_ptr(), exp_avg_base=Mo[i].data_ptr(), exp_avg_sq_base=V[i].data_ptr(), lp_base=LP[i].data_ptr(),
                               lr=1e-4, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.05, step=3, grad_scale=1.0)

    def desc(i, ko, fused):
        f = fz(i) if fused else None
        d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=M_, N=N_, K=NK, A=dY.data_ptr(), lda=M_, W=X.data_ptr(), ldw=N_, C=G[i].data_ptr(),
                           ldc=N_, k_group_offsets=ko.data_ptr(), num_k_groups=E, c_group_stride=M_ * N_, flags=L.GEMM_W_KN | L.GEMM_A_KM, adamw=None if f is None else C.pointer(f))
        return d, f
    rows = torch.randint(0, NK // 2, (NK,), dtype=torch.int32, device=dev)       # dW1 in the model gathers its W rows through the dispatch permutation

    def desc_g(i, ko):
        d, f = desc(i, ko, True)
        d.w_rows = rows.data_ptr()
        return d, f
    cases = {}
    # 0 = the ring kernel (round 5); else the wave-specialised persistent kernel (round 6) with 10 x ring slots + chunks per stream batch
    has_ws = lib.mode_set_option(b"adamw_ws", 1) == 0             # only the variant built by scripts/probe/build_trws_variant.sh knows the option
    for ws in ((0, 54) if has_ws else (0,)):
        tag = f"ws {ws // 10} slots, UB {ws % 10}" if ws else "ring"
        cases[f"fused GEMM + AdamW [{tag}]"] = (ws, lambda i: desc(i, offs, True))
        cases[f"fused, gathered W rows [{tag}]"] = (ws, lambda i: desc_g(i, offs))
        cases[f"epilogue alone (K = 0) [{tag}]"] = (ws, lambda i: desc(i, zero, True))
    if has_ws:
        cases["GEMM group alone [ws 5 slots, dbg]"] = (-54, lambda i: desc_g(i, offs))
        cases["GEMM group alone, no DMA [dbg 2]"] = (-1054, lambda i: desc_g(i, offs))
        cases["ws 5 slots, phase stamps [dbg 4]"] = (-3054, lambda i: desc_g(i, offs))
    cases["plain GEMM (ring)"] = (0, lambda i: desc(i, offs, False))
    lib.mode_set_option(b"gemm_tr_cfg", 7)                              # the ring kernels for the plain GEMM (what the fused launch is built on)
    for cname, (ws, mk) in cases.items():
        lib.mode_set_option(b"adamw_ws", 1 if ws else 0); lib.mode_set_option(b"adamw_ws_cfg", abs(ws) % 1000); lib.mode_set_option(b"adamw_ws_dbg", (1 + abs(ws) // 1000) if ws < 0 else 0)
        ds = [mk(i) for i in range(nl)]
        for d, _ in ds:
            L.check(lib.mode_gemm(C.byref(d), st), cname)
        torch.cuda.synchronize()
        best = 1e9
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(24):
                lib.mode_gemm(C.byref(ds[i % nl][0]), st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 24)
        if "dbg 4" in cname:
            ph = gsq[256:264].tolist()
            steps = max(ph[7], 1.0)
            print("    K-loop phases of wave 0 / workgroup 0, s_memtime ticks per step: " + "  ".join(f"{nm} {v / steps:.0f}" for nm, v in zip(
                ["waits", "signal+mma0", "spin", "reads-issue", "mma1", "wait-lgkm", "dma-issue+advance"], ph[:7])) + f"   ({steps:.0f} steps)")
        byts = n * 26 if "plain" not in cname else n * 4
        print(f"{name:24s} {cname:48s} {best:7.1f} us   {byts / best / 1e6:6.2f} TB/s of optimizer / gradient traffic")
    lib.mode_set_option(b"gemm_tr_cfg", 0); lib.mode_set_option(b"adamw_ws", 1); lib.mode_set_option(b"adamw_ws_cfg", 0)
    best = 1e9
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            j = i % nl
            lib.mode_adamw_step(P[j].data_ptr(), G[j].data_ptr(), Mo[j].data_ptr(), V[j].data_ptr(), n, 1e-4, 0.9, 0.95, 1e-8, 0.05, 3, 1.0, LP[j].data_ptr(), None, 0.0, st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 24)
    print(f"{name:24s} {'mode_adamw_step (30 B/el)':26s} {best:7.1f} us   {n * 30 / best / 1e6:6.2f} TB/s")
    del P, Mo, V, LP, G
    torch.cuda.empty_cache()
