"""A/B of the fused QKV + attention launch (qkv_attn.hip) against the two kernels it replaces, per batch size, each as a hipGraph of `reps` launches
(HIP events on the replay stream), interleaved rounds in ONE process.  Prints microseconds per launch.

    python scripts/qkv_attn_probe.py [--batches 32,64,96,128] [--rounds 5]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402
from mode_diffusion_policy_amd.engine import capture_graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32,48,64,96,128")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=120)
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda", 0)
    D, H, T, Ly = 1024, 8, 14, 12
    bf = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)
    W = [(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(bf).to(dev) for _ in range(Ly)]
    bias = [torch.randn(3 * D, generator=g).to(dev) * 0.1 for _ in range(Ly)]
    qg = torch.ones(D // H, device=dev); kg = torch.ones(D // H, device=dev)
    st_of = lambda: torch.cuda.current_stream().cuda_stream
    for B in [int(b) for b in a.batches.split(",")]:
        N = B * T
        h = torch.randn(N, D, generator=g).to(bf).to(dev)
        qkv = torch.empty(N, 3 * D, dtype=bf, device=dev); y = torch.empty(N, D, dtype=bf, device=dev); y2 = torch.empty(N, D, dtype=bf, device=dev)

        def fused(i, st, out=y):
            l = i % Ly
            d = L.ModeQkvAttnDesc(dtype=L.MODE_BF16, B=B, T=T, H=H, D=D, h=h.data_ptr(), ldh=D, wqkv=W[l].data_ptr(), ldw=D, bqkv=bias[l].data_ptr(),
                                  q_gain=qg.data_ptr(), k_gain=kg.data_ptr(), eps=1e-6, y=out.data_ptr(), ldy=D)
            L.check(lib.mode_qkv_attn_fwd(C.byref(d), st))

        def two(i, st):
            l = i % Ly
            d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_BIAS, out_dtype=L.MODE_BF16, M=N, N=3 * D, K=D, A=h.data_ptr(), lda=D, W=W[l].data_ptr(), ldw=D,
                               bias=bias[l].data_ptr(), C=qkv.data_ptr(), ldc=3 * D)
            L.check(lib.mode_gemm(C.byref(d), st))
            L.check(lib.mode_attn_block_fwd(qkv.data_ptr(), qg.data_ptr(), kg.data_ptr(), y2.data_ptr(), L.MODE_BF16, B, T, H, D // H, 1e-6, 0, 0.0, st))

        graphs = {}
        for name, fn, opt in (("4w/w3", fused, (4, 1)), ("4w/w2", fused, (4, 0)), ("8w/w3", fused, (8, 1)), ("8w/w2", fused, (8, 0)), ("gemm+attn", two, None)):
            if opt is not None:
                lib.mode_set_option(b"qkv_attn_waves", opt[0]); lib.mode_set_option(b"qkv_attn_w3", opt[1])
            for i in range(Ly):
                fn(i, st_of())
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with capture_graph(gr):
                cst = st_of()
                for i in range(a.reps):
                    fn(i, cst)
            gr.replay(); torch.cuda.synchronize()
            graphs[name] = gr
        lib.mode_set_option(b"qkv_attn_w3", 1); lib.mode_set_option(b"qkv_attn_waves", 4)
        assert torch.equal(y, y2), "fused result differs from the two kernels"
        best = {k: 1e9 for k in graphs}
        for _ in range(a.rounds):
            for name, gr in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                best[name] = min(best[name], e0.elapsed_time(e1) * 1e3 / a.reps)
        print(f"B={B:4d} ({N} rows): " + "  ".join(f"{k} {v:6.2f} us" for k, v in best.items()), flush=True)


if __name__ == "__main__":
    main()
