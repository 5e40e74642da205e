"""GPU A/B of the ring-kernel tile geometries on the QKV projection [14*B x 1024] x [3072 x 1024]^T (+ bias, bf16 out): bit-identity against
gemm_cfg 1 (128x128), then interleaved timing rounds.  Usage (GPU box): python scripts/qkv_tile_probe.py [--batch 128] [--cfgs 8,19,1,4]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--cfgs", default="8,19,1,4")
    ap.add_argument("--reps", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda:0")
    D, N = 1024, 14 * a.batch
    bf = torch.bfloat16
    torch.manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(N, D, device=dev).to(bf)
    nl = 12
    w = [torch.randn(3 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]
    b = torch.randn(3 * D, device=dev)
    out = torch.empty(N, 3 * D, dtype=bf, device=dev)

    def desc(i, o):
        return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_BIAS, out_dtype=L.MODE_BF16, M=N, N=3 * D, K=D, A=x.data_ptr(), lda=D, W=w[i].data_ptr(), ldw=D,
                              bias=b.data_ptr(), C=o.data_ptr(), ldc=3 * D)

    cfgs = [int(c) for c in a.cfgs.split(",")]
    lib.mode_set_option(b"gemm_cfg", 1)
    ref = torch.zeros_like(out); L.check(lib.mode_gemm(C.byref(desc(0, ref)), st)); torch.cuda.synchronize()
    for c in cfgs:
        lib.mode_set_option(b"gemm_cfg", c)
        o = torch.full_like(out, float("nan")); L.check(lib.mode_gemm(C.byref(desc(0, o)), st)); torch.cuda.synchronize()
        print(f"cfg {c}: bit-identical to cfg 1: {torch.equal(o.view(torch.int16), ref.view(torch.int16))}")
    ds = [desc(i, out) for i in range(nl)]
    res = {c: [] for c in cfgs}
    for r in range(a.rounds):
        for c in cfgs:
            lib.mode_set_option(b"gemm_cfg", c)
            for d in ds:
                lib.mode_gemm(C.byref(d), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps):
                lib.mode_gemm(C.byref(ds[i % nl]), st)
            e1.record(); torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    lib.mode_set_option(b"gemm_cfg", 0)
    fl = 2.0 * N * D * 3 * D
    for c, v in res.items():
        print(f"cfg {c:2d}: med {sorted(v)[len(v) // 2]:6.2f} min {min(v):6.2f} us ({fl / min(v) / 1e6:5.0f} TF/s)")


if __name__ == "__main__":
    main()
