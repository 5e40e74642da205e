"""K-sweep of the bf16 GEMM: fixed overhead vs per-K-step cost.  python scripts/gemm_ksweep.py --cfg N"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L

ap = argparse.ArgumentParser(); ap.add_argument("--cfg", type=int, default=0); ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
lib = L.load(); lib.mode_set_option(b"gemm_cfg", a.cfg)
dev = torch.device("cuda:0"); bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
for (M, N, epi, name) in [(1792, 3072, L.EPI_BIAS, "qkv-like"), (1792, 1024, L.EPI_NONE, "cproj-like"), (3584, 4096, L.EPI_SWIGLU, "gemm1-like"), (3584, 1024, L.EPI_NONE, "gemm2-like")]:
    res = []
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        nl = 4
        A = torch.randn(M, K, device=dev).to(bf)
        Ws = [torch.randn((2 * N if epi == L.EPI_SWIGLU else N), K, device=dev).to(bf) for _ in range(nl)]
        b = torch.randn(2 * N, device=dev)
        out = torch.empty(M, N, dtype=bf, device=dev)
        ds = [L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=epi, out_dtype=L.MODE_BF16, M=M, N=N, K=K, A=A.data_ptr(), lda=K, W=w.data_ptr(), ldw=K,
                             w_expert_stride=0, bias=b.data_ptr(), bias_expert_stride=0, resid=None, ldr=0, C=out.data_ptr(), ldc=N, a_rows=None,
                             expert_offsets=None, num_experts=0) for w in Ws]
        for d in ds: L.check(lib.mode_gemm(C.byref(d), st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.reps): lib.mode_gemm(C.byref(ds[i % nl]), st)
        e1.record(); torch.cuda.synchronize()
        res.append((K, e0.elapsed_time(e1) * 1e3 / a.reps))
    print(f"cfg{a.cfg} {name:11s} M={M} N={N}: " + "  ".join(f"K{k}:{t:6.1f}us" for k, t in res))
