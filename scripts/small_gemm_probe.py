"""Isolated timing of the small-batch chain's weight-streaming GEMMs (B=1: 14 tokens): hipGraph of launches cycling 12 layers' weights.
Variants via pp_flags (64 = coalesced-read timing probe, wrong results).  Usage: python scripts/small_gemm_probe.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402
from mode_diffusion_policy_amd.engine import capture_graph  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = L.load()
dev = torch.device("cuda:0")
D, E, k, T = 1024, 4, 2, 14
N, NK = B * T, B * T * k
bf = torch.bfloat16
torch.manual_seed(0)
nl = 12
x = torch.randn(N, D, device=dev).to(bf)
ss = torch.rand(N, D // 16, device=dev) + 0.5
wqkv = [torch.randn(3 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]; bq = torch.randn(3 * D, device=dev)
wo = [torch.randn(D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]
w1 = [torch.randn(E, 8 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]; b1 = torch.randn(E, 8 * D, device=dev)
w2 = [torch.randn(E, D, 4 * D, device=dev).to(bf) * 0.015 for _ in range(nl)]
hin = torch.randn(NK, 4 * D, device=dev).to(bf)
xr = torch.randn(N, D, device=dev); g2 = torch.ones(D, device=dev)
idx = torch.tensor([[1, 2]] * B, dtype=torch.int32, device=dev); w = torch.full((B, k), 0.5, device=dev)
ml = L.ModeMetaLayout(); lib.mode_moe_meta_layout(N, E, k, C.byref(ml))
meta = torch.empty(ml.total_words, dtype=torch.int32, device=dev)
st0 = torch.cuda.current_stream().cuda_stream
L.check(lib.mode_dit_dispatch(idx.data_ptr(), w.data_ptr(), 1, B * k, B, T, N, E, k, meta.data_ptr(), st0))
mp = meta.data_ptr()
FL = L.GEMM_SMALL_ROWS
qkv = torch.empty(N, 3 * D, dtype=bf, device=dev); xo = torch.empty(N, D, device=dev); h2 = torch.empty(N, D, dtype=bf, device=dev)
sso = torch.empty(N, D // 16, device=dev); hb = torch.empty(NK, 4 * D, dtype=bf, device=dev)
cases = {}
cases["qkv [14x1024]x[3072x1024] 6.3MB"] = (lambda i: L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_BIAS, out_dtype=L.MODE_BF16, M=N, N=3 * D, K=D, A=x.data_ptr(), lda=D,
    W=wqkv[i].data_ptr(), ldw=D, bias=bq.data_ptr(), C=qkv.data_ptr(), ldc=3 * D, flags=FL), 6.3)
cases["c_proj+resid+ln2 2.1MB"] = (lambda i: L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_RESIDUAL_NORM, out_dtype=L.MODE_F32, M=N, N=D, K=D, A=x.data_ptr(), lda=D,
    W=wo[i].data_ptr(), ldw=D, resid=xr.data_ptr(), ldr=D, C=xo.data_ptr(), ldc=D, C2=h2.data_ptr(), ldc2=D, gain=g2.data_ptr(), row_ss_out=sso.data_ptr(), flags=FL), 2.1)
cases["up swiglu+ln2 33.5MB"] = (lambda i: L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_SWIGLU, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=x.data_ptr(), lda=D,
    W=w1[i].data_ptr(), ldw=D, w_expert_stride=8 * D * D, bias=b1.data_ptr(), bias_expert_stride=8 * D, C=hb.data_ptr(), ldc=4 * D, a_rows=mp + 4 * ml.perm,
    expert_offsets=mp + 4 * ml.offsets, num_experts=E, row_ss=ss.data_ptr(), row_ss_n=D // 16, row_eps=1e-6, flags=FL), 33.5)
for S in (2, 4):
    Y = torch.empty(S, NK, D, dtype=bf, device=dev)
    cases[f"down S={S} 16.8MB"] = (lambda i, S=S, Y=Y: L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=NK, N=D, K=4 * D, A=hin.data_ptr(), lda=4 * D,
        W=w2[i].data_ptr(), ldw=4 * D, w_expert_stride=4 * D * D, C=Y.data_ptr(), ldc=D, expert_offsets=mp + 4 * ml.offsets, num_experts=E, split_k=S,
        split_stride=NK * D, flags=FL), 16.8)
reps = 120
for name, (mk, mb) in cases.items():
    ds = [mk(i) for i in range(nl)]
    out = []
    for fl in (0, 64):
        lib.mode_set_option(b"pp_flags", fl)
        for d in ds:
            L.check(lib.mode_gemm(C.byref(d), st0))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            cst = torch.cuda.current_stream().cuda_stream
            for i in range(reps):
                L.check(lib.mode_gemm(C.byref(ds[i % nl]), cst))
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out.append(f"flags {fl:2d}: {us:6.2f} us ({mb / us:5.2f} TB/s)")
    lib.mode_set_option(b"pp_flags", 0)
    print(f"B={B} {name:36s} " + "   ".join(out))

# ---- does a weight slab that is already in the Infinity Cache (MALL) stream faster?  touch(i) = a torch reduction over the ACTIVE weights of
# layer i right before gemm(i); the 12 layers cycle through more than the 256 MB cache, so the untouched runs read HBM.
STRIDE = int(os.environ.get("TOUCH_STRIDE", "0"))               # 0 = read every byte; n = one dword per n bytes (translation warm-up only)
sink = torch.zeros(1, device=dev)
views = {"qkv": [w.view(torch.int32).view(-1) for w in wqkv], "c_proj": [w.view(torch.int32).view(-1) for w in wo],
         "up": [w[1:3].view(torch.int32).view(-1) for w in w1], "down": [w[1:3].view(torch.int32).view(-1) for w in w2]}
for cname, key in (("qkv [14x1024]x[3072x1024] 6.3MB", "qkv"), ("c_proj+resid+ln2 2.1MB", "c_proj"), ("up swiglu+ln2 33.5MB", "up"), ("down S=4 16.8MB", "down")):
    ds = [cases[cname][0](i) for i in range(nl)]
    flat = views[key]
    def touch(i):
        sink.add_((flat[i][:: STRIDE // 4] if STRIDE else flat[i]).sum())
    def run(kind, cst):
        for i in range(reps):
            if kind in ("touch", "both"):
                touch(i % nl)
            if kind in ("gemm", "both"):
                L.check(lib.mode_gemm(C.byref(ds[i % nl]), cst))
    res = {}
    for kind in ("gemm", "touch", "both"):
        run(kind, st0); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            run(kind, torch.cuda.current_stream().cuda_stream)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res[kind] = e0.elapsed_time(e1) * 1e3 / reps
    print(f"MALL probe stride {STRIDE} {key:7s}: gemm alone {res['gemm']:.2f} us, touch alone {res['touch']:.2f} us, touch+gemm {res['both']:.2f} us -> gemm after touch {res['both'] - res['touch']:.2f} us")

# ---- where a launch's time goes: 100-MHz timestamps per workgroup (start, segment known, operands consumed, done), relative to the first workgroup's start
tr = torch.zeros(4096 * 4, dtype=torch.int64, device=dev)
ptr = tr.data_ptr()
for name, (mk, mb) in cases.items():
    ds = [mk(i) for i in range(nl)]
    for i in range(3):
        L.check(lib.mode_gemm(C.byref(ds[i]), st0))
    torch.cuda.synchronize(); tr.zero_()
    lib.mode_set_option(b"pp_trace_lo", C.c_int32(ptr & 0xffffffff).value); lib.mode_set_option(b"pp_trace_hi", C.c_int32((ptr >> 32) & 0xffffffff).value)
    L.check(lib.mode_gemm(C.byref(ds[5]), st0))
    torch.cuda.synchronize()
    lib.mode_set_option(b"pp_trace_lo", 0); lib.mode_set_option(b"pp_trace_hi", 0)
    t = tr.view(4096, 4).cpu().double()
    act = t[:, 3] > 0
    every = t[:, 0] > 0
    t0 = t[every, 0].min()
    a = (t[act] - t0) * 0.01          # us
    print(f"trace {name:34s} {int(act.sum()):4d} active / {int(every.sum())} workgroups; us since the first start: start med {a[:, 0].median():5.2f} max {a[:, 0].max():5.2f} | "
          f"segment known +{(a[:, 1] - a[:, 0]).median():4.2f} | operands consumed +{(a[:, 2] - a[:, 1]).median():5.2f} (max {(a[:, 2] - a[:, 1]).max():5.2f}) | "
          f"done +{(a[:, 3] - a[:, 2]).median():4.2f} | last done at {a[:, 3].max():5.2f}")
