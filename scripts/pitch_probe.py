"""Does a 2^11-byte row pitch of the operands (K = 1024 bf16) camp on L2 channels?  Times the expert up-projection (persistent ping-pong kernel) and
the QKV projection (128x64 ring kernel) with the weights / activations at pitch K and at pitch K + PAD elements.  Usage: python scripts/pitch_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402
from mode_diffusion_policy_amd.engine import capture_graph  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
B, D, E, k, T = 128, 1024, 4, 2, 14
N, NK = B * T, B * T * k
bf = torch.bfloat16
torch.manual_seed(0)
nl = 6
st0 = torch.cuda.current_stream().cuda_stream
idx = torch.tensor([[1, 2]] * B, dtype=torch.int32, device=dev); w = torch.full((B, k), 0.5, device=dev)
ml = L.ModeMetaLayout(); lib.mode_moe_meta_layout(N, E, k, C.byref(ml))
meta = torch.empty(ml.total_words, dtype=torch.int32, device=dev)
L.check(lib.mode_dit_dispatch(idx.data_ptr(), w.data_ptr(), 1, B * k, B, T, N, E, k, meta.data_ptr(), st0))
mp = meta.data_ptr()
b1 = torch.randn(E, 8 * D, device=dev); bq = torch.randn(3 * D, device=dev)
ss = torch.rand(N, D // 64, device=dev) + 0.5


def padded(shape, pad, scale):
    """tensor of `shape` whose last-dim pitch is shape[-1] + pad elements"""
    full = torch.randn(*shape[:-1], shape[-1] + pad, device=dev).to(bf) * scale
    return full, full[..., : shape[-1]]


reps = 60
for pad in (0, 64, 192, 32):
    xs, x = padded((N, D), pad, 1.0)
    w1s = [padded((E, 8 * D, D), pad, 0.03) for _ in range(nl)]
    wqs = [padded((3 * D, D), pad, 0.03) for _ in range(nl)]
    ld = D + pad
    out1 = torch.empty(NK, 4 * D, dtype=bf, device=dev); qkv = torch.empty(N, 3 * D, dtype=bf, device=dev)
    up = [L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_SWIGLU, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=x.data_ptr(), lda=ld, W=w1s[i][1].data_ptr(), ldw=ld,
                         w_expert_stride=8 * D * ld, bias=b1.data_ptr(), bias_expert_stride=8 * D, C=out1.data_ptr(), ldc=4 * D, a_rows=mp + 4 * ml.perm,
                         expert_offsets=mp + 4 * ml.offsets, num_experts=E, row_ss=ss.data_ptr(), row_ss_n=D // 64, row_eps=1e-6, flags=L.GEMM_UNIFORM_GROUPS) for i in range(nl)]
    qk = [L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_BIAS, out_dtype=L.MODE_BF16, M=N, N=3 * D, K=D, A=x.data_ptr(), lda=ld, W=wqs[i][1].data_ptr(), ldw=ld,
                         bias=bq.data_ptr(), C=qkv.data_ptr(), ldc=3 * D) for i in range(nl)]
    line = []
    for name, ds in (("up-projection (pp)", up), ("qkv (ring 128x64)", qk)):
        for d in ds:
            L.check(lib.mode_gemm(C.byref(d), st0))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            cst = torch.cuda.current_stream().cuda_stream
            for i in range(reps):
                L.check(lib.mode_gemm(C.byref(ds[i % nl]), cst))
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        line.append(f"{name}: {min(ts):6.2f} us")
    print(f"row pitch {ld * 2:5d} B (pad {pad:3d} elements)   " + "   ".join(line), flush=True)
