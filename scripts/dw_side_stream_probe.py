"""Can the HBM-bound weight-gradient GEMMs with the AdamW epilogue (ModeAdamWFuse) hide under the MFMA-bound data-gradient GEMMs of the backward when they run on
a SECOND stream?  Times, at the config-2 shapes (B = 128): A = the up-projection's data gradient dU = dP W1 (four K-slices; persistent ping-pong kernel or ring
kernel), B = dW1 with the fused optimizer epilogue, alone and concurrently (A on the current stream, B on a side stream, n launches each, joined by events).
Usage: python scripts/dw_side_stream_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
D, E, NK = 1024, 4, 3584
bf = torch.bfloat16
torch.manual_seed(0)
offs = torch.tensor([0, 871, 1796, 2699, 3584], dtype=torch.int32, device=dev)
nl = 6
dP = torch.randn(NK, 8 * D, device=dev).to(bf) * 0.1
U = torch.randn(NK, D, device=dev).to(bf)
W1 = [torch.randn(E, 8 * D, D, device=dev).to(bf) * 0.03 for _ in range(nl)]
dU = torch.empty(4, NK, D, device=dev)
n = E * 8 * D * D
P = [torch.randn(n, device=dev) * 0.02 for _ in range(nl)]; Mo = [torch.zeros(n, device=dev) for _ in range(nl)]; V = [torch.zeros(n, device=dev) for _ in range(nl)]
LP = [torch.zeros(n, dtype=bf, device=dev) for _ in range(nl)]; G = [torch.empty(n, device=dev) for _ in range(nl)]
keep = []


def desc_a(i):
    return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=NK, N=D, K=8 * D, A=dP.data_ptr(), lda=8 * D, W=W1[i].data_ptr(), ldw=D,
                          w_expert_stride=8 * D * D, C=dU.data_ptr(), ldc=D, expert_offsets=offs.data_ptr(), num_experts=E, flags=L.GEMM_W_KN, split_k=4, split_stride=NK * D)


def desc_b(i):
    f = L.ModeAdamWFuse(grad_base=G[i].data_ptr(), param_base=P[i].data_ptr(), exp_avg_base=Mo[i].data_ptr(), exp_avg_sq_base=V[i].data_ptr(), lp_base=LP[i].data_ptr(),
                        lr=1e-4, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.05, step=3, grad_scale=1.0)
    keep.append(f)
    return L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=8 * D, N=D, K=NK, A=dP.data_ptr(), lda=8 * D, W=U.data_ptr(), ldw=D, C=G[i].data_ptr(),
                          ldc=D, k_group_offsets=offs.data_ptr(), num_k_groups=E, c_group_stride=8 * D * D, flags=L.GEMM_W_KN | L.GEMM_A_KM, adamw=C.pointer(f))


side = torch.cuda.Stream()
cur = torch.cuda.current_stream()
N_ = 24
for cfg, name in ((0, "A = dU on the persistent ping-pong kernel"), (7, "A = dU on the ring kernels")):
    lib.mode_set_option(b"gemm_tr_cfg", cfg)
    da = [desc_a(i) for i in range(nl)]; db = [desc_b(i) for i in range(nl)]

    def run(a, b):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        side.wait_stream(cur)
        for i in range(N_):
            if a:
                L.check(lib.mode_gemm(C.byref(da[i % nl]), cur.cuda_stream))
            if b:
                L.check(lib.mode_gemm(C.byref(db[i % nl]), side.cuda_stream))
        cur.wait_stream(side)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / N_
    for _ in range(2):
        run(True, True)
    ta = min(run(True, False) for _ in range(4)); tb = min(run(False, True) for _ in range(4)); tab = min(run(True, True) for _ in range(4))
    print(f"{name}: A alone {ta:6.1f} us, B (dW1 + AdamW) alone {tb:6.1f} us, A || B {tab:6.1f} us per pair  (serial {ta + tb:6.1f}, ideal {max(ta, tb):6.1f})")
lib.mode_set_option(b"gemm_tr_cfg", 0)
