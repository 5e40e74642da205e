"""The four expert GEMMs of one training-backward layer (C2, B = 128, ragged multinomial segments): 128x128 transpose-read ring kernels
("gemm_tr_cfg" 7) vs the persistent ping-pong kernel of gemm_bf16_pptr.hip (6).  Every launch of the timed graph works on another layer's
operands (12 sets: HBM-cold like in the chain).  python scripts/pptr_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L
from mode_diffusion_policy_amd.engine import capture_graph
lib = L.load(); dev = "cuda"; bf = torch.bfloat16
D, E, Ly = 1024, 4, 12
counts = [871, 925, 903, 885]; NK = sum(counts)
off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=dev)
torch.manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)
dY = [rn(NK, D) for _ in range(Ly)]; W2 = [rn(E, D, 4 * D) for _ in range(Ly)]; H = [rn(NK, 4 * D) for _ in range(Ly)]
dP = [rn(NK, 8 * D) for _ in range(Ly)]; W1 = [rn(E, 8 * D, D) for _ in range(Ly)]; U = [rn(NK, D) for _ in range(Ly)]
dH = torch.empty(NK, 4 * D, dtype=bf, device=dev); dW2 = [torch.empty(E, D, 4 * D, device=dev) for _ in range(Ly)]
dU = torch.empty(4, NK, D, device=dev); dW1 = [torch.empty(E, 8 * D, D, device=dev) for _ in range(Ly)]
G = L.ModeGemmDesc
def descs(l, du_split):
    return {
        "dH  = dY W2   (30 GF)": G(dtype=0, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=NK, N=4 * D, K=D, A=dY[l].data_ptr(), lda=D, W=W2[l].data_ptr(), ldw=4 * D,
                                   w_expert_stride=4 * D * D, C=dH.data_ptr(), ldc=4 * D, expert_offsets=off.data_ptr(), num_experts=E, flags=L.GEMM_W_KN),
        "dW2 = dY^T H  (30 GF)": G(dtype=0, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=D, N=4 * D, K=NK, A=dY[l].data_ptr(), lda=D, W=H[l].data_ptr(), ldw=4 * D,
                                   C=dW2[l].data_ptr(), ldc=4 * D, k_group_offsets=off.data_ptr(), num_k_groups=E, c_group_stride=4 * D * D, flags=L.GEMM_W_KN | L.GEMM_A_KM),
        "dU  = dP W1   (60 GF)": G(dtype=0, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=NK, N=D, K=8 * D, A=dP[l].data_ptr(), lda=8 * D, W=W1[l].data_ptr(), ldw=D,
                                   w_expert_stride=8 * D * D, C=dU.data_ptr(), ldc=D, expert_offsets=off.data_ptr(), num_experts=E, flags=L.GEMM_W_KN, split_k=du_split,
                                   split_stride=NK * D),
        "dW1 = dP^T u  (60 GF)": G(dtype=0, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=8 * D, N=D, K=NK, A=dP[l].data_ptr(), lda=8 * D, W=U[l].data_ptr(), ldw=D,
                                   C=dW1[l].data_ptr(), ldc=D, k_group_offsets=off.data_ptr(), num_k_groups=E, c_group_stride=8 * D * D, flags=L.GEMM_W_KN | L.GEMM_A_KM),
    }
flops = {"dH": 2.0 * NK * D * 4 * D, "dW": 2.0 * NK * D * 4 * D, "dU": 2.0 * NK * D * 8 * D}
def timeit(cfg, name, du_split, reps=3):
    lib.mode_set_option(b"gemm_tr_cfg", cfg)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mode_gemm(C.byref(descs(0, du_split)[name]), st); torch.cuda.synchronize()
    if rc != 0:
        return None
    g = torch.cuda.CUDAGraph()
    with capture_graph(g):
        cst = torch.cuda.current_stream().cuda_stream
        for rep in range(reps):
            for l in range(Ly):
                lib.mode_gemm(C.byref(descs(l, du_split)[name]), cst)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * Ly))
    return best
for name in descs(0, 2):
    gf = 60.13e9 if "60 GF" in name else 30.06e9
    for cfg, split, tag in ((7, 2, "ring"), (6, 2, "pptr split2"), (6, 4, "pptr split4")):
        if "dU" not in name and split == 4:
            continue
        us = timeit(cfg, name, split)
        print(f"{name:24s} {tag:12s} {us:7.1f} us  {gf / us / 1e6:7.1f} TF/s" if us else f"{name} {tag} unsupported")
lib.mode_set_option(b"gemm_tr_cfg", 0)
