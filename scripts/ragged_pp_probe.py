"""Training-forward expert up-projection (ragged per-token routing, bias epilogue, pre-activation kept): 128x128 ring family vs the persistent
ping-pong kernel.  python scripts/ragged_pp_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L
from mode_diffusion_policy_amd.engine import capture_graph
lib = L.load(); dev = "cuda"; bf = torch.bfloat16
N, D, E, k, Ly = 1792, 1024, 4, 2, 12
NK = N * k
torch.manual_seed(0)
u = torch.randn(N, D, device=dev).to(bf)
W1 = [(torch.randn(E, 8 * D, D, device=dev) * D ** -0.5).to(bf) for _ in range(Ly)]; b1 = torch.randn(E, 8 * D, device=dev) * 0.1
idx = torch.stack([torch.randperm(E, device=dev)[:k] for _ in range(N)]).to(torch.int32)        # per-token draw without replacement
order = torch.argsort(idx.reshape(-1).long(), stable=True)
counts = torch.bincount(idx.reshape(-1).long(), minlength=E)
offsets = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), counts.cumsum(0)]).to(torch.int32)
perm = (order // k).to(torch.int32)
P = torch.empty(NK, 8 * D, dtype=bf, device=dev)
print("counts", counts.tolist())
def desc(l):
    return L.ModeGemmDesc(dtype=0, epilogue=L.EPI_BIAS, out_dtype=0, M=NK, N=8 * D, K=D, A=u.data_ptr(), lda=D, W=W1[l].data_ptr(), ldw=D, w_expert_stride=8 * D * D,
                          bias=b1.data_ptr(), bias_expert_stride=8 * D, C=P.data_ptr(), ldc=8 * D, a_rows=perm.data_ptr(), expert_offsets=offsets.data_ptr(), num_experts=E)
outs = {}
for cfg in (0, 17, 18, 13, 1):
    lib.mode_set_option(b"gemm_cfg", cfg)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mode_gemm(C.byref(desc(0)), st); torch.cuda.synchronize()
    if rc != 0:
        print("cfg", cfg, "rc", rc); continue
    outs[cfg] = P.clone()
    g = torch.cuda.CUDAGraph()
    with capture_graph(g):
        cst = torch.cuda.current_stream().cuda_stream
        for rep in range(5):
            for l in range(Ly):
                lib.mode_gemm(C.byref(desc(l)), cst)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * Ly)
    print("cfg", cfg, round(us, 1), "us", round(2.0 * NK * D * 8 * D / us / 1e6, 1), "TF/s", "equal to cfg0:", torch.equal(outs[cfg].view(torch.int16), outs[0].view(torch.int16)) if 0 in outs else None)
lib.mode_set_option(b"gemm_cfg", 0)
