"""The fp32 (router / embedding) products of one C2 training step, B = 128, on gemm_f32.hip: per-launch time with HBM-cold operands (every launch of the
timed graph reads another copy of the weight, 24 copies) and with the same operands every launch (cache-warm).  python scripts/gemm_f32_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L
from mode_diffusion_policy_amd.engine import capture_graph
lib = L.load(); dev = "cuda"
G = L.ModeGemmDesc
torch.manual_seed(0)
NC = 24


def problem(name, M, N, K, flags=0, kgroups=0, bias=False):
    a_km, w_kn = bool(flags & L.GEMM_A_KM), bool(flags & L.GEMM_W_KN)
    A = [torch.randn((K, M) if a_km else (M, K), device=dev) for _ in range(2)]
    nw = NC if M * K <= N * K else 2
    W = [torch.randn((K, N) if w_kn else (N, K), device=dev) for _ in range(NC if N * K * 4 <= (64 << 20) else 4)]
    b = torch.randn(N, device=dev) if bias else None
    ng = max(kgroups, 1)
    Cc = torch.empty(ng, M, N, device=dev)
    koffs = torch.arange(0, K + 1, K // ng, dtype=torch.int32, device=dev) if kgroups else None

    def desc(i):
        a, w = A[i % len(A)], W[i % len(W)]
        d = G(dtype=L.MODE_F32, epilogue=L.EPI_BIAS if bias else L.EPI_NONE, out_dtype=L.MODE_F32, M=M, N=N, K=K, A=a.data_ptr(), lda=a.shape[1], W=w.data_ptr(),
              ldw=w.shape[1], C=Cc.data_ptr(), ldc=N, flags=flags)
        if bias:
            d.bias = b.data_ptr()
        if kgroups:
            d.k_group_offsets = koffs.data_ptr(); d.num_k_groups = ng; d.c_group_stride = M * N
        return d
    keep = (A, W, b, Cc, koffs)
    return name, desc, 2.0 * M * N * K, (M * K + N * K + M * N * ng) * 4.0, keep


def timeit(desc, cold, n=24):
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mode_gemm(C.byref(desc(0)), st); torch.cuda.synchronize()
    assert rc == 0, rc
    g = torch.cuda.CUDAGraph()
    with capture_graph(g):
        cst = torch.cuda.current_stream().cuda_stream
        for i in range(n):
            lib.mode_gemm(C.byref(desc(i if cold else 0)), cst)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


D, B, Ly = 1024, 128, 12
KN, KM = L.GEMM_W_KN, L.GEMM_A_KM
probs = [
    problem("sigma_linear        [128 x 1024] K 1024", B, D, D),
    problem("tok_emb             [256 x 1024] K 2048", 2 * B, D, 2048, bias=True),
    problem("goal_emb            [128 x 1024] K  512", B, D, 512, bias=True),
    problem("router L x W0       [128 x 24576] K 1024", B, Ly * 2 * D, D, bias=True),
    problem("router dW0 (KM|KN)  [24576 x 1024] K 128", Ly * 2 * D, D, B, flags=KM | KN),
    problem("router dcond (KN,48 groups) [128 x 1024] K 24576", B, D, Ly * 2 * D, flags=KN, kgroups=48),
    problem("d w_tok (KM|KN)     [1024 x 2048] K 256", D, 2048, 2 * B, flags=KM | KN),
    problem("d w_sl (KM|KN)      [1024 x 1024] K 128", D, D, B, flags=KM | KN),
    problem("d e1 (KN)           [128 x 1024] K 1024", B, D, D, flags=KN),
]
for name, desc, fl, by, keep in probs:
    c, w = timeit(desc, True), timeit(desc, False)
    print(f"{name:52s} cold {c:7.1f} us ({fl / c / 1e6:6.1f} TF/s, {by / c / 1e3:6.1f} GB/s)   warm {w:7.1f} us")
