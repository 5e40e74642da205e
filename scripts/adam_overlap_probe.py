"""Can the HBM-bound AdamW pass run UNDER the backward GEMMs at all?  Stream A: a backward-like chain of bf16 GEMMs (dgrad W_KN + wgrad A_KM|W_KN
at the expert shapes), stream B: AdamW over an independent 54.6 M-parameter slice, no dependencies between the two.  Times: A alone, B alone,
A || B (wall of both).  If A || B ~ A + B the hardware is not co-scheduling them (or contention eats everything)."""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L
lib = L.load(); dev = "cuda"; bf = torch.bfloat16
R, D, H = 3584, 1024, 8192
dP = torch.randn(R, H, device=dev).to(bf); X = torch.randn(R, D, device=dev).to(bf)
W1 = (torch.randn(H, D, device=dev) * 0.03).to(bf)
dX = torch.empty(R, D, device=dev, dtype=bf); dW = torch.empty(H, D, device=dev)
n = 54_600_000 // 4 * 4
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
lp = torch.empty(n, device=dev, dtype=bf)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def dgrad(st):      # dX[R, D] = dP[R, H] @ W1[H, D]   (W is [K, N] row-major)
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=R, N=D, K=H, A=dP.data_ptr(), lda=H, W=W1.data_ptr(), ldw=D, C=dX.data_ptr(), ldc=D, flags=L.GEMM_W_KN)
    L.check(lib.mode_gemm(C.byref(d), st), "dgrad")

def wgrad(st):      # dW[H, D] = dP^T[H, R] @ X[R, D]
    d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=H, N=D, K=R, A=dP.data_ptr(), lda=H, W=X.data_ptr(), ldw=D, C=dW.data_ptr(), ldc=D, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    L.check(lib.mode_gemm(C.byref(d), st), "wgrad")

def chain_a(reps=24):
    for _ in range(reps):
        dgrad(sa.cuda_stream); wgrad(sa.cuda_stream)

def chain_b(reps=12):
    for i in range(reps):
        L.check(lib.mode_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.95, 1e-8, 0.05, i + 1, 1.0, lp.data_ptr(), None, 0.0, sb.cuda_stream), "adamw")

def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

for blocks in (256, 128, 64, 512):
    lib.mode_set_option(b"adamw_blocks", blocks)
    a = t(chain_a); b = t(chain_b); ab = t(lambda: (chain_a(), chain_b()))
    print(f"adamw_blocks {blocks:4d}: GEMM chain alone {a:6.2f} ms  AdamW alone {b:6.2f} ms ({12 * n * 30 / b / 1e9:5.2f} TB/s)  both {ab:6.2f} ms   (sum {a + b:6.2f}, max {max(a, b):6.2f})", flush=True)
