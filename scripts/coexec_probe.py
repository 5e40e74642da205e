"""What runs UNDER a backward-like chain of GEMM kernels issued on another stream?  A = 24 x (dgrad + wgrad) at the expert shapes; B = one of:
pure-ALU spin, pure read stream, read-modify-write stream (scripts/probe/coexec_probe.hip), each sized to ~3 ms alone.  Prints A, B, A || B."""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L
here = os.path.dirname(os.path.abspath(__file__))
px = C.CDLL(os.path.join(here, "probe", "libcoexec_probe.so"))
px.spin_alu.argtypes = [C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_void_p]; px.stream_read.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
px.stream_rmw.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p]
lib = L.load(); dev = "cuda"; bf = torch.bfloat16
R, D, H = 3584, 1024, 8192
dP = torch.randn(R, H, device=dev).to(bf); X = torch.randn(R, D, device=dev).to(bf); W1 = (torch.randn(H, D, device=dev) * 0.03).to(bf)
dX = torch.empty(R, D, device=dev, dtype=bf); dW = torch.empty(H, D, device=dev)
big = torch.zeros(1 << 30, device=dev)      # 4 GiB
out = torch.zeros(4, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "tr"

def chain_a(reps=24):
    for _ in range(reps):
        if mode == "tr":
            d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=R, N=D, K=H, A=dP.data_ptr(), lda=H, W=W1.data_ptr(), ldw=D, C=dX.data_ptr(), ldc=D, flags=L.GEMM_W_KN)
            L.check(lib.mode_gemm(C.byref(d), sa.cuda_stream), "dgrad")
            d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=H, N=D, K=R, A=dP.data_ptr(), lda=H, W=X.data_ptr(), ldw=D, C=dW.data_ptr(), ldc=D, flags=L.GEMM_W_KN | L.GEMM_A_KM)
            L.check(lib.mode_gemm(C.byref(d), sa.cuda_stream), "wgrad")
        else:                                                                 # forward GEMM (ring / ping-pong family): P[R, H] = X[R, D] @ W1^T
            d = L.ModeGemmDesc(dtype=L.MODE_BF16, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=R, N=H, K=D, A=X.data_ptr(), lda=D, W=W1.data_ptr(), ldw=D, C=dP.data_ptr(), ldc=H)
            L.check(lib.mode_gemm(C.byref(d), sa.cuda_stream), "fwd"); L.check(lib.mode_gemm(C.byref(d), sa.cuda_stream), "fwd")

def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

cases = []
for blocks, threads in ((256, 256), (64, 64), (1024, 256)):
    cases.append((f"ALU spin {blocks}x{threads}", lambda b=blocks, th=threads: px.spin_alu(b, th, 1_500_000, out.data_ptr(), sb.cuda_stream)))
for blocks in (256, 64, 1024):
    cases.append((f"read stream 4 GiB, {blocks} WGs", lambda b=blocks: px.stream_read(big.data_ptr(), big.numel() // 4, b, out.data_ptr(), sb.cuda_stream)))
for blocks in (256, 64):
    cases.append((f"RMW stream 2 GiB, {blocks} WGs", lambda b=blocks: px.stream_rmw(big.data_ptr(), big.numel() // 8, b, sb.cuda_stream)))
a = t(chain_a)
print(f"A = GEMM chain ({mode}) alone: {a:.2f} ms")
for name, fb in cases:
    b = t(fb); ab = t(lambda: (chain_a(), fb())); ba = t(lambda: (fb(), chain_a()))
    print(f"B = {name:32s}: alone {b:6.2f} ms   A then B enqueued: {ab:6.2f}   B then A enqueued: {ba:6.2f}   (sum {a + b:6.2f}, max {max(a, b):6.2f})", flush=True)
