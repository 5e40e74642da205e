"""1 x 1 / stride-1 convolutions of a ResNet-50 at B = 64, 224 x 224 (channels_last bf16): MIOpen through F.conv2d (forward, data gradient) against the same
products on the library's GEMMs (forward: mode_gemm; data gradient: MODE_GEMM_W_KN).  python scripts/conv1x1_probe.py"""
import ctypes as C, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L
lib = L.load(); dev = "cuda"; bf = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
st = lambda: torch.cuda.current_stream().cuda_stream
for (hw, cin, cout) in ((56, 64, 64), (56, 64, 256), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128), (28, 512, 256), (14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512)):
    B = 64; R = B * hw * hw
    x = torch.randn(B, cin, hw, hw, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device=dev) * cin ** -0.5).to(bf).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, hw, hw, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, cout, hw, hw, device=dev, dtype=bf).contiguous(memory_format=torch.channels_last)
    dx = torch.empty_like(x)
    t_f = timeit(lambda: F.conv2d(x, w))
    t_b = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False)))
    df = L.ModeGemmDesc(dtype=0, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=R, N=cout, K=cin, A=x.data_ptr(), lda=cin, W=w.data_ptr(), ldw=cin, C=y.data_ptr(), ldc=cout)
    db = L.ModeGemmDesc(dtype=0, epilogue=L.EPI_NONE, out_dtype=L.MODE_BF16, M=R, N=cin, K=cout, A=dy.data_ptr(), lda=cout, W=w.data_ptr(), ldw=cin, C=dx.data_ptr(), ldc=cin, flags=L.GEMM_W_KN)
    rf = lib.mode_gemm(C.byref(df), st()); rb = lib.mode_gemm(C.byref(db), st()); torch.cuda.synchronize()
    g_f = timeit(lambda: lib.mode_gemm(C.byref(df), st())) if rf == 0 else float("nan")
    g_b = timeit(lambda: lib.mode_gemm(C.byref(db), st())) if rb == 0 else float("nan")
    yr = F.conv2d(x, w); ok_f = float((y.float() - yr.float()).norm() / yr.float().norm()) if rf == 0 else -1
    gb = (float(R) * (cin + cout) * 2) / 1e3
    print(f"{hw:2d}x{hw:2d} {cin:4d}->{cout:4d}: fwd MIOpen {t_f:7.1f} us  ours {g_f:7.1f} us (rel {ok_f:.1e}) | dgrad MIOpen {t_b:7.1f} us  ours {g_b:7.1f} us | min traffic {gb / 5e3:6.1f} us at 5 TB/s")
