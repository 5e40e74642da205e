"""What the library's row-major weight-gradient GEMM makes of a 3 x 3 convolution's dW when all nine taps are ONE product with N = 9 * Cin (operand = an
explicit im2col matrix here: the upper bound of a tap-aware gather).  Shapes: ResNet-50 conv2 of each stage at B = 64, 224 x 224.  python scripts/conv_wgrad_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L
lib = L.load(); dev = "cuda"; bf = torch.bfloat16
def run(R, cout, cin, taps, G):
    A = torch.randn(R, cout, device=dev).to(bf); X = torch.randn(R, taps * cin, device=dev).to(bf)
    offs = torch.tensor([(R * i) // G for i in range(G + 1)], dtype=torch.int32, device=dev)
    part = torch.empty(G, cout, taps * cin, device=dev)
    d = L.ModeGemmDesc(dtype=0, epilogue=L.EPI_NONE, out_dtype=L.MODE_F32, M=cout, N=taps * cin, K=R, A=A.data_ptr(), lda=cout, W=X.data_ptr(), ldw=taps * cin,
                       C=part.data_ptr(), ldc=taps * cin, k_group_offsets=offs.data_ptr(), num_k_groups=G, c_group_stride=cout * taps * cin, flags=L.GEMM_W_KN | L.GEMM_A_KM)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): L.check(lib.mode_gemm(C.byref(d), st), "g")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.mode_gemm(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * R * cout * taps * cin
    print(f"R={R:7d} Cout={cout:4d} Cin={cin:4d} taps={taps} G={G:4d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
for cfg in (0, 1, 4, 5):
    lib.mode_set_option(b"gemm_tr_cfg", cfg if cfg else 7)
    print("gemm_tr_cfg", cfg if cfg else 7)
    for (R, c) in ((200704, 64), (50176, 128), (12544, 256), (3136, 512)):
        tiles = ((c + 127) // 128) * ((9 * c + 127) // 128)
        for G in sorted({max(1, min(R // 256, 768 // tiles)), max(1, min(R // 256, 1536 // tiles)), max(1, min(R // 512, 384 // tiles))}):
            run(R, c, c, 9, G)
lib.mode_set_option(b"gemm_tr_cfg", 0)
