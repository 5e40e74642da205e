"""K-slope of the fused QKV + attention kernel: the same 256-workgroup grid with K = 1024 (B = 128, 8 heads) and K = 2048 (B = 64, 16 heads) -
the difference is 16 K-steps of the loop (per-CU operand stream: 56 KiB per K-step).  Prints us per launch and cycles / bytes per clock per K-step."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L  # noqa: E402
from mode_diffusion_policy_amd.engine import capture_graph  # noqa: E402

lib = L.load()
dev = torch.device("cuda", 0)
bf = torch.bfloat16
T, reps = 14, 120
res = {}
for waves, w3 in ((4, 0), (4, 1), (8, 0), (8, 1)):
    lib.mode_set_option(b"qkv_attn_waves", waves); lib.mode_set_option(b"qkv_attn_w3", w3)
    for B, H in ((128, 8), (64, 16), (32, 32)):
        D = 128 * H
        N = B * T
        g = torch.Generator(device="cpu").manual_seed(0)
        h = torch.randn(N, D, generator=g).to(bf).to(dev)
        W = [(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(bf).to(dev) for _ in range(4)]
        bias = torch.zeros(3 * D, device=dev); qg = torch.ones(128, device=dev); y = torch.empty(N, D, dtype=bf, device=dev)

        def fused(i, st):
            d = L.ModeQkvAttnDesc(dtype=L.MODE_BF16, B=B, T=T, H=H, D=D, h=h.data_ptr(), ldh=D, wqkv=W[i % 4].data_ptr(), ldw=D, bqkv=bias.data_ptr(),
                                  q_gain=qg.data_ptr(), k_gain=qg.data_ptr(), eps=1e-6, y=y.data_ptr(), ldy=D)
            L.check(lib.mode_qkv_attn_fwd(C.byref(d), st))
        for i in range(4):
            fused(i, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with capture_graph(gr):
            cst = torch.cuda.current_stream().cuda_stream
            for i in range(reps):
                fused(i, cst)
        best = 1e9
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        res[(waves, w3, D)] = best
    a, b, c = res[(waves, w3, 1024)], res[(waves, w3, 2048)], res[(waves, w3, 4096)]
    per = (c - b) / 32
    print(f"{waves} waves, w3={w3}: K=1024 {a:.2f} us, K=2048 {b:.2f} us, K=4096 {c:.2f} us -> {per:.3f} us per K-step (from 2048->4096), "
          f"{(b - a) / 16:.3f} (1024->2048); fixed part {a - 16 * per:.2f} us; at 2.1 GHz {per * 2100:.0f} clk, {57344 / (per * 2100):.1f} B/clk/CU", flush=True)
