"""Micro-benchmark of mode_attn_block_bwd (B*H problems of T tokens), optionally stopping after phase N (profiling aid).
python scripts/attn_bwd_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, T, H, hd, p = 32, 14, 8, 128, 0.3
D = H * hd
qkv = torch.randn(B * T, 3 * D, device=dev).to(torch.bfloat16); dy = torch.randn(B * T, D, device=dev).to(torch.bfloat16)
g = torch.ones(hd, device=dev); dq = torch.empty_like(qkv); pq = torch.empty(B * H, hd, device=dev); pk = torch.empty_like(pq)
def run():
    L.check(lib.mode_attn_block_bwd(qkv.data_ptr(), g.data_ptr(), g.data_ptr(), dy.data_ptr(), dq.data_ptr(), pq.data_ptr(), pk.data_ptr(),
                                    L.MODE_BF16, B, T, H, hd, 1e-6, 7, p, st), "bwd")
for stop in [1, 2, 3, 4, 5, 6, 7, 8, 9, 0]:
    lib.mode_set_option(b"attn_bwd_stop", stop)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} stop_after={stop}: {e0.elapsed_time(e1) * 1e3 / 50:7.1f} us")
lib.mode_set_option(b"attn_bwd_stop", 0)
