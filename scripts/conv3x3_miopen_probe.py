import os, sys, torch, torch.nn.functional as F
dev="cuda"; bf=torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (hw, c, s) in ((56, 64, 1), (56, 128, 2), (28, 128, 1), (28, 256, 2), (14, 256, 1), (14, 512, 2), (7, 512, 1)):
    B=64
    x = torch.randn(B, c, hw, hw, device=dev).to(bf).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) * (9*c) ** -0.5).to(bf).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x, w, None, s, 1); dy = torch.randn_like(y)
    t_f = timeit(lambda: F.conv2d(x, w, None, s, 1))
    t_b = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
    fl = 2.0 * y.shape[0]*y.shape[2]*y.shape[3] * c * c * 9
    print(f"{hw}x{hw} {c}->{c} s{s}: fwd {t_f:7.1f} us ({fl/t_f/1e6:6.1f} TF/s)  dgrad {t_b:7.1f} us ({fl/t_b/1e6:6.1f} TF/s)")
