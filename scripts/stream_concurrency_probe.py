"""Do two HIP streams of one process execute kernels concurrently on this box?  torch.cuda._sleep (a one-thread spin kernel) on one and on two streams."""
import os, time, torch
torch.cuda.init(); x = torch.zeros(1, device="cuda")
cyc = int(4e8)
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
one = t(lambda: torch.cuda._sleep(cyc))
one = t(lambda: torch.cuda._sleep(cyc))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(sa):
        torch.cuda._sleep(cyc)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(cyc)
print(f"one spin kernel {one:.1f} ms; two streams, one spin kernel each: {t(two):.1f} ms  (concurrent = same, serialized = double)")
print({k: v for k, v in os.environ.items() if k.startswith(("HIP_", "HSA_", "GPU_", "ROC", "AMD_"))})
