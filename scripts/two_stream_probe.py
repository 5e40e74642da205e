"""Probe: does splitting the B=128 batch into concurrent half-batches on separate HIP streams fill kernel tail bubbles?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
dev = torch.device("cuda:0")
M, den = Bn.build_model(dev)
sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)
import copy
def make(Bsz):
    img, goal, x0 = Bn.synthetic_inputs(dev, Bsz)
    return ({"state_images": img}, x0, goal)
def run_single(Bsz, reps=20):
    st, x0, goal = make(Bsz)
    for _ in range(3): M.sample_ddim(den, st, x0, goal, sig, disable=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): M.sample_ddim(den, st, x0, goal, sig, disable=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
t128 = run_single(128); print(f"B=128 one stream: {t128*1e3:.2f} ms/chunk")
t64 = run_single(64); print(f"B=64  one stream: {t64*1e3:.2f} ms/chunk  (x2 serial = {2*t64*1e3:.2f})")
# two concurrent B=64 replays: second model instance sharing the same weights (separate engine/workspace/graph)
den2 = M.GCDenoiser(copy.copy(den.inner_model), 0.5).eval()
den2.inner_model._engine = None; den2.inner_model._route_cache = {}
stA, xA, gA = make(64); stB, xB, gB = make(64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    with torch.cuda.stream(s1): M.sample_ddim(den, stA, xA, gA, sig, disable=True)
    with torch.cuda.stream(s2): M.sample_ddim(den2, stB, xB, gB, sig, disable=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20):
    with torch.cuda.stream(s1): M.sample_ddim(den, stA, xA, gA, sig, disable=True)
    with torch.cuda.stream(s2): M.sample_ddim(den2, stB, xB, gB, sig, disable=True)
torch.cuda.synchronize(); t2 = (time.perf_counter() - t) / 20
print(f"2 x B=64 concurrent streams: {t2*1e3:.2f} ms per pair  -> speedup vs B=128: {t128/t2:.3f}x")
