"""Do two independent half-batch sampler chains on two streams beat one chain?  (A sample's result does not depend on the batch it is in - asserted bit for
bit in tests/test_gpu_model.py - so splitting the environments of a rollout batch over concurrent hipGraphs is free of numerical consequences.)  At the
rollout batch sizes every kernel of the chain is latency / fixed-cost bound and leaves most CUs idle; two chains interleave their fixed costs.
Probe form: two model instances with the same weights (separate workspaces and graphs).  Usage: python scripts/split_batch_probe.py [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
M, den_a = bench.build_model(dev)
_, den_b = bench.build_model(dev)
den_b.inner_model.load_state_dict(den_a.inner_model.state_dict())
sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)
for d in (den_a, den_b):
    for s_ in sig[:-1]:
        d.inner_model.precompute_experts_for_inference(s_)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
cur = torch.cuda.current_stream()
for B in [int(x) for x in sys.argv[1:]] or [32, 16, 64, 128, 8]:
    img, goal, x0 = bench.synthetic_inputs(dev, B)
    h = B // 2
    full = lambda: M.sample_ddim(den_a, {"state_images": img}, x0, goal, sig, disable=True)
    parts = [(den_a, sa, img[:h].contiguous(), goal[:h].contiguous(), x0[:h].contiguous()), (den_b, sb, img[h:].contiguous(), goal[h:].contiguous(), x0[h:].contiguous())]

    def split():
        outs = []
        for den, st, i_, g_, x_ in parts:
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(M.sample_ddim(den, {"state_images": i_}, x_, g_, sig, disable=True))
        for _, st, *_r in parts:
            cur.wait_stream(st)
        return torch.cat(outs)
    for _ in range(3):
        ref = full(); got = split()
    torch.cuda.synchronize()
    same = torch.equal(ref, got)
    res = {}
    for name, fn in (("one chain", full), ("two half-batch chains", split), ("one chain", full), ("two half-batch chains", split)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res.setdefault(name, []).append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"B = {B:3d}: " + "   ".join(f"{k} {min(v):6.3f} ms per chunk" for k, v in res.items()) + f"   bit-identical: {same}", flush=True)

# ---- round 6: the two half-batch chains as two BRANCHES OF ONE captured hipGraph (fork / join of the capture stream)
from mode_diffusion_policy_amd.engine import capture_graph  # noqa: E402
os.environ["MODE_HIP_GRAPH"] = "0"                          # inside the capture the chains are launched eagerly (the capture records them)
for B in [int(x) for x in sys.argv[1:]] or [32, 16, 64, 128]:
    img, goal, x0 = bench.synthetic_inputs(dev, B)
    h = B // 2
    parts = [(den_a, sa, img[:h].contiguous(), goal[:h].contiguous(), x0[:h].contiguous()), (den_b, sb, img[h:].contiguous(), goal[h:].contiguous(), x0[h:].contiguous())]
    os.environ["MODE_HIP_GRAPH"] = "1"
    ref = M.sample_ddim(den_a, {"state_images": img}, x0, goal, sig, disable=True)
    os.environ["MODE_HIP_GRAPH"] = "0"
    for den, st, i_, g_, x_ in parts:                       # warm-up outside the capture (schedule state, code objects)
        M.sample_ddim(den, {"state_images": i_}, x_, g_, sig, disable=True)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        outs = []
        with capture_graph(g):
            cap = torch.cuda.current_stream()
            for den, st, i_, g_, x_ in parts:
                st.wait_stream(cap)
                with torch.cuda.stream(st):
                    outs.append(M.sample_ddim(den, {"state_images": i_}, x_, g_, sig, disable=True))
            for _, st, *_r in parts:
                cap.wait_stream(st)
        g.replay(); torch.cuda.synchronize()
        got = torch.cat(outs)
        same = torch.equal(ref, got)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        os.environ["MODE_HIP_GRAPH"] = "1"
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            M.sample_ddim(den_a, {"state_images": img}, x0, goal, sig, disable=True)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / 20 * 1e3
        print(f"B = {B:3d}: ONE graph with two half-batch branches {min(ts):6.3f} ms per chunk   one chain {one:6.3f}   bit-identical: {same}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"B = {B:3d}: branched capture failed: {e!r}"[:400], flush=True)
    os.environ["MODE_HIP_GRAPH"] = "0"
