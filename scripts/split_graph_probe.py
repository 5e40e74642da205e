"""Round 6: the rollout's sampler chain as two half-batch BRANCHES of one captured hipGraph against the one-branch graph, per batch size.
python scripts/split_graph_probe.py [B ...]      (GPU box; needs scripts/probe/split_graph_branches.patch applied to mode_diffusion_policy_amd/modedit.py - the two-branch capture was measured and NOT shipped, profiles/r06_split_graph.txt)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
M, den = bench.build_model(dev)
sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)
for s_ in sig[:-1]:
    den.inner_model.precompute_experts_for_inference(s_)
for B in [int(x) for x in sys.argv[1:]] or [32, 16, 24, 48, 64, 128]:
    img, goal, x0 = bench.synthetic_inputs(dev, B)
    res, outs = {}, {}
    for rep in range(2):
        for name, env in (("one branch", "0"), ("two branches", "1")):
            os.environ["MODE_SPLIT_GRAPH"] = env
            os.environ["MODE_SPLIT_GRAPH_RANGE"] = "2,100000"
            for _ in range(3):
                out = M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True)
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / 30 * 1e3)
            outs[name] = out
    print(f"B = {B:3d}: " + "   ".join(f"{k} {min(v):6.3f} ms per chunk" for k, v in res.items()) + f"   bit-identical: {torch.equal(outs['one branch'], outs['two branches'])}", flush=True)
