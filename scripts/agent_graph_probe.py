"""Agent training step, eager towers vs GraphedTrainingTowers: GPU time per phase (events on the main stream) and host enqueue time per phase.
python scripts/agent_graph_probe.py eager|graphed"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mode_diffusion_policy_amd.optim import FlatAdamW, FusedAdamW  # noqa: E402
from mode_diffusion_policy_amd.perceptual_encoders import FiLMResNet50Policy, embed_visual_obs  # noqa: E402
try:                                                                       # the probe class is not part of the shipped package: scripts/probe/graphed_training_towers.py.txt
    from mode_diffusion_policy_amd.perceptual_encoders import GraphedTrainingTowers  # noqa: E402
except ImportError:
    GraphedTrainingTowers = None
from mode_diffusion_policy_amd.utils import rand_log_logistic  # noqa: E402

dev = torch.device("cuda:0")
M, den = bench.build_model(dev)
m = den.inner_model
den.train()
B = 64
torch.manual_seed(0)
enc_s, enc_g = FiLMResNet50Policy(512).to(dev).train(), FiLMResNet50Policy(512).to(dev).train()
g = torch.Generator().manual_seed(1)
rgb_s = torch.randn(B, 1, 3, 224, 224, generator=g).to(dev); rgb_g = torch.randn(B, 1, 3, 224, 224, generator=g).to(dev)
goal = torch.randn(B, 1, 512, generator=g).to(dev)
acts = torch.randn(B, 10, 7, generator=g).to(dev); noise = torch.randn(B, 10, 7, generator=g).to(dev)
opt = FusedAdamW(m, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, fuse_expert_step=True)
opt_e = FlatAdamW(list(enc_s.parameters()) + list(enc_g.parameters()), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
import gc
for mode in sys.argv[1:] or ["eager"]:                         # one mode per process: a capture behind eager steps of the same objects aborted in the runtime
    gc.collect(); gc.disable()
    towers = GraphedTrainingTowers(enc_s, enc_g, rgb_s, rgb_g, goal.squeeze(1)) if mode == "graphed" else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    host = [0.0] * 4
    gpu = [0.0] * 4

    def step(rec):
        sig = rand_log_logistic((B,), loc=math.log(0.5), scale=0.5, min_value=1e-3, max_value=80.0, device=dev)
        t = [time.perf_counter()]
        ev[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            emb = towers(rgb_s, rgb_g, goal.squeeze(1)) if towers is not None else embed_visual_obs(enc_s, enc_g, rgb_s, rgb_g, goal.squeeze(1))
            ev[1].record(); t.append(time.perf_counter())
            loss, _ = den.loss(emb, acts, goal, noise, sig)
        ev[2].record(); t.append(time.perf_counter())
        loss.backward()
        ev[3].record(); t.append(time.perf_counter())
        opt.step(); opt_e.step(); opt_e.zero_grad(set_to_none=True)
        ev[4].record(); t.append(time.perf_counter())
        if rec:
            torch.cuda.synchronize()
            for i in range(4):
                host[i] += (t[i + 1] - t[i]) * 1e3; gpu[i] += ev[i].elapsed_time(ev[i + 1])
    for _ in range(3):
        step(False)
    torch.cuda.synchronize()
    n = 5
    for _ in range(n):
        step(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        step(False)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 10 * 1e3
    names = ["encoders fwd", "denoiser fwd", "backward (denoiser + encoders)", "optimizers"]
    print(f"{mode:8s}: {tot:6.2f} ms per step | " + " | ".join(f"{nm}: gpu {gpu[i] / n:5.2f} host {host[i] / n:5.2f}" for i, nm in enumerate(names)), flush=True)
    del towers
