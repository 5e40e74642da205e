"""Same-process A/B of one mode_set_option switch on the training step of bench.py (C2, B = 128, fused AdamW overlapped): the two settings alternate,
three timed legs each.  python scripts/train_ab.py <option> [values ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
opt = sys.argv[1].encode(); vals = [int(x) for x in sys.argv[2:]] or [0, 1]
dev = torch.device("cuda:0")
M, den = bench.build_model(dev, "bf16")
lib = den.inner_model.engine.lib
res = {v: [] for v in vals}
for rep in range(3):
    for v in vals:
        assert lib.mode_set_option(opt, v) == 0
        t = bench.train_leg(den, dev, 1, 0, None, steps=10, warmup=3)
        res[v].append(t["train_ms_per_step"])
for v in vals:
    print(f"{opt.decode()} = {v}: " + " ".join(f"{x:.3f}" for x in res[v]) + " ms per step")
