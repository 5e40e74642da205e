"""Why is the default bench.py run's training leg slower than `bench.py --mode train` on the same box (VERDICT r05 #1a: 11.02 vs 10.36 ms)?
ONE process, the training leg re-timed after every earlier leg of the default run, then after an allocator flush and after a cool-down:

    python scripts/train_gap_probe.py [steps]        (GPU box)  ->  one JSON line per stage

Each stage prints the leg's block list, the socket clock / power over the leg and the allocator state, so that a clock / thermal effect can be told from a
memory-placement effect (reserved / allocated bytes, number of segments).
"""
import gc, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
M, den = bench.build_model(dev, "bf16")
img, goal, x0 = bench.synthetic_inputs(dev, 128)
sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)


def leg(tag):
    with bench.PowerSampler(0, period_s=0.01) as ps:
        t = bench.train_leg(den, dev, 1, 0, None, steps=steps)
    st = torch.cuda.memory_stats()
    print(json.dumps({"stage": tag, "train_ms": t["train_ms_per_step"], "blocks": t["train_ms_per_step_blocks"], "exposed": t["exposed_exchange_ms"],
                      "power": {k: v for k, v in ps.summary().items() if k in ("socket_w_avg", "sclk_mhz_avg", "sclk_mhz_min")},
                      "reserved_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 2), "allocated_gb": round(torch.cuda.memory_allocated() / 2 ** 30, 2),
                      "segments": st.get("segment.all.current"), "t": round(time.time() % 10000, 1)}), flush=True)


leg("fresh")
leg("fresh-again")
for _ in range(20):
    M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True)
torch.cuda.synchronize()
leg("after-sample")
bench.dominant_kernel_roofline(den, dev)
leg("after-roofline")
bench.sustained_mfma_peak(dev)
leg("after-burn")
bench.layer_kernel_breakdown(den, dev)
leg("after-layers")
bench.extra_measurements(M, den, dev)
leg("after-extras")
gc.collect(); torch.cuda.empty_cache()
leg("after-empty-cache")
time.sleep(20)
leg("after-20s-idle")
