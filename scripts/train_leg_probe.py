"""Which earlier leg of the default bench.py run slows the training leg down?  python scripts/train_leg_probe.py  (GPU box)"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
legs = sys.argv[1].split(",") if len(sys.argv) > 1 else []
M, den = bench.build_model(dev, "bf16")
img, goal, x0 = bench.synthetic_inputs(dev, 128)
sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)
out = {}
if "sample" in legs:
    for _ in range(5):
        M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True)
    torch.cuda.synchronize()
if "roof" in legs:
    bench.dominant_kernel_roofline(den, dev)
if "burn" in legs:
    bench.sustained_mfma_peak(dev)
if "layers" in legs:
    bench.layer_kernel_breakdown(den, dev)
if "extras" in legs:
    bench.extra_measurements(M, den, dev)
if "flush" in legs:
    import gc
    gc.collect(); torch.cuda.empty_cache()
if "prealloc" in legs:                      # (set before everything else by the env var below)
    pass
with bench.PowerSampler(0, period_s=0.01) as ps:
    t = bench.train_leg(den, dev, 1, 0, None, steps=int(os.environ.get("PROBE_STEPS", "10")))
st = torch.cuda.memory_stats()
print(json.dumps({"legs": legs, "train_ms": t["train_ms_per_step"], "blocks": t["train_ms_per_step_blocks"],
                  "power": {k: v for k, v in ps.summary().items() if k in ("socket_w_avg", "sclk_mhz_avg", "sclk_mhz_min")},
                  "reserved_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 2), "allocated_gb": round(torch.cuda.memory_allocated() / 2 ** 30, 2),
                  "segments": st.get("segment.all.current")}), flush=True)
