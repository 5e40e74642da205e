"""Latency of the perceptual encoders in the rollout (eval mode, the reference's B = 1 and a batch of 32 environments): embed_visual_obs with two
FiLM-ResNet-50s at 224 x 224 under autocast(bf16), eager vs one hipGraph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd.perceptual_encoders import FiLMResNet50Policy, embed_visual_obs
dev = "cuda"
es, eg = FiLMResNet50Policy(512).to(dev).eval(), FiLMResNet50Policy(512).to(dev).eval()
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (1, 32):
    s = torch.randn(B, 1, 3, 224, 224, device=dev); g = torch.randn(B, 1, 3, 224, 224, device=dev); goal = torch.randn(B, 512, device=dev)
    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return embed_visual_obs(es, eg, s, g, goal)["state_images"]
    eager = t(run)
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    try:
        with torch.cuda.graph(gr):
            out = run()
        graphed = t(gr.replay)
        ok = float((out.float() - run().float()).abs().max())
    except Exception as e:                                                      # noqa: BLE001
        graphed, ok = float("nan"), repr(e)[:200]
    print(f"B={B}: eager {eager:.2f} ms, one graph replay {graphed:.2f} ms (max |diff| vs eager {ok})", flush=True)
