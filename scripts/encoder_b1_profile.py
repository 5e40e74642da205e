"""The rollout's two FiLM-ResNet-50 encoders at B = 1 (224 x 224, bf16 autocast, eval), eager and as GraphedVisualEncoder replays: wall time per call.
Under rocprofv3 --kernel-trace the per-kernel durations of the replays show where a replanning call's encoder share goes.  python scripts/encoder_b1_profile.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mode_diffusion_policy_amd.perceptual_encoders import FiLMResNet50Policy, GraphedVisualEncoder, embed_visual_obs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
torch.manual_seed(0)
es, eg = FiLMResNet50Policy(512).to(dev).eval(), FiLMResNet50Policy(512).to(dev).eval()
rs = torch.randn(B, 1, 3, 224, 224, device=dev); rg = torch.randn(B, 1, 3, 224, 224, device=dev); goal = torch.randn(B, 512, device=dev)
gve = GraphedVisualEncoder(es, eg)
def eager():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return embed_visual_obs(es, eg, rs, rg, goal)
for fn, name in ((eager, "eager"), (lambda: gve(rs, rg, goal), "graphed")):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    print(f"B={B} {name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call")
