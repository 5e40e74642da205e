"""State-machine screen of ONE model instance: a seeded random sequence of sampler calls (changing batch sizes, step counts and sigma ranges -
each a different hipGraph / schedule state), eval forwards, EDM forwards, routing-cache warm-ups and resets, in-place weight updates, a
training step in between - every result against the oracle evaluated with the weights the model holds at that moment.  What it hunts: stale
graph or schedule caches, workspaces re-allocated under a captured graph, bf16 shadows not refreshed after an update."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mode_diffusion_policy_amd as M  # noqa: E402
from oracle import mode_oracle as O  # noqa: E402
from oracle.weights import get_config, make_inputs, make_state_dict  # noqa: E402
from tolerances import BF16_LOSS, BF16_OUT, FP32_LOSS, FP32_OUT  # noqa: E402


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("dtype,tol", [("fp32", FP32_OUT), ("bf16", BF16_OUT)])
@pytest.mark.parametrize("seed", range(int(os.environ.get("MODE_FUZZ_STATE_SEEDS", "2"))))
def test_random_call_sequence_vs_oracle(seed, dtype, tol):
    cfg = get_config("c1e4")
    sd = make_state_dict(cfg, 210)
    m = M.MoDeDiT(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cuda", goal_conditioned=True, action_dim=cfg.action_dim,
                  embed_dim=cfg.embed_dim, embed_pdrob=0, attn_pdrop=0.0, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1,
                  obs_seq_len=1, action_seq_len=cfg.action_seq_len, state_dim=None, mlp_pdrop=0.0, goal_drop=0.0, num_experts=cfg.num_experts,
                  top_k=cfg.top_k, use_argmax=True, compute_dtype=dtype)
    m.load_state_dict(sd)
    m = m.to("cuda").eval()
    den = M.GCDenoiser(m, 0.5).eval()
    r = random.Random(seed)
    cur = {k: v.clone() for k, v in sd.items()}                     # the oracle's view of the weights
    log = []
    for step in range(36):
        op = r.choice(["ddim", "ddim", "ddim", "forward", "denoise", "precompute", "reset", "bump", "reload", "train"])
        B = r.choice([1, 2, 5, 16, 33])
        inp = make_inputs(cfg, B, 1000 + step)
        c = {k: v.cuda() for k, v in inp.items()}
        st = {"state_images": c["state_images"]}
        log.append((op, B))
        if op == "ddim":
            n = r.choice([3, 5, 10]); smax = r.choice([80.0, 20.0])
            sched = M.get_sigmas_exponential(n, 1e-3, smax)
            x = M.sample_ddim(den, st, c["x0"], c["goals"], sched.cuda(), disable=True)
            ref = O.sample_ddim(cur, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sched)
            assert rel(x, ref) < tol, (step, log)
        elif op in ("forward", "denoise"):
            sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(step))
            with torch.no_grad():
                if op == "forward":
                    out = m(st, c["actions"], c["goals"], sig.cuda())
                    ref = O.dit_forward(cur, cfg, inp["state_images"], inp["actions"], inp["goals"], sig)
                else:
                    out = den(st, c["actions"], c["goals"], sig.cuda())
                    ref = O.denoiser_forward(cur, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], sig)
            assert rel(out, ref) < tol, (step, log)
        elif op == "precompute":
            for s_ in M.get_sigmas_exponential(r.choice([3, 10]), 1e-3, 80.0)[:-1]:
                m.precompute_experts_for_inference(s_.cuda())
        elif op == "reset":
            m.reset_all_caches()
        elif op == "bump":                                          # in-place update of a few tensors (what an optimizer step / EMA swap does)
            with torch.no_grad():
                for name in r.sample(sorted(cur), 4):
                    if name == "gripper_embed.weight":
                        continue
                    delta = 0.02 * torch.randn(cur[name].shape, generator=torch.Generator().manual_seed(step))
                    dict(m.named_parameters())[name].add_(delta.cuda())
                    cur[name] = cur[name] + delta
        elif op == "reload":
            sd2 = make_state_dict(cfg, 300 + step)
            m.load_state_dict(sd2)
            cur = {k: v.clone() for k, v in sd2.items()}
        elif op == "train":                                         # one score-matching step with plain SGD on the arena gradients, mirrored on the oracle
            m.train(); den.train()
            sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(step))
            loss, _ = den.loss(st, c["actions"], c["goals"], c["noise"], sig.cuda())
            m.zero_grad(set_to_none=True)
            loss.backward()
            sdg = {k: v.clone().requires_grad_(True) for k, v in cur.items()}
            ref_loss, _ = O.denoiser_loss(sdg, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], sig)
            ref_loss.backward()
            assert abs(float(loss) - float(ref_loss)) < (FP32_LOSS if dtype == "fp32" else BF16_LOSS) * abs(float(ref_loss)), (step, log)
            with torch.no_grad():
                for name, p in m.named_parameters():
                    if p.grad is not None and sdg[name].grad is not None:
                        p.add_(p.grad, alpha=-1e-3)                                  # the model steps on ITS gradients ...
                        cur[name] = cur[name] - 1e-3 * p.grad.detach().cpu()         # ... and the oracle follows the same numbers (gradient parity is test_gpu_train's job)
            m.zero_grad(set_to_none=True)
            m.eval(); den.eval()
