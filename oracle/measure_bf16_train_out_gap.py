"""The REFERENCE's own fp32 vs bf16-autocast gap of the TRAINING forward's model output F at full size - build container only; imports /root/reference.
TEST INFRASTRUCTURE: it grounds `BF16_TRAIN_OUT` (tests/tolerances.py), the tolerance tests/test_gpu_c2_full.py holds the HIP chain's training-mode output
to (VERDICT r05 "weak" #2: the number used to be "the envelope of the gradients", 4e-2, with no committed measurement behind it).

What is measured, on the reference's `GCDenoiser.loss` (mode/models/edm_diffusion/score_wrappers.py:45-63) over the full model (12 layers, d = 1024, 4 experts
top-2) in train mode, once in fp32 and once under `torch.autocast("cpu", dtype=torch.bfloat16)` (conf/config_calvin.yaml:37) with the router MLPs and the sigma embedding they
read kept in fp32 (this build's contract - and what makes "identical routing" obtainable over 1536 per-sample decisions, asserted):

  * PER-SAMPLE log-logistic sigma (mode_agent.py:452-466: loc log 0.5, scale 0.5, clipped to [1e-3, 80]) - small noise levels included, where c_in is
    large and the prediction target is the noise itself;
  * deterministic routing (`use_argmax=True`, dropouts off)  - the F18 configuration - at B = 16 and B = 128;
  * the stochastic path at B = 128: multinomial routing, attention dropout 0.3, expert dropout 0.1.  Both runs must see the SAME masks and draws, so
    `torch.nn.functional.dropout` is replaced for the duration by a mask drawn in fp32 from the global generator (the stock CPU kernels consume the
    generator differently for bf16 and fp32 inputs), and both runs start from the same seed; the multinomial draws come from the fp32 router's
    probabilities in both.  Routing equality is asserted, mask equality follows from the construction.

    python -m oracle.measure_bf16_train_out_gap       # writes tests/golden/bf16_train_out_gap.json
"""
from __future__ import annotations

import json
import math
import os
import time

import torch

from .gen_golden import OUT, _hook_router, _import_reference, _ref_model, _rel


def _fp32_mask_dropout(x, p=0.5, training=True, inplace=False):
    if not training or p <= 0.0:
        return x
    keep = (torch.rand(x.shape, dtype=torch.float32) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    modedit, sw, gs, ut = _import_reference()
    from oracle.mode_oracle import rand_log_logistic
    from oracle.weights import get_config, make_inputs, make_state_dict
    cfg = get_config("c2")
    rows = []
    cases = [("deterministic", 16, 400, dict(attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True)),
             ("deterministic", 128, 402, dict(attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True)),
             ("stochastic", 128, 403, dict(attn_pdrop=0.3, mlp_pdrop=0.1, goal_drop=0.0, use_argmax=False)),
             ("stochastic", 128, 404, dict(attn_pdrop=0.3, mlp_pdrop=0.1, goal_drop=0.0, use_argmax=False))]
    real_dropout = torch.nn.functional.dropout
    for name, B, seed, over in cases:
        sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
        sig = rand_log_logistic((B,), math.log(0.5), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(seed + 2))
        res = {}
        for mode in ("fp32", "bf16"):
            t0 = time.time()
            m = _ref_model(modedit, cfg, sd, train=True, **over)
            for blk in m.blocks:                                             # fp32 router inside the autocast region (this build's contract)
                orig = blk.router.forward

                def fwd(inputs, cond=None, _orig=orig):
                    with torch.autocast("cpu", enabled=False):
                        return _orig(inputs.float(), None if cond is None else cond.float())
                blk.router.forward = fwd
            orig_se = m.process_sigma_embeddings                             # ... and so is the sigma embedding the router reads (fp32 in this build: rowops.hip / gemm_f32)

            def se(sigma, _orig=orig_se):
                with torch.autocast("cpu", enabled=False):
                    return _orig(sigma.float())
            m.process_sigma_embeddings = se
            cap, hs = _hook_router(m)
            den = sw.GCDenoiser(m, 0.5).train()
            torch.nn.functional.dropout = _fp32_mask_dropout
            try:
                torch.manual_seed(seed + 3)
                ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "bf16" else torch.autocast("cpu", enabled=False)
                with torch.no_grad(), ctx:
                    loss, F_out = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
            finally:
                torch.nn.functional.dropout = real_dropout
            for h in hs:
                h.remove()
            res[mode] = dict(loss=float(loss), F=F_out.detach().float(), idx=torch.stack(cap["idx"]))
            print(name, B, mode, "done in", round(time.time() - t0, 1), "s", flush=True)
            del m, den
        same = bool(torch.equal(res["fp32"]["idx"], res["bf16"]["idx"]))
        F32, F16 = res["fp32"]["F"], res["bf16"]["F"]
        per_sample = ((F16 - F32).flatten(1).norm(dim=1) / F32.flatten(1).norm(dim=1).clamp_min(1e-30))
        lo = sig < 0.1
        row = dict(case=name, cfg="c2", layers=cfg.n_layers, B=B, seed=seed, same_routing=same, sigma_min=float(sig.min()), sigma_max=float(sig.max()),
                   loss_rel=abs(res["bf16"]["loss"] - res["fp32"]["loss"]) / abs(res["fp32"]["loss"]), F_rel=_rel(F16, F32),
                   F_rel_per_sample_max=float(per_sample.max()), F_rel_per_sample_median=float(per_sample.median()),
                   F_rel_sigma_below_0p1=(_rel(F16[lo], F32[lo]) if bool(lo.any()) else None), n_sigma_below_0p1=int(lo.sum()))
        assert same, "routing differs between the fp32 and the autocast run: the gap below would not be conditional on identical routing"
        rows.append(row)
        print(json.dumps(row), flush=True)
    env = max(r["F_rel"] for r in rows)
    out = dict(rows=rows, F_rel_envelope=env,
               note="reference GCDenoiser.loss, fp32 vs bf16 autocast with an fp32 router, full 12-layer C2 model, TRAIN mode, per-sample log-logistic sigma; "
                    "stochastic rows: identical multinomial draws and dropout masks in both runs (fp32-drawn masks)")
    with open(os.path.join(OUT, "bf16_train_out_gap.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
