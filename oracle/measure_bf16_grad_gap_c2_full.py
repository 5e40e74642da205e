"""The REFERENCE's own fp32 vs bf16-autocast gradient gap at FULL depth (12 layers, d = 1024, 4 experts top-2) - build container only; imports
/root/reference.  TEST INFRASTRUCTURE: it grounds the tolerance of tests/test_gpu_c2_full.py::test_c2_full_training_step_vs_oracle_*.

oracle/measure_bf16_grad_gap.py measured one- and two-block models (worst tensor 3.8e-2).  Through twelve blocks the bf16 rounding of the
activations accumulates on the way down AND on the way back up, so the first blocks' gradients carry more of it.  Here the reference's training
step (`GCDenoiser.loss` + backward; deterministic config: dropouts off, `use_argmax=True`) runs once in fp32 and once under
`torch.autocast("cpu", dtype=torch.bfloat16)` (conf/config_calvin.yaml:37) with ONE change that mirrors this build: the router MLP of every
block is kept in fp32 (autocast disabled inside `RouterCond.forward`, inputs cast to float) - the build's router is fp32 by contract, and only
then is "conditional on identical routing" (asserted) obtainable at this depth.  One noise level for the whole batch (12 routing decisions).

    python -m oracle.measure_bf16_grad_gap_c2_full      # writes tests/golden/bf16_grad_gap_c2_full.json
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

from .gen_golden import OUT, _hook_router, _import_reference, _ref_model, _rel


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    modedit, sw, gs, ut = _import_reference()
    from oracle.weights import get_config, make_inputs, make_state_dict
    cfg = get_config("c2")
    rows = []
    for seed, B, sigma in ((400, 16, 0.7), (401, 16, 2.5)):
        sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
        sig = torch.full((B,), float(sigma))
        res = {}
        for mode in ("fp32", "bf16"):
            t0 = time.time()
            m = _ref_model(modedit, cfg, sd, train=True, attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True)
            for blk in m.blocks:                                             # fp32 router inside the autocast region (this build's contract)
                orig = blk.router.forward

                def fwd(inputs, cond=None, _orig=orig):
                    with torch.autocast("cpu", enabled=False):
                        return _orig(inputs.float(), None if cond is None else cond.float())
                blk.router.forward = fwd
            cap, hs = _hook_router(m)
            den = sw.GCDenoiser(m, 0.5).train()
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "bf16" else torch.autocast("cpu", enabled=False)
            with ctx:
                loss, F_out = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
            loss.float().backward()
            for h in hs:
                h.remove()
            res[mode] = dict(loss=float(loss), F=F_out.detach().float(), idx=torch.stack(cap["idx"]),
                             grads={k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None})
            print(mode, "done in", round(time.time() - t0, 1), "s", flush=True)
            del m, den
        same = bool(torch.equal(res["fp32"]["idx"], res["bf16"]["idx"]))
        g32, g16 = res["fp32"]["grads"], res["bf16"]["grads"]
        per = {k: _rel(g16[k], a) for k, a in g32.items() if k in g16 and float(a.norm()) > 1e-6}
        v = np.array(list(per.values()))
        top = sorted(per, key=per.get, reverse=True)[:8]
        by_block = {}
        for k, e in per.items():
            if k.startswith("blocks."):
                b = int(k.split(".")[1])
                by_block[b] = max(by_block.get(b, 0.0), e)
        row = dict(cfg="c2", layers=cfg.n_layers, B=B, sigma=sigma, seed=seed, same_routing=same,
                   loss_rel=abs(res["bf16"]["loss"] - res["fp32"]["loss"]) / abs(res["fp32"]["loss"]), F_rel=_rel(res["bf16"]["F"], res["fp32"]["F"]),
                   grad_rel_median=float(np.median(v)), grad_rel_p90=float(np.percentile(v, 90)), grad_rel_max=float(v.max()),
                   worst_tensors={k: per[k] for k in top}, worst_per_block=[by_block[b] for b in sorted(by_block)], tensors=len(per))
        rows.append(row)
        print(json.dumps(row), flush=True)
    out = dict(rows=rows, note="reference fp32 vs reference under bf16 autocast with an fp32 router, full 12-layer C2 model, train-mode deterministic config")
    with open(os.path.join(OUT, "bf16_grad_gap_c2_full.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
