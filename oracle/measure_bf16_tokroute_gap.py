"""The REFERENCE's own fp32 vs bf16-autocast behaviour under TOKEN routing (``cond_router=False``, modedit.py:296-301, 322-325, 550-553) - build
container only; imports /root/reference.  TEST INFRASTRUCTURE: grounds tests/tolerances.py BF16_TOKROUTE_*.

With token routing the router reads the ln_2-normalised TOKEN states, which are themselves products of the block's GEMMs: in bf16 they carry
rounding noise, so tokens whose top-k margin is below that noise route differently than in fp32 - in the reference under autocast exactly as in
this build's bf16 mode (whose router MLP is fp32, but whose router INPUT went through bf16 attention / c_proj / expert GEMMs of the blocks before).
A bit-exact index match in bf16 is therefore not obtainable for this flag in any implementation that computes the blocks in bf16; this script
records what the reference itself does on fixture F14's configuration: the share of identical token decisions and the output gap, (a) autocast
everywhere like the reference trains, (b) autocast with the router MLP kept in fp32 (this build's arrangement).

    python -m oracle.measure_bf16_tokroute_gap      # writes tests/golden/bf16_tokroute_gap.json
"""
from __future__ import annotations

import dataclasses
import json
import os

import numpy as np
import torch

from .gen_golden import OUT, _hook_router, _import_reference, _margin, _ref_model, _rel


def main():
    torch.set_num_threads(8)
    modedit, sw, gs, ut = _import_reference()
    from oracle import mode_oracle as O
    from oracle.weights import get_config, make_inputs, make_state_dict
    rows = []
    for cfgname, seed, B in (("c1e4", 230, 8), ("c1e4", 231, 32), ("c2block", 300, 32)):
        cfg = dataclasses.replace(get_config(cfgname), cond_router=False)
        sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
        sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(seed))
        res = {}
        for mode in ("fp32", "autocast", "autocast_fp32_router"):
            m = _ref_model(modedit, cfg, sd, cond_router=False)
            if mode == "autocast_fp32_router":
                for blk in m.blocks:
                    orig = blk.router.forward

                    def fwd(inputs, cond=None, _orig=orig):
                        with torch.autocast("cpu", enabled=False):
                            return _orig(inputs.float(), None if cond is None else cond.float())
                    blk.router.forward = fwd
            cap, hs = _hook_router(m)
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode != "fp32" else torch.autocast("cpu", enabled=False)
            with torch.no_grad(), ctx:
                y = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sig)
            for h in hs:
                h.remove()
            res[mode] = dict(y=y.float(), idx=torch.stack(cap["idx"]), margin=_margin(cap["probs"], cfg.top_k))
        row = dict(cfg=cfgname, B=B, seed=seed, fp32_margin=res["fp32"]["margin"])
        for mode in ("autocast", "autocast_fp32_router"):
            a, b = res["fp32"]["idx"].sort(-1).values, res[mode]["idx"].sort(-1).values
            row[mode] = dict(same_decisions=float((a == b).all(-1).float().mean()), out_rel=_rel(res[mode]["y"], res["fp32"]["y"]))
        rows.append(row)
        print(json.dumps(row), flush=True)
    with open(os.path.join(OUT, "bf16_tokroute_gap.json"), "w") as f:
        json.dump(dict(rows=rows, note="reference fp32 vs reference under bf16 autocast, cond_router=False, eval forward at per-sample noise levels"), f, indent=1)


if __name__ == "__main__":
    main()
