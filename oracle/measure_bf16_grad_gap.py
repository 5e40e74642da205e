"""Measure the REFERENCE's own fp32 vs bf16-autocast GRADIENT gap on CPU (build container only; imports /root/reference).

SURVEY.md §8 a-bis measured this gap for the forward outputs (rel-L2 4-6e-3, hence the stated bf16 forward tolerance of 1e-2).  The
training tests also need a bf16 tolerance for gradients; rather than picking one by fiat this script runs the reference's training step
(`GCDenoiser.loss` + backward, deterministic config of fixture F5: dropouts off, `use_argmax=True`) once in fp32 and once under
`torch.autocast("cpu", dtype=torch.bfloat16)` — the reference trains with `trainer.precision: bf16` (conf/config_calvin.yaml:37) — and
reports per-tensor rel-L2 and gradient-norm differences conditional on identical routing.

    python -m oracle.measure_bf16_grad_gap          # prints the table recorded in DESIGN.md §5 and writes tests/golden/bf16_grad_gap.json
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .gen_golden import OUT, _hook_router, _import_reference, _ref_model, _rel


def main():
    torch.set_num_threads(8)
    modedit, sw, gs, ut = _import_reference()
    from oracle.weights import get_config, make_inputs, make_state_dict
    rows = []
    summary = {}
    for cfgname, seed, B in (("c1e4", 210, 8), ("c1e4", 211, 32), ("c2block", 300, 32), ("c2block", 301, 32), ("c2block", 302, 16)):
        cfg = get_config(cfgname); sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
        g = torch.Generator().manual_seed(seed)
        from oracle import mode_oracle as O
        sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=g)
        res = {}
        for mode in ("fp32", "bf16"):
            m = _ref_model(modedit, cfg, sd, train=True, attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True)
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
        same_routing = bool(torch.equal(res["fp32"]["idx"], res["bf16"]["idx"]))
        g32, g16 = res["fp32"]["grads"], res["bf16"]["grads"]
        per = {}
        for k, a in g32.items():
            if k in g16 and float(a.norm()) > 1e-6:
                per[k] = (_rel(g16[k], a), abs(float(g16[k].norm()) - float(a.norm())) / float(a.norm()))
        rel_all = np.array([v[0] for v in per.values()]); nrm_all = np.array([v[1] for v in per.values()])
        worst = max(per, key=lambda k: per[k][0])
        row = dict(cfg=cfgname, B=B, same_routing=same_routing, loss_rel=abs(res["bf16"]["loss"] - res["fp32"]["loss"]) / abs(res["fp32"]["loss"]),
                   F_rel=_rel(res["bf16"]["F"], res["fp32"]["F"]), grad_rel_median=float(np.median(rel_all)), grad_rel_p90=float(np.percentile(rel_all, 90)),
                   grad_rel_max=float(rel_all.max()), grad_rel_max_tensor=worst, grad_norm_rel_max=float(nrm_all.max()), tensors=len(per))
        rows.append(row)
        print(json.dumps(row))
        if not same_routing:
            continue                                            # the reference's router runs in bf16 under autocast and may flip near-ties; the build's router is fp32
        # the groups the GPU tests check with separate tolerances
        for grp, pred in (("router", lambda k: "router" in k), ("experts", lambda k: "experts" in k), ("attn", lambda k: ".attn." in k),
                          ("other", lambda k: "router" not in k and "experts" not in k and ".attn." not in k)):
            v = [per[k][0] for k in per if pred(k)]
            if v:
                summary.setdefault(grp, []).append(max(v))
    out = dict(rows=rows, worst_by_group={k: float(max(v)) for k, v in summary.items()})
    print(json.dumps(out["worst_by_group"]))
    with open(os.path.join(OUT, "bf16_grad_gap.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
