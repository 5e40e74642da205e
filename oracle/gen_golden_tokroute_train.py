"""Golden fixture F16: F12's composition (training loss + auxiliary router losses) with ``cond_router=False`` - every block routes each TOKEN on its
own ln_2-normalised state (modedit.py:296-301, 322-325, 550-553), so the router gradients also flow into the token states (build container only).

`MoDEAgent.training_step` (mode_agent.py:386-440) composes `act_loss + entropy_gamma * load_balancing_loss() + router_z_delta *
compute_router_z_loss()`; the YAML comment (conf/model/mode_agent.yaml:5) recommends entropy_gamma 0.01 for training from scratch.  The agent
itself cannot be imported here (Lightning etc.), so the composition is restated on the REAL reference modules (`GCDenoiser.loss`,
`MoDeDiT.load_balancing_loss`, `MoDeDiT.compute_router_z_loss`) and autograd produces the gradients.  Deterministic config of F5 (dropouts
off, use_argmax=True, cond_router=False).  Also checks the oracle's autograd of the same composition against it.

    python -m oracle.gen_golden_tokroute_train          # writes tests/golden/F16_c1e4_tokroute_loss_grad.npz
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .gen_golden import OUT, _import_reference, _ref_model, _rel

GAMMA, DELTA = 0.01, 0.001


def main():
    torch.set_num_threads(8)
    modedit, sw, gs, ut = _import_reference()
    from oracle import mode_oracle as O
    from oracle.weights import get_config, make_inputs, make_state_dict
    cfgname, seed, B = "c1e4", 210, 8
    import dataclasses
    cfgname, seed = "c1e4", 232
    cfg = dataclasses.replace(get_config(cfgname), cond_router=False); sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
    sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(seed))
    m = _ref_model(modedit, cfg, sd, train=True, attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True, cond_router=False)
    den = sw.GCDenoiser(m, 0.5).train()
    from .gen_golden import _hook_router, _margin
    cap, hs = _hook_router(m)
    img_in = inp["state_images"].clone().requires_grad_(True)
    act, _ = den.loss({"state_images": img_in}, inp["actions"], inp["goals"], inp["noise"], sig)
    for h in hs:
        h.remove()
    lb = m.load_balancing_loss(); z = m.compute_router_z_loss()
    total = act + GAMMA * lb + DELTA * z
    total.backward()
    ref_idx = torch.stack(cap["idx"])                                    # [L, B, T, k]
    print("token-routing top-k margin", _margin(cap["probs"], cfg.top_k))
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    gn = {k: float(g.norm()) for k, g in grads.items()}
    # oracle autograd of the same composition
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ot, oa, ol, oz = O.training_total_loss(sdg, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], sig, GAMMA, DELTA)
    ot.backward()
    worst = max(_rel(sdg[k].grad, grads[k]) for k in gn if gn[k] > 1e-6)
    print(dict(total=float(total), act=float(act), lb=float(lb), z=float(z), oracle_total_rel=abs(float(ot) - float(total)) / abs(float(total)),
               oracle_lb_abs=abs(float(ol) - float(lb)), oracle_z_abs=abs(float(oz) - float(z)), oracle_grad_worst_rel=worst))
    # how much the aux terms move the router gradients (sanity: the fixture must be able to tell "no gradient path" from the real thing)
    m2 = _ref_model(modedit, cfg, sd, train=True, attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True, cond_router=False)
    a2, _ = sw.GCDenoiser(m2, 0.5).train().loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
    a2.backward()
    k = "blocks.0.router.router.mlp.3.weight"
    print("router grad shift by the aux terms (rel):", _rel(grads[k], dict(m2.named_parameters())[k].grad) )
    keep = {}
    for kk, g in grads.items():
        if g.numel() <= 4096:
            keep["g:" + kk] = g.numpy()
        else:
            keep["gs:" + kk] = g.reshape(-1)[:2048].numpy()
    np.savez(os.path.join(OUT, f"F16_{cfgname}_tokroute_loss_grad.npz"), cfg=cfgname, B=B, seed=seed, sigma=sig.numpy(), gamma=GAMMA, delta=DELTA,
             total=float(total), act=float(act), lb=float(lb), z=float(z), idx=ref_idx.numpy().astype(np.int32), dimg=img_in.grad.numpy(),
             margin=_margin(cap["probs"], cfg.top_k), gn_keys=np.array(list(gn.keys())),
             gn_vals=np.array(list(gn.values()), dtype=np.float64), **keep)


if __name__ == "__main__":
    main()
