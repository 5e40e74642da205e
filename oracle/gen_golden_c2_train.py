"""F18: the score-matching TRAINING step of the full-size model run by the REFERENCE itself - BASELINE configs[2]'s model (12 layers, d = 1024, 8 heads,
4 experts top-2, obs 2048, goal 512) at B = 16 (the reference's autograd on the host: ~1 min), deterministic config (dropouts off, `use_argmax=True`: the
stochastic path cannot be pinned to the reference - its RNG streams are torch's), per-sample log-logistic sigma, both auxiliary router losses in the total
(`mode_agent.py:386-440`: act + 0.01 lb + 0.001 z).  TEST INFRASTRUCTURE; build container only (imports /root/reference).

    python -m oracle.gen_golden_c2_train       # writes tests/golden/F18_c2_train.npz and prints oracle-vs-reference errors

Recorded: the three losses, the model output F [16, 10, 7], the expert ids of every (layer, sample), the gradient norm of EVERY parameter and the leading 512
elements (or all) of every gradient - twelve layers of backward depth pinned to the reference directly (F5 / F12 pin two layers).  The generator asserts
oracle-autograd-vs-reference <= 1e-4 on every gradient with a non-negligible norm."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .gen_golden import OUT, _hook_router, _import_reference, _margin, _ref_model, _rel

B, SEED, GAMMA, DELTA = 16, 400, 0.01, 0.001


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    modedit, sw, gs, ut = _import_reference()
    from oracle import mode_oracle as O
    from oracle.weights import get_config, make_inputs, make_state_dict
    cfg = get_config("c2")
    sd = make_state_dict(cfg, SEED)
    inp = make_inputs(cfg, B, SEED + 1)
    sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(SEED))
    t0 = time.time()
    m = _ref_model(modedit, cfg, sd, train=True, attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True)
    cap, hs = _hook_router(m)
    den = sw.GCDenoiser(m, 0.5).train()
    act, F_out = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
    lb = m.load_balancing_loss(); z = m.compute_router_z_loss()
    total = act + GAMMA * lb + DELTA * z
    total.backward()
    for h in hs:
        h.remove()
    print(f"reference training step: {time.time() - t0:.1f} s", flush=True)
    idx = torch.stack(cap["idx"])                                              # [L, B, T, k]
    assert (idx == idx[:, :, :1, :]).all(), "conditioning-row routing: every token of a sample routes alike"
    margin = _margin(cap["probs"], cfg.top_k)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    none = sorted(k for k, p in m.named_parameters() if p.grad is None)
    gn = {k: float(g.norm()) for k, g in grads.items()}
    t0 = time.time()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ot, oa, ol, oz = O.training_total_loss(sdg, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], sig, GAMMA, DELTA)
    ot.backward()
    print(f"oracle training step: {time.time() - t0:.1f} s", flush=True)
    worst = max((_rel(sdg[k].grad, grads[k]), k) for k in gn if gn[k] > 1e-6)
    e_t = abs(float(ot) - float(total)) / abs(float(total))
    print(f"oracle-vs-reference at full C2 depth: total loss {e_t:.2e}, lb {abs(float(ol) - float(lb)):.2e}, z {abs(float(oz) - float(z)):.2e}, worst gradient {worst[0]:.2e} ({worst[1]}), "
          f"top-k margin {margin:.2e}, parameters without gradient: {none}")
    assert e_t <= 1e-5 and worst[0] <= 1e-4
    keep = {}
    for k, g in grads.items():
        if g.numel() <= 4096:
            keep["g:" + k] = g.numpy()
        else:
            keep["gs:" + k] = g.reshape(-1)[:512].numpy()
    np.savez_compressed(os.path.join(OUT, "F18_c2_train.npz"), cfg="c2", B=B, seed=SEED, sigma=sig.numpy(), gamma=GAMMA, delta=DELTA, total=float(total), act=float(act),
                        lb=float(lb), z=float(z), F=F_out.detach().numpy(), topk_idx=idx[:, :, 0, :].numpy(), margin=margin, none=np.array(none),
                        gn_keys=np.array(list(gn.keys())), gn_vals=np.array(list(gn.values()), dtype=np.float64), oracle_err=np.array([e_t, worst[0]]), **keep)


if __name__ == "__main__":
    main()
