"""Golden fixture F14: the reference with ``cond_router=False`` (every block routes each token on its own normalised state, modedit.py:296-301,
322-325, 550-553) — eval forward at a shared and at per-sample noise levels and the 10-step DDIM chunk, with the per-layer per-token expert ids
and their top-k margins (build container only; imports /root/reference).  Also checks the oracle's token-routing branch against it.

    python -m oracle.gen_golden_tokroute      # writes tests/golden/F14_c1e4_token_routing.npz
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .gen_golden import OUT, _hook_router, _import_reference, _margin, _ref_model, _rel


def main():
    torch.set_num_threads(8)
    modedit, sw, gs, ut = _import_reference()
    import dataclasses
    from oracle import mode_oracle as O
    from oracle.weights import get_config, make_inputs, make_state_dict
    cfgname, seed, B = "c1e4", 230, 8
    cfg = dataclasses.replace(get_config(cfgname), cond_router=False)
    sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, B, seed + 1)
    m = _ref_model(modedit, cfg, sd, cond_router=False)
    out = {}
    sig_u = torch.tensor(1.857)
    sig_p = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(seed))
    for tag, sig in (("uniform", sig_u * torch.ones(B)), ("persample", sig_p)):
        cap, hs = _hook_router(m)
        with torch.no_grad():
            y = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sig)
        for h in hs:
            h.remove()
        idx = torch.stack(cap["idx"])                                    # [L, B, T, k]
        oy, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sig, return_aux=True)
        oidx = torch.stack(aux.topk_idx)
        assert torch.equal(idx, oidx), f"{tag}: oracle token-routing indices differ from the reference"
        print(tag, "oracle vs reference rel-L2", _rel(oy, y), "top-k margin", _margin(cap["probs"], cfg.top_k),
              "distinct routings per layer", [int(torch.unique(idx[l].reshape(-1, cfg.top_k), dim=0).shape[0]) for l in range(cfg.n_layers)])
        out[f"{tag}_sigma"] = sig.numpy(); out[f"{tag}_out"] = y.numpy(); out[f"{tag}_idx"] = idx.numpy().astype(np.int32)
        out[f"{tag}_margin"] = _margin(cap["probs"], cfg.top_k)
    # 10-step DDIM through the reference sampler
    den = sw.GCDenoiser(m, 0.5).eval()
    sig = gs.get_sigmas_exponential(10, 1e-3, 80.0)
    with torch.no_grad():
        x = gs.sample_ddim(den, {"state_images": inp["state_images"]}, inp["x0"], inp["goals"], sig, disable=True)
    ox = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sig)
    print("ddim oracle vs reference rel-L2", _rel(ox, x))
    np.savez(os.path.join(OUT, "F14_c1e4_token_routing.npz"), cfg=cfgname, B=B, seed=seed, sigmas=sig.numpy(), x_final=x.numpy(), **out)


if __name__ == "__main__":
    main()
