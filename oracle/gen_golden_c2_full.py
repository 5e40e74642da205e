"""F17: the BENCHMARKED workload run by the REFERENCE itself - BASELINE configs[1] at full size (12 layers, d = 1024, 8 heads, 4 experts
top-2, obs 2048, goal 512, B = 128, seed 400 = the inputs of tests/test_gpu_c2_full.py).  TEST INFRASTRUCTURE; build container only
(imports /root/reference; the reference never travels to the GPU box).

    python -m oracle.gen_golden_c2_full        # writes tests/golden/F17_c2_full.npz (~80 KB) and prints oracle-vs-reference errors

What is recorded, all produced by the reference classes (`MoDeDiT.forward` modedit.py:741-809 inside `GCDenoiser` score_wrappers.py:18-100
inside `sample_ddim` gc_sampling.py:922-951) on CPU fp32:
  * `forward`   [128, 10, 7]  MoDeDiT.forward at sigma = sched[3] for the whole batch
  * `x_final`   [128, 10, 7]  the 10-step DDIM chunk from `x0`
  * `action_in` [10, 4, 10, 7] the sampler's `action` callback value (the input of step i = the DDIM update of step i-1) of samples 0-3
                               (localises a drift to a step)
  * `topk_idx`  [10, 12, 2]    the router's expert ids per (sampler step, layer) - one sigma per step, so every token of every sample routes alike
                               (asserted before the reduction)
  * `fwd_topk_idx` [12, 2], `margin` (smallest top-k probability gap over all 11 forwards)
Weights / inputs are regenerated from oracle/weights.py seeds; the fixture holds seeds + outputs only.  The generator ASSERTS
oracle-vs-reference <= 1e-5 on both outputs and identical expert ids, so twelve layers of depth are pinned to the reference directly
(before this fixture the oracle was pinned at 2 layers / one C2-sized block and the full-size comparison was HIP <-> oracle only)."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .gen_golden import OUT, _hook_router, _import_reference, _margin, _ref_model, _rel

B, SEED = 128, 400


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    modedit, sw, gs, ut = _import_reference()
    from oracle import mode_oracle as O
    from oracle.weights import get_config, make_inputs, make_state_dict
    cfg = get_config("c2")
    sd = make_state_dict(cfg, SEED)
    inp = make_inputs(cfg, B, SEED + 1)
    sched = gs.get_sigmas_exponential(10, 1e-3, 80.0)
    m = _ref_model(modedit, cfg, sd)
    st = {"state_images": inp["state_images"]}

    t0 = time.time()
    cap, hs = _hook_router(m)
    with torch.no_grad():
        fwd = m(st, inp["actions"], inp["goals"], sched[3] * torch.ones(B))
    for h in hs:
        h.remove()
    fidx = torch.stack(cap["idx"])                                             # [L, B, T, k]
    assert (fidx == fidx[:, :1, :1, :]).all(), "one sigma for the batch: every token must route alike"
    fmargin = _margin(cap["probs"], cfg.top_k)
    print(f"reference forward: {time.time() - t0:.1f} s", flush=True)

    den = sw.GCDenoiser(m, 0.5).eval()
    cap, hs = _hook_router(m)
    trace = []
    t0 = time.time()
    with torch.no_grad():
        x = gs.sample_ddim(den, st, inp["x0"], inp["goals"], sched, disable=True, callback=lambda d: trace.append(d["action"][:4].clone()))
    for h in hs:
        h.remove()
    print(f"reference 10-step DDIM: {time.time() - t0:.1f} s", flush=True)
    idx = torch.stack(cap["idx"]).reshape(10, cfg.n_layers, B, cfg.seq_len, cfg.top_k)
    assert (idx == idx[:, :, :1, :1, :]).all()
    margin = min(fmargin, _margin(cap["probs"], cfg.top_k))

    # ---- the oracle on the same inputs: the pin
    t0 = time.time()
    with torch.no_grad():
        o_f, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sched[3] * torch.ones(B), return_aux=True)
        o_x, o_trace = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sched, trace=True)
    print(f"oracle forward + DDIM: {time.time() - t0:.1f} s", flush=True)
    assert torch.equal(torch.stack(aux.topk_idx), fidx), "oracle router ids differ from the reference (forward)"
    emb = O.sigma_embedding(sd, sched[:-1])
    for l in range(cfg.n_layers):
        _, p = O.router_probs(sd, l, emb)
        wi, _ = O.topk_route(p, cfg.top_k, cfg.router_normalize)
        assert torch.equal(wi, idx[:, l, 0, 0, :]), f"oracle router ids differ from the reference (layer {l})"
    e_f, e_x = _rel(o_f, fwd), _rel(o_x, x)
    e_t = max(_rel(o_trace[i - 1][:4], trace[i]) for i in range(1, 10))
    print(f"oracle-vs-reference at full C2 size: forward {e_f:.3e}, 10-step DDIM {e_x:.3e}, worst per-step action {e_t:.3e}, top-k margin {margin:.3e}")
    assert e_f <= 1e-5 and e_x <= 1e-5 and e_t <= 1e-5
    np.savez_compressed(os.path.join(OUT, "F17_c2_full.npz"), cfg="c2", B=B, seed=SEED, sigmas=sched.numpy(), sigma_fwd=float(sched[3]),
                        forward=fwd.numpy(), x_final=x.numpy(), action_in=torch.stack(trace).numpy(), topk_idx=idx[:, :, 0, 0, :].numpy(),
                        fwd_topk_idx=fidx[:, 0, 0, :].numpy(), margin=margin, oracle_err=np.array([e_f, e_x, e_t]))


if __name__ == "__main__":
    main()
