"""GPU: the BENCHMARKED workload itself against the oracle - BASELINE configs[1] / configs[2] at full size (12 layers, d = 1024, 8 heads,
4 experts top-2, obs 2048, goal 512, B = 128): one forward at a schedule noise level, the whole 10-step DDIM chunk, and one score-matching
training step (stochastic path: multinomial routing + both dropouts, SHARED randomness) - twelve layers of error accumulation asserted as a
number, not a property.  The oracle (fp32 CPU restatement, pinned to the reference by tests/golden) takes ~1-3 s per forward on the host.

Tolerances: tests/tolerances.py (fp32 1e-3; bf16 outputs 1e-2 / loss 1e-2 / full-depth gradients 6e-2, conditional on identical routing -
which is asserted bit-exact for every layer and every sampler step)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mode_diffusion_policy_amd as M  # noqa: E402
from oracle import mode_oracle as O  # noqa: E402
from oracle.weights import get_config, make_inputs, make_state_dict  # noqa: E402
from tolerances import BF16_GRAD_FULL_DEPTH, BF16_TRAIN_OUT, FP32_GRAD, FP32_OUT, LOSS, OUT  # noqa: E402

B, SEED = 128, 400


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _model(cfg, sd, dtype, **over):
    kw = dict(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cuda", goal_conditioned=True, action_dim=7, embed_dim=cfg.embed_dim,
              embed_pdrob=0, attn_pdrop=0.3, mlp_pdrop=0.1, goal_drop=0.0, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1,
              obs_seq_len=1, action_seq_len=10, num_experts=cfg.num_experts, top_k=cfg.top_k, compute_dtype=dtype)
    kw.update(over)
    m = M.MoDeDiT(**kw)
    m.load_state_dict(sd)
    return m.to("cuda")


@pytest.fixture(scope="module")
def c2():
    cfg = get_config("c2")
    sd = make_state_dict(cfg, SEED)
    inp = make_inputs(cfg, B, SEED + 1)
    sched = M.get_sigmas_exponential(10, 1e-3, 80.0)
    # precondition of "bit-exact routing": the fixture's smallest top-k margin over the whole schedule (fp32 CPU vs fp32 MFMA differ by ~1e-7)
    emb = O.sigma_embedding(sd, sched[:-1])
    margin = 1.0
    for l in range(cfg.n_layers):
        _, p = O.router_probs(sd, l, emb)
        top = p.sort(-1, descending=True).values
        margin = min(margin, float((top[:, cfg.top_k - 1] - top[:, cfg.top_k]).min()))
    assert margin > 1e-5, margin
    with torch.no_grad():
        ref_f, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sched[3] * torch.ones(B), return_aux=True)
        ref_x, ref_trace = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sched, trace=True)
    return dict(cfg=cfg, sd=sd, inp=inp, sched=sched, ref_f=ref_f, aux=aux, ref_x=ref_x, ref_trace=ref_trace, margin=margin)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c2_full_forward_and_ddim_vs_oracle(c2, golden, dtype):
    cfg, sd, inp, sched = c2["cfg"], c2["sd"], c2["inp"], c2["sched"]
    m = _model(cfg, sd, dtype).eval()
    c = {k: v.cuda() for k, v in inp.items()}
    st = {"state_images": c["state_images"]}
    with torch.no_grad():
        f = m(st, c["actions"], c["goals"], (sched[3] * torch.ones(B)).cuda())
    idx = m._last_topk.cpu().long()                                             # [L, R, k]
    want = torch.stack(c2["aux"].topk_idx)[:, :, 0, :]                            # [L, B, k]: every sample routes alike at a shared sigma
    assert torch.equal(idx.expand_as(want) if idx.shape[1] == 1 else idx, want)
    e_f = rel(f, c2["ref_f"])
    den = M.GCDenoiser(m, 0.5).eval()
    steps = []
    x = M.sample_ddim(den, st, c["x0"], c["goals"], sched.cuda(), disable=True, callback=lambda d: steps.append(d["action"].clone()))
    xg = M.sample_ddim(den, st, c["x0"], c["goals"], sched.cuda(), disable=True)                # the benchmarked path: one hipGraph replay
    e_x, e_g = rel(x, c2["ref_x"]), rel(xg, c2["ref_x"])
    # routing of ALL sampler steps (resolved up front by the fused sampler): [L, n, k] against the oracle's router at every schedule level
    emb = O.sigma_embedding(sd, sched[:-1])
    for l in range(cfg.n_layers):
        _, p = O.router_probs(sd, l, emb)
        wi, _ = O.topk_route(p, cfg.top_k, cfg.router_normalize)
        assert torch.equal(m._last_topk[l].cpu().long(), wi), l
    print(f"C2 B=128 {dtype}: forward rel-L2 {e_f:.2e}, 10-step DDIM rel-L2 {e_x:.2e} (graph {e_g:.2e}); tol {OUT[dtype]:g}; top-k margin {c2['margin']:.1e}")
    assert e_f < OUT[dtype] and e_x < OUT[dtype] and e_g < OUT[dtype]
    # ... and against the REFERENCE itself: fixture F17 = the reference's MoDeDiT / GCDenoiser / sample_ddim on these very inputs
    # (oracle/gen_golden_c2_full.py; modedit.py:741-809, gc_sampling.py:922-951) - outputs, per-step sampler inputs, expert ids of every (step, layer)
    g = golden("F17_c2_full")
    assert int(g["B"]) == B and int(g["seed"]) == SEED and np.array_equal(g["sigmas"], sched.numpy())
    r_f, r_x, r_g = rel(f, g["forward"]), rel(x, g["x_final"]), rel(xg, g["x_final"])
    assert np.array_equal(m._last_topk.cpu().numpy().transpose(1, 0, 2), g["topk_idx"])            # [n, L, k]: bit-exact vs the reference's router
    assert np.array_equal(idx[:, 0, :].numpy(), g["fwd_topk_idx"])
    r_s = max(rel(steps[i][:4], g["action_in"][i]) for i in range(10))
    print(f"C2 B=128 {dtype} vs REFERENCE fixture F17: forward {r_f:.2e}, DDIM {r_x:.2e} (graph {r_g:.2e}), worst per-step sampler input {r_s:.2e}")
    assert r_f < OUT[dtype] and r_x < OUT[dtype] and r_g < OUT[dtype] and r_s < OUT[dtype]
    assert rel(x, xg) < OUT[dtype]                                              # generic (callback) path vs the fused hipGraph path


def test_c2_full_training_step_vs_oracle_shared_randomness(c2):
    """configs[2]: the score-matching loss of the full model at B = 128 on the path the bench times (train mode, per-token multinomial routing,
    attention + expert dropout), loss and a sample of gradients from every part of the network against the oracle's autograd, for both compute
    modes on the SAME draw (the fp32 router makes the sampled expert ids identical in both)."""
    cfg, sd, inp = c2["cfg"], c2["sd"], c2["inp"]
    sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(3))
    c = {k: v.cuda() for k, v in inp.items()}
    names = ["tok_emb.weight", "goal_emb.weight", "action_emb.weight", "sigma_linear.weight", "pos_emb", "out.weight", "ln.g",
             "blocks.0.attn.query.weight", "blocks.0.attn.key.bias", "blocks.0.attn.c_proj.weight", "blocks.0.ln_1.g", "blocks.5.ln_2.g",
             "blocks.0.router.router.mlp.0.weight", "blocks.6.router.router.mlp.3.weight", "blocks.3.experts.expert_1.mlp.0.project.weight",
             "blocks.11.experts.expert_2.mlp.2.weight", "blocks.11.experts.expert_0.mlp.0.project.bias", "blocks.7.attn.q_norm.g"]
    runs = {}
    for dtype in ("bf16", "fp32"):
        m = _model(cfg, sd, dtype).train()
        den = M.GCDenoiser(m, 0.5).train()
        torch.manual_seed(77)                                                   # multinomial draw (device generator) + dropout step seed (host generator)
        torch.cuda.manual_seed(77)
        loss, F = den.loss({"state_images": c["state_images"]}, c["actions"], c["goals"], c["noise"], sig.cuda())
        loss.backward()
        runs[dtype] = dict(loss=float(loss), F=F.detach().cpu(), idx=m._last_topk.cpu().long(), seed=m._last_seed,
                           grads={n: p.grad.detach().cpu() for n, p in m.named_parameters() if n in names})
        del m, den
        torch.cuda.empty_cache()
    assert torch.equal(runs["bf16"]["idx"], runs["fp32"]["idx"]) and runs["bf16"]["seed"] == runs["fp32"]["seed"]
    idx = runs["bf16"]["idx"]
    sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    ref_loss, ref_F = O.denoiser_loss(sdg, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], sig,
                                      topk_idx=[idx[l].view(B, cfg.seq_len, cfg.top_k) for l in range(cfg.n_layers)],
                                      dropout=dict(seed=runs["bf16"]["seed"], attn_p=0.3, mlp_p=0.1))
    ref_loss.backward()
    for dtype in ("fp32", "bf16"):
        r = runs[dtype]
        e_l = abs(r["loss"] - float(ref_loss)) / abs(float(ref_loss))
        e_F = rel(r["F"], ref_F.detach())
        worst = max((rel(r["grads"][n], sdg[n].grad), n) for n in names)
        print(f"C2 B=128 train {dtype}: loss rel err {e_l:.2e}, F rel-L2 {e_F:.2e}, worst sampled gradient {worst[0]:.2e} ({worst[1]})")
        assert e_l < LOSS[dtype] and e_F < (FP32_OUT if dtype == "fp32" else BF16_TRAIN_OUT)
        for n in names:
            assert rel(r["grads"][n], sdg[n].grad) < (FP32_GRAD if dtype == "fp32" else BF16_GRAD_FULL_DEPTH), (dtype, n, rel(r["grads"][n], sdg[n].grad))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c2_full_deterministic_training_step_vs_reference_fixture(golden, dtype):
    """F18 = the REFERENCE's own modules + autograd on the full-size model (12 layers, d = 1024, B = 16, dropouts off, top-k routing in training, both auxiliary
    router losses; oracle/gen_golden_c2_train.py): losses, model output, expert ids (bit-exact) and EVERY parameter gradient of the HIP training chain against it.
    bf16 tolerances = the reference's own fp32-vs-autocast gap at this depth (tests/tolerances.py)."""
    g = golden("F18_c2_train")
    cfg = get_config(str(g["cfg"])); Bt, seed = int(g["B"]), int(g["seed"])
    sd = make_state_dict(cfg, seed); inp = make_inputs(cfg, Bt, seed + 1)
    m = _model(cfg, sd, dtype, attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True).train()
    den = M.GCDenoiser(m, 0.5).train()
    c = {k: v.cuda() for k, v in inp.items()}
    sig = torch.from_numpy(g["sigma"]).cuda()
    act, F = den.loss({"state_images": c["state_images"]}, c["actions"], c["goals"], c["noise"], sig)
    lb, z = m.load_balancing_loss(), m.compute_router_z_loss()
    total = act + float(g["gamma"]) * lb + float(g["delta"]) * z
    total.backward()
    idx = m._last_topk.cpu().long()                                             # [L, B, k]
    assert np.array_equal(idx.numpy().reshape(cfg.n_layers, Bt, cfg.top_k), g["topk_idx"])
    tl = LOSS[dtype]
    e_t, e_F = abs(float(total) - float(g["total"])) / abs(float(g["total"])), rel(F.detach(), g["F"])
    assert e_t < tl and abs(float(lb) - float(g["lb"])) < 1e-4 * abs(float(g["lb"])) and abs(float(z) - float(g["z"])) < 1e-4 * abs(float(g["z"]))
    assert e_F < (FP32_OUT if dtype == "fp32" else BF16_TRAIN_OUT)             # per-sample log-logistic sigma (small levels included): the training-forward tolerance
    gn = dict(zip(g["gn_keys"].tolist(), g["gn_vals"].tolist()))
    grads = {n: p.grad for n, p in m.named_parameters()}
    tol_t = FP32_GRAD if dtype == "fp32" else BF16_GRAD_FULL_DEPTH
    worst = (0.0, "")
    for key in g.files:
        if key.startswith("g:") and gn[key[2:]] > 1e-6:
            worst = max(worst, (rel(grads[key[2:]], g[key]), key[2:]))
        if key.startswith("gs:") and gn[key[3:]] > 1e-6:
            got = grads[key[3:]].reshape(-1)[:len(g[key])].cpu()
            worst = max(worst, (float((got - torch.from_numpy(g[key])).norm()) / gn[key[3:]], key[3:] + "[:512]/|g|"))
    nworst = max((abs(float(grads[n].norm()) - ref) / ref, n) for n, ref in gn.items() if ref > 1e-6)
    print(f"C2 full-depth training step vs REFERENCE fixture F18, {dtype}: total loss {e_t:.2e}, F {e_F:.2e}, worst gradient {worst[0]:.2e} ({worst[1]}), worst norm {nworst[0]:.2e} ({nworst[1]})")
    # cancellation-dominated tensors (router MLPs, key bias: the reference's own bf16 gap there is O(1), bf16_grad_gap_c2_full.json) are held to their norms in bf16
    skip = (lambda n: dtype == "bf16" and ("router" in n or "key.bias" in n))
    for key in g.files:
        name = key[2:] if key.startswith("g:") else key[3:] if key.startswith("gs:") else None
        if name is None or gn[name] <= 1e-6 or skip(name):
            continue
        if key.startswith("g:"):
            assert rel(grads[name], g[key]) < tol_t, (name, rel(grads[name], g[key]))
        else:
            got = grads[name].reshape(-1)[:len(g[key])].cpu()
            assert float((got - torch.from_numpy(g[key])).norm()) < tol_t * gn[name], name


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c2_b32_precached_rollout_vs_oracle(c2, dtype):
    """configs[4] exactly: the full model, 32 environments, routing pre-cached per noise level (``precompute_experts_for_inference`` over the
    schedule, as MoDEAgent does on its first inference call, mode_agent.py:733-745, modedit.py:607-633), one replanning call of
    ``ChunkedRolloutPolicy.denoise_actions`` (one hipGraph replay) against the oracle's 10-step DDIM from the same initial noise.  The cached
    expert ids / weights the sampler consumed are compared with the oracle's router at every level, bit for bit."""
    from mode_diffusion_policy_amd import rollout
    cfg, sd, sched = c2["cfg"], c2["sd"], c2["sched"]
    nb = 32
    inp = make_inputs(cfg, nb, SEED + 7)
    m = _model(cfg, sd, dtype).eval()
    den = M.GCDenoiser(m, 0.5).eval()
    c = {k: v.cuda() for k, v in inp.items()}
    obs = {"state_images": c["state_images"]}
    pol = rollout.ChunkedRolloutPolicy(den, num_sampling_steps=10, sigma_min=1e-3, sigma_max=80.0, noise_scheduler="exponential", sampler_type="ddim",
                                       multistep=10, generator=torch.Generator(device="cuda").manual_seed(11))
    assert pol.need_precompute_experts_for_inference
    plan = pol.denoise_actions(obs, c["goals"])
    assert not pol.need_precompute_experts_for_inference and plan.shape == (nb, 10, 7)
    # every block now holds one cache entry per schedule level (modedit.py:607-633)
    assert all(len(blk.routing_info) == 10 for blk in m.blocks)
    x0 = torch.randn((nb, 10, 7), device="cuda", generator=torch.Generator(device="cuda").manual_seed(11)) * 80.0
    with torch.no_grad():
        ref = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], x0.cpu(), inp["goals"], sched)
    emb = O.sigma_embedding(sd, sched[:-1])
    for l in range(cfg.n_layers):
        _, p = O.router_probs(sd, l, emb)
        wi, ww = O.topk_route(p, cfg.top_k, cfg.router_normalize)
        assert torch.equal(m._last_topk[l].cpu().long(), wi), l
    e = rel(plan, ref)
    print(f"C2 B=32 pre-cached rollout {dtype}: 10-step DDIM plan rel-L2 {e:.2e}; tol {OUT[dtype]:g}")
    assert e < OUT[dtype]
    # a second replanning call replays the same graph on fresh noise and stays on the oracle
    plan2 = pol.denoise_actions(obs, c["goals"])
    assert not torch.equal(plan2, plan) and torch.isfinite(plan2).all()


# ---------------------------------------------------------------------------------------------- the >= 6x leg's numerics: bf16 on the wire, full depth
DP_NAMES = ["tok_emb.weight", "goal_emb.weight", "action_emb.weight", "sigma_linear.weight", "pos_emb", "out.weight", "ln.g",
            "blocks.0.attn.query.weight", "blocks.0.attn.c_proj.weight", "blocks.0.ln_1.g", "blocks.5.ln_2.g", "blocks.11.attn.value.weight",
            "blocks.3.experts.expert_1.mlp.0.project.weight", "blocks.11.experts.expert_2.mlp.2.weight", "blocks.7.attn.q_norm.g",
            "blocks.0.experts.expert_0.mlp.2.weight", "blocks.6.attn.value.bias"]


def _bf16wire_worker(rank, world, port, outdir, Bt):
    """One rank of `bench.py --gpus N`'s bf16wire leg on cuda:0 (both ranks share the GPU; gloo carries the device tensors): the FULL-SIZE model on its half
    of the batch, deterministic mode, gradients exchanged by ArenaGradReducer with bf16 on the wire (the sum is formed in bf16)."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mode_diffusion_policy_amd.ddp import ArenaGradReducer
        torch.cuda.set_device(0)
        cfg = get_config("c2")
        sd = make_state_dict(cfg, SEED)
        inp = make_inputs(cfg, Bt, SEED + 5)
        sig = O.rand_log_logistic((Bt,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(4))
        m = _model(cfg, sd, "bf16", attn_pdrop=0.0, mlp_pdrop=0.0, goal_drop=0.0, use_argmax=True).train()
        den = M.GCDenoiser(m, 0.5).train()
        red = ArenaGradReducer.for_model(m, mode="allreduce", comm_dtype=torch.bfloat16)
        sl = slice(rank * Bt // world, (rank + 1) * Bt // world)
        c = {k: v[sl].cuda() for k, v in inp.items() if k != "x0"}
        loss, _ = den.loss({"state_images": c["state_images"]}, c["actions"], c["goals"], c["noise"], sig[sl].cuda())
        loss.backward()
        scale = red.reduce()
        torch.cuda.synchronize()
        g = {n: (p.grad.detach() * scale).cpu() for n, p in m.named_parameters() if n in DP_NAMES}
        g["__idx__"] = m._last_topk.cpu().reshape(cfg.n_layers, -1, cfg.top_k)
        torch.save(g, os.path.join(outdir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_c2_full_bf16_wire_exchange_vs_oracle_whole_batch_gradients(tmp_path):
    """configs[3]'s claim leg (bench.py `bf16wire_*`, DESIGN.md section 6: the only exchange whose wire time fits under the step at N = 8) sums the per-rank
    gradients in bf16 on the links - a numerics change the reference's fp32 DDP all-reduce (mode/training_calvin.py:92-103) does not make.  Pinned here at FULL
    depth: two ranks, each the full-size model on half of a B = 16 batch, gradient exchange with bf16 on the wire; the exchanged mean gradient of a sample
    of tensors from every part of the network stays inside BF16_GRAD_FULL_DEPTH of the ORACLE's fp32 autograd on the concatenated batch (the tolerance the
    single-process bf16 backward is held to, tests/tolerances.py) and is identical on both ranks."""
    import socket
    import torch.multiprocessing as mp
    Bt = 16
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_bf16wire_worker, args=(r, 2, port, str(tmp_path), Bt)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    g = [torch.load(tmp_path / f"g{r}.pt") for r in range(2)]
    cfg = get_config("c2")
    sd = make_state_dict(cfg, SEED)
    inp = make_inputs(cfg, Bt, SEED + 5)
    sig = O.rand_log_logistic((Bt,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(4))
    # the ranks' routing decisions (fp32 router, per-sample sigma): rank r holds rows of its half
    idx = torch.cat([g[0].pop("__idx__"), g[1].pop("__idx__")], dim=1).long()          # [L, B, k]
    sdg = {k: (v.clone().requires_grad_(True) if k in DP_NAMES else v) for k, v in sd.items()}
    ref_loss, _ = O.denoiser_loss(sdg, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], sig,
                                  topk_idx=[idx[l].view(Bt, 1, cfg.top_k).expand(Bt, cfg.seq_len, cfg.top_k) for l in range(cfg.n_layers)])
    ref_loss.backward()
    worst = (0.0, "")
    for n in DP_NAMES:
        assert torch.equal(g[0][n], g[1][n]), n                                          # every rank holds the same exchanged gradient, bit for bit
        e = rel(g[0][n], sdg[n].grad)
        worst = max(worst, (e, n))
        assert e < BF16_GRAD_FULL_DEPTH, (n, e)
    print(f"C2 full depth, world 2, bf16 on the wire: worst exchanged gradient vs the oracle's whole-batch autograd {worst[0]:.2e} ({worst[1]}); tol {BF16_GRAD_FULL_DEPTH:g}")
