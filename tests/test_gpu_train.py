"""GPU parity of the score-matching training step (GCDenoiser.loss -> MoDeDiT train forward -> HIP backward) against the golden
loss / output / gradients produced by the REAL reference's autograd (fixtures F5, deterministic config: all dropouts 0,
use_argmax=True — dropout and the multinomial draw cannot be bit-matched).
Tolerances: fp32 compute mode 1e-3 (observed ~1e-5); bf16 mode: loss/F 1e-2, gradient norms 3e-2, per-tensor rel-L2 6e-2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mode_diffusion_policy_amd as M  # noqa: E402
from oracle import mode_oracle as O  # noqa: E402
from oracle.weights import get_config, make_inputs, make_state_dict  # noqa: E402


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def build_train(cfgname, seed, dtype, **over):
    cfg = get_config(cfgname)
    kw = dict(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cuda", goal_conditioned=True, action_dim=7, embed_dim=cfg.embed_dim,
              embed_pdrob=0, attn_pdrop=0.0, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1, obs_seq_len=1, action_seq_len=10,
              mlp_pdrop=0.0, goal_drop=0.0, num_experts=cfg.num_experts, top_k=cfg.top_k, use_argmax=True, compute_dtype=dtype)
    kw.update(over)
    m = M.MoDeDiT(**kw)
    sd = make_state_dict(cfg, seed)
    m.load_state_dict(sd)
    return cfg, sd, m.to("cuda").train()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cfgname", ["c1e4", "c1"])
def test_loss_and_gradients_vs_reference(golden, cfgname, dtype):
    g = golden(f"F5_{cfgname}_loss_grad")
    cfg, sd, m = build_train(cfgname, int(g["seed"]), dtype)
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 8, int(g["seed"]) + 1).items()}
    den = M.GCDenoiser(m, 0.5).train()
    sig = torch.from_numpy(g["sigma"]).cuda()
    loss, F = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
    tolF = 1e-3 if dtype == "fp32" else 1e-2
    assert rel(F.detach(), g["F"]) < tolF
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < tolF
    # aux-loss side channels (values)
    lb = m.load_balancing_loss(); z = m.compute_router_z_loss()
    assert abs(float(lb) - float(g["lb"])) < 1e-4 * max(1.0, abs(float(g["lb"])))
    assert abs(float(z) - float(g["z"])) < 1e-4 * max(1.0, abs(float(g["z"])))
    loss.backward()
    none = set(g["none"].tolist())
    gn = dict(zip(g["gn_keys"].tolist(), g["gn_vals"].tolist()))
    grads = {n: p.grad for n, p in m.named_parameters()}
    assert grads["gripper_embed.weight"] is None                                   # dead parameter, as in the reference
    for n in none:
        if n != "gripper_embed.weight":
            assert grads[n] is None or float(grads[n].abs().max()) == 0.0, n       # un-routed experts: exact zeros (zero-filled buckets)
    tol_n, tol_t = (1e-3, 1e-3) if dtype == "fp32" else (3e-2, 6e-2)
    bad = []
    for n, ref_norm in gn.items():
        if ref_norm <= 1e-6:
            continue
        got = float(grads[n].norm())
        if abs(got - ref_norm) / ref_norm > tol_n:
            bad.append((n, got, ref_norm))
    assert not bad, bad[:8]
    for key in g.files:
        if key.startswith("g:") and gn[key[2:]] > 1e-6:
            assert rel(grads[key[2:]], g[key]) < tol_t, key
        if key.startswith("gs:") and gn[key[3:]] > 1e-6:
            got = grads[key[3:]].reshape(-1)[:2048].cpu()
            assert float((got - torch.from_numpy(g[key])).norm()) < tol_t * gn[key[3:]], key


def test_training_step_decreases_loss_and_refreshes_shadows():
    """A few AdamW steps on one batch with the default (stochastic) training config: multinomial routing + dropouts on."""
    torch.manual_seed(0)
    cfg, sd, m = build_train("c1e4", 210, "bf16", attn_pdrop=0.3, mlp_pdrop=0.1, goal_drop=0.1, use_argmax=False)
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 16, 5).items()}
    den = M.GCDenoiser(m, 0.5).train()
    from mode_diffusion_policy_amd.ddp import optimizer_param_groups
    opt = torch.optim.AdamW(optimizer_param_groups(m, 0.05), lr=3e-4, betas=(0.9, 0.95))
    sig = O.rand_log_logistic((16,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(1)).cuda()
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss, _ = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and np.mean(losses[-3:]) < 0.8 * np.mean(losses[:3]), losses
    assert m.blocks[0].probs["top_k_hot"].shape == (16, cfg.seq_len, cfg.num_experts)


def test_parameters_live_in_one_arena_and_state_dict_is_untouched():
    cfg, sd, m = build_train("c1e4", 7, "bf16")
    eng = m.engine
    ar = eng.arena
    lo, hi = ar.flat.data_ptr(), ar.flat.data_ptr() + ar.flat.numel() * 4
    for n, p in m.named_parameters():
        assert lo <= p.data_ptr() < hi and p.data_ptr() % 16 == 0, n
        assert torch.equal(p.detach().cpu(), sd[n]), n                           # adoption does not change a single value
    assert list(m.state_dict().keys()) == list(sd.keys())
    # q/k/v of a block are adjacent rows of one packed operand; experts are stacked; routers of all layers are adjacent
    a = m.blocks[1].attn
    assert a.key.weight.data_ptr() == a.query.weight.data_ptr() + a.query.weight.numel() * 4
    assert m.blocks[1].router.router.mlp[0].weight.data_ptr() == m.blocks[0].router.router.mlp[0].weight.data_ptr() + 2 * cfg.embed_dim * cfg.embed_dim * 4
    # staleness: an in-place update through a Parameter view must reach the bf16 compute shadow
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 4, 1).items()}
    m.eval()
    s = torch.full((4,), 0.7, device="cuda")
    F0 = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s).clone()
    with torch.no_grad():
        m.out.weight.mul_(2.0); m.out.bias.mul_(2.0)
    F1 = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s)
    assert rel(F1, 2.0 * F0) < 1e-6
    with torch.no_grad():
        m.blocks[0].attn.c_proj.weight.zero_()                                    # a bf16-shadowed GEMM operand
    F2 = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s)
    assert rel(F2, F1) > 1e-4
    # re-allocation (module.to(dtype) and back) breaks aliasing -> the arena is rebuilt transparently
    m.double().float()
    F3 = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s)
    assert m.engine.arena is not ar and rel(F3, F2) < 1e-6


@pytest.mark.parametrize("dtype,overlap", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_fused_adamw_matches_torch_adamw(dtype, overlap):
    """FusedAdamW over the arena == torch.optim.AdamW with MoDEAgent.get_optim_groups' two groups (mode_agent.py:365-392)."""
    from mode_diffusion_policy_amd.ddp import optimizer_param_groups
    from mode_diffusion_policy_amd.optim import FusedAdamW
    cfg, sd, m = build_train("c1e4", 11, dtype)
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 8, 3).items()}
    den = M.GCDenoiser(m, 0.5).train()
    sig = torch.full((8,), 0.9, device="cuda")
    opt = FusedAdamW(m, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    # shadow copy driven by torch's optimizer with the SAME gradients
    ref = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}

    class _Holder(torch.nn.Module):
        def named_parameters(self_inner, *a, **k):
            return iter(ref.items())
    topt = torch.optim.AdamW(optimizer_param_groups(_Holder(), 0.05), lr=1e-3, betas=(0.9, 0.95))
    for step in range(3):
        loss, _ = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
        loss.backward()
        for n, p in m.named_parameters():
            ref[n].grad = None if p.grad is None else p.grad.detach().clone()
        opt.step(overlap=overlap)                           # overlap: per-block updates on a side stream behind the backward's block events
        topt.step()
        for n, p in m.named_parameters():
            assert rel(p.detach(), ref[n].detach()) < 2e-6, (step, n)
    assert torch.equal(m.gripper_embed.weight.detach().cpu(), sd["gripper_embed.weight"])          # dead parameter: never updated
    if dtype == "bf16":                                                                              # shadow written by the optimizer itself
        ar = m.engine.arena
        assert ar.lp_synced and torch.equal(ar.lp[: ar.bounds["no_decay"]], ar.flat[: ar.bounds["no_decay"]].to(torch.bfloat16))
    # the loss keeps moving: the next forward reads the updated weights (transposed shadows refreshed as well)
    l2, _ = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
    assert float(l2) != float(loss)


def test_arena_reducer_overlap_slices_and_events():
    """Data-parallel exchange on the gradient arena (ddp.ArenaGradReducer): per-block slices in backward order, each gated by the event the
    backward chain records for that block; the slices tile the optimised part of the arena exactly once.  Runs the real stream / event /
    RCCL code path on one GPU (process group of size 1; the reducer is told it has 2 ranks so it does not short-circuit)."""
    import os
    import socket
    import torch.distributed as dist
    from mode_diffusion_policy_amd.ddp import ArenaGradReducer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg, sd, m = build_train("c1e4", 31, "bf16")
        inp = {k: v.cuda() for k, v in make_inputs(cfg, 8, 3).items()}
        den = M.GCDenoiser(m, 0.5).train()
        sig = torch.full((8,), 0.9, device="cuda")
        red = ArenaGradReducer.for_model(m)
        ar = m.engine.arena
        cover = sorted((lo, hi) for lo, hi, _ in red.slices)
        assert cover[0][0] == 0 and cover[-1][1] == ar.bounds["no_decay"]
        assert all(a[1] == b[0] for a, b in zip(cover[:-1], cover[1:]))                   # contiguous, no overlap, nothing missing
        evs = [ev for _, _, ev in red.slices]
        assert sum(e is not None for e in evs) == cfg.n_layers and evs[0] is not None      # block L-1 first: it finishes first
        assert red.slices[0][0] == ar.offset(f"l{cfg.n_layers - 1}.wqkv")
        loss, _ = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
        loss.backward()
        red.world = 2                                                                       # exercise the collective path
        scale = red.reduce()
        torch.cuda.synchronize()
        want = ar.grad.clone()
        loss, _ = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
        loss.backward()
        torch.cuda.synchronize()
        assert scale == 0.5 and torch.equal(ar.grad, want)                                  # sum over the single rank = identity, deterministic backward
    finally:
        dist.destroy_process_group()


def test_arena_ema_fused_standalone_and_swap():
    """ArenaEMA == the EMA callback's non-apex update rule (mode/callbacks/ema.py:83-126: e -= (1 - decay_t)(e - w), warm-up decay schedule)
    restated with torch on clones of the weights; fused into FusedAdamW.step and as a stand-alone pass; swap() round-trips bit-exactly."""
    from mode_diffusion_policy_amd.optim import ArenaEMA, FusedAdamW
    cfg, sd, m = build_train("c1e4", 41, "bf16")
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 8, 3).items()}
    den = M.GCDenoiser(m, 0.5).train()
    sig = torch.full((8,), 0.9, device="cuda")
    opt = FusedAdamW(m, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    ema = ArenaEMA(m, decay=0.999, apply_ema_every_n_steps=1, start_step=0)
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}                       # on_train_start: a copy of the weights
    for step in range(1, 5):
        loss, _ = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
        loss.backward()
        if step % 2:
            opt.step(ema=ema)                                                             # fused into the optimizer pass
        else:
            opt.step()
            ema.update(step)                                                              # stand-alone pass after a (foreign) optimizer step
        d = ema.get_decay(step)
        for n, p in m.named_parameters():
            ref[n] = ref[n] - (1.0 - d) * (ref[n] - p.detach())
    assert abs(ema.get_decay(1) - 0.0) < 1e-12 and abs(ema.get_decay(3) - (1 - 3.0 ** (-2 / 3))) < 1e-12
    from mode_diffusion_policy_amd.arena import param_views
    views = param_views(m, {k: ema.flat[off: off + int(torch.Size(shp).numel())].view(shp) for k, shp, off in m.engine.arena.layout})
    for n in ref:
        assert rel(views[n], ref[n].reshape(views[n].shape)) < 1e-6, n
    # swap(): evaluate with the averaged weights, then restore the live ones bit-exactly
    m.eval()
    live = m.engine.arena.flat.clone(); avg = ema.flat.clone()
    F0 = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sig).clone()
    ema.swap()
    assert torch.equal(m.engine.arena.flat, avg) and torch.equal(ema.flat, live)          # arena now holds the average, the EMA object the live weights
    F1 = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sig).clone()
    ema.swap()
    F2 = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sig)
    assert torch.equal(m.engine.arena.flat, live) and torch.equal(F2, F0) and not torch.equal(F1, F0)


def test_fused_adamw_respects_frozen_router():
    """Fine-tuning mode (MoDEAgent.prepare_model_for_finetuning -> inner_model.freeze_router(), mode_agent.py:762-766): router tensors have
    requires_grad False, get no gradient and must not move (torch's AdamW skips them; decoupled weight decay included)."""
    from mode_diffusion_policy_amd.optim import FusedAdamW
    cfg, sd, m = build_train("c1e4", 51, "bf16")
    m.freeze_router()
    inp = {k: v.cuda() for k, v in make_inputs(cfg, 8, 3).items()}
    den = M.GCDenoiser(m, 0.5).train()
    sig = torch.full((8,), 0.9, device="cuda")
    opt = FusedAdamW(m, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    for _ in range(2):
        loss, _ = den.loss({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], inp["noise"], sig)
        loss.backward()
        opt.step()
    moved = 0
    for n, p in m.named_parameters():
        if "router" in n:
            assert p.grad is None and torch.equal(p.detach(), before[n]), n
        elif n != "gripper_embed.weight":
            moved += int(not torch.equal(p.detach(), before[n]))
    assert moved > 50


@pytest.mark.parametrize("B", [1, 3, 37])
def test_training_step_ragged_batches_vs_oracle_autograd(B):
    """Loss and gradients at batch sizes that leave partial tiles / nearly empty expert segments, against the oracle's autograd (fp32 mode)."""
    cfg, sd, m = build_train("c1e4", 210, "fp32")
    inp = make_inputs(cfg, B, 70 + B)
    sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(B))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss, _ = O.denoiser_loss(sdg, cfg, 0.5, inp["state_images"], inp["actions"], inp["goals"], inp["noise"], sig)
    ref_loss.backward()
    c = {k: v.cuda() for k, v in inp.items()}
    den = M.GCDenoiser(m, 0.5).train()
    loss, _ = den.loss({"state_images": c["state_images"]}, c["actions"], c["goals"], c["noise"], sig.cuda())
    loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    checked = 0
    for n, p in m.named_parameters():
        r = sdg[n].grad
        if r is None or float(r.norm()) < 1e-7:
            assert p.grad is None or float(p.grad.abs().max()) < 1e-6, n           # un-routed experts / dead parameter
            continue
        assert rel(p.grad, r) < 2e-3, (n, rel(p.grad, r))
        checked += 1
    assert checked > 40
