"""GPU parity of the whole denoising path (MoDeDiT / GCDenoiser / sample_ddim mirrors -> C-ABI -> HIP kernels) against the
golden vectors generated from the reference and against the oracle.

Tolerances (SURVEY.md §8 a-bis, grounded in the reference's own fp32-vs-bf16-autocast gap):
  fp32 compute mode : rel-L2 <= 1e-3 vs the fp32 reference (observed ~1e-6)
  bf16 compute mode : rel-L2 <= tolerances.BF16_OUT CONDITIONAL on identical router indices (router runs in fp32 in both modes)
  router top-k indices / dispatch permutation: bit-exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mode_diffusion_policy_amd as M  # noqa: E402
from oracle import mode_oracle as O  # noqa: E402
from oracle.weights import get_config, make_inputs, make_state_dict  # noqa: E402

from tolerances import BF16_OUT, BF16_TOKROUTE_AGREE, BF16_TOKROUTE_OUT, FP32_OUT, OUT as TOL  # noqa: E402  (one number per quantity: tests/tolerances.py)


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def build(cfgname, seed, dtype, **over):
    cfg = get_config(cfgname)
    kw = dict(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cuda", goal_conditioned=True, action_dim=cfg.action_dim,
              embed_dim=cfg.embed_dim, embed_pdrob=0, attn_pdrop=0.3, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=1,
              obs_seq_len=1, action_seq_len=cfg.action_seq_len, state_dim=None, num_experts=cfg.num_experts, top_k=cfg.top_k,
              compute_dtype=dtype)
    kw.update(over)
    m = M.MoDeDiT(**kw)
    sd = make_state_dict(cfg, seed)
    m.load_state_dict(sd)
    return cfg, sd, m.to("cuda").eval()


def cuda_inputs(inp):
    return {k: v.cuda() for k, v in inp.items()}


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["F2_blocks_tiny", "F3_c1_forward_uniform", "F3_c1_forward_persample", "F3_c1e4_forward_persample",
                                  "F7_c2block"])
def test_forward_vs_golden(golden, name, dtype):
    g = golden(name)
    cfgname = str(g["cfg"]); B = int(g["B"]); seed = int(g["seed"])
    cfg, sd, m = build(cfgname, seed, dtype)
    inp = cuda_inputs(make_inputs(cfg, B, seed + 1))
    sigma = torch.from_numpy(g["sigma"]).cuda()
    with torch.no_grad():
        out = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], sigma)
    idx = m._last_topk.cpu().long()                                        # [L, R, k]
    ref_idx = torch.from_numpy(g["topk_idx"])[:, :, 0, :]                   # reference: same experts for all T tokens of a sample
    assert torch.equal(idx, ref_idx), "router top-k indices must be bit-identical to the fp32 reference"
    assert rel(out, g["out"]) < TOL[dtype]
    if dtype == "fp32":
        assert rel(out, g["out"]) < 2e-5                                    # what fp32 MFMA actually achieves


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["nonoise", "goalroute", "nonorm", "b1"])
def test_flag_variants_vs_golden(golden, tag, dtype):
    """T=13 (no noise token), goal-conditioned routing, un-normalised router weights, B=1 — against the real reference's outputs."""
    g = golden(f"F9_{tag}")
    over = {k: bool(g[k]) for k in ("use_noise_token_as_input", "use_goal_in_routing", "router_normalize") if k in g.files}
    B = int(g["B"])
    cfg, sd, m = build("c1e4", int(g["seed"]), dtype, **over)
    inp = cuda_inputs(make_inputs(cfg, B, int(g["seed"]) + 1))
    with torch.no_grad():
        out = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], torch.from_numpy(g["sigma"]).cuda())
    assert torch.equal(m._last_topk.cpu().long(), torch.from_numpy(g["topk_idx"])[:, :, 0, :])
    assert rel(out, g["out"]) < TOL[dtype]
    # the fused EDM forward and the fused sampler take the same flags
    den = M.GCDenoiser(m, 0.5).eval()
    sig = M.get_sigmas_exponential(10, 1e-3, 80.0).cuda()
    x = M.sample_ddim(den, {"state_images": inp["state_images"]}, inp["x0"], inp["goals"], sig, disable=True)
    import dataclasses
    from oracle import mode_oracle as OO
    ocfg = dataclasses.replace(cfg, **over)
    ci = make_inputs(cfg, B, int(g["seed"]) + 1)
    ref = OO.sample_ddim(sd, ocfg, 0.5, ci["state_images"], ci["x0"], ci["goals"], sig.cpu())
    assert rel(x, ref) < TOL[dtype]


def test_large_batch_and_goal_2d():
    """B=1024 (N=14336 tokens; the global batch of config 4 on one GPU) + goals given as (B, G): finite, and the first 16 samples
    agree with running them alone (samples are independent)."""
    cfg, sd, m = build("c1e4", 210, "bf16")
    inp = cuda_inputs(make_inputs(cfg, 1024, 3))
    s = torch.full((1024,), 0.7, device="cuda")
    with torch.no_grad():
        a = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"].reshape(1024, -1), s)
        b = m({"state_images": inp["state_images"][:16]}, inp["actions"][:16], inp["goals"][:16], s[:16])
    assert torch.isfinite(a).all() and rel(a[:16], b) < 1e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_uniform_scalar_sigma_equals_vector_sigma(dtype):
    """sigma given as a 0-d tensor (shared routing row, the sampler's fast path) == the same sigma repeated per sample."""
    cfg, sd, m = build("c1e4", 210, dtype)
    inp = cuda_inputs(make_inputs(cfg, 8, 5))
    s = torch.tensor(1.857, device="cuda")
    with torch.no_grad():
        a = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s)
        b = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s * torch.ones(8, device="cuda"))
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cfgname", ["c1", "c1e4"])
@pytest.mark.parametrize("graph", ["1", "0"])
def test_ddim_vs_golden(golden, cfgname, dtype, graph, monkeypatch):
    monkeypatch.setenv("MODE_HIP_GRAPH", graph)
    g = golden(f"F4_{cfgname}_ddim")
    cfg, sd, m = build(cfgname, int(g["seed"]), dtype)
    inp = cuda_inputs(make_inputs(cfg, 8, int(g["seed"]) + 1))
    den = M.GCDenoiser(m, 0.5).eval()
    sig = torch.from_numpy(g["sigmas"]).cuda()
    assert np.array_equal(M.get_sigmas_exponential(10, 1e-3, 80.0).numpy(), g["sigmas"])
    x = M.sample_ddim(den, {"state_images": inp["state_images"]}, inp["x0"], inp["goals"], sig, disable=True)
    ref_idx = torch.from_numpy(g["topk_idx"])[:, :, 0, 0, :].permute(1, 0, 2)     # [steps, L, B, T, k] -> [L, steps, k]
    assert torch.equal(m._last_topk.cpu().long(), ref_idx)
    assert rel(x, g["x_final"]) < TOL[dtype]
    # replay (graph path: second call reuses the captured hipGraph) must be bit-identical
    x2 = M.sample_ddim(den, {"state_images": inp["state_images"]}, inp["x0"], inp["goals"], sig, disable=True)
    assert torch.equal(x, x2)
    # generic (un-fused) path through the callback hook follows the reference step order and must agree
    trace = []
    x3 = M.sample_ddim(den, {"state_images": inp["state_images"]}, inp["x0"], inp["goals"], sig, disable=True,
                       callback=lambda d: trace.append(d["denoised"]))
    assert len(trace) == 10
    assert rel(torch.stack(trace), g["denoised"]) < TOL[dtype]
    assert rel(x3, x) < (1e-5 if dtype == "fp32" else BF16_OUT)


def test_tagged_schedules_edited_in_place_are_not_confused():
    """Two schedules from the same generator call carry the same value tag; edited in place DIFFERENTLY (both at version 1) they must not share a
    schedule state: the tag only vouches for the values the generator wrote (round-3 advisor finding)."""
    cfg, sd, m = build("c1e4", 210, "fp32")
    inp = cuda_inputs(make_inputs(cfg, 4, 9))
    den = M.GCDenoiser(m, 0.5).eval()
    st = {"state_images": inp["state_images"]}
    cpu = {k: v.cpu() for k, v in inp.items()}
    base = M.get_sigmas_exponential(6, 1e-3, 80.0, "cuda")
    x_base = M.sample_ddim(den, st, inp["x0"], inp["goals"], base, disable=True)
    s1 = M.get_sigmas_exponential(6, 1e-3, 80.0, "cuda"); s1[2] = 7.5
    s2 = M.get_sigmas_exponential(6, 1e-3, 80.0, "cuda"); s2[4] = 0.02
    assert s1._version == s2._version
    x1 = M.sample_ddim(den, st, inp["x0"], inp["goals"], s1, disable=True)
    x2 = M.sample_ddim(den, st, inp["x0"], inp["goals"], s2, disable=True)
    for x, s in ((x_base, base), (x1, s1), (x2, s2)):
        ref = O.sample_ddim(sd, cfg, 0.5, cpu["state_images"], cpu["x0"], cpu["goals"], s.cpu())
        assert rel(x, ref) < FP32_OUT
    assert not torch.equal(x1, x2) and not torch.equal(x1, x_base)
    # an untouched twin of the generator call still takes the tag path and replays the unchanged schedule
    assert torch.equal(M.sample_ddim(den, st, inp["x0"], inp["goals"], M.get_sigmas_exponential(6, 1e-3, 80.0, "cuda"), disable=True), x_base)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gcdenoiser_forward_vs_oracle(dtype):
    cfg, sd, m = build("c1e4", 210, dtype)
    inp = make_inputs(cfg, 8, 33)
    sig = O.rand_log_logistic((8,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(3))
    ref = O.denoiser_forward(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sig)
    c = cuda_inputs(inp)
    den = M.GCDenoiser(m, 0.5).eval()
    out = den({"state_images": c["state_images"]}, c["x0"], c["goals"], sig.cuda())
    assert rel(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fused_cache_fixture(golden, dtype):
    """precompute_experts_for_inference caches exactly the reference's (e0,e1,p0,p1) per layer per sigma (modedit.py:607-633)."""
    g = golden("F6_fused_cache")
    cfg, sd, m = build("c1e4", int(g["seed"]), dtype)
    for si, s in enumerate(g["sigmas"]):
        m.reset_all_caches()
        m.precompute_experts_for_inference(torch.tensor([s]))
        for l, blk in enumerate(m.blocks):
            (info,) = blk.routing_info.values()
            assert info["indices"].tolist() == g["idx"][si, l].tolist()
            assert np.allclose(info["probs"], g["p"][si, l], rtol=1e-5)
    for B in (1, 8):
        inp = cuda_inputs(make_inputs(cfg, B, 777))
        with torch.no_grad():
            out = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], torch.tensor(g["sigmas"][2]).cuda() * torch.ones(B).cuda())
        assert rel(out, g[f"loop_B{B}"]) < TOL[dtype]


def test_weight_shadow_refresh():
    """In-place parameter updates (optimizer step / EMA swap / load_state_dict) must invalidate the bf16 shadows."""
    cfg, sd, m = build("c1e4", 210, "bf16")
    inp = cuda_inputs(make_inputs(cfg, 4, 1))
    s = torch.full((4,), 0.5, device="cuda")
    with torch.no_grad():
        a = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s)
        m.out.bias.add_(1.0)
        b = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s)
        assert rel(b - 1.0, a) < 1e-5
        m.load_state_dict(sd)
        c = m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], s)
    assert torch.equal(a, c)


def test_no_cpu_fallback():
    cfg = get_config("tiny")
    m = M.MoDeDiT(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cpu", goal_conditioned=True, action_dim=7, embed_dim=cfg.embed_dim,
                  embed_pdrob=0, attn_pdrop=0.3, n_layers=2, n_heads=4, goal_seq_len=1, obs_seq_len=1, action_seq_len=10).eval()
    inp = make_inputs(cfg, 2, 1)
    with pytest.raises(Exception):
        m({"state_images": inp["state_images"]}, inp["actions"], inp["goals"], torch.ones(2))


# ------------------------------------------------------------------------------------------ full-size (config 2) properties
@pytest.fixture(scope="module")
def c2_model():
    torch.manual_seed(0)
    cfg = get_config("c2")
    m = M.MoDeDiT(obs_dim=cfg.obs_dim, goal_dim=cfg.goal_dim, device="cuda", goal_conditioned=True, action_dim=7, embed_dim=1024,
                  embed_pdrob=0, attn_pdrop=0.3, n_layers=12, n_heads=8, goal_seq_len=1, obs_seq_len=1, action_seq_len=10,
                  num_experts=4, top_k=2, compute_dtype="bf16")
    with torch.no_grad():                      # non-trivial gains / pos_emb, stronger router so margins are healthy
        for n_, p in m.named_parameters():
            if n_.endswith(".g"):
                p.add_(0.1 * torch.randn_like(p))
            if "router.router.mlp.3.weight" in n_:
                p.mul_(20.0)
        m.pos_emb.normal_(0, 0.1)
    return cfg, m.to("cuda").eval()


def test_c2_full_size_properties(c2_model):
    """BASELINE config 2 (B=128, 12 layers, D=1024, 4 experts top-2): size-independent properties at full size."""
    cfg, m = c2_model
    B = 128
    inp = cuda_inputs(make_inputs(cfg, B, 9))
    den = M.GCDenoiser(m, 0.5).eval()
    sig = M.get_sigmas_exponential(10, 1e-3, 80.0).cuda()
    st = {"state_images": inp["state_images"]}
    x = M.sample_ddim(den, st, inp["x0"], inp["goals"], sig, disable=True)
    assert torch.isfinite(x).all()
    # (1) determinism / graph replay idempotence
    assert torch.equal(x, M.sample_ddim(den, st, inp["x0"], inp["goals"], sig, disable=True))
    # (2) samples are independent: a permutation of the batch permutes the result bit-exactly
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    xp = M.sample_ddim(den, {"state_images": inp["state_images"][perm]}, inp["x0"][perm], inp["goals"][perm], sig, disable=True)
    assert torch.equal(xp, x[perm])
    # (3) batch-slice consistency: the first 8 samples alone give the same result as inside the batch of 128
    x8 = M.sample_ddim(den, {"state_images": inp["state_images"][:8]}, inp["x0"][:8], inp["goals"][:8], sig, disable=True)
    assert torch.equal(x8, x[:8])
    # (3a) BASELINE configs[4]'s batch: 32 environments (other GEMM tile configurations than B=128 / B=8: 64x64 K-sliced down-projection, ring-2
    #      128x128 up-projection) - finite, deterministic across replays, bit-identical to the same samples inside the batch of 128
    st32 = {"state_images": inp["state_images"][:32]}
    x32 = M.sample_ddim(den, st32, inp["x0"][:32], inp["goals"][:32], sig, disable=True)
    assert torch.isfinite(x32).all() and torch.equal(x32, x[:32])
    assert torch.equal(x32, M.sample_ddim(den, st32, inp["x0"][:32], inp["goals"][:32], sig, disable=True))
    # (3b) one and two environments take the weight-streaming GEMMs (another fp32 summation order): equal to bf16 rounding, not bit for bit
    for nb in (1, 2):
        xs = M.sample_ddim(den, {"state_images": inp["state_images"][:nb]}, inp["x0"][:nb], inp["goals"][:nb], sig, disable=True)
        assert rel(xs, x[:nb]) < BF16_OUT, nb
    # (4) last DDIM step has r = 0: x_final == denoised of the last step (gc_sampling.py:948-950 with sigma_next = 0)
    trace = []
    M.sample_ddim(den, st, inp["x0"], inp["goals"], sig, disable=True, callback=lambda d: trace.append(d["denoised"]))
    assert rel(trace[-1], x) < BF16_OUT
    # (5) fp32 compute mode vs bf16 at full size, conditional on identical routing
    idx_bf16 = m._last_topk.clone()
    m.compute_dtype = "fp32"
    with torch.no_grad():
        f32 = m(st, inp["actions"], inp["goals"], sig[4] * torch.ones(B, device="cuda"))
        i32 = m._last_topk.clone()
        m.compute_dtype = "bf16"
        b16 = m(st, inp["actions"], inp["goals"], sig[4] * torch.ones(B, device="cuda"))
    assert torch.equal(i32, m._last_topk)
    assert rel(b16, f32) < BF16_OUT


def test_c2_block_oracle_large_batch(golden):
    """One C2-sized block at B=128 (N=1792 tokens: every GEMM tile shape of the benchmark) against the oracle."""
    cfg, sd, m = build("c2block", 300, "bf16")
    inp = make_inputs(cfg, 128, 41)
    sig = O.rand_log_logistic((128,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(4))
    ref, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sig, return_aux=True)
    c = cuda_inputs(inp)
    with torch.no_grad():
        out = m({"state_images": c["state_images"]}, c["actions"], c["goals"], sig.cuda())
    assert torch.equal(m._last_topk.cpu().long()[0], aux.topk_idx[0][:, 0, :])
    assert rel(out, ref) < BF16_OUT
    m.compute_dtype = "fp32"
    with torch.no_grad():
        out32 = m({"state_images": c["state_images"]}, c["actions"], c["goals"], sig.cuda())
    assert rel(out32, ref) < 2e-5


def test_empty_batch_is_a_no_op():
    """B = 0 (an evaluation loop whose last shard is empty): empty outputs of the right shape, no kernel launched, no error."""
    cfg, sd, m = build("c1e4", 3, "bf16")
    den = M.GCDenoiser(m, 0.5).eval()
    st = {"state_images": torch.zeros(0, 2, cfg.obs_dim, device="cuda")}
    a = torch.zeros(0, 10, 7, device="cuda"); g = torch.zeros(0, 1, cfg.goal_dim, device="cuda")
    sig = M.get_sigmas_exponential(10, 1e-3, 80.0).cuda()
    assert m(st, a, g, torch.zeros(0, device="cuda")).shape == (0, 10, 7)
    assert den(st, a, g, torch.zeros(0, device="cuda")).shape == (0, 10, 7)
    assert M.sample_ddim(den, st, a, g, sig, disable=True).shape == (0, 10, 7)


@pytest.mark.parametrize("B", [1, 3, 5, 37, 129])
def test_ragged_batch_sizes_vs_oracle(B):
    """Batch sizes that are no multiple of any tile edge (N = 14 B tokens: 14 … 1806): partial GEMM tiles, partially filled expert
    segments, attention problem counts that do not fill a workgroup — fp32 mode against the oracle, per-sample sigma, router indices exact."""
    cfg, sd, m = build("c1e4", 210, "fp32")
    inp = make_inputs(cfg, B, 40 + B)
    sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(B))
    ref, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sig, return_aux=True)
    c = cuda_inputs(inp)
    with torch.no_grad():
        out = m({"state_images": c["state_images"]}, c["actions"], c["goals"], sig.cuda())
    assert torch.equal(m._last_topk.cpu().long(), torch.stack(aux.topk_idx)[:, :, 0, :])
    assert rel(out, ref) < 1e-4
    den = M.GCDenoiser(m, 0.5).eval()
    sched = M.get_sigmas_exponential(5, 1e-3, 80.0)
    x = M.sample_ddim(den, {"state_images": c["state_images"]}, c["x0"], c["goals"], sched.cuda(), disable=True)
    assert rel(x, O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sched)) < 1e-4


@pytest.mark.parametrize("B", [4, 37])
def test_down_projection_split_k_slices(B):
    """The inference path cuts the expert down-projection's K into 1-8 slices (dit.hip down_proj_split; bf16 partial slabs summed in slice
    order by the combine / head kernels): every slice count stays inside the bf16 tolerance against the fp32 oracle, routes identically, and
    agrees with the unsplit path to bf16 rounding of the partial sums."""
    from mode_diffusion_policy_amd import _lib as L
    cfg, sd, m = build("c1e4", 77, "bf16")
    inp = make_inputs(cfg, B, 90 + B)
    sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(B))
    ref, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sig, return_aux=True)
    c = cuda_inputs(inp)
    lib = L.load()
    outs = {}
    try:
        for s in (1, 2, 4, 8, 0):
            assert lib.mode_set_option(b"dn_split_k", s) == 0
            with torch.no_grad():
                outs[s] = m({"state_images": c["state_images"]}, c["actions"], c["goals"], sig.cuda()).float().cpu()
            assert torch.equal(m._last_topk.cpu().long(), torch.stack(aux.topk_idx)[:, :, 0, :])
            assert rel(outs[s], ref) < BF16_OUT, s
    finally:
        lib.mode_set_option(b"dn_split_k", 0)
    for s in (2, 4, 8, 0):
        assert rel(outs[s], outs[1]) < 4e-3, s
    assert lib.mode_set_option(b"dn_split_k", 9) != 0


@pytest.mark.parametrize("B", [1, 2, 3])
def test_single_environment_bf16_chain_vs_oracle(B):
    """The reference's own rollout is ONE environment (mode_agent.py:630): 14-28 token rows take the weight-streaming GEMM, ln_2 stays a kernel of its
    own, B = 3 crosses back to the tiled kernels with the fused ln_2.  bf16 chain + 10-step DDIM (hipGraph) against the fp32 oracle, routing exact."""
    cfg, sd, m = build("c1e4", 210, "bf16")
    inp = make_inputs(cfg, B, 70 + B)
    sig = O.rand_log_logistic((B,), float(np.log(0.5)), 0.5, 1e-3, 80.0, generator=torch.Generator().manual_seed(B))
    ref, aux = O.dit_forward(sd, cfg, inp["state_images"], inp["actions"], inp["goals"], sig, return_aux=True)
    c = cuda_inputs(inp)
    with torch.no_grad():
        out = m({"state_images": c["state_images"]}, c["actions"], c["goals"], sig.cuda())
    assert torch.equal(m._last_topk.cpu().long(), torch.stack(aux.topk_idx)[:, :, 0, :])
    assert rel(out, ref) < BF16_OUT
    den = M.GCDenoiser(m, 0.5).eval()
    sched = M.get_sigmas_exponential(10, 1e-3, 80.0)
    st = {"state_images": c["state_images"]}
    x = M.sample_ddim(den, st, c["x0"], c["goals"], sched.cuda(), disable=True)
    want = O.sample_ddim(sd, cfg, 0.5, inp["state_images"], inp["x0"], inp["goals"], sched)
    assert rel(x, want) < BF16_OUT
    assert torch.equal(x, M.sample_ddim(den, st, c["x0"], c["goals"], sched.cuda(), disable=True))      # graph replay is deterministic


# ------------------------------------------------------------------------------------------------- cond_router=False: token routing
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_token_routing_vs_reference(golden, dtype):
    """``cond_router=False`` (modedit.py:296-301, 322-325, 550-553): every block routes each token on its own ln_2-normalised state - routing
    is resolved INSIDE the launch chain, per layer (fp32 router MLP on the token states, top-k, dispatch).  Fixture F14 = the reference with that
    flag (eval forward at a shared and at per-sample noise levels, 10-step DDIM).  fp32: every token's experts identical, outputs <= 1e-3.
    bf16: the router input itself carries bf16 GEMM noise, so near-tied tokens may legitimately flip (the fixture's smallest top-k margin is
    5e-5): at least 97 % of the token decisions identical and a looser output tolerance."""
    g = golden("F14_c1e4_token_routing")
    cfg, sd, m = build(str(g["cfg"]), int(g["seed"]), dtype, cond_router=False)
    B = int(g["B"])
    inp = cuda_inputs(make_inputs(cfg, B, int(g["seed"]) + 1))
    st = {"state_images": inp["state_images"]}
    for tag in ("uniform", "persample"):
        sig = torch.from_numpy(g[f"{tag}_sigma"]).cuda()
        with torch.no_grad():
            out = m(st, inp["actions"], inp["goals"], sig)
        want_idx = torch.from_numpy(g[f"{tag}_idx"]).reshape(cfg.n_layers, -1, cfg.top_k)          # [L, B*T, k]
        got_idx = m._last_topk.cpu().long()
        same = (got_idx.sort(-1).values == want_idx.long().sort(-1).values).all(-1).float().mean().item()
        if dtype == "fp32":
            assert same == 1.0 and torch.equal(got_idx, want_idx.long()), tag
            assert rel(out, g[f"{tag}_out"]) < FP32_OUT, tag
        else:
            assert same >= BF16_TOKROUTE_AGREE, (tag, same)
            assert rel(out, g[f"{tag}_out"]) < BF16_TOKROUTE_OUT, tag
        with torch.no_grad():
            assert torch.equal(out, m(st, inp["actions"], inp["goals"], sig))                          # deterministic
    den = M.GCDenoiser(m, 0.5).eval()
    sig = torch.from_numpy(g["sigmas"]).cuda()
    x = M.sample_ddim(den, st, inp["x0"], inp["goals"], sig, disable=True)
    assert rel(x, g["x_final"]) < (FP32_OUT if dtype == "fp32" else BF16_TOKROUTE_OUT)
    m.precompute_experts_for_inference(sig[0])                                                          # nothing to cache per noise level: a no-op
    assert all(not blk.fused_experts for blk in m.blocks)
    m.train()                                                                                           # the training chain exists too (tests/test_gpu_train_dropin.py)
    assert m(st, inp["actions"], inp["goals"], sig[:1].expand(B)).shape == out.shape
