"""GPU: no entry point of the path may accumulate device memory from call to call.

Round 3 found that the training node had pinned every step's activation stash since round 1 (an uncollectable reference cycle through the autograd
node; tests/test_gpu_train_dropin.py holds that regression test).  This file screens every OTHER repeated call of the path the same way: warm up, switch
Python's cyclic collector off (only reference counts may free memory), repeat, and require `torch.cuda.memory_allocated()` to stay flat."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu

import mode_diffusion_policy_amd as M  # noqa: E402
from mode_diffusion_policy_amd import gc_sampling, rollout  # noqa: E402
from oracle import resnet_oracle as R  # noqa: E402
from oracle.weights import get_config, make_inputs  # noqa: E402
from test_gpu_model import build  # noqa: E402
from test_gpu_train import build_train  # noqa: E402


def _flat(fn, warm=3, reps=8, slack=1 << 20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        fn()
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        seen = []
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()
            seen.append(torch.cuda.memory_allocated())
    finally:
        if was:
            gc.enable()
    assert max(seen) <= base + slack, (base, seen)


def test_inference_calls_hold_no_memory():
    cfg, sd, m = build("c1e4", 210, "bf16")
    den = M.GCDenoiser(m, 0.5).eval()
    B = 6
    inp = {k: v.cuda() for k, v in make_inputs(cfg, B, 3).items()}
    st = {"state_images": inp["state_images"]}
    sig10 = M.get_sigmas_exponential(10, 1e-3, 80.0).cuda()
    x0 = torch.randn(B, 10, 7, device="cuda") * 80
    sg = torch.full((B,), 1.3, device="cuda")
    with torch.no_grad():
        _flat(lambda: den(st, inp["actions"], inp["goals"], sg))                                              # one denoiser call (eager chain)
        _flat(lambda: M.sample_ddim(den, st, x0, inp["goals"], sig10, disable=True))                           # fused DDIM, same schedule tensor
        _flat(lambda: M.sample_ddim(den, st, x0, inp["goals"], M.get_sigmas_exponential(10, 1e-3, 80.0).cuda(), disable=True))   # schedule rebuilt per call (the agent's pattern)
        _flat(lambda: gc_sampling.sample_euler(den, st, x0, inp["goals"], sig10, disable=True))
        _flat(lambda: gc_sampling.sample_dpmpp_2m(den, st, x0, inp["goals"], sig10, disable=True))
        pol = rollout.ChunkedRolloutPolicy(den, num_sampling_steps=10, multistep=2, act_window_size=10)
        _flat(lambda: pol.step(st, inp["goals"].squeeze(1)), warm=4, reps=12)
        polb = rollout.ChunkedRolloutPolicy(den, sampler_type="dpmpp_2m", noise_scheduler="karras", multistep=1)
        _flat(lambda: polb.step(st, inp["goals"]))


@pytest.mark.parametrize("tokr,argmax", [(False, False), (False, True), (True, False)])
def test_training_variants_hold_no_memory(tokr, argmax):
    over = dict(use_argmax=argmax)
    if tokr:
        over["cond_router"] = False
    cfg, sd, m = build_train("c1e4", 210, "bf16", **over)
    den = M.GCDenoiser(m, 0.5).train()
    B = 8
    inp = {k: v.cuda() for k, v in make_inputs(cfg, B, 3).items()}
    sg = torch.full((B,), 0.7, device="cuda")
    enc = torch.nn.Linear(16, cfg.n_img_tokens * cfg.obs_dim).cuda()
    raw = torch.randn(B, 16, device="cuda")
    opt = torch.optim.SGD(list(m.parameters()) + list(enc.parameters()), lr=1e-5)

    def step():                                                                  # trainable encoder upstream, auxiliary losses in the graph, two losses / one backward
        img = enc(raw).view(B, cfg.n_img_tokens, cfg.obs_dim)
        l1, _ = den.loss({"state_images": img}, inp["actions"], inp["goals"], inp["noise"], sg)
        aux = m.load_balancing_loss() + m.compute_router_z_loss()
        l2, _ = den.loss({"state_images": img}, inp["actions"], inp["goals"], inp["noise"], sg * 0.5)
        (l1 + 0.01 * aux + l2).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    _flat(step)


def test_encoder_steps_hold_no_memory():
    enc = M.FiLMResNet18Policy(32).cuda().train()
    enc.load_state_dict({k: v.cuda() for k, v in R.fill_encoder_state_dict(enc.state_dict(), 1).items()})
    x = torch.randn(4, 3, 64, 64, device="cuda"); c = torch.randn(4, 1, 32, device="cuda")
    opt = torch.optim.SGD(enc.parameters(), lr=1e-5)

    def step():
        enc(x, c).square().mean().backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    _flat(step)
