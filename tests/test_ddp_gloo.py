"""CPU, world_size 2, gloo: the data-parallel gradient exchange (mode_diffusion_policy_amd/ddp.py) — static buckets in backward order,
zero-fill of parameters that got no gradient on a rank (un-routed experts, dead `gripper_embed`), mean semantics equal to the
single-process gradient on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mode_diffusion_policy_amd.ddp import ArenaGradReducer, BucketedGradReducer, optimizer_param_groups
from oracle import mode_oracle as O


class ToyMoE(torch.nn.Module):
    """Tiny stand-in with the path's gradient pattern: a dead parameter and experts that only some ranks route to."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.emb = torch.nn.Linear(6, 16, bias=False)
        self.gripper_embed = torch.nn.Linear(6, 16, bias=False)                      # never used -> grad None everywhere
        self.experts = torch.nn.ModuleDict({f"expert_{i}": torch.nn.Linear(16, 16) for i in range(4)})
        self.out = torch.nn.Linear(16, 3)

    def forward(self, x, expert_ids):
        h = self.emb(x)
        y = torch.zeros_like(h)
        for e in range(4):
            m = expert_ids == e
            if m.any():
                y[m] = self.experts[f"expert_{e}"](h[m])
        return self.out(h + y)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, bucket_mb, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        X = torch.randn(8, 6); T = torch.randn(8, 3)
        ids = torch.tensor([0, 0, 1, 1, 2, 2, 2, 0])           # rank 0 rows 0-3 -> experts {0,1}; rank 1 rows 4-7 -> {2,0}; expert 3 unused
        model = ToyMoE()
        red = BucketedGradReducer(model, bucket_mb=bucket_mb, mode=mode)
        sl = slice(rank * 4, rank * 4 + 4)
        for step in range(2):                                   # two steps: the reducer must re-arm itself
            model.zero_grad(set_to_none=True)
            loss = (model(X[sl], ids[sl]) - T[sl]).pow(2).mean()
            loss.backward()
            red.finish()
        grads = {n: p.grad.clone() for n, p in model.named_parameters()}
        # reference: single process on the concatenated batch; per-rank losses are means over 4 rows -> global mean of the two
        ref = ToyMoE()
        l = 0.5 * ((ref(X[:4], ids[:4]) - T[:4]).pow(2).mean() + (ref(X[4:], ids[4:]) - T[4:]).pow(2).mean())
        l.backward()
        ok = True
        for n, p in ref.named_parameters():
            want = p.grad if p.grad is not None else torch.zeros_like(p)
            ok &= torch.allclose(grads[n], want, rtol=1e-5, atol=1e-7)
        zero_dead = float(grads["gripper_embed.weight"].abs().max()) == 0.0 and float(grads["experts.expert_3.weight"].abs().max()) == 0.0
        names = [b.names for b in red.buckets]
        q.put((rank, ok, zero_dead, names))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,bucket_mb", [("allreduce", 64.0), ("allreduce", 0.0005), ("rs_ag", 0.0005)])
def test_bucketed_reducer_world2(mode, bucket_mb):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, bucket_mb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, zero_dead, names in res:
        assert ok, f"rank {rank}: averaged gradients differ from the single-process gradient on the concatenated batch"
        assert zero_dead, "parameters without a gradient on any rank must come out as exact zeros"
    flat = [n for b in res[0][3] for n in b]
    assert flat[0].startswith("out.") and flat[-1] == "emb.weight"           # backward (reverse registration) order
    if bucket_mb < 1:
        assert len(res[0][3]) > 1                                             # small cap -> several buckets
    assert res[0][3] == res[1][3]                                             # identical static layout on every rank


def test_single_process_passthrough_and_param_groups():
    model = ToyMoE()
    red = BucketedGradReducer(model)
    x = torch.randn(4, 6)
    model(x, torch.tensor([0, 1, 0, 1])).sum().backward()
    g = model.out.weight.grad.clone()
    red.finish()
    assert torch.equal(model.out.weight.grad, g)
    assert float(model.gripper_embed.weight.grad.abs().max()) == 0.0         # zero-filled, not None
    groups = optimizer_param_groups(model, 0.05)
    decayed = {id(p) for p in groups[0]["params"]}
    for n, p in model.named_parameters():
        assert (id(p) in decayed) == O.uses_weight_decay(n)
    assert groups[0]["weight_decay"] == 0.05 and groups[1]["weight_decay"] == 0.0


def _arena_bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, n_reduce = 1000, 896
        g = torch.Generator().manual_seed(5)
        base = torch.randn(n, generator=g)
        flat = base * (rank + 1)
        red = ArenaGradReducer(flat, n_reduce, slice_mb=0.001, comm_dtype=torch.bfloat16)
        scale = red.reduce()
        want = (base.to(torch.bfloat16) * 1 + (base * 2).to(torch.bfloat16)).float()     # bf16 operands, bf16 sum
        err = float((flat[:n_reduce] - (base * 3.0)[:n_reduce]).abs().max() / (base * 3.0).abs().max())
        ok = err < 2 ** -7 and torch.equal(flat[n_reduce:], base[n_reduce:] * (rank + 1)) and flat.dtype == torch.float32
        ok = ok and float((flat[:n_reduce] - want[:n_reduce]).abs().max()) <= float(want.abs().max()) * 2 ** -7
        q.put((rank, bool(ok), scale, err))
    finally:
        dist.destroy_process_group()


def test_arena_reducer_bf16_exchange_world2():
    """comm_dtype=bfloat16: the slices travel as bf16 (half the bytes on the links), the arena stays fp32; the sum is exact to bf16 rounding."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_arena_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, scale, err in res:
        assert ok and scale == 0.5, (rank, err)
    with pytest.raises(ValueError):
        ArenaGradReducer(torch.zeros(8), 8, comm_dtype=torch.float16)


def _arena_worker(rank, world, port, mode, average, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, n_reduce = 1000, 896                                   # tail = the arena's dead region: must not be exchanged
        base = torch.arange(n, dtype=torch.float32)
        flat = base * (rank + 1)
        red = ArenaGradReducer(flat, n_reduce, slice_mb=0.001, mode=mode, average=average)   # 262 elements per slice -> 4 slices, ragged tail
        scale = red.reduce()
        want = base * 3.0 * (0.5 if average else 1.0)
        ok = torch.allclose(flat[:n_reduce], want[:n_reduce]) and torch.equal(flat[n_reduce:], base[n_reduce:] * (rank + 1))
        q.put((rank, bool(ok), scale, len(red.slices)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,average", [("allreduce", False), ("allreduce", True), ("rs_ag", False)])
def test_arena_reducer_world2(mode, average):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, mode, average, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, scale, nsl in res:
        assert ok, f"rank {rank}: arena slices not summed correctly"
        assert scale == (1.0 if average else 0.5)
        assert nsl == 4
