"""Data-parallel gradient exchange for the score-matching training step (replaces Lightning's
``Trainer(strategy="ddp_find_unused_parameters_true")``, mode/training_calvin.py:92-103).

One process per GPU; the only exchange step of the path is the gradient mean over ranks.  MI355X-first choices:

* **Static flat buckets.**  Parameters are laid out once, in *reverse registration order* (the order backward produces
  gradients: head, last block … first block, embeddings), into contiguous fp32 (or bf16) bucket buffers of ``bucket_mb`` each.
  The layout never changes, so no per-step graph search is needed.
* **Unrouted experts / dead parameters are zero-filled** inside the static bucket (``gripper_embed.weight`` never receives a
  gradient and experts no token selected have ``grad is None`` — that is what ``find_unused_parameters=True`` papers over in the
  reference).  After the exchange every rank holds the same mean gradient, including exact zeros for globally-unused tensors.
* **Overlap with backward.**  ``register_post_accumulate_grad_hook`` marks a parameter ready; when the last parameter of a bucket is
  ready the bucket's collective is issued on a side stream (``async_op``), so communication of late layers overlaps the backward of
  early layers.  ``finish()`` waits, zero-fills whatever never fired, and scatters the averaged buckets back into ``p.grad``.
* **xGMI is point-to-point** (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound (~31 ms for 2.74 GB fp32), so the
  default collective is reduce-scatter + all-gather (``mode="rs_ag"``), which RCCL can spread over all 7 links (~4.5 ms fp32,
  ~2.2 ms with bf16 buckets); ``mode="allreduce"`` is kept for backends without reduce_scatter_tensor (gloo in the CPU tests).

The reducer is host logic on top of ``torch.distributed`` (backend "nccl" == RCCL on ROCm); it is exercised on CPU with gloo,
world_size 2, in ``tests/test_ddp_gloo.py``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


@dataclass
class _Bucket:
    index: int
    params: List[torch.nn.Parameter] = field(default_factory=list)
    names: List[str] = field(default_factory=list)
    offsets: List[int] = field(default_factory=list)
    numel: int = 0
    padded: int = 0
    buf: Optional[torch.Tensor] = None
    pending: int = 0
    ready: List[bool] = field(default_factory=list)
    work: Optional[object] = None
    shard: Optional[torch.Tensor] = None


class BucketedGradReducer:
    def __init__(self, module: torch.nn.Module, process_group=None, bucket_mb: float = 64.0, grad_dtype: torch.dtype = torch.float32,
                 mode: str = "auto", dead_params=("gripper_embed",)):
        if any(type(m).__name__ == "MoDeDiT" and hasattr(m, "engine") for m in module.modules()):
            # the HIP backward writes gradients straight into the gradient arena and hands autograd None for every parameter: the
            # post-accumulate hooks this reducer is driven by never fire, and finish() would overwrite the arena-aliased p.grad with zeros
            raise TypeError("BucketedGradReducer is hook-driven and cannot reduce a HIP MoDeDiT (its gradients live in the gradient arena): "
                            "use ArenaGradReducer.for_model(model)")
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.grad_dtype = grad_dtype
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        dev = named[0][1].device
        if mode == "auto":
            mode = "rs_ag" if dev.type == "cuda" else "allreduce"
        self.mode = mode
        cap = max(1, int(bucket_mb * 1024 * 1024 / torch.empty((), dtype=grad_dtype).element_size()))
        self.buckets: List[_Bucket] = []
        cur = _Bucket(0)
        for n, p in reversed(named):                       # backward order: last registered parameter first
            if cur.numel and cur.numel + p.numel() > cap:
                self.buckets.append(cur)
                cur = _Bucket(len(self.buckets))
            cur.params.append(p); cur.names.append(n); cur.offsets.append(cur.numel); cur.numel += p.numel()
        self.buckets.append(cur)
        self._where: Dict[int, tuple] = {}
        # tensors that never receive a gradient (the reference's dead `gripper_embed`, modedit.py:684): complete from the start
        self._dead = {id(p) for n, p in named if any(d in n for d in dead_params)}
        for b in self.buckets:
            b.padded = (b.numel + self.world - 1) // self.world * self.world      # reduce_scatter needs equal shards
            b.buf = torch.zeros(b.padded, dtype=grad_dtype, device=dev)
            b.ready = [False] * len(b.params)
            for i, p in enumerate(b.params):
                self._where[id(p)] = (b, i)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for _, p in named]
        self._comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.prepare()

    # ------------------------------------------------------------------ per-step protocol
    def prepare(self) -> None:
        """Call before backward (done automatically after ``finish``)."""
        self._next = 0
        for b in self.buckets:
            b.pending = len(b.params)
            b.ready = [False] * len(b.params)
            b.work = None
            for i, p in enumerate(b.params):                # statically dead tensors are complete (zero) up front
                if id(p) in self._dead:
                    b.buf[b.offsets[i]: b.offsets[i] + p.numel()].zero_()
                    b.ready[i] = True
                    b.pending -= 1

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        b, i = self._where[id(p)]
        if b.ready[i]:
            return                                          # gradient accumulation fired twice: keep the latest value at finish()
        off = b.offsets[i]
        b.buf[off: off + p.numel()].copy_(p.grad.reshape(-1))
        b.ready[i] = True
        b.pending -= 1
        self._launch_ready()

    def _launch_ready(self) -> None:
        """Collectives must be issued in the SAME order on every rank: buckets go out strictly in bucket order, each as soon as it
        and all earlier buckets are complete (a bucket holding a tensor that got no gradient on this rank waits for finish())."""
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b: _Bucket) -> None:
        if self.world == 1:
            return
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self._comm_stream)
        else:
            from contextlib import nullcontext
            ctx = nullcontext()
        with ctx:
            if self.mode == "rs_ag":
                shard = b.buf.new_empty(b.padded // self.world)
                b.shard = shard
                b.work = dist.reduce_scatter_tensor(shard, b.buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            else:
                b.work = dist.all_reduce(b.buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def finish(self) -> None:
        """Call after backward: completes the exchange and leaves the rank-mean gradient in every ``p.grad``."""
        for b in self.buckets[self._next:]:
            if b.pending > 0:                               # parameters that produced no gradient on THIS rank: zeros
                for i, p in enumerate(b.params):
                    if not b.ready[i]:
                        b.buf[b.offsets[i]: b.offsets[i] + p.numel()].zero_()
                b.pending = 0
        self._launch_ready()
        from contextlib import nullcontext
        for b in self.buckets:
            if self.world > 1:
                # everything that touches the bucket after its collective is enqueued on the COMMUNICATION stream, and work.wait() is called
                # with that stream current: wait() only orders the *current* stream behind the async collective, so waiting on the compute
                # stream and then dividing on the side stream would let div_ race the reduce_scatter that is still writing the shard
                with (torch.cuda.stream(self._comm_stream) if self._comm_stream is not None else nullcontext()):
                    b.work.wait()
                    if self.mode == "rs_ag":
                        b.shard.div_(self.world)
                        dist.all_gather_into_tensor(b.buf, b.shard, group=self.pg)
                    else:
                        b.buf.div_(self.world)
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        for b in self.buckets:
            for i, p in enumerate(b.params):
                g = b.buf[b.offsets[i]: b.offsets[i] + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.to(p.dtype).clone()
                else:
                    p.grad.copy_(g)
        self.prepare()

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()


class ArenaGradReducer:
    """Data-parallel gradient exchange for a MoDeDiT whose gradients live in the flat gradient arena (``arena.py``).

    The backward chain writes gradients straight into the arena in backward order (per-layer slices, last layer first), so the
    exchange is a handful of large collectives over static flat slices — no bucket copies at all.  ``reduce()`` is called once after
    ``loss.backward()`` returns (the backward kernels are still running): every block's slice is queued on a side stream behind the
    event the backward chain records when that block's gradients are complete, so communication overlaps the rest of the backward.  The SUM is left in place; the division by ``world`` is folded into ``FusedAdamW.step(grad_scale=1/world)``
    (or applied here with ``average=True`` for foreign optimizers).  xGMI is point-to-point: slices of ``slice_mb`` (default 256 MB,
    ~ one transformer block) keep each ring step large enough to run at link rate.
    """

    def __init__(self, grad_flat: torch.Tensor, n_reduce: int, process_group=None, slice_mb: float = 256.0, mode: str = "auto",
                 average: bool = False, comm_dtype: torch.dtype = torch.float32):
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be torch.float32 or torch.bfloat16")
        self.comm_dtype = comm_dtype
        self._stage = None                                  # bf16 staging buffer of the largest slice (comm_dtype=bfloat16)
        self.grad = grad_flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        dev = grad_flat.device
        if mode == "auto":
            mode = "allreduce"                              # RCCL picks ring/tree for the xGMI mesh itself; "rs_ag" = explicit in-place halves
        self.mode = mode
        per = max(self.world, int(slice_mb * 1024 * 1024 / grad_flat.element_size()) // self.world * self.world)
        self.slices = []
        o = 0
        while o < n_reduce:
            e = min(n_reduce, o + per)
            self.slices.append((o, e))
            o = e
        self._comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    @classmethod
    def for_model(cls, model, overlap: bool = True, **kw) -> "ArenaGradReducer":
        """Reducer over the model's gradient arena.  With ``overlap`` the slices follow the arena's per-block layout (blocks are stored in
        backward order) and each block slice waits only on the event the backward chain records when that block's gradients are complete,
        so its collective runs while the earlier blocks are still back-propagating."""
        eng = model.engine
        ar = eng.arena
        ar.ensure_grad(model)
        model.grad_mode = "arena"                                               # the exchange runs over the gradient arena the backward chain writes
        if overlap and dist.is_available() and dist.is_initialized() and dist.get_world_size(kw.get("process_group")) > 1:
            eng.lib.mode_set_option(b"bwd_coexec", 1)                           # collectives run beside the backward: its GEMMs leave CU room (mode_hip.h)
        red = cls(ar.grad, ar.bounds["no_decay"], **kw)
        if overlap and ar.grad.device.type == "cuda":
            from .training import TrainState
            if getattr(eng, "_train", None) is None:
                eng._train = TrainState(eng)
            eng._train.layer_events()
            evs = eng._train.events
            L = model.num_layers
            starts = [ar.offset(f"l{i}.wqkv") for i in range(L)]
            first, last_end = starts[L - 1], ar.offset("pos")                   # blocks are laid out L-1 ... 0, then the embeddings
            bounds = sorted(starts) + [last_end]
            block = {lo: hi for lo, hi in zip(bounds[:-1], bounds[1:])}
            red.slices = [(starts[i], block[starts[i]], evs[i]) for i in reversed(range(L))]
            red.slices += [(0, first, None), (last_end, ar.bounds["no_decay"], None)]   # stacked routers / gains; embeddings, head, biases
        return red

    def _exchange(self, g: torch.Tensor) -> None:
        """Sum one flat slice over the ranks, in place (runs on whatever stream is current: the communication stream)."""
        if self.comm_dtype != torch.float32:
            if self._stage is None or self._stage.numel() < g.numel():
                longest = max(hi - lo for lo, hi, *_ in self.slices)
                self._stage = torch.empty(max(longest, g.numel()), dtype=self.comm_dtype, device=g.device)
            st = self._stage[: g.numel()]
            st.copy_(g)                                                      # fp32 -> bf16, one pass
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.pg)
            g.copy_(st)
        elif self.mode == "rs_ag" and g.numel() % self.world == 0:
            shard = g.view(self.world, -1)[dist.get_rank(self.pg)]
            dist.reduce_scatter_tensor(shard, g, op=dist.ReduceOp.SUM, group=self.pg)      # in place: shard aliases its own slot
            dist.all_gather_into_tensor(g, shard, group=self.pg)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)
        if self.average:
            g.div_(self.world)

    def reduce_async(self):
        """Queue every slice's collective on the communication stream (each behind its block event) WITHOUT joining the current stream;
        returns {(lo, hi): torch.cuda.Event recorded after that slice's exchange} so that an optimizer can chain per-slice updates
        (``FusedAdamW.step(reducer=...)``).  The caller is responsible for the final join."""
        assert self._comm_stream is not None, "reduce_async needs CUDA streams"
        cur = torch.cuda.current_stream()
        done = {}
        for sl in self.slices:
            lo, hi, ev = sl if len(sl) == 3 else (sl[0], sl[1], None)
            if ev is not None:
                self._comm_stream.wait_event(ev)
            else:
                self._comm_stream.wait_stream(cur)
            with torch.cuda.stream(self._comm_stream):
                g = self.grad[lo:hi]
                self._exchange(g)
                e = torch.cuda.Event()
                e.record(self._comm_stream)
                done[(lo, hi)] = e
        return done

    def reduce_scatter_async(self, only=None):
        """ZeRO-1 exchange: every slice (or only slice ``only``) is reduce-scattered IN PLACE (rank r ends up with the sum of its 1/world part of
        the slice, at its own offset of the gradient arena - RCCL's in-place layout recvbuff == sendbuff + rank * count); queued on the
        communication stream behind the slice's block event.  Returns [(lo, hi, shard_lo, shard_hi, event)] in slice order.  Half the bytes of an
        all-reduce on the wire; the optimizer then updates only the shard and all-gathers the new weights (``FusedAdamW.step(zero1=...)``)."""
        cur = torch.cuda.current_stream() if self._comm_stream is not None else None
        rank = dist.get_rank(self.pg) if self.world > 1 else 0
        out = []
        from contextlib import nullcontext
        for sl in (self.slices if only is None else [self.slices[only]]):
            lo, hi, ev = sl if len(sl) == 3 else (sl[0], sl[1], None)
            if (hi - lo) % self.world:
                raise ValueError(f"slice [{lo}, {hi}) does not split evenly over {self.world} ranks")
            per = (hi - lo) // self.world
            slo, shi = lo + rank * per, lo + (rank + 1) * per
            if self._comm_stream is not None:
                if ev is not None:
                    self._comm_stream.wait_event(ev)
                else:
                    self._comm_stream.wait_stream(cur)
            with (torch.cuda.stream(self._comm_stream) if self._comm_stream is not None else nullcontext()):
                if self.world > 1:
                    g = self.grad[lo:hi]
                    if self.comm_dtype != torch.float32:
                        if self._stage is None or self._stage.numel() < g.numel():
                            longest = max(h - l for l, h, *_ in self.slices)
                            self._stage = torch.empty(longest, dtype=self.comm_dtype, device=g.device)
                        st = self._stage[: g.numel()]
                        st.copy_(g)
                        dist.reduce_scatter_tensor(st[rank * per:(rank + 1) * per], st, op=dist.ReduceOp.SUM, group=self.pg)
                        self.grad[slo:shi].copy_(st[rank * per:(rank + 1) * per])
                    else:
                        dist.reduce_scatter_tensor(self.grad[slo:shi], g, op=dist.ReduceOp.SUM, group=self.pg)
                e = None
                if self._comm_stream is not None:
                    e = torch.cuda.Event()
                    e.record(self._comm_stream)
            out.append((lo, hi, slo, shi, e))
        return out

    def all_gather_async(self, flat: torch.Tensor, lo: int, hi: int, after=None):
        """All-gather ``flat[lo:hi]`` in place (every rank contributes its 1/world part) on the communication stream, after ``after`` (an event
        on another stream: the shard's optimizer update).  Returns the event recorded behind the collective."""
        from contextlib import nullcontext
        per = (hi - lo) // self.world
        rank = dist.get_rank(self.pg) if self.world > 1 else 0
        if self._comm_stream is not None and after is not None:
            self._comm_stream.wait_event(after)
        with (torch.cuda.stream(self._comm_stream) if self._comm_stream is not None else nullcontext()):
            if self.world > 1:
                dist.all_gather_into_tensor(flat[lo:hi], flat[lo + rank * per: lo + (rank + 1) * per], group=self.pg)
            e = None
            if self._comm_stream is not None:
                e = torch.cuda.Event()
                e.record(self._comm_stream)
        return e

    def reduce(self) -> float:
        """Sum (or average) the gradient arena over ranks; returns the scale the optimizer still has to apply.  Call right after
        ``loss.backward()`` returned: the backward kernels are still executing, the collectives queue up behind their events."""
        if self.world == 1:
            return 1.0
        cur = torch.cuda.current_stream() if self._comm_stream is not None else None
        for sl in self.slices:
            lo, hi, ev = sl if len(sl) == 3 else (sl[0], sl[1], None)
            if self._comm_stream is not None:
                if ev is not None:
                    self._comm_stream.wait_event(ev)
                else:
                    self._comm_stream.wait_stream(cur)
                ctx = torch.cuda.stream(self._comm_stream)
            else:
                from contextlib import nullcontext
                ctx = nullcontext()
            with ctx:
                g = self.grad[lo:hi]
                self._exchange(g)
        if self._comm_stream is not None:
            cur.wait_stream(self._comm_stream)
        return 1.0 if self.average else 1.0 / self.world


def optimizer_param_groups(model: torch.nn.Module, weight_decay: float):
    """AdamW grouping rule of MoDEAgent.get_optim_groups (mode/models/mode_agent.py:365-384): decay every denoiser parameter whose
    NAME contains none of 'bias' / 'LayerNorm' / 'embedding' (so RMSNorm gains, pos_emb and *_emb.weight ARE decayed)."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        (decay if all(s not in name for s in ("bias", "LayerNorm", "embedding")) else no_decay).append(p)
    return [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]
