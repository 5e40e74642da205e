"""Host-side driver of the HIP denoiser: weight shadows, workspaces and the launch chain.

PyTorch is plumbing here (device memory, streams); all arithmetic of the hot path runs in libmode_hip.so.
Parameters live in one flat HBM arena (``arena.py``): the module's Parameters are views into it, q/k/v and the experts of a
block are adjacent (packed operands without copies) and the bf16 compute shadow is one cast of the arena.  The ``state_dict``
layout is never changed; the shadow is refreshed when a parameter version changes (optimizer step, ``load_state_dict``, EMA
swap — SURVEY.md §7 "weight-layout staleness").
"""
from __future__ import annotations

import contextlib
import ctypes as C
import gc
from typing import Optional, Tuple

import torch

from . import _lib as L

_DT = {"bf16": (L.MODE_BF16, torch.bfloat16), "fp32": (L.MODE_F32, torch.float32)}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Raw handle of torch's current stream on the current device.  `torch.cuda.current_stream()` builds a Stream object per call - 8.7 us, 3 ms of host time per
    agent training step (scripts/agent_host_profile.py) - the raw accessor returns the same handle (also inside a graph capture) in well under a microsecond."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


@contextlib.contextmanager
def capture_graph(graph: "torch.cuda.CUDAGraph"):
    """hipGraph stream capture with the Python cyclic collector held off.  A collection that happens to fire inside the capture window can
    run the destructor of an older, unreachable ``CUDAGraph`` (hipGraphExecDestroy + release of its private pool), which the runtime refuses
    while a capture is open and the process aborts; so dead cycles are collected BEFORE the capture opens and the collector stays disabled
    until it closes.  ``thread_local`` error mode: other threads (the RCCL watchdog) may touch the runtime during capture."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            yield
    finally:
        if was:
            gc.enable()


class DitEngine:
    def __init__(self, model, compute_dtype: str = "bf16"):
        if compute_dtype not in _DT:
            raise ValueError(f"compute_dtype must be one of {list(_DT)}")
        self.model = model
        self.compute_dtype = compute_dtype
        self.dt, self.tdt = _DT[compute_dtype]
        self.lib = L.load()
        self._wkey = None
        self._pv = None
        self._structs_for = None
        self.arena = None
        self._ws: Optional[torch.Tensor] = None
        self._ws_pin: Optional[torch.Tensor] = None            # workspace owned by a hipGraph being warmed up / captured (see pinned_workspace)
        m = model
        self.dims = L.ModeDims(D=m.embed_dim, H=m.n_heads, L=m.num_layers, E=m.num_experts, k=m.top_k, T=m.seq_len,
                               A_len=m.action_seq_len, A_dim=m.action_dim, O=m.obs_dim, G=m.goal_dim, n_img=m.n_img_tokens,
                               use_noise_token=int(m.use_noise_token_as_input), router_normalize=int(m.router_normalize),
                               eps=1e-6)

    # ------------------------------------------------------------------ weights
    def param_index(self):
        """[(owner module, attribute name, parameter, dotted name)] of the model, cached.  ``Module.named_parameters()`` walks the module tree
        through nested generators - three walks per training step cost ~10 ms of host time under a profiler (the eager training chain is the
        one host-sensitive leg: a slow host then makes the step host-bound).  The cache is validated by identity against the owners'
        ``_parameters`` dicts (a re-assigned Parameter object rebuilds it); re-allocated storage is caught by ``ParamArena.owns``."""
        idx = getattr(self, "_pidx", None)
        if idx is not None and all(mod._parameters.get(n) is p for mod, n, p, _ in idx):
            return idx
        idx = []
        nmod = 0
        for mname, mod in self.model.named_modules():
            nmod += 1
            for n, p in mod._parameters.items():
                if p is not None:
                    idx.append((mod, n, p, f"{mname}.{n}" if mname else n))
        order = {nm: i for i, (nm, _) in enumerate(self.model.named_parameters())}      # registration order = what named_parameters() yields
        idx.sort(key=lambda t: order[t[3]])
        self._pidx, self._pidx_modules = idx, nmod
        return idx

    def named_params(self):
        return [(nm, p) for _, _, p, nm in self.param_index()]

    def _key(self):
        ar = self.arena
        return (id(ar), ar.version, tuple(p._version for _, _, p, _ in self.param_index()))

    def ensure_weights(self) -> None:
        """Adopt the module's parameters into the flat arena (once) and keep the compute shadow current.

        Staleness: an in-place update through the Parameter views (torch optimizers, ``load_state_dict``, EMA swap) bumps
        their version counters -> the bf16 shadow is re-cast; ``FusedAdamW`` writes the shadow itself and bumps ``arena.version``;
        ``.to()`` / ``.half()`` re-allocate parameters -> they no longer alias the arena and it is rebuilt."""
        m = self.model
        ar = self.arena
        if ar is None or not ar.owns(m, self.named_params()):
            dev = m.pos_emb.device
            if dev.type != "cuda":
                raise L.ModeHipUnavailable("MoDeDiT parameters must live on a ROCm device: the denoising path has no CPU implementation")
            from .arena import ParamArena
            ar = self.arena = ParamArena(m, dev)
            self.device = dev
            self._wkey = self._pv = None
            self._structs_for = None
        key = self._key()
        if key == self._wkey:
            return
        if self._pv is not None and key[2] != self._pv:
            ar.lp_synced = False                                    # somebody wrote through the Parameter views
        if self.compute_dtype == "bf16":
            ar.ensure_lp()
        if self._structs_for != (id(ar), ar.lp is not None):
            self._build_structs()
        self._wkey, self._pv = key, key[2]

    def _build_structs(self) -> None:
        m, ar = self.model, self.arena
        mat = ar.wl if self.compute_dtype == "bf16" else ar.w      # GEMM operands in the compute dtype; everything else fp32
        w = ar.w
        layers = (L.ModeLayerWeights * m.num_layers)()
        for i in range(m.num_layers):
            k = f"l{i}."
            lw = layers[i]
            lw.ln1_g, lw.ln2_g, lw.qn_g, lw.kn_g = (_ptr(w[n][i]) for n in ("ln1_g", "ln2_g", "qn_g", "kn_g"))
            lw.wqkv, lw.bqkv, lw.wo = _ptr(mat[k + "wqkv"]), _ptr(w[k + "bqkv"]), _ptr(mat[k + "wo"])
            lw.r_w0, lw.r_b0, lw.r_w3, lw.r_b3 = _ptr(w["r_w0"][i]), _ptr(w["r_b0"][i]), _ptr(w["r_w3"][i]), _ptr(w["r_b3"][i])
            lw.w1, lw.b1, lw.w2 = _ptr(mat[k + "w1"]), _ptr(w[k + "b1"]), _ptr(mat[k + "w2"])
        mw = L.ModeModelWeights()
        for nm in ("pos", "w_se", "b_se", "w_sl", "w_tok", "w_goal", "w_act", "ln_g", "w_out", "b_out"):
            setattr(mw, nm, _ptr(w[nm]))
        mw.layers = C.cast(layers, C.POINTER(L.ModeLayerWeights))
        self._layers, self._mw = layers, mw
        self._structs_for = (id(ar), ar.lp is not None)

    def weights_updated(self, lp_synced: bool) -> None:
        """Called by an optimizer that wrote the arena directly (no autograd version bump)."""
        self.arena.version += 1
        self.arena.lp_synced = bool(lp_synced) and self.arena.lp is not None

    # ------------------------------------------------------------------ workspace
    def workspace_bytes(self, B: int, R: int) -> int:
        need = self.lib.mode_dit_workspace_bytes(C.byref(self.dims), B, R, self.dt)
        if need == 0:
            raise RuntimeError("unsupported MoDeDiT dimensions for the HIP path")
        return need

    def workspace(self, B: int, R: int) -> Tuple[int, int]:
        need = self.workspace_bytes(B, R)
        pin = self._ws_pin
        if pin is not None:                                     # a captured graph bakes this pointer in: it must never be re-allocated
            if pin.numel() < need:
                raise RuntimeError("pinned workspace too small for this launch chain")
            return pin.data_ptr(), pin.numel()
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws.data_ptr(), self._ws.numel()

    def pinned_workspace(self, ws: torch.Tensor):
        """Context manager: every launch chain issued inside uses ``ws`` (a tensor the caller keeps alive next to its hipGraph) instead of
        the engine's growable scratch buffer — a replay must not read a pointer that a later, larger request freed."""
        eng = self

        class _Pin:
            def __enter__(self_inner):
                self_inner.prev = eng._ws_pin
                eng._ws_pin = ws

            def __exit__(self_inner, *exc):
                eng._ws_pin = self_inner.prev
                return False
        return _Pin()

    def meta_layout(self, N: int) -> L.ModeMetaLayout:
        ml = L.ModeMetaLayout()
        L.check(self.lib.mode_moe_meta_layout(N, self.dims.E, self.dims.k, C.byref(ml)), "meta_layout")
        return ml

    # ------------------------------------------------------------------ building blocks (each = a few launches, no sync)
    def sigma_embed(self, sigma: torch.Tensor) -> torch.Tensor:
        R = sigma.numel()
        emb = torch.empty(R, self.dims.D, dtype=torch.float32, device=self.device)
        ws, wsn = self.workspace(0, R)
        L.check(self.lib.mode_dit_sigma_embed(C.byref(self.dims), C.byref(self._mw), sigma.data_ptr(), R, emb.data_ptr(), ws, wsn,
                                              _stream()), "sigma_embed")
        return emb

    def embed_obs(self, state_images: torch.Tensor, goals: torch.Tensor, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
        B = state_images.shape[0]
        if out is not None:
            img_e, goal_e = out
        else:
            img_e = torch.empty(B * self.dims.n_img, self.dims.D, dtype=torch.float32, device=self.device)
            goal_e = torch.empty(B, self.dims.D, dtype=torch.float32, device=self.device)
        L.check(self.lib.mode_dit_embed_obs(C.byref(self.dims), C.byref(self._mw), state_images.data_ptr(), goals.data_ptr(), B,
                                            img_e.data_ptr(), goal_e.data_ptr(), _stream()), "embed_obs")
        return img_e, goal_e

    def route(self, cond: torch.Tensor, want_probs: bool = False, want_pre: bool = False):
        R, d = cond.shape[0], self.dims
        idx = torch.empty(d.L, R, d.k, dtype=torch.int32, device=self.device)
        w = torch.empty(d.L, R, d.k, dtype=torch.float32, device=self.device)
        probs = torch.empty(d.L, R, d.E, dtype=torch.float32, device=self.device) if want_probs else None
        shifted = torch.empty(d.L, R, d.E, dtype=torch.float32, device=self.device) if want_probs else None
        pre = torch.empty(R, d.L, 2 * d.D, dtype=torch.float32, device=self.device) if want_pre else None      # [R][L][2D]
        ws, wsn = self.workspace(0, R)
        L.check(self.lib.mode_dit_route(C.byref(d), C.byref(self._mw), cond.data_ptr(), R, idx.data_ptr(), w.data_ptr(),
                                        _ptr(probs), _ptr(shifted), _ptr(pre), ws, wsn, _stream()), "route")
        if want_pre:
            return idx, w, probs, shifted, pre
        return idx, w, probs, shifted

    def dispatch(self, idx: torch.Tensor, w: torch.Tensor, nbatch: int, R: int, tokens_per_row: int, N: int) -> torch.Tensor:
        ml = self.meta_layout(N)
        meta = torch.empty(nbatch, ml.total_words, dtype=torch.int32, device=self.device)
        L.check(self.lib.mode_dit_dispatch(idx.data_ptr(), w.data_ptr(), nbatch, R * self.dims.k, R, tokens_per_row, N, self.dims.E,
                                           self.dims.k, meta.data_ptr(), _stream()), "dispatch")
        return meta

    def forward(self, B: int, emb_t, emb_stride: int, cond, cond_stride: int, meta_ptr: int, meta_layer_stride: int, goal_e, img_e,
                actions, c_in=None, c_in_stride: int = 0, scal_ptr: Optional[int] = None, scal_stride: int = 0, F=None,
                denoised=None, x_next=None, topk_out=None, uniform: bool = False, den_prev=None, lin_ptr: Optional[int] = None, aux1=None, aux2=None) -> None:
        """``meta_ptr=None``: token routing (cond_router=False) - every block routes inside the chain; ``topk_out`` int32 [L, B*T, k] receives the experts.
        ``uniform``: the dispatch records were built from ONE routing row for the whole batch (``dispatch(..., R=1, ...)``) - see ModeForwardArgs.
        ``den_prev``: the previous step's ``denoised`` for the head's two-point multistep update (scal[3] = its weight; ModeHeadDesc.den_prev).
        ``lin_ptr`` / ``aux1`` / ``aux2``: the head's general linear update of the two-stage solvers (ModeHeadDesc.lin)."""
        a = L.ModeForwardArgs(B=B, dtype=self.dt, emb_t=_ptr(emb_t), emb_row_stride=emb_stride, cond=_ptr(cond),
                              cond_row_stride=cond_stride, meta=meta_ptr, meta_layer_stride=meta_layer_stride,
                              goal_e=_ptr(goal_e), img_e=_ptr(img_e), actions=_ptr(actions), c_in=_ptr(c_in) if torch.is_tensor(c_in) else c_in,
                              c_in_stride=c_in_stride, scal=scal_ptr, scal_stride=scal_stride, F=_ptr(F), denoised=_ptr(denoised),
                              x_next=_ptr(x_next), topk_idx_out=_ptr(topk_out), uniform_routing=int(bool(uniform) and meta_ptr is not None),
                              den_prev=_ptr(den_prev), lin=lin_ptr, aux1=_ptr(aux1), aux2=_ptr(aux2))
        ws, wsn = self.workspace(B, 0)
        L.check(self.lib.mode_dit_forward(C.byref(self.dims), C.byref(self._mw), C.byref(a), ws, wsn, _stream()), "dit_forward")
