"""Randomised screen of mode_gemm under the auto dispatch (scripts/gemm_fuzz.py): every regime boundary (streamer / register-resident /
ring / ping-pong), grouped incl. empty experts, gathered, uniform-group hints, split-K, all epilogues - against an fp32 reference of the same
bf16 operands, with canary rows (out-of-bounds writes) and a NaN pre-fill (unwritten elements)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1])
def test_gemm_auto_dispatch_fuzz(seed):
    import gemm_fuzz
    failed, unsupported = gemm_fuzz.run(cases=150, seed=seed, verbose=False)
    assert failed == 0
    assert unsupported == 0        # every generated case is inside the documented contract of mode_gemm


@pytest.mark.parametrize("seed", [0, 1])
def test_gemm_f32_fuzz(seed):
    """The fp32 MFMA GEMM (gemm_f32.hip: 64 x 64 / 32 x 32 tiles, 64 / 128-k swizzled LDS fills, [rows][K] and [K][cols] operand layouts) on random shapes -
    M, N, K off every tile / fill / vector boundary, all epilogues, bf16 and fp32 outputs, grouped (incl. empty experts), gathered rows, K-groups - against a
    float64 reference, with canary rows around the output and a NaN pre-fill.  MODE_FUZZ_F32_CASES widens it."""
    import ctypes as C
    import random
    import torch
    from mode_diffusion_policy_amd import _lib as L
    lib = L.load(); dev = "cuda"
    rng = random.Random(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    p = lambda t: None if t is None else t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    n_cases = int(os.environ.get("MODE_FUZZ_F32_CASES", "70"))
    worst = 0.0
    for case in range(n_cases):
        kind = rng.choice(["plain", "plain", "grouped", "gathered", "kn", "kn", "kgroups"])
        M = rng.choice([1, 7, 17, 31, 33, 63, 64, 65, 100, 128, 130, 257, 600]) if kind != "kn" else rng.choice([4, 20, 64, 68, 132, 512])
        N = rng.choice([17, 18, 32, 33, 64, 66, 100, 128, 200, 260]) if kind not in ("kn", "kgroups") else rng.choice([20, 64, 68, 128, 260])
        K = rng.choice([4, 7, 16, 30, 48, 64, 100, 128, 130, 192, 200, 256, 500, 1024])
        if kind in ("kn", "kgroups"):
            K = (K + 3) // 4 * 4
        epi = rng.choice([L.EPI_NONE, L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_RESIDUAL, L.EPI_SWIGLU]) if kind in ("plain", "grouped", "gathered") else L.EPI_NONE
        ob = rng.random() < 0.3 and kind in ("plain", "grouped", "gathered")
        E = rng.choice([2, 4]) if kind == "grouped" else 1
        wrows = 2 * N if epi == L.EPI_SWIGLU else N
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(E, wrows, K, generator=g) * K ** -0.5).to(dev)
        bias = torch.randn(E, wrows, generator=g).to(dev) if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_SWIGLU) else None
        resid = torch.randn(M, N, generator=g).to(dev) if epi == L.EPI_RESIDUAL else None
        groups = 1
        out = torch.full((M + 2, N), float("nan"), dtype=torch.bfloat16 if ob else torch.float32, device=dev)
        d = L.ModeGemmDesc(dtype=L.MODE_F32, epilogue=epi, out_dtype=L.MODE_BF16 if ob else L.MODE_F32, M=M, N=N, K=K, A=p(A), lda=K, W=p(W), ldw=K, bias=p(bias),
                           resid=p(resid), ldr=N, C=out[1:].data_ptr(), ldc=N)
        Af = A.double().cpu()
        seg = [(0, M, 0)]
        keep = [A, W, bias, resid]
        if kind == "grouped":
            cuts = sorted(rng.randint(0, M) for _ in range(E - 1))
            offs = [0] + cuts + [M]
            if rng.random() < 0.3:
                offs[1] = offs[0]                                           # an empty expert
                offs = sorted(offs)
            off_t = torch.tensor(offs, dtype=torch.int32, device=dev); keep.append(off_t)
            d.expert_offsets = p(off_t); d.num_experts = E; d.w_expert_stride = wrows * K; d.bias_expert_stride = wrows
            seg = [(offs[e], offs[e + 1], e) for e in range(E)]
        elif kind == "gathered":
            src = torch.randn(M + 9, K, generator=g).to(dev)
            rows = torch.randint(0, M + 9, (M,), generator=g).int().to(dev); keep += [src, rows]
            d.A = p(src); d.a_rows = p(rows)
            Af = src.double().cpu()[rows.cpu().long()]
        elif kind in ("kn", "kgroups"):
            a_km = rng.random() < 0.6 or kind == "kgroups"
            w_kn = rng.random() < 0.7 or not a_km
            At = A.t().contiguous() if a_km else A
            Wt = W[0].t().contiguous() if w_kn else W[0]
            keep += [At, Wt]
            d.A = p(At); d.lda = At.stride(0); d.W = p(Wt); d.ldw = Wt.stride(0)
            d.flags = (L.GEMM_A_KM if a_km else 0) | (L.GEMM_W_KN if w_kn else 0)
            if (a_km and M % 4) or (w_kn and N % 4):
                continue                                                    # [K][cols] operands need 16-byte aligned rows (documented)
            if kind == "kgroups":
                groups = rng.choice([2, 3, 5])
                ko = sorted(rng.randint(0, K // 4) * 4 for _ in range(groups - 1))
                ko = [0] + ko + [K]
                ko_t = torch.tensor(ko, dtype=torch.int32, device=dev); keep.append(ko_t)
                out = torch.full((groups, M + 2, N), float("nan"), device=dev)
                d.C = out[0, 1:].data_ptr(); d.k_group_offsets = p(ko_t); d.num_k_groups = groups; d.c_group_stride = (M + 2) * N
        rc = lib.mode_gemm(C.byref(d), st)
        torch.cuda.synchronize()
        assert rc == 0, (case, kind, M, N, K, epi, rc)
        Wd = W.double().cpu()
        if kind == "kgroups":
            for gi in range(groups):
                k0, k1 = ko[gi], ko[gi + 1]
                ref = Af[:, k0:k1] @ Wd[0][:, k0:k1].t()
                got = out[gi, 1:M + 1].double().cpu()
                assert torch.isnan(out[gi, 0]).all() and torch.isnan(out[gi, M + 1]).all(), (case, "canary")
                err = float((got - ref).norm() / max(float(ref.norm()), 1e-9)) if k1 > k0 else float(got.abs().max())
                worst = max(worst, err)
                assert err < 2e-6, (case, kind, M, N, K, gi, err)
            continue
        ref = torch.empty(M, N, dtype=torch.float64)
        for lo, hi, e in seg:
            z = Af[lo:hi] @ Wd[e].t()
            if bias is not None:
                z = z + bias[e].double().cpu()
            if epi == L.EPI_BIAS_GELU:
                z = torch.nn.functional.gelu(z)
            if epi == L.EPI_SWIGLU:
                z = z[:, :N] * torch.nn.functional.silu(z[:, N:])
            if epi == L.EPI_RESIDUAL:
                z = z + resid[lo:hi].double().cpu()
            ref[lo:hi] = z
        got = out[1:M + 1].double().cpu()
        assert torch.isnan(out[0].float()).all() and torch.isnan(out[M + 1].float()).all(), (case, "canary rows overwritten")
        assert not torch.isnan(got).any(), (case, kind, M, N, K, epi, "unwritten output")
        err = float((got - ref).norm() / max(float(ref.norm()), 1e-9))
        worst = max(worst, err)
        assert err < (6e-3 if ob else 3e-6), (case, kind, M, N, K, epi, ob, err)
    print(f"gemm_f32 fuzz seed {seed}: {n_cases} cases, worst fp32 rel-L2 error {worst:.2e}")
