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
