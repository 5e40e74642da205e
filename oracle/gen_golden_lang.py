"""Golden fixture F13: the reference's language-goal embedding cache (mode/utils/lang_buffer.py:6-71) driven through a scripted sequence with a
deterministic stand-in encoder (build container only; imports /root/reference).  Recorded: every returned batch, the encoder's call log, the
buffer's key order and size after every operation, and a save/load round trip through a smaller buffer.

    python -m oracle.gen_golden_lang         # writes tests/golden/F13_lang_buffer.npz
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import tempfile

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DIM = 6


class FakeEncoder:
    """texts -> [len, DIM] from a hash of the text (what CLIP's text tower is to the cache: a pure function of the string)"""
    output_dim = DIM

    def __init__(self):
        self.calls = []

    def __call__(self, texts):
        self.calls.append(list(texts))
        if any(t == "<boom>" for t in texts):
            raise RuntimeError("encoder failure")
        rows = [np.frombuffer(hashlib.sha256(t.encode()).digest()[: DIM * 4], dtype=np.uint32).astype(np.float64) / 2 ** 32 for t in texts]
        return torch.tensor(np.stack(rows), dtype=torch.float32)


# the scripted sequence: (op, argument)
SCRIPT = [("batch", ["open the drawer", "push the blue block", "open the drawer"]), ("one", "turn on the light"), ("batch", ["push the blue block"]),
          ("batch", ["a", "b", "c"]), ("batch", ["open the drawer", "d"]), ("batch", ["<boom>", "a"]), ("str", "lift the red block"),
          ("preload", ["e", "f", "a"]), ("batch", ["turn on the light", "e"]), ("clear", None), ("batch", ["a", "a"])]
CAPACITY = 6


def run(cls):
    enc = FakeEncoder()
    buf = cls(enc, CAPACITY)
    outs, keys, sizes = [], [], []
    for op, arg in SCRIPT:
        if op == "batch":
            r = buf.get_goal_instruction_embeddings(arg)
        elif op == "one":
            r = buf.get_goal_instruction_embedding(arg)
        elif op == "str":
            r = buf.get_or_encode_batch(arg)
        elif op == "preload":
            buf.preload_common_strings(arg); r = torch.zeros(0, DIM)
        elif op == "clear":
            buf.clear_buffer(); r = torch.zeros(0, DIM)
        outs.append(r.detach().cpu().numpy().astype(np.float32))
        keys.append("|".join(buf.goal_instruction_buffer.keys()))
        sizes.append(buf.get_buffer_size())
    # save / load round trip into a smaller buffer: the newest entries survive, in order
    for t in ("x1", "x2", "x3", "x4"):
        buf.get_goal_instruction_embedding(t)
    path = os.path.join(tempfile.mkdtemp(), "buf.pkl")
    buf.save_buffer(path)
    small = cls(FakeEncoder(), 3)
    small.load_buffer(path)
    loaded_keys = "|".join(small.goal_instruction_buffer.keys())
    loaded_vals = torch.stack([v for v in small.goal_instruction_buffer.values()]).cpu().numpy().astype(np.float32)
    return dict(outs=outs, keys=keys, sizes=sizes, calls=["|".join(c) for c in enc.calls], loaded_keys=loaded_keys, loaded_vals=loaded_vals,
                saved_keys="|".join(buf.goal_instruction_buffer.keys()))


def main():
    spec = importlib.util.spec_from_file_location("ref_lang_buffer", "/root/reference/mode/utils/lang_buffer.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    ref = run(mod.AdvancedLangEmbeddingBuffer)
    from mode_diffusion_policy_amd.lang_buffer import AdvancedLangEmbeddingBuffer as Mine
    mine = run(Mine)
    for k in ("keys", "sizes", "calls", "loaded_keys", "saved_keys"):
        assert ref[k] == mine[k], (k, ref[k], mine[k])
    for a, b in zip(ref["outs"], mine["outs"]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(ref["loaded_vals"], mine["loaded_vals"])
    print("reference == build on the scripted sequence;", len(SCRIPT), "operations,", len(ref["calls"]), "encoder calls")
    np.savez(os.path.join(OUT, "F13_lang_buffer.npz"), capacity=CAPACITY, dim=DIM, n_ops=len(SCRIPT),
             keys=np.array(ref["keys"]), sizes=np.array(ref["sizes"]), calls=np.array(ref["calls"]), loaded_keys=ref["loaded_keys"],
             saved_keys=ref["saved_keys"], loaded_vals=ref["loaded_vals"], **{f"out{i}": o for i, o in enumerate(ref["outs"])})


if __name__ == "__main__":
    main()
