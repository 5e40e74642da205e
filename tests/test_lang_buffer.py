"""Language-goal embedding cache: the build's AdvancedLangEmbeddingBuffer replayed against fixture F13, which records the REFERENCE
class (mode/utils/lang_buffer.py) on the same scripted sequence (oracle/gen_golden_lang.py: the two were compared operation by operation when
the fixture was written)."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle.gen_golden_lang import CAPACITY, DIM, SCRIPT, FakeEncoder, run
from mode_diffusion_policy_amd.lang_buffer import AdvancedLangEmbeddingBuffer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "F13_lang_buffer.npz")


def test_scripted_sequence_matches_reference_fixture(capsys):
    g = np.load(GOLD)
    assert int(g["capacity"]) == CAPACITY and int(g["dim"]) == DIM and int(g["n_ops"]) == len(SCRIPT)
    mine = run(AdvancedLangEmbeddingBuffer)
    assert mine["keys"] == list(g["keys"]) and mine["sizes"] == list(g["sizes"])              # FIFO eviction order, hit does not refresh, clear
    assert mine["calls"] == list(g["calls"])                                                   # the encoder only ever sees texts not in the cache
    for i, o in enumerate(mine["outs"]):
        assert np.array_equal(o, g[f"out{i}"]), i                                              # values, request order, the zero fallback on errors
    assert mine["saved_keys"] == str(g["saved_keys"]) and mine["loaded_keys"] == str(g["loaded_keys"])
    assert np.array_equal(mine["loaded_vals"], g["loaded_vals"])                               # load keeps the newest `capacity` entries
    assert "Error encoding texts" in capsys.readouterr().out                                   # the reference prints and falls back; so does this


def test_one_table_one_gather_and_pickle_format():
    enc = FakeEncoder()
    buf = AdvancedLangEmbeddingBuffer(enc, 4)
    out = buf.get_goal_instruction_embeddings(["p", "q", "p", "r"])
    assert out.shape == (4, DIM) and torch.equal(out[0], out[2])
    assert buf._table.shape == (4, DIM) and buf.get_buffer_size() == 3                          # one allocation for the whole cache
    assert enc.calls == [["p", "q", "r"]] or enc.calls == [["p", "q", "p", "r"]]
    n_calls = len(enc.calls)
    again = buf.get_goal_instruction_embeddings(["r", "q"])
    assert len(enc.calls) == n_calls and torch.equal(again, torch.stack([out[3], out[1]]))      # pure hit: no encoder call
    path = os.path.join(tempfile.mkdtemp(), "b.pkl")
    buf.save_buffer(path)
    import pickle
    from collections import OrderedDict
    with open(path, "rb") as f:
        raw = pickle.load(f)
    assert isinstance(raw, OrderedDict) and list(raw.keys()) == ["p", "q", "r"] and all(v.shape == (DIM,) for v in raw.values())


@pytest.mark.gpu
def test_table_lives_on_the_encoder_device():
    enc = FakeEncoder()
    dev_enc = lambda texts: enc(texts).cuda()
    buf = AdvancedLangEmbeddingBuffer(dev_enc, 8)
    out = buf.get_goal_instruction_embeddings(["open the drawer", "push the blue block"])
    assert out.is_cuda and buf._table.is_cuda and torch.equal(out.cpu(), enc(["open the drawer", "push the blue block"]))
