"""Pins the oracle's beam search (OnlineBeamSearchLayer restatement) against HF beam search on the tiny golden model
(tests/golden/tiny_gptneox_beam.npz, made by tests/golden/make_golden.py) and checks the layer's step semantics."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import GOLDEN, load_tiny, weight_list_to_layers

CASES = (("a", 3, 8), ("b", 4, 6), ("c", 2, 10))


@pytest.fixture(scope="module")
def tiny():
    cfg, w, _ = load_tiny()
    layers, glob = weight_list_to_layers(cfg, w)
    return cfg, layers, glob, np.load(os.path.join(GOLDEN, "tiny_gptneox_beam.npz"))


def _model(cfg, layers, glob, **kw):
    c = dict(cfg)
    c.update(kw)
    return orc.Model(c, layers, glob)


@pytest.mark.parametrize("name,K,n_new", CASES)
def test_fp32_beam_search_matches_hf(tiny, name, K, n_new):
    cfg, layers, glob, g = tiny
    m = _model(cfg, layers, glob, fp16=0)
    ids, lens = g[f"ids_{name}"], g[f"lens_{name}"]
    r = m.generate_beam(ids, lens, n_new, K)
    S = ids.shape[1]
    for b in range(ids.shape[0]):
        n = int(lens[b])
        for k in range(K):
            row = r["output_ids"][b, k]
            assert row[:n].tolist() == ids[b, :n].tolist()
            assert row[n:n + n_new].tolist() == g[f"hf_beam_tokens_{name}"][b, k].tolist()
            assert row[n + n_new:].tolist() == [cfg["end_id"]] * (S - n)
        np.testing.assert_allclose(r["cum_log_probs"][b], g[f"hf_beam_scores_{name}"][b], atol=2e-4)
    assert (r["sequence_lengths"] == S + n_new).all()


def test_beam_width_one_is_greedy(tiny):
    cfg, layers, glob, g = tiny
    m = _model(cfg, layers, glob, fp16=1)
    ids, lens = g["ids_b"], g["lens_b"]
    r1 = m.generate_beam(ids, lens, 6, 1)
    r0 = m.generate(ids, lens, 6)
    assert np.array_equal(r1["output_ids"][:, 0], r0["output_ids"])


def _state(B, K, V, total, s_max, seed=0):
    rng = np.random.RandomState(seed)
    return dict(logits=rng.randn(B * K, V).astype(np.float32), output_ids=np.zeros((total, B * K), np.int32),
                parent_ids=np.zeros((total, B * K), np.int32), finished=np.zeros(B * K, np.uint8),
                seq_len=np.full(B * K, 4, np.int32), cum=np.zeros(B * K, np.float32),
                src=np.zeros((B, K, s_max), np.int32), tgt=np.zeros((B, K, s_max), np.int32))


def _step(s, K, step, bp, end_id=2, max_input_len=4):
    B = s["logits"].shape[0] // K
    orc.beam_search_step(s["logits"], K, step, max_input_len, np.full(B, max_input_len, np.int32), bp, end_id,
                         s["output_ids"], s["parent_ids"], s["finished"], s["seq_len"], s["cum"], s["src"], s["tgt"])


def test_step_matches_numpy_topk_of_log_softmax():
    B, K, V, total = 2, 3, 50, 8
    s = _state(B, K, V, total, total, seed=1)
    s["cum"][:] = np.random.RandomState(2).randn(B * K).astype(np.float32)
    lg = s["logits"].astype(np.float64)
    cum0 = s["cum"].copy()
    _step(s, K, 4, orc.BeamParams(B))
    lp = lg - np.log(np.exp(lg - lg.max(1, keepdims=True)).sum(1, keepdims=True)) - lg.max(1, keepdims=True)
    for b in range(B):
        sc = (lp[b * K:(b + 1) * K] + cum0[b * K:(b + 1) * K, None]).reshape(-1)
        top = np.argsort(-sc, kind="stable")[:K]
        assert s["output_ids"][4, b * K:(b + 1) * K].tolist() == (top % V).tolist()
        assert s["parent_ids"][4, b * K:(b + 1) * K].tolist() == (top // V).tolist()
        np.testing.assert_allclose(s["cum"][b * K:(b + 1) * K], sc[top], atol=1e-5)
    assert (s["seq_len"] == 5).all() and not s["finished"].any()


def test_finished_beam_keeps_score_and_emits_end_id():
    B, K, V, total = 1, 2, 20, 8
    s = _state(B, K, V, total, total, seed=3)
    s["finished"][0] = 1
    s["cum"][:] = [-0.5, -30.0]
    s["seq_len"][:] = [5, 6]
    _step(s, K, 6, orc.BeamParams(B))
    # the finished beam offers only (end_id, cum + 0): it stays on top, its length does not grow
    assert s["output_ids"][6, 0] == 2 and s["parent_ids"][6, 0] == 0
    assert s["cum"][0] == np.float32(-0.5) and s["finished"][0] == 1 and s["seq_len"][0] == 5
    assert s["parent_ids"][6, 1] == 1 and s["seq_len"][1] == 7


def test_cache_indirection_follows_parents():
    B, K, V, total = 1, 3, 30, 8
    s = _state(B, K, V, total, total, seed=5)
    s["src"][0] = np.arange(K)[:, None] * np.ones(total, np.int32)  # beam k so far read its own rows
    s["cum"][:] = [-50.0, 0.0, -50.0]  # every survivor descends from beam 1
    _step(s, K, 5, orc.BeamParams(B))
    assert s["parent_ids"][5].tolist() == [1, 1, 1]
    for k in range(K):
        assert s["tgt"][0, k, :5].tolist() == [1] * 5 and s["tgt"][0, k, 5] == k


def test_diversity_rate_and_temperature_change_ranking():
    B, K, V, total = 1, 2, 10, 6
    s = _state(B, K, V, total, total, seed=7)
    s["logits"][:] = -20.0
    s["logits"][0, 3], s["logits"][0, 4] = 5.0, 4.9  # beam 0's two best beat everything of beam 1 ...
    s["logits"][1, 7] = 4.0
    s["cum"][:] = [0.0, -5.0]
    a = {k: v.copy() for k, v in s.items()}
    _step(a, K, 4, orc.BeamParams(B))
    assert a["parent_ids"][4].tolist() == [0, 0] and a["output_ids"][4].tolist() == [3, 4]
    b = {k: v.copy() for k, v in s.items()}
    _step(b, K, 4, orc.BeamParams(B, diversity_rate=-10.0))  # ... until the second candidate of a row is charged
    assert b["parent_ids"][4].tolist() == [0, 1] and b["output_ids"][4].tolist() == [3, 7]


def test_fp16_emulation_beam_search_runs_and_orders_scores(tiny):
    cfg, layers, glob, g = tiny
    m = _model(cfg, layers, glob, fp16=1)
    r = m.generate_beam(g["ids_b"], g["lens_b"], 6, 4)
    assert (np.diff(r["cum_log_probs"], axis=1) <= 0).all()
    assert r["output_ids"].shape == (2, 4, 18)
