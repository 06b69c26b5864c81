"""-m gpu: beam search (beam_width > 1) through GptNeoXOp / the C ABI against the oracle's OnlineBeamSearchLayer
restatement and the HF beam-search goldens (tests/golden/tiny_gptneox_beam.npz)."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before the engine library: see tests/test_gpu_engine.py)

from oracle import oracle as orc
from tests.helpers import GOLDEN, load_tiny, quantize_layers, random_model, weight_list_to_layers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


@pytest.fixture(scope="module")
def tiny():
    cfg, w, _ = load_tiny()
    layers, glob = weight_list_to_layers(cfg, w)
    return cfg, w, layers, glob, np.load(os.path.join(GOLDEN, "tiny_gptneox_beam.npz"))


@pytest.mark.parametrize("name,K,n_new", [("a", 3, 8), ("b", 4, 6), ("c", 2, 10)])
def test_tiny_fp16_beam_search_matches_hf_golden_and_oracle(gh, tiny, name, K, n_new):
    cfg, w, layers, glob, g = tiny
    op = gh.make_op(cfg, w)
    ids, lens = g[f"ids_{name}"], g[f"lens_{name}"]
    r = gh.run_op_beam(op, ids, lens, n_new, cfg["vocab_size"], K)
    assert op.stats()["decode_path"] == 2  # beams read K/V through the cache indirection: general path
    for b in range(ids.shape[0]):
        n = int(lens[b])
        for k in range(K):
            assert r["output_ids"][b, k, n:n + n_new].tolist() == g[f"hf_beam_tokens_{name}"][b, k].tolist()
    o = orc.Model(dict(cfg, fp16=1), layers, glob).generate_beam(ids, lens, n_new, K)
    assert r["output_ids"].tolist() == o["output_ids"].tolist()
    assert r["sequence_lengths"].tolist() == o["sequence_lengths"].tolist()
    np.testing.assert_allclose(r["cum_log_probs"], o["cum_log_probs"], rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(r["cum_log_probs"], g[f"hf_beam_scores_{name}"], rtol=2e-2, atol=5e-2)


def _pick_end_id(g, name, b=0, t=3):
    """A token the best beam emits mid-way: used as end_id so that beams finish during the search."""
    return int(g[f"hf_beam_tokens_{name}"][b, 0, t])


@pytest.mark.parametrize("kw", [
    dict(),
    dict(temperature=0.7),
    dict(repetition_penalty=1.3),
    dict(beam_search_diversity_rate=-0.8),
    dict(len_penalty=0.7),
    dict(temperature=[1.2, 0.6], beam_search_diversity_rate=[-0.3, 0.0]),
    dict(repetition_penalty=[1.1, 1.5]),
])
def test_tiny_beams_finish_on_end_id_and_follow_runtime_args(gh, tiny, kw):
    cfg, w, layers, glob, g = tiny
    cfg = dict(cfg, end_id=_pick_end_id(g, "b"))
    op = gh.make_op(cfg, w)
    ids, lens = g["ids_b"], g["lens_b"]
    K, n_new = 4, 10
    r = gh.run_op_beam(op, ids, lens, n_new, cfg["vocab_size"], K, **kw)
    okw = {k: v for k, v in kw.items() if k != "beam_search_diversity_rate"}
    bp = orc.BeamParams(ids.shape[0], diversity_rate=kw.get("beam_search_diversity_rate"), **okw)
    o = orc.Model(dict(cfg, fp16=1), layers, glob).generate_beam(ids, lens, n_new, K, bp)
    assert (o["output_ids"][0, :, lens[0]:lens[0] + n_new] == cfg["end_id"]).any()  # the case really finishes beams
    # The device kernels are checked exactly in test_beam_search_kernels_are_exact_given_the_same_logits; end to end the
    # fp16 engine's logits differ from the oracle's emulation in the last bits, which may reorder two beams whose scores
    # are closer than that: the best beam must agree, the others up to such an event.
    assert r["output_ids"][:, 0].tolist() == o["output_ids"][:, 0].tolist()
    assert r["sequence_lengths"][:, 0].tolist() == o["sequence_lengths"][:, 0].tolist()
    np.testing.assert_allclose(r["cum_log_probs"][:, 0], o["cum_log_probs"][:, 0], rtol=2e-2, atol=3e-2)
    if r["output_ids"].tolist() != o["output_ids"].tolist():
        # (a candidate pruned at an intermediate step by a margin below the fp16 noise: the lower beams then follow
        # another hypothesis; not visible in the final scores)
        assert (r["output_ids"] == o["output_ids"]).mean() > 0.7
    else:
        assert r["sequence_lengths"].tolist() == o["sequence_lengths"].tolist()
        np.testing.assert_allclose(r["cum_log_probs"], o["cum_log_probs"], rtol=2e-2, atol=3e-2)


def _replay(cfg, ids, lens, n_new, K, gpu_logits, bp):
    """The GPU's own raw logits of every step pushed through the oracle's beam-search layer: what the device kernels must
    have produced from them (no model noise in the comparison)."""
    B, S = ids.shape
    total, BK = S + n_new, B * K
    out = np.zeros((total, BK), np.int32)
    par = np.zeros((total, BK), np.int32)
    for bb in range(BK):
        out[:S, bb] = ids[bb // K]
    fin = np.zeros(BK, np.uint8)
    seq = np.full(BK, S - 1, np.int32)
    cum = np.where(np.arange(BK) % K == 0, 0.0, -1e20).astype(np.float32)
    ind = [np.zeros((B, K, total), np.int32), np.zeros((B, K, total), np.int32)]
    tl = np.repeat(np.asarray(lens, np.int32), K)
    for t in range(n_new):
        lg = np.ascontiguousarray(gpu_logits[t], dtype=np.float32).copy()
        orc.beam_search_step(lg, K, S + t, S, tl, bp, cfg["end_id"], out, par, fin, seq, cum, ind[t % 2], ind[1 - t % 2])
        if fin.all():
            break
    o_ids, o_len = orc.gather_tree_beam(out, par, seq, tl, K, S, cfg["end_id"])
    return o_ids, o_len, cum.reshape(B, K)


@pytest.mark.parametrize("kw", [
    dict(),
    dict(temperature=[0.8, 0.6], repetition_penalty=[1.2, 1.5], beam_search_diversity_rate=[-0.2, 0.0], len_penalty=0.3),
    dict(temperature=[1.2, 0.6], repetition_penalty=[1.1, 1.5], beam_search_diversity_rate=[-0.3, 0.0]),
    dict(len_penalty=[1.0, 0.5], beam_search_diversity_rate=0.4),
])
def test_beam_search_kernels_are_exact_given_the_same_logits(gh, tiny, kw):
    cfg, w, layers, glob, g = tiny
    cfg = dict(cfg, end_id=_pick_end_id(g, "b"))
    op = gh.make_op(cfg, w)
    ids, lens = g["ids_b"], g["lens_b"]
    K, n_new = 4, 10
    stop = np.array([[[63, 376], [2, -1]], [[191, 219], [2, -1]]], dtype=np.int32)
    r = gh.run_op_beam(op, ids, lens, n_new, cfg["vocab_size"], K, return_logits=True, stop_words=stop, **kw)
    okw = {k: v for k, v in kw.items() if k != "beam_search_diversity_rate"}
    bp = orc.BeamParams(ids.shape[0], diversity_rate=kw.get("beam_search_diversity_rate"), stop_words=stop, **okw)
    o_ids, o_len, o_cum = _replay(cfg, ids, lens, n_new, K, r["logits"], bp)
    assert r["output_ids"].tolist() == o_ids.tolist()
    assert r["sequence_lengths"].tolist() == o_len.tolist()
    np.testing.assert_allclose(r["cum_log_probs"], o_cum, rtol=1e-5, atol=1e-4)


def test_tiny_beam_stop_words_optional_tokens_and_callback(gh, tiny):
    cfg, w, layers, glob, g = tiny
    op = gh.make_op(cfg, w)
    ids, lens = g["ids_a"], g["lens_a"]
    K, n_new = 3, 8
    best = g["hf_beam_tokens_a"][0, 0].tolist()
    stop = np.array([[[best[2], best[3]], [2, -1]]], dtype=np.int32)  # the best beam stops after its 4th token
    allowed = np.array([[best[0], int(g["hf_beam_tokens_a"][0, 1, 1]), 7, -1]], dtype=np.int32)
    events = []
    r = gh.run_op_beam(op, ids, lens, n_new, cfg["vocab_size"], K, stop_words=stop, optional_last_tokens=allowed,
                       callback=events.append)
    bp = orc.BeamParams(1, stop_words=stop, optional_last_tokens=allowed)
    o = orc.Model(dict(cfg, fp16=1), layers, glob).generate_beam(ids, lens, n_new, K, bp)
    assert r["output_ids"].tolist() == o["output_ids"].tolist()
    assert r["sequence_lengths"].tolist() == o["sequence_lengths"].tolist()
    first = r["output_ids"][0, :, lens[0]]
    assert set(first.tolist()) <= set(allowed[0, :3].tolist())
    assert events and len(events[0]["last_tokens"]) == 1 and len(events[0]["last_tokens"][0]) == K


MID = dict(head_num=8, size_per_head=128, inter_size=4096, num_layer=2, vocab_size=2048, rotary_dim=32, start_id=0,
           end_id=2)


@pytest.mark.parametrize("int8_mode", [0, 1])
@pytest.mark.parametrize("B,K", [(1, 2), (2, 4), (2, 10)])
def test_mid_model_beam_search_follows_oracle(gh, B, K, int8_mode):
    """H=1024/Dh=128 with ragged prompts: rows = B*K in {2, 8, 20} take the split-K GEMV, small-m and tiled MFMA GEMM
    forms of the general decode path.  fp16-level noise may flip a near-tie, after which histories differ: require
    the cumulative log-probs of all beams to agree and the large majority of tokens."""
    cfg = MID
    w = random_model(cfg, seed=31 + B + K + int8_mode, std=0.04)
    layers, glob = weight_list_to_layers(cfg, w)
    if int8_mode:
        layers = quantize_layers(layers)
    rng = np.random.RandomState(B * 7 + K)
    S, out = 29, 10
    lens = rng.randint(12, S + 1, size=B).astype(np.int32)
    lens[0] = S
    ids = np.full((B, S), cfg["end_id"], dtype=np.int32)
    for b in range(B):
        ids[b, :lens[b]] = rng.randint(3, cfg["vocab_size"], size=lens[b])
    op = gh.make_op(cfg, w, int8_mode=int8_mode)
    r = gh.run_op_beam(op, ids, lens, out, cfg["vocab_size"], K, return_logits=True)
    o = orc.Model(dict(cfg, fp16=1, int8_mode=int8_mode), layers, glob).generate_beam(ids, lens, out, K)
    np.testing.assert_allclose(r["cum_log_probs"], o["cum_log_probs"], rtol=2e-2, atol=6e-2)
    assert r["sequence_lengths"].tolist() == o["sequence_lengths"].tolist()
    p_ids, p_len, p_cum = _replay(cfg, ids, lens, out, K, r["logits"], orc.BeamParams(B))
    assert r["output_ids"].tolist() == p_ids.tolist()  # exact given the GPU's own logits
    np.testing.assert_allclose(r["cum_log_probs"], p_cum, rtol=1e-5, atol=1e-4)
    agree = (r["output_ids"] == o["output_ids"]).mean()
    assert agree > 0.9, agree
    for b in range(B):  # prompt copied into every beam, un-padded
        for k in range(K):
            assert r["output_ids"][b, k, :lens[b]].tolist() == ids[b, :lens[b]].tolist()


def test_beam_width_out_of_range_is_rejected(gh, tiny):
    cfg, w, layers, glob, g = tiny
    op = gh.make_op(cfg, w)
    with pytest.raises(RuntimeError):
        gh.run_op_beam(op, g["ids_a"], g["lens_a"], 4, cfg["vocab_size"], 65)


@pytest.mark.parametrize("B,K,out", [(1, 16, 6), (2, 33, 5), (1, 64, 4)])
def test_wide_beams_are_exact_given_the_same_logits(gh, tiny, B, K, out):
    """beam_width up to the online beam search limit (64): rows = B*K reach the tiled GEMM path; the K*K candidate pool of the
    batch kernel grows to 4096."""
    cfg, w, layers, glob, g = tiny
    op = gh.make_op(cfg, w)
    z = g["ids_a"][0]
    ids = np.stack([np.roll(z, i) for i in range(B)]).astype(np.int32)
    lens = np.full(B, ids.shape[1], np.int32)
    r = gh.run_op_beam(op, ids, lens, out, cfg["vocab_size"], K, return_logits=True, repetition_penalty=1.2,
                       beam_search_diversity_rate=-0.1)
    bp = orc.BeamParams(B, repetition_penalty=1.2, diversity_rate=-0.1)
    p_ids, p_len, p_cum = _replay(cfg, ids, lens, out, K, r["logits"], bp)
    assert r["output_ids"].tolist() == p_ids.tolist()
    assert r["sequence_lengths"].tolist() == p_len.tolist()
    np.testing.assert_allclose(r["cum_log_probs"], p_cum, rtol=1e-4, atol=1e-3)
