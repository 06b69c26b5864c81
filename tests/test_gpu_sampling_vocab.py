"""-m gpu: the dynamic-decode kernels at the HEADLINE vocabulary (V = 100864, CodeFuse-13B's), replayed exactly through the oracle.

The engine tests replay the sampling / beam kernels at V = 512; the limits the kernels were tuned around live at the big vocabulary
(k_topk_stage1: 8 slices x 256 threads x up to 60 logits in registers, the radix select of a slice's k best, the sorted top-p walk over
128 candidates per slice with its full-row fallback, the optional-token bitmask, the repetition penalty's staging of a long history).
Here a one-layer H = 256 model with the 100864-row LM head (51 MB) generates from a 1500-token prompt; the GPU's own per-step logits
go through the oracle's DynamicDecodeLayer / OnlineBeamSearchLayer restatement (same counter-based uniforms, same (value desc, index
asc) order) and must pick the same tokens.  Reference: sampling_topk_kernels.cu:131-311, sampling_topp_kernels.cu,
sampling_penalty_kernels.cu:485-520, select_optional_last_tokens.cu:22-85, online_softmax_beamsearch_kernels.cu,
examples/pytorch/codefuse/codefuse_example.py:799 (top_k = 50 is the harness default)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import random_model

pytestmark = pytest.mark.gpu

V = 100864
CFG = dict(head_num=2, size_per_head=128, inter_size=1024, num_layer=1, vocab_size=V, rotary_dim=32, start_id=0, end_id=2)
S = 1500  # prompt tokens: the history the repetition penalty walks


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


@pytest.fixture(scope="module")
def model(gh):
    w = random_model(CFG, seed=4242, std=0.05)
    return w, gh.make_op(CFG, w)


def _prompts(B, seed):
    rng = np.random.RandomState(seed)
    # a history with repeats (the penalty hits a token once however often it occurred) and ids on both sides of 65536
    ids = rng.randint(3, V, size=(B, S)).astype(np.int32)
    ids[:, 100:400] = ids[:, 500:800]
    return ids


def _forward(op, ids_np, out, **kw):
    """ftcf_gptneox_forward through its argument block: every runtime argument incl. min_length, and the per-step logits."""
    import torch
    from fastertransformer4codefuse_amd import capi
    B = ids_np.shape[0]
    keep = []

    def host(v, dt):
        if v is None:
            return None, 0
        a = np.ascontiguousarray(np.asarray(v, dtype=dt).reshape(-1))
        keep.append(a)
        return a.ctypes.data, int(a.size)

    def dev(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.int32)).cuda()
        keep.append(t)
        return t

    ids = dev(ids_np)
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    out_ids = torch.zeros((B, 1, S + out), dtype=torch.int32, device="cuda")
    seq = torch.zeros((B, 1), dtype=torch.int32, device="cuda")
    cum = torch.zeros((B, 1), dtype=torch.float32, device="cuda")
    dbg = torch.zeros((out, B, V), dtype=torch.float32, device="cuda")
    fa = capi.ForwardArgs()
    fa.input_ids, fa.input_lengths = ids.data_ptr(), lens.data_ptr()
    fa.batch_size, fa.max_input_len, fa.output_len, fa.beam_width = B, S, out, 1
    fa.top_k, fa.n_top_k = host(kw.get("top_k"), np.int32)
    fa.top_p, fa.n_top_p = host(kw.get("top_p"), np.float32)
    fa.temperature, fa.n_temperature = host(kw.get("temperature"), np.float32)
    fa.repetition_penalty, fa.n_repetition_penalty = host(kw.get("repetition_penalty"), np.float32)
    fa.random_seed, fa.n_random_seed = host(kw.get("random_seed"), np.uint64)
    fa.min_length, fa.n_min_length = host(kw.get("min_length"), np.int32)
    if kw.get("stop_words") is not None:
        sw = dev(kw["stop_words"])
        fa.stop_words_list, fa.stop_words_len = sw.data_ptr(), int(kw["stop_words"].shape[2])
    if kw.get("optional_last_tokens") is not None:
        ol = dev(kw["optional_last_tokens"])
        fa.optional_last_tokens, fa.optional_last_tokens_count = ol.data_ptr(), int(kw["optional_last_tokens"].shape[1])
    fa.return_cum_log_probs = 1
    fa.output_ids, fa.sequence_lengths, fa.cum_log_probs = out_ids.data_ptr(), seq.data_ptr(), cum.data_ptr()
    fa.debug_logits = dbg.data_ptr()
    capi.check(capi.lib().ftcf_gptneox_forward(op._h, C.byref(fa)))
    torch.cuda.synchronize()
    return dict(output_ids=out_ids[:, 0].cpu().numpy(), sequence_lengths=seq[:, 0].cpu().numpy(), cum_log_probs=cum[:, 0].cpu().numpy(),
                logits=dbg.cpu().numpy(), steps=op.stats()["decode_steps"])


def _replay(r, ids, out, sp, end_id):
    """The GPU's logits of every step through orc_dynamic_decode; returns (positions where the tokens differ, the oracle's state)."""
    B = ids.shape[0]
    total = S + out
    step_ids = np.zeros((total, B), np.int32)
    step_ids[:S] = ids.T
    fin = np.zeros(B, np.uint8)
    seq = np.full(B, S - 1, np.int32)
    cum = np.zeros(B, np.float32)
    draws = np.zeros(B, np.uint64)
    lens = np.full(B, S, np.int32)
    bad = []
    steps = 0
    for t in range(out):
        lg = np.ascontiguousarray(r["logits"][t], dtype=np.float32).copy()
        orc.dynamic_decode(lg, S + t, S, lens, sp, end_id, step_ids, fin, seq, cum, draws)
        steps += 1
        got = r["output_ids"][:, S + t]
        for b in np.nonzero(step_ids[S + t] != got)[0]:
            bad.append((t, int(b), int(step_ids[S + t, b]), int(got[b])))
        if fin.all():
            break
    return bad, dict(fin=fin, seq=seq, cum=cum, steps=steps, ids=step_ids)


# (top_k, top_p, temperature, repetition_penalty): greedy, the harness default with a penalty, the widest top-k, the top-p layer on a
# peaked and on a flat distribution, top-k followed by top-p, small k with a tight p
SETTINGS = [(1, 0.0, 1.0, 1.0), (50, 0.0, 1.0, 1.2), (1024, 0.0, 0.7, 1.0), (0, 0.9, 1.0, 1.0), (0, 0.9, 0.35, 1.3), (50, 0.9, 1.3, 1.0),
            (1024, 0.5, 1.0, 1.1), (8, 0.3, 1.0, 1.0)]


@pytest.mark.parametrize("B", [1, 16])
def test_sampling_kernels_at_the_headline_vocabulary(gh, model, B):
    w, op = model
    out = 6
    ids = _prompts(B, 7 + B)
    rows = [SETTINGS[(b + (1 if B == 1 else 0)) % len(SETTINGS)] for b in range(B)]  # (one row: the harness default)
    kw = dict(top_k=[r[0] for r in rows], top_p=[r[1] for r in rows], temperature=[r[2] for r in rows],
              repetition_penalty=[r[3] for r in rows], random_seed=[1000 + 17 * b for b in range(B)])
    r = _forward(op, ids, out, **kw)
    bad, st = _replay(r, ids, out, orc.Sampling(B, **kw), CFG["end_id"])
    assert not bad, bad
    np.testing.assert_allclose(r["cum_log_probs"], st["cum"], rtol=1e-4, atol=2e-4)
    assert (r["output_ids"][:, S:S + out] >= 0).all() and (r["output_ids"][:, S:S + out] < V).all()
    # the big top-k rows really sampled beyond the head of the distribution somewhere (not a disguised arg max)
    if B == 16:
        lg = r["logits"]
        rank = [(lg[t, b] > lg[t, b, r["output_ids"][b, S + t]]).sum() for t in range(out) for b in (2, 10)]
        assert max(rank) >= 1, rank


def test_min_length_stop_words_and_optional_tokens_at_the_headline_vocabulary(gh, model):
    """Rows end on end_id / stop words at different steps; min_length masks end_id; the optional-token list (ids on both sides of
    65536) confines the first step -- all of it equal to the oracle's replay, the lengths and the loop count included."""
    w, op = model
    B, out = 4, 7
    ids = _prompts(B, 99)
    free = _forward(op, ids, out, top_k=[1] * B)
    g = free["output_ids"][:, S:]
    # end_id := what row 0 emits third when it runs free; stop words: row 1's 4th+5th tokens, row 2's 2nd token
    cfg2 = dict(CFG, end_id=int(g[0, 2]))
    from tests import gpu_helpers
    op2 = gpu_helpers.make_op(cfg2, w)
    stop = np.full((B, 2, 3), -1, np.int32)
    stop[:, 0, :] = 0
    stop[1, 0, :2] = g[1, 3:5]
    stop[1, 1, 0] = 2
    stop[2, 0, 0] = g[2, 1]
    stop[2, 1, 0] = 1
    for name, kw in {
            "end_id": dict(top_k=[1] * B),
            "min_length": dict(top_k=[1] * B, min_length=[5] * B),
            "stop_words": dict(top_k=[1] * B, stop_words=stop),
            "optional": dict(top_k=[1, 50, 1, 0], top_p=[0.0, 0.0, 0.0, 0.8], random_seed=[5, 6, 7, 8],
                             optional_last_tokens=np.array([[70001, 5, 100863, 31], [99999, 65536, 65535, 12], [3, 4, 5, 6],
                                                            [100000, 90000, 80000, 70000]], np.int32)),
    }.items():
        r = _forward(op2, ids, out, **kw)
        bad, st = _replay(r, ids, out, orc.Sampling(B, **kw), cfg2["end_id"])
        assert not bad, (name, bad)
        assert r["steps"] == st["steps"], (name, r["steps"], st["steps"])
        assert np.array_equal(r["sequence_lengths"], st["seq"] + 1) or np.array_equal(r["sequence_lengths"], st["seq"]), (
            name, r["sequence_lengths"], st["seq"])
        if name == "end_id":
            assert r["output_ids"][0, S + 2] == cfg2["end_id"] and r["sequence_lengths"][0] < S + out
        if name == "min_length":
            assert (r["output_ids"][0, S:S + 5] != cfg2["end_id"]).all()
        if name == "optional":
            allowed = kw["optional_last_tokens"]
            assert all(r["output_ids"][b, S] in allowed[b] for b in range(B)), r["output_ids"][:, S]


def test_beam_search_kernels_at_the_headline_vocabulary(gh, model):
    """beam_width 4 at V = 100864: every hypothesis, length and score equal to the oracle's replay of the GPU's logits."""
    from tests.test_gpu_beam import _replay as beam_replay
    w, op = model
    B, K, out = 2, 4, 5
    rng = np.random.RandomState(3)
    Sb = 64
    ids = rng.randint(3, V, size=(B, Sb)).astype(np.int32)
    lens = np.full(B, Sb, np.int32)
    kw = dict(temperature=[0.9, 1.1], repetition_penalty=[1.2, 1.0], beam_search_diversity_rate=[-0.1, 0.0], len_penalty=0.5)
    r = gh.run_op_beam(op, ids, lens, out, V, K, return_logits=True, **kw)
    okw = {k: v for k, v in kw.items() if k != "beam_search_diversity_rate"}
    bp = orc.BeamParams(B, diversity_rate=kw["beam_search_diversity_rate"], **okw)
    o_ids, o_len, o_cum = beam_replay(CFG, ids, lens, out, K, r["logits"], bp)
    assert r["output_ids"].tolist() == o_ids.tolist()
    assert r["sequence_lengths"].tolist() == o_len.tolist()
    np.testing.assert_allclose(r["cum_log_probs"], o_cum, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B", [1, 3])
def test_the_one_launch_top_k_step_equals_the_general_launches(gh, model, monkeypatch, B):
    """The harness default (top_k = 50, nothing else: codefuse_example.py:799-806) takes ONE launch behind the LM head (k_topk_decode);
    FTCF_TOPK_FUSED=0 sends the same request through k_decode_prep / k_topk_stage1 / k_sample / k_decode_finish.  Same tokens, lengths
    and loop count from both, scores to 1e-4 (the row's soft-max denominator is summed over 32 slices instead of 8), and both what the
    oracle picks from the GPU's logits -- with min_length holding end_id back and stop words ending rows."""
    w, op = model
    out = 8
    ids = _prompts(B, 311 + B)
    ks = [50, 8, 2][:B]
    free = _forward(op, ids, out, top_k=ks, random_seed=[77 + b for b in range(B)])
    g = free["output_ids"][:, S:]
    cfg2 = dict(CFG, end_id=int(g[0, 3]))  # row 0 draws it as its fourth token
    stop = np.full((B, 2, 2), -1, np.int32)
    stop[:, 0, :] = 0
    stop[B - 1, 0, :2] = g[B - 1, 4:6]
    stop[B - 1, 1, 0] = 2
    cases = {"plain": {}, "min_length": dict(min_length=[6] * B), "stop_words": dict(stop_words=stop)}
    for name, extra in cases.items():
        kw = dict(top_k=ks, random_seed=[77 + b for b in range(B)], **extra)
        got = {}
        for form in ("1", "0"):
            monkeypatch.setenv("FTCF_TOPK_FUSED", form)
            got[form] = _forward(gh.make_op(cfg2, w), ids, out, **kw)
        a, b = got["1"], got["0"]
        assert np.array_equal(a["output_ids"], b["output_ids"]), (name, a["output_ids"][:, S:], b["output_ids"][:, S:])
        assert np.array_equal(a["sequence_lengths"], b["sequence_lengths"]) and a["steps"] == b["steps"], name
        np.testing.assert_allclose(a["cum_log_probs"], b["cum_log_probs"], rtol=1e-4, atol=1e-4, err_msg=name)
        bad, st = _replay(a, ids, out, orc.Sampling(B, **kw), cfg2["end_id"])
        assert not bad, (name, bad)
        assert a["steps"] == st["steps"], (name, a["steps"], st["steps"])
        np.testing.assert_allclose(a["cum_log_probs"], st["cum"], rtol=1e-4, atol=2e-4, err_msg=name)
        if name == "plain":
            assert a["output_ids"][0, S + 3] == cfg2["end_id"] and a["sequence_lengths"][0] < S + out
