"""-m gpu: the fp32 instantiation of the engine (FTGptNeoX<float>, th_op/gptneox/GptNeoXOp.cc:56-70) -- BASELINE config 1, the
tiny fp32 model the reference's own CPU-side checks use -- through the product API.  With no half rounding anywhere the
engine must reproduce HuggingFace's fp32 forward to float accuracy: the bound is the one the fp32 oracle is pinned with
(tests/test_oracle_golden.py), 2e-4 of max|logit|, where the fp16 engine needs 4e-3."""
import os
import threading

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import load_tiny, shard_weights, weight_list_to_layers

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP32_FRAC = 2e-4


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


@pytest.fixture(scope="module")
def tiny():
    cfg, w, z = load_tiny()
    layers, glob = weight_list_to_layers(cfg, w)
    return cfg, w, layers, glob, z


def _close(a, b, frac=FP32_FRAC):
    err = np.abs(a - b).max() / np.abs(b).max()
    assert err <= frac, err


def test_tiny_fp32_greedy_reproduces_hf_fp32(gh, tiny):
    cfg, w, layers, glob, z = tiny
    op = gh.make_op(cfg, w, dtype=torch.float32)
    r = gh.run_op(op, z["prompt"][None, :], [16], 8, cfg["vocab_size"], top_k=1)
    assert op.stats()["decode_path"] == 2
    assert r["output_ids"][0, 16:].tolist() == z["hf_tokens"].tolist()
    _close(r["logits"][:, 0, :], z["hf_logits"])  # HF's own fp32 logits (tests/golden/make_golden.py)
    o = orc.Model(dict(cfg, fp16=0), layers, glob).generate(z["prompt"][None, :], [16], 8, return_logits=True)
    assert r["output_ids"].tolist() == o["output_ids"].tolist()
    _close(r["logits"], o["logits"], 5e-5)  # the fp32 oracle: same arithmetic up to summation order
    assert r["sequence_lengths"].tolist() == o["sequence_lengths"].tolist()
    np.testing.assert_allclose(r["cum_log_probs"], o["cum_log_probs"], rtol=1e-4, atol=1e-4)
    # same request again on the same engine: bit-identical
    r2 = gh.run_op(op, z["prompt"][None, :], [16], 8, cfg["vocab_size"], top_k=1)
    assert np.array_equal(r["logits"], r2["logits"])


def test_tiny_fp32_ragged_batch_and_single_token_prompt(gh, tiny):
    cfg, w, layers, glob, z = tiny
    op = gh.make_op(cfg, w, dtype=torch.float32)
    end_id = cfg["end_id"]
    ids = np.full((2, 16), end_id, dtype=np.int32)
    ids[0] = z["prompt"]
    ids[1, :11] = z["prompt_b"]
    r = gh.run_op(op, ids, [16, 11], 8, cfg["vocab_size"], top_k=1)
    assert r["output_ids"][0, :24].tolist() == z["prompt"].tolist() + z["hf_tokens"].tolist()
    assert r["output_ids"][1, :19].tolist() == z["prompt_b"].tolist() + z["hf_tokens_b"].tolist()
    assert r["output_ids"][1, 19:].tolist() == [end_id] * 5
    assert r["sequence_lengths"].tolist() == [24, 24]
    o = orc.Model(dict(cfg, fp16=0), layers, glob).generate(ids, [16, 11], 8, return_logits=True)
    _close(r["logits"], o["logits"], 5e-5)
    r1 = gh.run_op(op, z["prompt_1"][None, :], [1], 6, cfg["vocab_size"], top_k=1)
    assert r1["output_ids"][0, 1:].tolist() == z["hf_tokens_1"].tolist()


def test_tiny_fp32_sequential_residual_follows_hf(gh, tiny):
    from tests.test_oracle_golden import _sequential_weights
    cfg, w, _, _, z = tiny
    w2, s = _sequential_weights(cfg, w)
    layers, glob = weight_list_to_layers(cfg, w2)
    op = gh.make_op(cfg, w2, use_gptj_residual=False, dtype=torch.float32)
    ids = np.stack([z["prompt"], z["prompt"][::-1]]).astype(np.int32)
    r = gh.run_op(op, ids, [16, 16], 8, cfg["vocab_size"], top_k=1)
    assert r["output_ids"][0, 16:].tolist() == s["hf_tokens"].tolist()
    o = orc.Model(dict(cfg, fp16=0, use_gptj_residual=0), layers, glob).generate(ids, [16, 16], 8, return_logits=True)
    assert r["output_ids"].tolist() == o["output_ids"].tolist()
    _close(r["logits"], o["logits"], 5e-5)


@pytest.mark.parametrize("name,K,n_new", [("a", 3, 8), ("b", 4, 6)])
def test_tiny_fp32_beam_search_matches_hf(gh, tiny, name, K, n_new):
    cfg, w, layers, glob, _ = tiny
    g = np.load(os.path.join(GOLDEN, "tiny_gptneox_beam.npz"))
    op = gh.make_op(cfg, w, dtype=torch.float32)
    ids, lens = g[f"ids_{name}"], g[f"lens_{name}"]
    r = gh.run_op_beam(op, ids, lens, n_new, cfg["vocab_size"], K)
    for b in range(ids.shape[0]):
        n = int(lens[b])
        for k in range(K):
            assert r["output_ids"][b, k, n:n + n_new].tolist() == g[f"hf_beam_tokens_{name}"][b, k].tolist()
    np.testing.assert_allclose(r["cum_log_probs"], g[f"hf_beam_scores_{name}"], rtol=2e-3, atol=2e-3)
    o = orc.Model(dict(cfg, fp16=0), layers, glob).generate_beam(ids, lens, n_new, K)
    assert r["output_ids"].tolist() == o["output_ids"].tolist()
    np.testing.assert_allclose(r["cum_log_probs"], o["cum_log_probs"], rtol=1e-4, atol=1e-4)


def test_tiny_fp32_sampling_arguments_run_and_agree_with_the_oracle_replay(gh, tiny):
    """Dynamic decode is shared with the fp16 engine (it works on fp32 logits either way): temperature / repetition penalty /
    min-length through the fp32 engine, greedy so that the tokens can be compared with the oracle's."""
    cfg, w, layers, glob, z = tiny
    op = gh.make_op(cfg, w, dtype=torch.float32)
    kw = dict(temperature=0.7, repetition_penalty=1.3)
    r = gh.run_op(op, z["prompt"][None, :], [16], 8, cfg["vocab_size"], top_k=1, **kw)
    sp = orc.Sampling(1, top_k=1, temperature=0.7, repetition_penalty=1.3)
    o = orc.Model(dict(cfg, fp16=0), layers, glob).generate(z["prompt"][None, :], [16], 8, sp)
    assert r["output_ids"].tolist() == o["output_ids"].tolist()


def test_fp32_tensor_parallel_local_group_matches_tp1(gh, tiny):
    """tensor_para_size = 2 of the fp32 engine (ranks = engine instances on this GPU): fp32 all-reduce, x / TP
    residual, vocabulary split of the LM head."""
    from fastertransformer4codefuse_amd.gptneox_op import LocalTensorParallelGroup
    cfg, w, layers, glob, z = tiny
    ids = np.stack([z["prompt"], z["prompt"][::-1]]).astype(np.int32)
    op1 = gh.make_op(cfg, w, dtype=torch.float32)
    r1 = gh.run_op(op1, ids, [16, 16], 6, cfg["vocab_size"], top_k=1)
    group = LocalTensorParallelGroup()
    res, err = [None] * 2, []

    def worker(r):
        try:
            op = gh.make_op(cfg, shard_weights(cfg, w, 2, r), tp=2, rank=r, comm=group, dtype=torch.float32)
            res[r] = gh.run_op(op, ids, [16, 16], 6, cfg["vocab_size"], top_k=1)
        except BaseException as e:  # noqa: BLE001
            err.append((r, repr(e)))

    ths = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not err, err
    assert res[0] is not None and res[1] is not None
    assert res[0]["output_ids"].tolist() == res[1]["output_ids"].tolist() == r1["output_ids"].tolist()
    _close(res[0]["logits"], r1["logits"], 5e-5)


def test_fp32_refuses_int8_mode(gh, tiny):
    cfg, w, _, _, _ = tiny
    with pytest.raises(RuntimeError):
        gh.make_op(cfg, w, int8_mode=1, dtype=torch.float32)
