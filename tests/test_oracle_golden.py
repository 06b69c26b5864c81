"""Pins the CPU oracle against the golden vectors (HF GPTNeoXForCausalLM + the reference's own loader)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import load_tiny, quantize_layers, weight_list_to_layers


@pytest.fixture(scope="module")
def tiny():
    cfg, w, z = load_tiny()
    layers, glob = weight_list_to_layers(cfg, w)
    return cfg, layers, glob, z


def _model(cfg, layers, glob, **kw):
    c = dict(cfg)
    c.update(kw)
    return orc.Model(c, layers, glob)


def test_fp32_greedy_matches_hf_tokens_and_logits(tiny):
    cfg, layers, glob, z = tiny
    m = _model(cfg, layers, glob, fp16=0)
    prompt = z["prompt"][None, :]
    r = m.generate(prompt, [prompt.shape[1]], 8, return_logits=True)
    assert r["output_ids"][0, 16:].tolist() == z["hf_tokens"].tolist()
    assert r["output_ids"][0, :16].tolist() == z["prompt"].tolist()
    np.testing.assert_allclose(r["logits"][:, 0, :], z["hf_logits"], atol=2e-4, rtol=1e-4)
    assert r["sequence_lengths"].tolist() == [24]


def test_fp32_ragged_batch_matches_unpadded_hf_rows(tiny):
    cfg, layers, glob, z = tiny
    m = _model(cfg, layers, glob, fp16=0)
    end_id = cfg["end_id"]
    pa, pb = z["prompt"], z["prompt_b"]
    ids = np.full((2, 16), end_id, dtype=np.int32)
    ids[0, :16] = pa
    ids[1, :11] = pb
    r = m.generate(ids, [16, 11], 8, return_logits=True)
    # output layout (gatherTree): [input without pad gap | generated | end_id fill]
    assert r["output_ids"][0, :24].tolist() == pa.tolist() + z["hf_tokens"].tolist()
    assert r["output_ids"][1, :19].tolist() == pb.tolist() + z["hf_tokens_b"].tolist()
    assert r["output_ids"][1, 19:].tolist() == [end_id] * 5
    np.testing.assert_allclose(r["logits"][:, 1, :], z["hf_logits_b"], atol=2e-4, rtol=1e-4)
    assert r["sequence_lengths"].tolist() == [24, 24]  # S_max_in + n_generated


def test_fp32_single_token_prompt_skips_prefill(tiny):
    cfg, layers, glob, z = tiny
    m = _model(cfg, layers, glob, fp16=0)
    r = m.generate(z["prompt_1"][None, :], [1], 6, return_logits=True)
    assert r["output_ids"][0, 1:].tolist() == z["hf_tokens_1"].tolist()
    np.testing.assert_allclose(r["logits"][:, 0, :], z["hf_logits_1"], atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("which", ["", "_b", "_1"])
def test_fp16_emulation_stays_close_to_fp32(tiny, which):
    """The oracle's fp16 mode (binary16 at every point the reference stores `half`) against HF's fp32 logits, all three
    golden prompts.  Measured 0.9e-3 .. 1.5e-3 of max|logit| per step (the half rounding of activations and of the
    KV cache); the bound is 2x that -- a rounding point restated wrongly (e.g. LayerNorm output kept in fp32, or the
    attention probabilities rounded to half) moves the error outside it or flips tokens."""
    cfg, layers, glob, z = tiny
    m = _model(cfg, layers, glob, fp16=1)
    prompt = z["prompt" + which][None, :]
    n = prompt.shape[1]
    ref = z["hf_logits" + which]
    r = m.generate(prompt, [n], ref.shape[0], return_logits=True)
    err = np.abs(r["logits"][:, 0, :] - ref).max(axis=1) / np.abs(ref).max()
    assert err.max() < 3e-3, err
    assert err.min() > 1e-4, "suspiciously exact: is the fp16 mode on?"
    assert r["output_ids"][0, n:].tolist() == z["hf_tokens" + which].tolist()


@pytest.mark.parametrize("which", ["", "_b", "_1"])
def test_int8_weight_only_stays_close(tiny, which):
    """Weight-only int8 (per-column symmetric scales) against HF's fp32 logits: the quantisation error itself, measured
    1.0e-2 .. 2.0e-2 of max|logit| on this model; bound 3e-2, tokens equal on all three golden prompts."""
    cfg, layers, glob, z = tiny
    m = _model(cfg, quantize_layers(layers), glob, fp16=1, int8_mode=1)
    prompt = z["prompt" + which][None, :]
    n = prompt.shape[1]
    ref = z["hf_logits" + which]
    r = m.generate(prompt, [n], ref.shape[0], return_logits=True)
    rel = np.abs(r["logits"][:, 0, :] - ref).max(axis=1) / np.abs(ref).max()
    assert rel.max() < 3e-2, rel
    assert r["output_ids"][0, n:].tolist() == z["hf_tokens" + which].tolist()


def test_end_id_finishes_row_and_fills(tiny):
    cfg, layers, glob, z = tiny
    c = dict(cfg)
    c["end_id"] = int(z["hf_tokens"][2])  # third generated token becomes EOS
    m = orc.Model(dict(c, fp16=0), layers, glob)
    r = m.generate(z["prompt"][None, :], [16], 8)
    out = r["output_ids"][0]
    assert out[16:19].tolist() == z["hf_tokens"][:3].tolist()
    assert out[19:].tolist() == [c["end_id"]] * 5
    assert r["steps"] == 3
    assert r["sequence_lengths"].tolist() == [19]


def _sequential_weights(cfg, w):
    """Weight list of the use_gptj_residual = 0 form of the tiny model: attention / FFN output biases kept apart."""
    import os
    from tests.helpers import GOLDEN
    s = np.load(os.path.join(GOLDEN, "tiny_gptneox_seq.npz"))
    L = cfg["num_layer"]
    w2 = list(w)
    for l in range(L):
        w2[5 * L + l] = s["out_b"][l].astype(np.float32)
        w2[9 * L + l] = s["ffn2_b"][l].astype(np.float32)
    return w2, s


def test_fp32_sequential_residual_matches_hf():
    cfg, w, z = load_tiny()
    w2, s = _sequential_weights(cfg, w)
    layers, glob = weight_list_to_layers(cfg, w2)
    m = orc.Model(dict(cfg, fp16=0, use_gptj_residual=0), layers, glob)
    prompt = z["prompt"][None, :]
    r = m.generate(prompt, [16], 8, return_logits=True)
    assert r["output_ids"][0, 16:].tolist() == s["hf_tokens"].tolist()
    np.testing.assert_allclose(r["logits"][:, 0, :], s["hf_logits"], atol=2e-4, rtol=1e-4)
