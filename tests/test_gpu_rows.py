"""-m gpu: the persistent decode layers for 3..16 rows (csrc/rows_device.hip.h, `decode_path` 3) against the oracle, against the
general path (per-GEMM launches, FTCF_ROWS=0) and against themselves (bit-identical repeats).  Reference:
models/gptneox/GptNeoXDecoder.cc:245-384 (one layer loop for any batch size)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import load_tiny, quantize_layers, random_model, weight_list_to_layers

pytestmark = pytest.mark.gpu

LOGIT_FRAC = 5e-3  # the engine tests' end-to-end bound (tests/test_gpu_engine.py)

MID = dict(head_num=8, size_per_head=128, inter_size=4096, num_layer=2, vocab_size=2048, rotary_dim=32, start_id=0, end_id=2)
SMALL64 = dict(head_num=6, size_per_head=64, inter_size=1536, num_layer=3, vocab_size=512, rotary_dim=16, start_id=0, end_id=2)


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


def _batch(cfg, B, S, seed, ragged=True):
    rng = np.random.RandomState(seed)
    lens = (rng.randint(max(1, S // 2), S + 1, size=B) if ragged else np.full(B, S)).astype(np.int32)
    lens[0] = S
    ids = np.full((B, S), cfg["end_id"], dtype=np.int32)
    for b in range(B):
        ids[b, :lens[b]] = rng.randint(3, cfg["vocab_size"], size=lens[b])
    return ids, lens


def _follows(r, o, lens, out, frac=LOGIT_FRAC):
    checked = 0
    for b in range(len(lens)):
        gen_r = r["output_ids"][b, lens[b]:lens[b] + out]
        gen_o = o["output_ids"][b, lens[b]:lens[b] + out]
        for t in range(out):
            ref = o["logits"][t, b]
            scale = np.abs(ref).max()
            assert np.abs(r["logits"][t, b] - ref).max() <= frac * scale, (b, t, np.abs(r["logits"][t, b] - ref).max() / scale)
            checked += 1
            if gen_r[t] != gen_o[t]:  # a near-tie may flip one arg max: the histories part there
                top2 = np.sort(ref)[-2:]
                assert top2[1] - top2[0] < 2 * frac * scale, "token flip without a near tie"
                break
    assert checked >= 2 * len(lens)


@pytest.mark.parametrize("int8_mode", [0, 1])
@pytest.mark.parametrize("cfg_name,B", [("mid", 3), ("mid", 5), ("mid", 16), ("small64", 4), ("small64", 11)])
def test_rows_kernel_follows_the_oracle(gh, cfg_name, B, int8_mode):
    """Ragged batches of 3..16 rows, size_per_head 128 and 64, fp16 and int8 weights: logits of every step within 5e-3 of the
    oracle's range, tokens equal while the histories agree."""
    cfg = MID if cfg_name == "mid" else SMALL64
    w = random_model(cfg, seed=7 * B + int8_mode, std=0.04)
    layers, glob = weight_list_to_layers(cfg, w)
    if int8_mode:
        layers = quantize_layers(layers)
    ids, lens = _batch(cfg, B, 29, B)
    out = 10
    op = gh.make_op(cfg, w, int8_mode=int8_mode)
    r = gh.run_op(op, ids, lens, out, cfg["vocab_size"], top_k=1)
    assert op.stats()["decode_path"] == 3  # the rows kernel really ran
    o = orc.Model(dict(cfg, fp16=1, int8_mode=int8_mode), layers, glob).generate(ids, lens, out, return_logits=True)
    _follows(r, o, lens, out)


@pytest.mark.parametrize("int8_mode", [0, 1])
def test_rows_kernel_against_the_general_path_and_itself(gh, monkeypatch, int8_mode):
    """The same request on the rows kernel twice (bit-identical logits: every sum has a fixed order) and on the general path
    (FTCF_ROWS=0: per-GEMM launches; only the accumulation order differs)."""
    cfg = MID
    B, out = 9, 8
    w = random_model(cfg, seed=77 + int8_mode, std=0.04)
    ids, lens = _batch(cfg, B, 33, 5)
    op = gh.make_op(cfg, w, int8_mode=int8_mode)
    a = gh.run_op(op, ids, lens, out, cfg["vocab_size"], top_k=1)
    assert op.stats()["decode_path"] == 3
    b = gh.run_op(op, ids, lens, out, cfg["vocab_size"], top_k=1)
    assert np.array_equal(a["logits"], b["logits"]) and np.array_equal(a["output_ids"], b["output_ids"])
    monkeypatch.setenv("FTCF_ROWS", "0")
    op0 = gh.make_op(cfg, w, int8_mode=int8_mode)
    g = gh.run_op(op0, ids, lens, out, cfg["vocab_size"], top_k=1)
    assert op0.stats()["decode_path"] == 2
    scale = np.abs(g["logits"]).max()
    # first decode step: same prompt phase, one layer stack of different summation order
    assert np.abs(a["logits"][1] - g["logits"][1]).max() <= 3e-3 * scale
    same = (a["output_ids"] == g["output_ids"]).mean()
    assert same >= 0.9, same


def test_rows_kernel_with_sampling_and_finished_rows(gh):
    """Rows that finish early (end_id drawn) stay finished while the others go on; sampled rows are reproducible."""
    cfg, w, z = load_tiny()
    layers, glob = weight_list_to_layers(cfg, w)
    B = 6
    ids = np.stack([np.roll(z["prompt"], i) for i in range(B)]).astype(np.int32)
    kw = dict(top_k=[1, 5, 1, 0, 3, 1], top_p=[0.0, 0.0, 0.0, 0.8, 0.5, 0.0], temperature=[1.0, 0.8, 1.0, 1.1, 1.0, 1.0],
              random_seed=[1, 2, 3, 4, 5, 6])
    op = gh.make_op(cfg, w)
    r1 = gh.run_op(op, ids, [16] * B, 12, cfg["vocab_size"], **kw)
    assert op.stats()["decode_path"] == 3
    r2 = gh.run_op(op, ids, [16] * B, 12, cfg["vocab_size"], **kw)
    assert np.array_equal(r1["output_ids"], r2["output_ids"])
    o = orc.Model(dict(cfg, fp16=1), layers, glob).generate(ids, [16] * B, 12, sampling=orc.Sampling(B, **kw), return_logits=True)
    greedy = [0, 2, 5]
    for b in greedy:
        assert r1["output_ids"][b].tolist() == o["output_ids"][b].tolist()


def test_rows_kernel_with_long_ragged_contexts_on_few_workgroups(gh, monkeypatch):
    """Sixteen rows of up to 300 cached keys on 52 workgroups (FTCF_ROWS_NB): two whole (row, head) pairs + a third on 24 of them, 19
    key blocks per pair -- the K/V ring's steady rotations across a workgroup's pairs, padded (ragged) prompts and the padding-key
    mask, out-proj / FFN2 K pieces of an odd size -- against the oracle, and bit-identical when repeated."""
    cfg = MID
    B, S, out = 16, 300, 4
    monkeypatch.setenv("FTCF_ROWS_NB", "52")
    w = random_model(cfg, seed=123, std=0.04)
    layers, glob = weight_list_to_layers(cfg, w)
    layers = quantize_layers(layers)
    ids, lens = _batch(cfg, B, S, 41)
    op = gh.make_op(cfg, w, int8_mode=1)
    r = gh.run_op(op, ids, lens, out, cfg["vocab_size"], top_k=1)
    assert op.stats()["decode_path"] == 3
    r2 = gh.run_op(op, ids, lens, out, cfg["vocab_size"], top_k=1)
    assert np.array_equal(r["logits"], r2["logits"]) and np.array_equal(r["output_ids"], r2["output_ids"])
    o = orc.Model(dict(cfg, fp16=1, int8_mode=1), layers, glob).generate(ids, lens, out, return_logits=True)
    _follows(r, o, lens, out)
