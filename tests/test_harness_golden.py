"""CPU: the host-side harness / converter counterparts reproduce the I/O captured from the reference's Python
(tests/golden/harness_io.json, tiny_gptneox_fp32.npz, tiny_gptneox_tp2.json -- see tests/golden/make_golden.py)."""
import contextlib
import hashlib
import importlib.util
import io
import json
import os

import numpy as np
import pytest
import torch

from fastertransformer4codefuse_amd import convert, harness
from tests.helpers import GOLDEN, load_tiny


def _load_make_golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def io_golden():
    with open(os.path.join(GOLDEN, "harness_io.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tok():
    return _load_make_golden().FakeTok()


def test_to_word_list_format(io_golden, tok):
    for case in io_golden["to_word_list_format"]:
        out = harness.to_word_list_format(case["in"], tok)
        assert out.dtype == torch.int32
        assert out.numpy().tolist() == case["out"]


def test_trie_suggestions(io_golden, tok):
    trie = harness.Trie(tok.get_vocab())
    for case in io_golden["trie"]:
        res = []
        code = trie.printAutoSuggestions(case["key"], res)
        assert code == case["code"]
        assert sorted(i for _, i in res) == case["ids"]


def test_is_garbage(io_golden):
    for case in io_golden["is_garbage"]:
        assert harness.is_garbage(case["cp"]) == case["out"], case


def test_token_stream_convertor(io_golden, tok):
    for case in io_golden["stream"]:
        conv = harness.token_stream_2_str_stream_convertor(2, tok, 0)
        chunks = []
        for t in case["tokens"]:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                conv.append_token(t)
            chunks.append(buf.getvalue())
        assert chunks == case["chunks"]


def test_get_data_package(io_golden):
    for key in ("get_data_package", "get_data_package2"):
        case = io_golden[key]
        assert harness.get_data_package(case["in"], case["seed"]) == case["out"]


@pytest.fixture(scope="module")
def hf_model():
    pytest.importorskip("transformers")
    return _load_make_golden().build_hf()


def _load(dir_, cfg, tp, rank):
    nh = cfg.num_attention_heads
    w = harness.GptNeoXWeights(nh, cfg.hidden_size // nh, cfg.num_hidden_layers, cfg.vocab_size, 1024, tp, 1, True,
                               int8_mode=0, inference_data_type="fp32", weights_data_type=np.float32,
                               inter_size=cfg.intermediate_size)
    assert w.load(dir_, tensor_para_rank=rank, pipeline_para_rank=0)
    return [t.numpy() for t in w.w]


def test_converter_and_loader_reproduce_reference_tensors(hf_model, tmp_path):
    cfg, m = hf_model
    d1 = str(tmp_path / "1-gpu")
    convert.convert_model(m, d1, 1, "fp32", "tiny")
    _, gold, _ = load_tiny()
    got = _load(d1, cfg, 1, 0)
    assert len(got) == len(gold)
    for i, (a, b) in enumerate(zip(got, gold)):
        assert a.size == b.size, i
        np.testing.assert_array_equal(a.reshape(-1), b.reshape(-1), err_msg=str(i))
    with open(os.path.join(GOLDEN, "tiny_gptneox_tp2.json")) as f:
        tp2 = json.load(f)
    assert sorted(f for f in os.listdir(d1) if f != "config.ini") == tp2["files_tp1"]
    # config.ini is written even though transformers 5.x has no `rotary_pct` (the reference skips it there)
    from configparser import ConfigParser
    c = ConfigParser()
    c.read(os.path.join(d1, "config.ini"))
    assert c["gptneox"]["rotary_embedding"] == "16" and c["gptneox"]["use_gptj_residual"] == "1"
    d2 = str(tmp_path / "2-gpu")
    convert.convert_model(m, d2, 2, "fp32", "tiny")
    assert sorted(f for f in os.listdir(d2) if f != "config.ini") == tp2["files_tp2"]
    for r in range(2):
        ws = _load(d2, cfg, 2, r)
        for i, (a, g) in enumerate(zip(ws, tp2[f"rank{r}"])):
            assert list(a.shape) == g["shape"], (r, i)
            sha = hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()
            assert sha == g["sha256"], (r, i)


def test_quant_and_save_round_trip(hf_model, tmp_path):
    cfg, m = hf_model
    d1 = str(tmp_path / "ck")
    convert.convert_model(m, d1, 1, "fp16", "tiny")
    dq = str(tmp_path / "ckq")
    convert.quant_and_save(d1, dq, 1)
    nh = cfg.num_attention_heads
    kw = dict(inference_data_type="fp16", weights_data_type=np.float16, inter_size=cfg.intermediate_size)
    a = harness.GptNeoXWeights(nh, cfg.hidden_size // nh, cfg.num_hidden_layers, cfg.vocab_size, 1024, 1, 1, True,
                               int8_mode=1, **kw)
    assert a.load(d1, 0, 0)  # quantise at load
    b = harness.GptNeoXWeights(nh, cfg.hidden_size // nh, cfg.num_hidden_layers, cfg.vocab_size, 1024, 1, 1, True,
                               int8_mode=1, enable_int8_weights=True, **kw)
    assert b.load(dq, 0, 0)  # pre-quantised files
    for x, y in zip(a.int8_w, b.int8_w):
        assert torch.equal(x.reshape(-1), y.reshape(-1))
    for x, y in zip(a.scale, b.scale):
        assert torch.equal(x, y)
    assert all(t.numel() == 0 for g in (2, 4, 6, 8) for t in a.w[g * 2:(g + 1) * 2])


def _saw(model):
    return json.loads(json.dumps(model.saw))


def test_generate_marshals_and_post_processes_like_the_reference(io_golden, tok):
    """generate() around a recording stand-in for the model (make_golden.FakeGpt): the tensors the model is called with --
    padded prompts, per-row sampling tensors and their dtypes, the stop-word and optional-last-token tensors -- and the texts /
    lengths / scores cut out of its hypotheses equal what the reference's generate() did with the same stand-in."""
    mg = _load_make_golden()
    trie = harness.Trie(tok.get_vocab())
    for (kwargs, hyps), want in zip(mg.generate_cases(tok), io_golden["generate"]):
        gpt = mg.FakeGpt(hyps)
        texts, lengths, scores, latency = harness.generate(gpt, tok, trie=trie, **kwargs)
        assert _saw(gpt) == want["model_saw"]
        assert texts == want["texts"] and lengths == want["lengths"]
        assert scores == want["cum_log_probs"] and latency >= 0.0


def test_predict_streams_and_answers_like_the_reference(io_golden, tok):
    mg = _load_make_golden()
    req, hyps = mg.predict_case(tok)
    handler = harness.CodeFuseHandler.__new__(harness.CodeFuseHandler)
    handler.local_rank, handler.world_size, handler.tokenizer = 0, 1, tok
    handler.trie, handler.model = harness.Trie(tok.get_vocab()), mg.FakeGpt(hyps)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        code, msg, out = handler.predict(req, "trace-0")
    want = io_golden["predict"]
    body = json.loads(out["res"])
    assert body.pop("latency") >= 0.0
    assert (code, msg, body) == (want["code"], want["message"], want["response"])
    assert buf.getvalue() == want["printed"]
    assert _saw(handler.model) == want["model_saw"]
    # a failing request is answered, not raised
    code, msg, out = handler.predict({"prompts": [{"prompt": 3}], "out_seq_length": 1}, "trace-1")
    assert code == 1 and out == {"res": ""} and "Traceback" in msg
