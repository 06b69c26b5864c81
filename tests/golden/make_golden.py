#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ -- run ONLY in the build container.

Imports (never copies) the reference's Python from /root/reference and HF `GPTNeoXForCausalLM` to produce
input/output vectors that pin the oracle (oracle/) and the host-side counterparts of the reference harness:

  tiny_gptneox_fp32.npz   config 1 of BASELINE.json: L=2,H=256,NH=4,I=1024,V=512, rotary 16, parallel residual.
                          - `w###`  : the 12L+4 tensors exactly as the reference's `GptNeoXWeights.load`
                                      (codefuse_example.py:336-419) returns them after the reference's
                                      `split_and_convert_process` (huggingface_convert.py:22-81) wrote the .bin files
                                      (stored as fp16: every HF parameter was made fp16-representable first).
                          - HF fp32 greedy: prompt(s), per-step logits and tokens (16-in / 8-out, and a ragged B=2 case
                            whose rows were run through HF one by one, un-padded).
  tiny_gptneox_beam.npz   HF beam search (num_beams = K, no EOS, length_penalty 0) on the same model: prompts, the K returned
                          continuations (best first) and their cumulative log-probs, for three (B, K, out_len) cases.
  tiny_gptneox_seq.npz    the same parameters with use_parallel_residual=False (FT use_gptj_residual = 0): the separate
                          attention / FFN output biases and HF's greedy logits + tokens on `prompt`.
  tiny_gptneox_tp2.json   sha256 of every tensor the reference loader returns for tensor_para_size=2, rank 0 and 1.
  harness_io.json         I/O of to_word_list_format / Trie.printAutoSuggestions / is_garbage /
                          token_stream_2_str_stream_convertor / get_data_package / generate / CodeFuseHandler.predict
                          captured from the reference (the last two around a recording stand-in for the model: FakeGpt).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference/examples/pytorch/codefuse"
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))

CFG = dict(hidden_size=256, num_attention_heads=4, num_hidden_layers=2, intermediate_size=1024, vocab_size=512,
           rotary_pct=0.25, use_parallel_residual=True, hidden_act="gelu_new", layer_norm_eps=1e-5,
           max_position_embeddings=64, tie_word_embeddings=False, bos_token_id=0, eos_token_id=2, attention_bias=True)


def build_hf(**overrides):
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    cfg = GPTNeoXConfig(**dict(CFG, **overrides))
    torch.manual_seed(1234)
    m = GPTNeoXForCausalLM(cfg).eval()
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layernorm.weight" in n or "layer_norm.weight" in n:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "embed" in n or "lm_head" in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.12 * torch.randn(p.shape, generator=g))
            p.copy_(p.half().float())  # fp16-representable so the fixture can store fp16
    return cfg, m


def export_with_reference(m, cfg, saved_dir, factor):
    import huggingface_convert as hc
    hf_config = vars(m.config)
    for name, param in m.named_parameters():
        array = param.detach().cpu().numpy().astype(np.float32)
        if name == "gpt_neox.embed_in.weight":
            array.tofile(saved_dir + "/model.wte.bin")
        elif name == "gpt_neox.final_layer_norm.bias":
            array.tofile(saved_dir + "/model.final_layernorm.bias.bin")
        elif name == "gpt_neox.final_layer_norm.weight":
            array.tofile(saved_dir + "/model.final_layernorm.weight.bin")
        elif name in ("embed_out.weight", "lm_head.weight"):
            array.tofile(saved_dir + "/model.lm_head.weight.bin")
        else:
            hc.split_and_convert_process(saved_dir, factor, name.replace("gpt_neox.", ""), None, hf_config, array.T)
    # post-processing of huggingface_convert.py:192-206 (use_gptj_residual)
    for l in range(cfg.num_hidden_layers):
        a = np.fromfile(saved_dir + f"/model.layers.{l}.attention.dense.bias.bin", dtype=np.float32)
        b = np.fromfile(saved_dir + f"/model.layers.{l}.mlp.dense_4h_to_h.bias.bin", dtype=np.float32)
        (a + b).astype(np.float32).tofile(saved_dir + f"/model.layers.{l}.mlp.attention.bias.sum.bin")


def load_with_reference(saved_dir, cfg, tp, rank):
    import codefuse_example as ce
    nh = cfg.num_attention_heads
    w = ce.GptNeoXWeights(nh, cfg.hidden_size // nh, cfg.num_hidden_layers, cfg.vocab_size, 1024, tp, 1, True,
                          int8_mode=0, inference_data_type="fp32", weights_data_type=np.float32)
    assert w.load(saved_dir, tensor_para_rank=rank, pipeline_para_rank=0)
    return [t.numpy() for t in w.w]


def hf_greedy(m, ids, n_new):
    logits_steps, toks = [], []
    cur = torch.tensor([ids], dtype=torch.long)
    with torch.no_grad():
        for _ in range(n_new):
            lg = m(cur).logits[0, -1].float()
            t = int(torch.argmax(lg))
            logits_steps.append(lg.numpy().copy())
            toks.append(t)
            cur = torch.cat([cur, torch.tensor([[t]])], dim=1)
    return np.stack(logits_steps), toks


def hf_beam(m, prompts, n_new, K):
    """HF beam search without EOS / length penalty (score == cumulative log-prob): the same search the reference's
    OnlineBeamSearchLayer does when no beam finishes.  Returns tokens [B][K][n_new] and scores [B][K], best first."""
    L = max(len(p) for p in prompts)
    toks, scores = [], []
    for p in prompts:  # row by row, un-padded
        out = m.generate(torch.tensor([p], dtype=torch.long), num_beams=K, num_return_sequences=K, do_sample=False,
                         max_new_tokens=n_new, min_new_tokens=n_new, eos_token_id=None, pad_token_id=0,
                         length_penalty=0.0, early_stopping=False, output_scores=True, return_dict_in_generate=True)
        toks.append(out.sequences[:, len(p):].numpy().astype(np.int32))
        scores.append(out.sequences_scores.numpy().astype(np.float32))
    return np.stack(toks), np.stack(scores)


def beam_golden(m, cfg):
    rng = np.random.RandomState(4242)
    res = {}
    for name, lens, K, n_new in (("a", [16], 3, 8), ("b", [12, 7], 4, 6), ("c", [1], 2, 10)):
        prompts = [rng.randint(3, cfg.vocab_size, size=n).tolist() for n in lens]
        toks, scores = hf_beam(m, prompts, n_new, K)
        S = max(lens)
        ids = np.zeros((len(lens), S), dtype=np.int32)
        for i, p in enumerate(prompts):
            ids[i, :len(p)] = p
        res[f"ids_{name}"] = ids
        res[f"lens_{name}"] = np.array(lens, dtype=np.int32)
        res[f"hf_beam_tokens_{name}"] = toks
        res[f"hf_beam_scores_{name}"] = scores
        print("beam", name, toks.tolist(), scores.tolist())
    np.savez_compressed(os.path.join(OUT, "tiny_gptneox_beam.npz"), **res)


def sequential_golden(prompt):
    """The same parameters with use_parallel_residual=False (FT: use_gptj_residual = 0): the two biases the parallel form
    only stores as a sum, and HF's greedy logits / tokens."""
    cfg, m = build_hf(use_parallel_residual=False)
    L = cfg.num_hidden_layers
    sd = dict(m.named_parameters())
    out_b = np.stack([sd[f"gpt_neox.layers.{l}.attention.dense.bias"].detach().numpy() for l in range(L)])
    ffn2_b = np.stack([sd[f"gpt_neox.layers.{l}.mlp.dense_4h_to_h.bias"].detach().numpy() for l in range(L)])
    logits, toks = hf_greedy(m, prompt, 8)
    np.savez_compressed(os.path.join(OUT, "tiny_gptneox_seq.npz"), out_b=out_b.astype(np.float16),
                        ffn2_b=ffn2_b.astype(np.float16), hf_logits=logits.astype(np.float32),
                        hf_tokens=np.array(toks, np.int32))
    print("sequential", toks)


class FakeTok:
    """Deterministic stand-in tokenizer (the reference helpers only call encode/decode/get_vocab)."""

    def __init__(self):
        self.vocab = {}
        words = ["\n", "}", "for", " (", "int", " i", "=", "0", ";", "def", " ", "re", "ret", "return", "retu", "r",
                 "x", "y", "\n}", "中", "文", "a", "b", "ab", "abc", "éé"]
        for i, w_ in enumerate(words):
            self.vocab[w_] = i + 3

    def get_vocab(self):
        return dict(self.vocab)

    def encode(self, text):
        out, i = [], 0
        keys = sorted(self.vocab, key=len, reverse=True)
        while i < len(text):
            for k in keys:
                if text.startswith(k, i):
                    out.append(self.vocab[k])
                    i += len(k)
                    break
            else:
                i += 1
        return out

    def decode(self, ids):
        inv = {v: k for k, v in self.vocab.items()}
        return "".join(inv.get(int(i), "") for i in ids)


class FakeGpt:
    """Stands in for the GptNeoX module inside generate() / CodeFuseHandler.predict(): records the keyword arguments it is
    called with and returns fixed hypotheses (token by token through the callback first, like the op's streaming callback)."""
    end_id = 2

    def __init__(self, hypotheses):
        self.hypotheses = hypotheses  # [B][beam] lists of generated token ids (end_id where a hypothesis ends)
        self.saw = None

    def __call__(self, **kw):
        self.saw = {k: ({"dtype": str(v.dtype), "value": v.tolist()} if torch.is_tensor(v) else v)
                    for k, v in kw.items() if k != "callback"}
        self.saw["has_callback"] = kw.get("callback") is not None
        ids, n_out, beam = kw["start_ids"], kw["output_len"], kw["beam_width"]
        B, S = ids.shape
        out = torch.full((B, beam, S + n_out), self.end_id, dtype=torch.int32)
        for b in range(B):
            for w in range(beam):
                out[b, w, :S] = ids[b]
                hyp = self.hypotheses[b][w][:n_out]
                out[b, w, S:S + len(hyp)] = torch.tensor(hyp, dtype=torch.int32)
        if kw.get("callback") is not None:
            for t in range(n_out):
                kw["callback"]({"last_tokens": out[:, :, S + t].tolist()})
        lengths = torch.full((B, beam), S + n_out, dtype=torch.int32)
        scores = torch.tensor([[-(1.0 + b) - 0.25 * w for w in range(beam)] for b in range(B)])
        return out, lengths, scores


def generate_cases(tok):
    """(keyword arguments of generate(), the stand-in model's hypotheses) -- shared with tests/test_harness_golden.py"""
    v = tok.vocab
    return [
        (dict(texts=["def", "for ("], output_len=4, beam_width=1, top_k=[3, 50], temperature=[1.0, 0.5],
              random_seed=[1, 2], stop_words_list=[["\n}"], ["x"]]),
         [[[v[" i"], v["="], 2, v["x"]]], [[v["int"], v[" i"], v["="], v["éé"]]]]),
        (dict(texts=None, input_ids_list=[[v["ret"], v["x"]], [v["a"]]], output_len=3, beam_width=2, top_k=5, top_p=0.9,
              repetition_penalty=1.25, last_token_list=["re", "zzz"]),
         [[[v["y"], v["\n"], v["}"]], [2, 2, 2]], [[v["中"], v["文"], 2], [v["b"], v["éé"], v["éé"]]]]),
    ]


def predict_case(tok):
    v = tok.vocab
    req = {"out_seq_length": 4, "stream": True,
           "prompts": [{"prompt": "for (", "random_seed": 11, "top_k": 4, "stop_words": ["\n}"]},
                       {"prompt": "def", "random_seed": 12, "top_k": 2, "stop_words": ["x"]}]}
    return req, [[[v["int"], v[" i"], v["\n"], v["x"]]], [[v[" "], v["a"], v["b"], 2]]]


def harness_io():
    import contextlib
    import io
    import codefuse_example as ce
    tok = FakeTok()
    res = {"vocab": tok.get_vocab()}
    cases = [[["\n}", "for ("]], [["return", "x"], ["abc"]], [[""], ["}"]]]
    res["to_word_list_format"] = [{"in": c, "out": ce.to_word_list_format(c, tok).numpy().tolist()} for c in cases]
    trie = ce.Trie(tok.get_vocab())
    tr = []
    for key in ["re", "ret", "a", "zzz", "abc", ""]:
        r = []
        code = trie.printAutoSuggestions(key, r)
        tr.append({"key": key, "code": code, "ids": sorted(t_i for _, t_i in r)})
    res["trie"] = tr
    cps = [65, 0x4E2D, 200, 65292, 8230, 0x1F600, 127, 128, 183, 12290, 0x3400, 0x2A700, 233]
    res["is_garbage"] = [{"cp": c, "out": bool(ce.is_garbage(c))} for c in cps]
    streams = []
    for toks in ([tok.vocab["for"], tok.vocab[" ("], tok.vocab["int"], tok.vocab[" i"], tok.vocab["\n"], tok.vocab["中"],
                  tok.vocab["文"], tok.vocab["x"], tok.vocab["éé"], 2],
                 [tok.vocab["a"], tok.vocab[" "], tok.vocab["b"], 2, tok.vocab["x"]]):
        conv = ce.token_stream_2_str_stream_convertor(2, tok, 0)
        chunks = []
        for t in toks:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                conv.append_token(t)
            chunks.append(buf.getvalue())
        streams.append({"tokens": toks, "chunks": chunks})
    res["stream"] = streams
    req = {"out_seq_length": 8, "prompts": [{"prompt": "def", "top_k": 3, "stop_words": ["\n}"]},
                                           {"prompt": "for (", "temperature": 0.5, "stop_words": ["x"]}]}
    res["get_data_package"] = {"in": req, "seed": 77, "out": ce.get_data_package(req, 77)}
    req2 = {"out_seq_length": 4, "beam_width": 1, "prompts": [{"prompt": "a"}]}
    res["get_data_package2"] = {"in": req2, "seed": 5, "out": ce.get_data_package(req2, 5)}
    gen = []
    for kwargs, hyps in generate_cases(tok):
        gpt = FakeGpt(hyps)
        texts, lengths, scores, _ = ce.generate(gpt, tok, trie=trie, **kwargs)
        gen.append({"model_saw": gpt.saw, "texts": texts, "lengths": lengths, "cum_log_probs": scores})
    res["generate"] = gen
    req3, hyps3 = predict_case(tok)
    handler = ce.CodeFuseHandler.__new__(ce.CodeFuseHandler)
    handler.local_rank, handler.world_size, handler.tokenizer, handler.trie = 0, 1, tok, trie
    handler.model = FakeGpt(hyps3)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        code, msg, out = handler.predict(req3, "trace-0")
    body = json.loads(out["res"])
    body.pop("latency")
    res["predict"] = {"code": code, "message": msg, "response": body, "printed": buf.getvalue(), "model_saw": handler.model.saw}
    return res


def main():
    if "--harness-only" in sys.argv:
        with open(os.path.join(OUT, "harness_io.json"), "w") as f:
            json.dump(harness_io(), f, indent=1, ensure_ascii=False)
        return
    cfg, m = build_hf()
    beam_golden(m, cfg)
    if "--beam-only" in sys.argv:
        return
    if "--seq-only" in sys.argv:
        rng0 = np.random.RandomState(42)
        sequential_golden(rng0.randint(3, cfg.vocab_size, size=16).tolist())
        return
    rng = np.random.RandomState(42)
    prompt = rng.randint(3, cfg.vocab_size, size=16).tolist()
    logits, toks = hf_greedy(m, prompt, 8)
    sequential_golden(prompt)
    # ragged batch: rows of length 16 and 11, each run through HF separately (no padding semantics involved)
    prompt_b = rng.randint(3, cfg.vocab_size, size=11).tolist()
    logits_b, toks_b = hf_greedy(m, prompt_b, 8)
    prompt_1 = [int(rng.randint(3, cfg.vocab_size))]
    logits_1, toks_1 = hf_greedy(m, prompt_1, 6)

    with tempfile.TemporaryDirectory() as d1:
        export_with_reference(m, cfg, d1, 1)
        w1 = load_with_reference(d1, cfg, 1, 0)
        files = sorted(os.listdir(d1))
    # fp16 storage wherever it is exact (all HF parameters); the converter's bias sums stay fp32
    arrays = {}
    for i, a in enumerate(w1):
        h = a.astype(np.float16)
        arrays[f"w{i:03d}"] = h if np.array_equal(h.astype(np.float32), a) else a.astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "tiny_gptneox_fp32.npz"), **arrays,
        cfg=np.array([cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads, cfg.intermediate_size,
                      cfg.num_hidden_layers, cfg.vocab_size, 16, 0, 2], dtype=np.int32),
        prompt=np.array(prompt, dtype=np.int32), hf_logits=logits.astype(np.float32), hf_tokens=np.array(toks, np.int32),
        prompt_b=np.array(prompt_b, dtype=np.int32), hf_logits_b=logits_b.astype(np.float32),
        hf_tokens_b=np.array(toks_b, np.int32), prompt_1=np.array(prompt_1, dtype=np.int32),
        hf_logits_1=logits_1.astype(np.float32), hf_tokens_1=np.array(toks_1, np.int32))

    tp2 = {"files_tp1": files}
    with tempfile.TemporaryDirectory() as d2:
        export_with_reference(m, cfg, d2, 2)
        tp2["files_tp2"] = sorted(os.listdir(d2))
        for r in range(2):
            ws = load_with_reference(d2, cfg, 2, r)
            tp2[f"rank{r}"] = [{"shape": list(a.shape), "sha256": hashlib.sha256(
                np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()} for a in ws]
    with open(os.path.join(OUT, "tiny_gptneox_tp2.json"), "w") as f:
        json.dump(tp2, f, indent=1)
    with open(os.path.join(OUT, "harness_io.json"), "w") as f:
        json.dump(harness_io(), f, indent=1, ensure_ascii=False)
    print("tokens", toks, toks_b, toks_1)


if __name__ == "__main__":
    main()
