"""Host-side harness of the CodeFuse path: the counterpart of the reference's `examples/pytorch/codefuse/
codefuse_example.py` (request marshalling, checkpoint loading, post-processing) written against this engine.

Same names, argument meaning and I/O as the reference so that its call sites read the same:
  to_word_list_format (codefuse_example.py:26-53)      is_garbage / is_chinese_char (:56-81)
  token_stream_2_str_stream_convertor (:83-130)        Trie (:137-172)
  GptNeoXWeights (:182-419)                            GptNeoX (:422-611)
  init_model_and_tokenizer (:619-663)                  generate (:666-770)
  get_data_package (:779-812)                          CodeFuseHandler (:814-905)
The behaviour is pinned by tests/golden/harness_io.json and tests/golden/tiny_gptneox_*.{npz,json}, captured from the
reference's own functions (tests/golden/make_golden.py).
"""
import json
import logging
import os
import random
import time
import traceback
from configparser import ConfigParser

import numpy as np
import torch

_CJK_RANGES = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F),
               (0x2B820, 0x2CEAF), (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))
# full-width / typographic punctuation the reference whitelists (codefuse_example.py:76-80)
_PUNCT_OK = frozenset(ord(c) for c in "，。？！、；：“”‘’（）《》【】{}[]<>|-=_+*&^%$#@￥~·`…")


def is_chinese_char(cp):
    return any(lo <= cp <= hi for lo, hi in _CJK_RANGES)


def is_garbage(cp):
    """True for a trailing code point that should be dropped from a decoded string (a half-decoded byte sequence)."""
    return not (is_chinese_char(cp) or cp < 128 or cp in _PUNCT_OK)


def to_word_list_format(words_list, tokenizer):
    """[[word, ...] per request] -> int32 tensor [B, 2, L]: row 0 the concatenated token ids, row 1 the cumulative
    end offsets of every word; rows are padded with 0 / -1 (the layout stop_words_criterion consumes,
    kernels/stop_criteria_kernels.cu:24-83)."""
    id_rows, off_rows = [], []
    for words in words_list:
        ids, lens = [], []
        for word in words:
            enc = tokenizer.encode(word)
            if not enc:
                continue
            ids.extend(enc)
            lens.append(len(enc))
        id_rows.append(ids)
        off_rows.append(np.cumsum(lens).astype(np.int64).tolist() if lens else [])
    width = max(1, max(len(r) for r in id_rows))
    out = np.zeros((len(id_rows), 2, width), dtype=np.int32)
    out[:, 1, :] = -1
    for i, (ids, offs) in enumerate(zip(id_rows, off_rows)):
        out[i, 0, :len(ids)] = ids
        out[i, 1, :len(offs)] = offs
    return torch.from_numpy(out)


class token_stream_2_str_stream_convertor:
    """Incremental detokeniser behind the streaming callback: tokens in, printable text out.

    Contract (pinned by tests/golden/harness_io.json "stream", captured from the reference's class of the same name):
    the pending tokens are decoded as a whole after every arrival and the text is released up to a *safe point* --
    everything when it ends in a newline (the pending tokens are then dropped) or in a CJK character, otherwise up to
    and including its last space; `end_id` releases the rest (minus a trailing half-decoded character), prints the
    end marker and closes the stream.  Only rank 0 prints."""

    END_MARKER = "\n\nend\n\n"

    def __init__(self, end_id, tokenizer, local_rank):
        self.end_id, self.tokenizer, self.local_rank = end_id, tokenizer, local_rank
        self._pending = []   # tokens whose text has not been released completely
        self._released = 0   # characters of decode(self._pending) already printed
        self._closed = False

    def _emit(self, text):
        if self.local_rank == 0:
            print(text, end="", flush=True)

    def send_str(self, str_to_send):
        self._emit(str_to_send)

    def send_finish(self):
        self._emit(self.END_MARKER)

    @staticmethod
    def _safe_point(text):
        """(characters that may be shown, whether the pending tokens can be forgotten)."""
        if text.endswith("\n"):
            return len(text), True
        if text and is_chinese_char(ord(text[-1])):
            return len(text), False
        return text.rfind(" ") + 1, False

    def _release(self, text, upto):
        chunk = text[self._released:upto] if upto > self._released else ""
        self._released = max(self._released, upto)
        self.send_str(chunk)

    def append_token(self, token):
        if self._closed:
            return
        if token == self.end_id:
            text = self.tokenizer.decode(self._pending)
            tail = len(text)
            if tail > self._released and is_garbage(ord(text[-1])):
                tail -= 1
            self._release(text, tail)
            self._pending, self._released, self._closed = [], 0, True
            self.send_finish()
            return
        self._pending.append(token)
        text = self.tokenizer.decode(self._pending)
        upto, forget = self._safe_point(text)
        self._release(text, upto)
        if forget:
            self._pending, self._released = [], 0


class Trie:
    """Prefix tree over the tokenizer vocabulary for the "optional last tokens" completion feature."""

    class _Node:
        __slots__ = ("children", "last")

        def __init__(self):
            self.children, self.last = {}, False

    def __init__(self, vocab):
        self.vocab = vocab
        self.root = Trie._Node()
        for key in vocab.keys():
            self.insert(key)

    def insert(self, key):
        node = self.root
        for ch in key:
            nxt = node.children.get(ch)
            if nxt is None:
                nxt = node.children[ch] = Trie._Node()
            node = nxt
        node.last = True

    def printAutoSuggestions(self, key, results):
        """Appends (word, id) of every vocabulary entry that extends `key`.  Returns 0 if `key` is not a prefix of
        anything, -1 if it is a complete entry with no extension, 1 otherwise."""
        node = self.root
        for ch in key:
            node = node.children.get(ch)
            if node is None:
                return 0
        if not node.children:
            return -1
        stack = [(node, key)]
        found = []
        while stack:
            cur, word = stack.pop()
            if cur.last:
                found.append((word, self.vocab[word]))
            for ch, child in cur.children.items():
                stack.append((child, word + ch))
        results.extend(found)
        return 1


str_type_map = {"fp32": torch.float32, "fp16": torch.float16}
_NP_TYPES = {"fp16": np.float16, "fp32": np.float32, "float16": np.float16, "float32": np.float32}

# checkpoint tensors of one layer, in the GptNeoXOp weight-list order (GptNeoXOp.h:121-146)
_LAYER_FILES = (("input_layernorm.bias", False), ("input_layernorm.weight", False),
                ("attention.query_key_value.weight", True), ("attention.query_key_value.bias", True),
                ("attention.dense.weight", True), ("attention.dense.bias", False),
                ("mlp.dense_h_to_4h.weight", True), ("mlp.dense_h_to_4h.bias", True),
                ("mlp.dense_4h_to_h.weight", True), ("mlp.dense_4h_to_h.bias", False),
                ("post_attention_layernorm.bias", False), ("post_attention_layernorm.weight", False))
_KERNEL_GROUPS = (2, 4, 6, 8)


class GptNeoXWeights:
    """Weight container + loader for the `.bin` + config.ini checkpoint format (codefuse_example.py:182-419)."""

    def __init__(self, head_num, size_per_head, layer_num, vocab_size, max_seq_len, tensor_para_size,
                 pipeline_para_size, use_gptj_residual, int8_mode=0, inference_data_type="fp16",
                 weights_data_type=np.float32, enable_int8_weights=False, use_pybind11=False, inter_size=None):
        assert head_num % tensor_para_size == 0
        assert int8_mode in (0, 1), "Invalid int8 mode for GPT. Must be 0 or 1"
        if int8_mode == 1:
            assert str_type_map[inference_data_type] == torch.float16, \
                "Weight only quant only supported for infer type fp16 or bf16."
            from .gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8
            self.weight_transpose_calibrate_quantize = symmetric_quantize_last_axis_of_batched_matrix_int8
        if isinstance(weights_data_type, str):
            if weights_data_type not in _NP_TYPES:
                raise ValueError(f"Don't know how to interpret weights_data_type: {weights_data_type}")
            weights_data_type = _NP_TYPES[weights_data_type]
        assert weights_data_type in (np.float32, np.float16)
        self.head_num, self.size_per_head, self.layer_num = head_num, size_per_head, layer_num
        self.vocab_size, self.max_seq_len = vocab_size, max_seq_len
        self.tensor_para_size, self.pipeline_para_size = tensor_para_size, pipeline_para_size
        self.layers_per_device = layer_num // pipeline_para_size
        self.use_gptj_residual = use_gptj_residual
        self.int8_mode, self.enable_int8_weights, self.use_pybind11 = int8_mode, enable_int8_weights, use_pybind11
        self.weights_data_type = weights_data_type
        self.inference_data_type = str_type_map[inference_data_type]
        self.global_head_num = head_num
        self.local_head_num = head_num // tensor_para_size
        self.global_hidden_units = head_num * size_per_head
        self.local_hidden_units = self.local_head_num * size_per_head
        # the reference hard-wires inter = 4 * hidden (codefuse_example.py:216); config.ini's inter_size wins here
        self.local_inter_size = (inter_size // tensor_para_size) if inter_size else self.local_hidden_units * 4
        H, hl, il, L, dt = (self.global_hidden_units, self.local_hidden_units, self.local_inter_size, layer_num,
                            self.inference_data_type)
        self._shapes = [(H,), (H,), (H, 3 * hl), (3 * hl,), (hl, H), (H,) if not use_gptj_residual else (0,),
                        (H, il), (il,), (il, H), (H,), (H,), (H,)]
        self.w = [torch.zeros(shape, dtype=dt) for shape in self._shapes for _ in range(L)]
        self._global_shapes = [(vocab_size, H), (H,), (H,), (vocab_size, H)]
        self.w += [torch.zeros(shape, dtype=dt) for shape in self._global_shapes]
        self.int8_w, self.scale = [], []
        if int8_mode:
            for g in _KERNEL_GROUPS:
                self.int8_w += [torch.zeros(self._shapes[g], dtype=torch.int8) for _ in range(L)]
                self.scale += [torch.zeros(self._shapes[g][1], dtype=torch.float) for _ in range(L)]
            if enable_int8_weights:
                for g in _KERNEL_GROUPS:
                    for l in range(L):
                        self.w[g * L + l] = torch.empty(0).to(dt)

    def __getitem__(self, idx):
        return self.w[idx]

    def __setitem__(self, idx, val):
        self.w[idx] = val

    def __len__(self):
        return len(self.w)

    def _map(self, func):
        self.w = [func(t) for t in self.w]

    def _map_int8(self, func):
        self.int8_w = [func(t) for t in self.int8_w]
        self.scale = [func(t) for t in self.scale]

    def _map_int8_scales(self, func):
        self.scale = [func(t) for t in self.scale]

    def _file_name(self, group, tensor_para_rank):
        name, sharded = _LAYER_FILES[group]
        if group in _KERNEL_GROUPS and self.enable_int8_weights:
            return None
        if group == 5 and self.use_gptj_residual:
            return None
        if group == 9 and self.use_gptj_residual:
            return "mlp.attention.bias.sum"
        return f"{name}.{tensor_para_rank}" if sharded else name

    def load(self, ckpt_path, tensor_para_rank, pipeline_para_rank):
        if not os.path.exists(ckpt_path):
            return False
        L, dt = self.layer_num, self.inference_data_type
        lo = self.layers_per_device * pipeline_para_rank
        hi = lo + self.layers_per_device

        def read(path, np_dtype=None):
            return torch.from_numpy(np.fromfile(path, dtype=np_dtype or self.weights_data_type))

        loaded = []
        for g in range(12):
            fname = self._file_name(g, tensor_para_rank)
            for l in range(L):
                if fname is not None and lo <= l < hi:
                    loaded.append(read(f"{ckpt_path}/model.layers.{l}.{fname}.bin").to(dt))
                else:
                    loaded.append(torch.empty(0).to(dt))
        # slot 12L+1 <- final_layernorm.weight, 12L+2 <- .bias (GptNeoXOp.h:172-173)
        for fname in ("wte", "final_layernorm.weight", "final_layernorm.bias", "lm_head.weight"):
            loaded.append(read(f"{ckpt_path}/model.{fname}.bin").to(dt))
        for i, t in enumerate(loaded):
            if t.nelement() > 0:
                try:
                    self.w[i] = t.reshape(self.w[i].shape)
                except RuntimeError:
                    raise RuntimeError(
                        "head_num, size_per_head, vocab_size, and max_seq_len must be the same as the ones during "
                        f"training (idx: {i} expected shape: {self.w[i].shape} got shape: {t.shape}).")
            else:
                self.w[i] = t
        if self.int8_mode:
            for j, g in enumerate(_KERNEL_GROUPS):
                for l in range(L):
                    if not self.enable_int8_weights:
                        q, s = self.weight_transpose_calibrate_quantize(self.w[g * L + l].contiguous())
                        self.int8_w[j * L + l], self.scale[j * L + l] = q, s
                        self.w[g * L + l] = torch.empty(0).to(dt)  # the fp kernels are no longer needed
                    else:
                        base = f"{ckpt_path}/model.layers.{l}.{_LAYER_FILES[g][0]}.{tensor_para_rank}"
                        self.int8_w[j * L + l] = read(base + ".q.bin", np.int8).to(torch.int8)
                        self.scale[j * L + l] = read(base + ".s.bin").to(dt)
        return True


class GptNeoX(torch.nn.Module):
    """nn.Module face of the op (codefuse_example.py:422-611); `lib_path` points at the directory holding the
    `libth_gptneox` / `libth_common` modules (fastertransformer4codefuse_amd/lib)."""

    def __init__(self, head_num, size_per_head, vocab_size, rotary_embedding_dim, start_id, end_id, layer_num,
                 max_seq_len, tensor_para_size, pipeline_para_size, use_gptj_residual, lib_path=None, int8_mode=0,
                 inference_data_type="fp16", weights_data_type=np.float32, enable_int8_weights=False,
                 use_pybind11=True, inter_size=None):
        super().__init__()
        import torch.distributed as dist
        self.head_num, self.size_per_head = head_num, size_per_head
        self.inter_size = inter_size or 4 * head_num * size_per_head
        self.vocab_size, self.rotary_embedding_dim = vocab_size, rotary_embedding_dim
        self.start_id, self.end_id, self.max_seq_len, self.layer_num = start_id, end_id, max_seq_len, layer_num
        self.use_gptj_residual, self.int8_mode = use_gptj_residual, int8_mode
        self.enable_int8_weights = enable_int8_weights
        self.tensor_para_size, self.pipeline_para_size = tensor_para_size, pipeline_para_size
        self.build_model = False
        self.weights_data_type, self.inference_data_type = weights_data_type, inference_data_type
        assert torch.cuda.is_available(), "CUDA is required for this model."  # torch-ROCm reports the HIP device here
        assert head_num % tensor_para_size == 0, "head_num must be a multiple of tensor_para_size."
        assert layer_num % pipeline_para_size == 0, "layer_num must be a multiple of pipeline_para_size."
        from .gptneox_op import GptNeoXOp
        self.GptNeoXOp = GptNeoXOp
        self.weights = GptNeoXWeights(head_num, size_per_head, layer_num, vocab_size, max_seq_len, tensor_para_size,
                                      pipeline_para_size, use_gptj_residual, int8_mode=int8_mode,
                                      weights_data_type=weights_data_type, inference_data_type=inference_data_type,
                                      use_pybind11=True, enable_int8_weights=enable_int8_weights,
                                      inter_size=self.inter_size)
        if not dist.is_initialized() and tensor_para_size * pipeline_para_size > 1:
            dist.init_process_group(backend="nccl")
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device_count = torch.cuda.device_count()
        self.device = self.rank % self.device_count
        torch.cuda.set_device(self.device)
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        assert self.world_size == tensor_para_size * pipeline_para_size, \
            "tensor_para_size * pipeline_para_size must be equal to world_size."
        self.tensor_para_rank = self.rank % self.tensor_para_size
        self.pipeline_para_rank = self.rank // self.tensor_para_size

    def load(self, ckpt_path):
        ok = self.weights.load(ckpt_path, tensor_para_rank=self.tensor_para_rank,
                               pipeline_para_rank=self.pipeline_para_rank)
        self.cuda()
        return ok

    def half(self):
        self.weights._map(lambda w: w.half())
        self.cuda()

    def cuda(self):
        import torch.distributed as dist
        self.weights._map(lambda w: w.cuda(self.device))
        if self.int8_mode:
            self.weights._map_int8(lambda w: w.cuda(self.device))
        if self.build_model:
            del self.model
            self.build_model = False
        comm = dist.distributed_c10d._get_default_group() if dist.is_initialized() else None
        self.model = self.GptNeoXOp(comm, self.rank, self.head_num, self.size_per_head, self.inter_size,
                                    self.layer_num, self.vocab_size, self.rotary_embedding_dim, self.start_id,
                                    self.end_id, self.tensor_para_size, self.pipeline_para_size, self.int8_mode,
                                    self.max_seq_len, self.use_gptj_residual, self.weights.w, self.weights.int8_w,
                                    self.weights.scale)
        self.build_model = True

    def forward(self, start_ids, start_lengths, output_len, beam_width=1, top_k=None, top_p=None,
                beam_search_diversity_rate=None, temperature=None, len_penalty=None, repetition_penalty=None,
                random_seed=None, stop_words_list=None, optional_last_tokens=None, return_output_length=False,
                return_cum_log_probs=0, callback=None):
        if not self.build_model:
            self.cuda()
        assert start_ids.size(1) > 0, \
            "input len must be larger than zero. For an unconditional case, use start_id as the first token."
        to_dev = lambda t: None if t is None else t.cuda(self.device)
        outputs = self.model.forward(to_dev(start_ids), to_dev(start_lengths), output_len, beam_width, top_k, top_p,
                                     beam_search_diversity_rate, temperature, len_penalty, repetition_penalty,
                                     random_seed, to_dev(stop_words_list), to_dev(optional_last_tokens),
                                     return_cum_log_probs, callback)
        if not return_output_length:
            return outputs[0]
        return tuple(outputs) if return_cum_log_probs > 0 else (outputs[0], outputs[1])


def init_model_and_tokenizer(lib_path, ckpt_path, tokenizer_file_path, tensor_parallel, int8_mode=0,
                             enable_int8_weights=False, trie_needed=False, end_id=None, tokenizer=None):
    from transformers import AutoTokenizer
    config = ConfigParser()
    config.read(os.path.join(ckpt_path, "config.ini"))
    sec = config["gptneox"]
    head_num, size_per_head = int(sec["head_num"]), int(sec["size_per_head"])
    end_id = int(sec["end_id"]) if end_id is None else end_id
    tokenizer = tokenizer or AutoTokenizer.from_pretrained(tokenizer_file_path)
    gpt = GptNeoX(head_num, size_per_head, int(sec["vocab_size"]), int(sec["rotary_embedding"]), int(sec["start_id"]),
                  end_id, int(sec["num_layer"]), max_seq_len=1024, tensor_para_size=tensor_parallel,
                  pipeline_para_size=1, use_gptj_residual=(sec["use_gptj_residual"] == "1"), lib_path=lib_path,
                  int8_mode=int8_mode, inference_data_type="fp16", weights_data_type=sec["weight_data_type"],
                  use_pybind11=True, enable_int8_weights=enable_int8_weights,
                  inter_size=int(sec["inter_size"]) if "inter_size" in sec else None)
    if not gpt.load(ckpt_path=ckpt_path):
        print("[WARNING] Checkpoint file not found. Model loading is skipped.")
    if not trie_needed:
        return gpt, tokenizer
    return gpt, tokenizer, Trie(tokenizer.get_vocab())


# per-request sampling arguments of GptNeoX.forward: (keyword, python scalar type, tensor constructor)
_SAMPLING_ARGS = (("top_k", int, torch.IntTensor), ("top_p", float, torch.FloatTensor),
                  ("beam_search_diversity_rate", float, torch.FloatTensor), ("temperature", float, torch.FloatTensor),
                  ("len_penalty", float, torch.FloatTensor), ("repetition_penalty", float, torch.FloatTensor),
                  ("random_seed", int, torch.LongTensor))


def _row_tensor(value, scalar_type, ctor):
    """None | scalar | per-row list -> None | 1-D tensor (anything else is a caller error, as in the reference)."""
    if value is None:
        return None
    if isinstance(value, scalar_type):
        value = [value]
    if not isinstance(value, list):
        raise RuntimeError("type don't match with %s" % str(scalar_type))
    return ctor(value)


def _pad_rows(rows, pad):
    """list of int lists -> (int32 tensor [B, longest] padded with `pad`, list of lengths)."""
    lens = [len(r) for r in rows]
    out = torch.full((len(rows), max(lens)), pad, dtype=torch.int32)
    for b, row in enumerate(rows):
        out[b, :lens[b]] = torch.tensor(row, dtype=torch.int32)
    return out, lens


def _trim_and_decode(ids, prompt_len, end_id, tokenizer):
    """One returned hypothesis -> (text, generated length): the prompt is dropped, generation ends in front of the first
    `end_id`, a trailing half-decoded character is not shown."""
    gen = list(ids[prompt_len:])
    if end_id in gen:
        del gen[gen.index(end_id):]
    text = tokenizer.decode(gen)
    if text and is_garbage(ord(text[-1])):
        text = text[:-1]
    return text, len(gen)


def generate(gpt, tokenizer, texts, output_len, beam_width, top_k=None, top_p=None, beam_search_diversity_rate=None,
             temperature=None, len_penalty=None, repetition_penalty=None, random_seed=None, input_ids_list=None,
             callback=None, stop_words_list=None, last_token_list=None, trie=None):
    """Prompts (texts, or token id lists) -> ([B][beam] texts, [B][beam] generated lengths, cum_log_probs, seconds).
    Counterpart of codefuse_example.py:666-770; the request / response shapes are what CodeFuseHandler.predict serves."""
    assert texts is not None or input_ids_list is not None
    prompts = [list(tokenizer.encode(t)) for t in texts] if texts is not None else [list(r) for r in input_ids_list]
    start_ids, prompt_lens = _pad_rows(prompts, gpt.end_id)
    sampling = {}
    given = dict(top_k=top_k, top_p=top_p, beam_search_diversity_rate=beam_search_diversity_rate, temperature=temperature,
                 len_penalty=len_penalty, repetition_penalty=repetition_penalty, random_seed=random_seed)
    for name, scalar_type, ctor in _SAMPLING_ARGS:
        sampling[name] = _row_tensor(given[name], scalar_type, ctor)
    stop_words = None if stop_words_list is None else to_word_list_format(stop_words_list, tokenizer)
    last_tokens = None
    if last_token_list is not None:
        assert trie is not None, "trie is None, can't select last token"
        candidates = []
        for prefix in last_token_list:  # every vocabulary entry that extends the prefix; none: the end token
            hits = []
            trie.printAutoSuggestions(prefix, hits)
            candidates.append([tid for _, tid in hits] or [gpt.end_id])
        last_tokens, _ = _pad_rows(candidates, -1)
    t0 = time.time()
    with torch.no_grad():
        out = gpt(start_ids=start_ids, start_lengths=torch.IntTensor(prompt_lens), output_len=output_len,
                  beam_width=beam_width, stop_words_list=stop_words, optional_last_tokens=last_tokens,
                  return_output_length=True, return_cum_log_probs=1, callback=callback, **sampling)
    latency = time.time() - t0
    token_rows, cum_log_probs = out[0].detach().cpu().tolist(), out[2].detach().cpu().tolist()
    outputs, output_lengths = [], []
    for beams, n_prompt in zip(token_rows, prompt_lens):
        decoded = [_trim_and_decode(ids, n_prompt, gpt.end_id, tokenizer) for ids in beams]
        outputs.append([text for text, _ in decoded])
        output_lengths.append([n for _, n in decoded])
    return outputs, output_lengths, cum_log_probs, latency


_BATCHED_DEFAULTS = (("top_k", 50), ("top_p", 0.), ("beam_search_diversity_rate", 0.), ("temperature", 1.),
                     ("len_penalty", 0.), ("repetition_penalty", 1.))


def get_data_package(request_dict, default_random_seed):
    """request JSON -> keyword arguments of generate() (per-prompt keys gathered into per-row lists)."""
    prompts = request_dict["prompts"]

    def gather(key, default=None):
        present = [key in p for p in prompts]
        if default is None:
            if not any(present):
                return None
            if not all(present):
                raise RuntimeError("default_value is None while %s is also None." % key)
        return [p.get(key, default) for p in prompts]

    for p in prompts:
        assert isinstance(p["prompt"], str)
    pkg = {"texts": [p["prompt"] for p in prompts], "output_len": request_dict["out_seq_length"],
           "beam_width": request_dict.get("beam_width", 1)}
    for key, default in _BATCHED_DEFAULTS:
        pkg[key] = gather(key, default)
    pkg["random_seed"] = gather("random_seed", default_random_seed)
    pkg["stop_words_list"] = gather("stop_words")
    pkg["last_token_list"] = gather("last_token")
    return pkg


class CodeFuseHandler:
    """Service shim: predict(request_dict, trace_id) -> (code, message, {"res": json}) (codefuse_example.py:814-905)."""

    def __init__(self, lib_path, ckpt_path, tokenizer_path, int8_mode, enable_int8_weights, world_size=1, local_rank=0,
                 end_id=None, tokenizer=None):
        self.local_rank, self.world_size = local_rank, world_size
        logging.info("start init rank: %d" % local_rank)
        try:
            self.model, self.tokenizer, self.trie = init_model_and_tokenizer(
                lib_path=lib_path, ckpt_path=ckpt_path, tokenizer_file_path=tokenizer_path,
                tensor_parallel=world_size, int8_mode=int8_mode, enable_int8_weights=enable_int8_weights,
                trie_needed=True, end_id=end_id, tokenizer=tokenizer)
            generate(self.model, self.tokenizer, ["demo"], 2, 1)  # warm-up, like the reference (:832)
        except BaseException as err:  # noqa: B902 -- the reference logs and carries on
            logging.exception(err)

    def _shared_seed(self):
        """One random seed per request, the same on every tensor-parallel rank (rank 0 draws, the others receive)."""
        seed = random.randint(0, 1048576)
        if self.world_size > 1:
            import torch.distributed as dist
            box = torch.IntTensor([seed]).to("cuda")
            dist.broadcast(box, src=0)
            seed = int(box.cpu()[0])
        return seed

    def _stream_printers(self, batch, beam):
        """(callback for GptNeoX.forward, closer) that print every hypothesis' text as its tokens arrive."""
        grid = [[token_stream_2_str_stream_convertor(self.model.end_id, self.tokenizer, self.local_rank)
                 for _ in range(beam)] for _ in range(batch)]

        def feed(tokens):  # tokens[b][w]
            for row, row_tokens in zip(grid, tokens):
                for printer, token in zip(row, row_tokens):
                    printer.append_token(token)

        def on_step(message):
            try:
                feed(message["last_tokens"])
            except BaseException as err:  # noqa: B902 -- a printing problem must not end the generation
                logging.error("callback error: %s" % str(err))

        return on_step, lambda: feed([[self.model.end_id] * beam] * batch)

    def predict(self, request_dict, trace_id):
        logging.info("%s request: %s" % (trace_id, json.dumps(request_dict, ensure_ascii=False)))
        try:
            pkg = get_data_package(request_dict, self._shared_seed())
            on_step = close = None
            if request_dict.get("stream") and self.local_rank == 0:
                on_step, close = self._stream_printers(len(pkg["texts"]), pkg["beam_width"])
            texts, lengths, cum_log_probs, latency = generate(self.model, self.tokenizer, trie=self.trie,
                                                              callback=on_step, **pkg)
            if close is not None:
                close()
            body = json.dumps({"latency": latency, "random_seed": pkg["random_seed"], "generated_code": texts,
                               "length": lengths, "cum_log_prob": cum_log_probs}, ensure_ascii=False)
            logging.info("%s response: %s" % (trace_id, body))
            return 0, "ok", {"res": body}
        except BaseException:  # noqa: B902
            return 1, traceback.format_exc(), {"res": ""}
