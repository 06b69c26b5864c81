"""ctypes front-end of oracle/libftcf_oracle.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(),
never from the product package (fastertransformer4codefuse_amd/).  See ftcf_oracle.h for how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libftcf_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("ftcf_oracle.c", "ftcf_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_long, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_long, C.c_void_p)


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "head_num", "size_per_head", "inter_size", "num_layer", "vocab_size", "rotary_dim", "start_id", "end_id",
        "tp_size", "tp_rank", "int8_mode", "fp16", "use_gptj_residual")] + [
        ("allreduce", ALLREDUCE_FN), ("allgather", ALLGATHER_FN), ("comm_ctx", C.c_void_p)]


_PP = C.POINTER(C.c_void_p)


class OrcWeights(C.Structure):
    _fields_ = [(n, _PP) for n in (
        "ln1_g", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b", "ln2_g", "ln2_b",
        "qkv_q", "qkv_s", "out_q", "out_s", "ffn1_q", "ffn1_s", "ffn2_q", "ffn2_s")] + [
        (n, C.c_void_p) for n in ("wte", "final_ln_g", "final_ln_b", "lm_head")]


class OrcSampling(C.Structure):
    _fields_ = [("top_k", C.c_void_p), ("top_p", C.c_void_p), ("temperature", C.c_void_p),
                ("repetition_penalty", C.c_void_p), ("min_length", C.c_void_p), ("random_seed", C.c_void_p),
                ("stop_words", C.c_void_p), ("stop_len", C.c_int), ("optional_last_tokens", C.c_void_p),
                ("optional_count", C.c_int), ("return_cum_log_probs", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_round_half.restype = C.c_float
        _lib.orc_round_half.argtypes = [C.c_float]
        _lib.orc_round_half_soft.restype = C.c_float
        _lib.orc_round_half_soft.argtypes = [C.c_float]
        _lib.orc_uniform.restype = C.c_float
        _lib.orc_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        _lib.orc_generate.restype = C.c_int
        _lib.orc_generate_beam.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def round_half(x):
    """Round a float32 array to binary16 and back (numpy does RNE, same as the C helper)."""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def symmetric_quantize_int8(w, weight_is_half=True):
    """cutlass_preprocessors.cc:576-673.  Returns (q int8 [K,N] row-major UNPROCESSED, scale float32 [N])."""
    w = _f32(w)
    K, N = w.shape
    q = np.empty((K, N), dtype=np.int8)
    s = np.empty((N,), dtype=np.float32)
    lib().orc_symmetric_quantize_int8(_ptr(w), C.c_int(K), C.c_int(N), C.c_int(int(weight_is_half)), _ptr(q), _ptr(s))
    return q, s


def gemm(A, W=None, q=None, scale=None, bias=None, act=0, fp16=True, out_fp32=False):
    A = _f32(A)
    m, k = A.shape
    if q is not None:
        q = np.ascontiguousarray(q, dtype=np.int8)
        scale = _f32(scale)
        n = q.shape[1]
    else:
        W = _f32(W)
        n = W.shape[1]
    bias = None if bias is None else _f32(bias)
    Cm = np.empty((m, n), dtype=np.float32)
    lib().orc_gemm(_ptr(A), C.c_int(m), C.c_int(k), C.c_int(n), _ptr(W), _ptr(q), _ptr(scale), _ptr(bias),
                   C.c_int(act), _ptr(Cm), C.c_int(int(fp16)), C.c_int(int(out_fp32)))
    return Cm


def lm_head(A, Wt):
    A = _f32(A)
    Wt = _f32(Wt)
    m, k = A.shape
    n = Wt.shape[0]
    Cm = np.empty((m, n), dtype=np.float32)
    lib().orc_lm_head(_ptr(A), C.c_int(m), C.c_int(k), C.c_int(n), _ptr(Wt), _ptr(Cm))
    return Cm


def layernorm(x, gamma, beta, eps=1e-5, fp16=True):
    x = _f32(x)
    m, n = x.shape
    out = np.empty_like(x)
    gamma = _f32(gamma)
    beta = None if beta is None else _f32(beta)
    lib().orc_layernorm(_ptr(x), _ptr(gamma), _ptr(beta), C.c_int(m), C.c_int(n), C.c_float(eps), _ptr(out),
                        C.c_int(int(fp16)))
    return out


def add_bias_gelu(x, bias, fp16=True):
    x = _f32(x).copy()
    m, n = x.shape
    bias = None if bias is None else _f32(bias)
    lib().orc_add_bias_gelu(_ptr(x), _ptr(bias), C.c_int(m), C.c_int(n), C.c_int(int(fp16)))
    return x


def add_bias_attn_ffn_residual(ffn, attn, x, bias, tp=1, inplace_variant=False, fp16=True):
    ffn, attn, x, bias = _f32(ffn), _f32(attn), _f32(x), _f32(bias)
    m, n = x.shape
    out = np.empty_like(x)
    lib().orc_add_bias_attn_ffn_residual(_ptr(out), _ptr(ffn), _ptr(attn), _ptr(x), _ptr(bias), C.c_int(m), C.c_int(n),
                                         C.c_int(tp), C.c_int(int(inplace_variant)), C.c_int(int(fp16)))
    return out


def mmha_step(qkv, qkv_bias, k_cache, v_cache, seq_len, pad_count, masked_tokens, finished, nh, dh, rot, step,
              fp16=True):
    """k_cache/v_cache float32 [B, nh, S_max, dh], modified in place.  Returns ctx [B, nh*dh]."""
    qkv = _f32(qkv)
    B = qkv.shape[0]
    s_max = k_cache.shape[2]
    assert k_cache.dtype == np.float32 and k_cache.flags.c_contiguous
    assert v_cache.dtype == np.float32 and v_cache.flags.c_contiguous
    seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
    pad_count = None if pad_count is None else np.ascontiguousarray(pad_count, dtype=np.int32)
    masked_tokens = None if masked_tokens is None else np.ascontiguousarray(masked_tokens, dtype=np.uint8)
    finished = None if finished is None else np.ascontiguousarray(finished, dtype=np.uint8)
    qkv_bias = None if qkv_bias is None else _f32(qkv_bias)
    ctx = np.zeros((B, nh * dh), dtype=np.float32)
    lib().orc_mmha_step(_ptr(qkv), _ptr(qkv_bias), _ptr(k_cache), _ptr(v_cache), _ptr(seq_len), _ptr(pad_count),
                        _ptr(masked_tokens), _ptr(finished), C.c_int(B), C.c_int(nh), C.c_int(dh), C.c_int(rot),
                        C.c_int(s_max), C.c_int(step), _ptr(ctx), C.c_int(int(fp16)))
    return ctx


def context_attention(qkv, qkv_bias, input_lengths, k_cache, v_cache, B, S, nh, dh, rot, fp16=True):
    qkv = _f32(qkv)
    s_max = k_cache.shape[2]
    input_lengths = np.ascontiguousarray(input_lengths, dtype=np.int32)
    qkv_bias = None if qkv_bias is None else _f32(qkv_bias)
    ctx = np.zeros((B * S, nh * dh), dtype=np.float32)
    lib().orc_context_attention(_ptr(qkv), _ptr(qkv_bias), _ptr(input_lengths), _ptr(k_cache), _ptr(v_cache),
                                C.c_int(B), C.c_int(S), C.c_int(nh), C.c_int(dh), C.c_int(rot), C.c_int(s_max),
                                _ptr(ctx), C.c_int(int(fp16)))
    return ctx


class Sampling:
    """Keeps the numpy buffers alive behind an OrcSampling struct."""

    def __init__(self, B, top_k=1, top_p=0.0, temperature=1.0, repetition_penalty=1.0, random_seed=0,
                 min_length=None, stop_words=None, optional_last_tokens=None, return_cum_log_probs=1):
        def bc(v, dt):
            a = np.asarray(v, dtype=dt).reshape(-1)
            return np.ascontiguousarray(np.broadcast_to(a, (B,)) if a.size == 1 else a, dtype=dt)

        self.top_k = bc(top_k, np.int32)
        self.top_p = bc(top_p, np.float32)
        self.temperature = bc(temperature, np.float32)
        self.repetition_penalty = None if repetition_penalty is None else bc(repetition_penalty, np.float32)
        self.random_seed = bc(random_seed, np.uint64)
        self.min_length = None if min_length is None else bc(min_length, np.int32)
        self.stop_words = None if stop_words is None else np.ascontiguousarray(stop_words, dtype=np.int32)
        self.optional = None if optional_last_tokens is None else np.ascontiguousarray(optional_last_tokens,
                                                                                        dtype=np.int32)
        self.struct = OrcSampling(
            _ptr(self.top_k), _ptr(self.top_p), _ptr(self.temperature), _ptr(self.repetition_penalty),
            _ptr(self.min_length), _ptr(self.random_seed), _ptr(self.stop_words),
            0 if self.stop_words is None else int(self.stop_words.shape[2]), _ptr(self.optional),
            0 if self.optional is None else int(self.optional.shape[1]), int(return_cum_log_probs))


class OrcBeamParams(C.Structure):
    _fields_ = [("temperature", C.c_void_p), ("repetition_penalty", C.c_void_p), ("diversity_rate", C.c_void_p),
                ("len_penalty", C.c_void_p), ("min_length", C.c_void_p), ("stop_words", C.c_void_p),
                ("stop_len", C.c_int), ("optional_last_tokens", C.c_void_p), ("optional_count", C.c_int)]


class BeamParams:
    """Keeps the numpy buffers alive behind an orc_beam_params struct (runtime args of the beam-search layer)."""

    def __init__(self, B, temperature=None, repetition_penalty=None, diversity_rate=None, len_penalty=None,
                 min_length=None, stop_words=None, optional_last_tokens=None):
        def bc(v, dt):
            if v is None:
                return None
            a = np.asarray(v, dtype=dt).reshape(-1)
            return np.ascontiguousarray(np.broadcast_to(a, (B,)) if a.size == 1 else a, dtype=dt)

        self.temperature = bc(temperature, np.float32)
        self.repetition_penalty = bc(repetition_penalty, np.float32)
        self.diversity_rate = bc(diversity_rate, np.float32)
        self.len_penalty = bc(len_penalty, np.float32)
        self.min_length = bc(min_length, np.int32)
        self.stop_words = None if stop_words is None else np.ascontiguousarray(stop_words, dtype=np.int32)
        self.optional = None if optional_last_tokens is None else np.ascontiguousarray(optional_last_tokens,
                                                                                        dtype=np.int32)
        self.struct = OrcBeamParams(
            _ptr(self.temperature), _ptr(self.repetition_penalty), _ptr(self.diversity_rate), _ptr(self.len_penalty),
            _ptr(self.min_length), _ptr(self.stop_words),
            0 if self.stop_words is None else int(self.stop_words.shape[2]), _ptr(self.optional),
            0 if self.optional is None else int(self.optional.shape[1]))


def beam_search_step(logits, K, step, max_input_len, input_lengths, bp, end_id, output_ids, parent_ids, finished,
                     seq_len, cum_log_probs, src_indir, tgt_indir):
    """One OnlineBeamSearchLayer step, in place on the state arrays.  logits [B*K, V] fp32; output_ids / parent_ids
    time-major [total, B*K] int32; src_indir / tgt_indir [B, K, s_max] int32."""
    BK, V = logits.shape
    B = BK // K
    assert logits.dtype == np.float32 and output_ids.dtype == np.int32 and parent_ids.dtype == np.int32
    assert finished.dtype == np.uint8 and seq_len.dtype == np.int32 and cum_log_probs.dtype == np.float32
    assert src_indir.dtype == np.int32 and tgt_indir.dtype == np.int32
    input_lengths = np.ascontiguousarray(input_lengths, dtype=np.int32)
    lib().orc_beam_search_step(_ptr(logits), C.c_int(B), C.c_int(K), C.c_int(V), C.c_int(step), C.c_int(max_input_len),
                               _ptr(input_lengths), C.byref(bp.struct), C.c_int(end_id), _ptr(output_ids),
                               _ptr(parent_ids), _ptr(finished), _ptr(seq_len), _ptr(cum_log_probs), _ptr(src_indir),
                               _ptr(tgt_indir), C.c_int(src_indir.shape[2]))


def gather_tree_beam(ids, parents, seq_len, tiled_lengths, K, max_input_len, end_id):
    """ids / parents time-major [total, B*K] -> (output_ids [B, K, total], sequence_lengths [B, K])."""
    total, BK = ids.shape
    B = BK // K
    out = np.zeros((B, K, total), dtype=np.int32)
    sl = np.zeros((B, K), dtype=np.int32)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    parents = np.ascontiguousarray(parents, dtype=np.int32)
    seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
    tiled_lengths = np.ascontiguousarray(tiled_lengths, dtype=np.int32)
    lib().orc_gather_tree_beam(_ptr(ids), _ptr(parents), _ptr(seq_len), _ptr(tiled_lengths), C.c_int(B), C.c_int(K),
                               C.c_int(max_input_len), C.c_int(total), C.c_int(end_id), _ptr(out), _ptr(sl))
    return out, sl


def dynamic_decode(logits, step, max_input_len, input_lengths, sampling, end_id, output_ids, finished, seq_len,
                   cum_log_probs, draw_counter):
    """In-place on all state arrays (numpy, C-contiguous). output_ids is time-major [total, B] int32."""
    B, V = logits.shape
    assert logits.dtype == np.float32 and output_ids.dtype == np.int32 and finished.dtype == np.uint8
    assert seq_len.dtype == np.int32 and cum_log_probs.dtype == np.float32 and draw_counter.dtype == np.uint64
    input_lengths = np.ascontiguousarray(input_lengths, dtype=np.int32)
    lib().orc_dynamic_decode(_ptr(logits), C.c_int(B), C.c_int(V), C.c_int(step), C.c_int(max_input_len),
                             _ptr(input_lengths), C.byref(sampling.struct), C.c_int(end_id), _ptr(output_ids),
                             _ptr(finished), _ptr(seq_len), _ptr(cum_log_probs), _ptr(draw_counter),
                             C.c_int(output_ids.shape[0]))


_LAYER_KEYS = ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b", "ln2_g",
               "ln2_b", "qkv_q", "qkv_s", "out_q", "out_s", "ffn1_q", "ffn1_s", "ffn2_q", "ffn2_s")


class Model:
    """Weights of one tensor-parallel rank for the oracle.

    `layers` is a list (len L) of dicts with float32/int8 numpy arrays keyed by _LAYER_KEYS (missing -> NULL);
    `globals_` has wte, final_ln_g, final_ln_b, lm_head."""

    def __init__(self, cfg: dict, layers, globals_, allreduce=None, allgather=None):
        self.cfg_dict = dict(cfg)
        self._keep = []
        self.L = len(layers)
        w = OrcWeights()
        for key in _LAYER_KEYS:
            arr = (C.c_void_p * self.L)()
            for l, lay in enumerate(layers):
                a = lay.get(key)
                if a is None:
                    arr[l] = None
                else:
                    a = np.ascontiguousarray(a, dtype=np.int8 if key.endswith("_q") else np.float32)
                    self._keep.append(a)
                    arr[l] = a.ctypes.data
            self._keep.append(arr)
            setattr(w, key, C.cast(arr, _PP))
        for key in ("wte", "final_ln_g", "final_ln_b", "lm_head"):
            a = _f32(globals_[key])
            self._keep.append(a)
            setattr(w, key, a.ctypes.data)
        self.w = w
        self._ar = ALLREDUCE_FN(allreduce) if allreduce else ALLREDUCE_FN()
        self._ag = ALLGATHER_FN(allgather) if allgather else ALLGATHER_FN()
        c = cfg
        self.c = OrcConfig(c["head_num"], c["size_per_head"], c["inter_size"], self.L, c["vocab_size"],
                           c["rotary_dim"], c.get("start_id", 0), c["end_id"], c.get("tp_size", 1),
                           c.get("tp_rank", 0), c.get("int8_mode", 0), int(c.get("fp16", 1)),
                           int(c.get("use_gptj_residual", 1)), self._ar, self._ag, None)

    def generate(self, input_ids, input_lengths, out_len, sampling=None, return_logits=False):
        input_ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        B, S = input_ids.shape
        input_lengths = np.ascontiguousarray(input_lengths, dtype=np.int32)
        sampling = sampling or Sampling(B)
        total = S + out_len
        out = np.zeros((B, total), dtype=np.int32)
        sl = np.zeros((B,), dtype=np.int32)
        cum = np.zeros((B,), dtype=np.float32)
        V = self.c.vocab_size
        H = self.c.head_num * self.c.size_per_head
        dbg = np.zeros((out_len, B, V), dtype=np.float32) if return_logits else None
        hid = np.zeros((B, H), dtype=np.float32)
        n = lib().orc_generate(C.byref(self.c), C.byref(self.w), _ptr(input_ids), _ptr(input_lengths), C.c_int(B),
                               C.c_int(S), C.c_int(out_len), C.byref(sampling.struct), _ptr(out), _ptr(sl), _ptr(cum),
                               _ptr(dbg), _ptr(hid))
        res = {"output_ids": out, "sequence_lengths": sl, "cum_log_probs": cum, "steps": n, "last_hidden": hid}
        if return_logits:
            res["logits"] = dbg
        return res

    def generate_beam(self, input_ids, input_lengths, out_len, beam_width, beam_params=None):
        """Beam search (beam_width > 1): output_ids [B, K, S+out_len], sequence_lengths [B, K], cum_log_probs [B, K]."""
        input_ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        B, S = input_ids.shape
        K = int(beam_width)
        input_lengths = np.ascontiguousarray(input_lengths, dtype=np.int32)
        bp = beam_params or BeamParams(B)
        out = np.zeros((B, K, S + out_len), dtype=np.int32)
        sl = np.zeros((B, K), dtype=np.int32)
        cum = np.zeros((B, K), dtype=np.float32)
        n = lib().orc_generate_beam(C.byref(self.c), C.byref(self.w), _ptr(input_ids), _ptr(input_lengths), C.c_int(B),
                                    C.c_int(S), C.c_int(out_len), C.c_int(K), C.byref(bp.struct), _ptr(out), _ptr(sl),
                                    _ptr(cum))
        return {"output_ids": out, "sequence_lengths": sl, "cum_log_probs": cum, "steps": n}

    def decoder_step(self, x, k_cache, v_cache, seq_len, pad_count, masked_tokens, finished, step):
        """k_cache/v_cache float32 [L, B, nhl, S_max, dh] in place."""
        x = _f32(x)
        B = x.shape[0]
        s_max = k_cache.shape[3]
        y = np.empty_like(x)
        seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        pad_count = np.ascontiguousarray(pad_count, dtype=np.int32)
        masked_tokens = np.ascontiguousarray(masked_tokens, dtype=np.uint8)
        finished = np.ascontiguousarray(finished, dtype=np.uint8)
        lib().orc_decoder_step(C.byref(self.c), C.byref(self.w), _ptr(x), _ptr(k_cache), _ptr(v_cache), _ptr(seq_len),
                               _ptr(pad_count), _ptr(masked_tokens), _ptr(finished), C.c_int(B), C.c_int(s_max),
                               C.c_int(step), _ptr(y))
        return y


def sm80_preprocess_int8(q_rowmajor, step=None):
    """CUDA-build layout of a row-major int8 [K,N] matrix (cutlass_preprocessors.cc:500-539); `step` selects one of the
    four steps ("permute", "transpose", "interleave", "bias") for the reference's known-answer tests."""
    q = np.ascontiguousarray(q_rowmajor, dtype=np.int8)
    L = lib()
    i8p = C.POINTER(C.c_int8)
    if step == "bias":
        out = q.copy().reshape(-1)
        L.orc_sm80_add_bias_interleave_int8(out.ctypes.data_as(i8p), C.c_size_t(out.size))
        return out.reshape(q.shape)
    K, N = q.shape
    out = np.empty((K, N) if step in (None, "permute") else (N, K), dtype=np.int8)
    if step is None:
        out = np.empty(K * N, dtype=np.int8)
    fn = {None: L.orc_sm80_preprocess_int8, "permute": L.orc_sm80_permute_rows, "transpose": L.orc_sm80_transpose,
          "interleave": L.orc_sm80_interleave_columns}[step]
    if step == "interleave":  # input is already column major [N][K]
        N, K = q.shape
        out = np.empty(K * N, dtype=np.int8)
    fn(q.ctypes.data_as(i8p), K, N, out.ctypes.data_as(i8p))
    return out
