"""ctypes binding of libftcf.so (include/ftcf.h).  Fails loudly when the HIP library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("FTCF_LIB_NAME", "libftcf.so"))  # env: kernel-variant experiments

UNIQUE_ID_BYTES = 128
FP32, FP16, BF16 = 0, 1, 2
ACT_NONE, ACT_GELU = 0, 1


class FtcfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libftcf error {code}: {msg}")
        self.code = code


TOKEN_CALLBACK = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p)
# ftcf_host_allgather_fn (include/ftcf.h): int (*)(void* user, const void* send, void* recv, size_t bytes_per_rank)
HOST_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
BATCHER_TOKEN_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_long, C.c_int, C.c_int)


class GptNeoXConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "head_num", "size_per_head", "inter_size", "num_layer", "vocab_size", "rotary_embedding_dim", "start_id",
        "end_id", "tensor_para_size", "tensor_para_rank", "pipeline_para_size", "int8_mode", "dtype",
        "use_gptj_residual", "device")] + [("stream", C.c_void_p), ("comm", C.c_void_p), ("use_hip_graph", C.c_int)]


class GptNeoXWeights(C.Structure):
    _fields_ = [("weights", C.POINTER(C.c_void_p)), ("n_weights", C.c_int),
                ("int8_weights", C.POINTER(C.c_void_p)), ("n_int8_weights", C.c_int),
                ("scales", C.POINTER(C.c_void_p)), ("n_scales", C.c_int)]


class ForwardArgs(C.Structure):
    _fields_ = [
        ("input_ids", C.c_void_p), ("input_lengths", C.c_void_p),
        ("batch_size", C.c_int), ("max_input_len", C.c_int), ("output_len", C.c_int), ("beam_width", C.c_int),
        ("top_k", C.c_void_p), ("n_top_k", C.c_int),
        ("top_p", C.c_void_p), ("n_top_p", C.c_int),
        ("beam_search_diversity_rate", C.c_void_p), ("n_beam_search_diversity_rate", C.c_int),
        ("temperature", C.c_void_p), ("n_temperature", C.c_int),
        ("len_penalty", C.c_void_p), ("n_len_penalty", C.c_int),
        ("repetition_penalty", C.c_void_p), ("n_repetition_penalty", C.c_int),
        ("random_seed", C.c_void_p), ("n_random_seed", C.c_int),
        ("min_length", C.c_void_p), ("n_min_length", C.c_int),
        ("stop_words_list", C.c_void_p), ("stop_words_len", C.c_int),
        ("optional_last_tokens", C.c_void_p), ("optional_last_tokens_count", C.c_int),
        ("return_cum_log_probs", C.c_int),
        ("callback", TOKEN_CALLBACK), ("callback_user", C.c_void_p),
        ("output_ids", C.c_void_p), ("sequence_lengths", C.c_void_p), ("cum_log_probs", C.c_void_p),
        ("debug_logits", C.c_void_p)]


class ForwardStats(C.Structure):
    _fields_ = [("prefill_ms", C.c_float), ("decode_ms", C.c_float), ("decode_steps", C.c_int),
                ("gemv_ms_sum", C.c_float), ("gemv_launches", C.c_long), ("gemv_bytes", C.c_double),
                ("gemv_kind", C.c_int), ("decode_path", C.c_int), ("prefill_overlap", C.c_int),
                ("prefill_ms_plain", C.c_float), ("prefill_ms_overlapped", C.c_float), ("window_allreduces", C.c_int),
                ("decode_overlap", C.c_int), ("decode_step_ms_plain", C.c_float), ("decode_step_ms_overlapped", C.c_float),
                ("persist_layout", C.c_int)]


# every symbol include/ftcf.h declares (tests/test_capi_host.py checks the list against the header and the library)
EXPORTED = [
    "ftcf_last_error", "ftcf_version", "ftcf_device_count", "ftcf_symmetric_quantize_int8",
    "ftcf_int8_rowmajor_to_tiled", "ftcf_int8_tiled_to_rowmajor", "ftcf_int8_cuda_sm80_to_rowmajor",
    "ftcf_int8_rowmajor_to_cuda_sm80", "ftcf_fp16_rowmajor_to_tiled",
    "ftcf_fpA_intB_gemm", "ftcf_fp16_gemm", "ftcf_lm_head", "ftcf_layernorm", "ftcf_add_bias_attn_ffn_residual",
    "ftcf_masked_multihead_attention", "ftcf_masked_multihead_attention_workspace", "ftcf_context_attention",
    "ftcf_comm_get_unique_id", "ftcf_comm_init", "ftcf_comm_destroy", "ftcf_comm_local_unique_id",
    "ftcf_comm_init_local", "ftcf_comm_init_host_exchange", "ftcf_comm_allreduce_sum",
    "ftcf_comm_allgather", "ftcf_gptneox_create", "ftcf_gptneox_forward", "ftcf_gptneox_begin", "ftcf_gptneox_step", "ftcf_gptneox_finish",
    "ftcf_gptneox_get_stats",
    "ftcf_gptneox_set_profiling", "ftcf_gptneox_destroy",
    "ftcf_batcher_create", "ftcf_batcher_submit", "ftcf_batcher_submit_ex", "ftcf_batcher_submit_beam", "ftcf_batcher_submit_beam_ex", "ftcf_batcher_beam_result", "ftcf_batcher_step", "ftcf_batcher_set_token_callback", "ftcf_batcher_status", "ftcf_batcher_cancel", "ftcf_batcher_destroy"]

_lib = None


def lib():
    """Loads libftcf.so.  Import torch first when both live in one process so that a single HIP runtime is shared."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FtcfError(-5, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc, gfx950). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.ftcf_last_error.restype = C.c_char_p
        _lib.ftcf_masked_multihead_attention_workspace.restype = C.c_size_t
    return _lib


def check(code):
    if code != 0:
        raise FtcfError(code, lib().ftcf_last_error().decode("utf-8", "replace"))


def device_count():
    return int(lib().ftcf_device_count())


def require_gpu():
    if device_count() <= 0:
        raise FtcfError(-5, "no HIP device visible; the MI355X engine has no CPU fallback")


def vp(x):
    """data pointer of a torch tensor / numpy array / int / None as c_void_p"""
    if x is None:
        return C.c_void_p(None)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)
