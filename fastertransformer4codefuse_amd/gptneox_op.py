"""`GptNeoXOp` -- Python face of the engine with the reference's pybind11 signature.

Mirrors `th_op/gptneox/GptNeoXOp.cc:25-185` (`GptNeoXOp::GptNeoXOp`, `GptNeoXOp::forward`) and
`GptNeoXOp.h:69-231,246-381` (`FTGptNeoX<T>`): same positional arguments, same tensor checks, same outputs.  All
compute goes through libftcf.so (C ABI); nothing here touches tensor contents on the CPU.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi


def _check_input(t, name, dtype=None):
    # th_utils.h:32-49 CHECK_INPUT / CHECK_TH_CUDA / CHECK_CONTIGUOUS
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} has invalid dtype {t.dtype}, expected {dtype}")


class LocalTensorParallelGroup:
    """Test infrastructure: the ranks of a tensor-parallel job inside ONE process on ONE device (one host thread per rank,
    see include/ftcf.h `ftcf_comm_init_local`).  Pass one shared instance as `comm` to every rank's GptNeoXOp."""

    def __init__(self):
        ids = np.zeros(capi.UNIQUE_ID_BYTES, dtype=np.uint8)
        capi.check(capi.lib().ftcf_comm_local_unique_id(ids.ctypes.data_as(C.POINTER(C.c_uint8))))
        self.id = ids


def init_tensor_parallel_comm(group, rank, world_size, device):
    """Counterpart of nccl_inherit::ftNcclInitialize (th_op/gptneox/utils/nccl_inherit_utils.cc:25-68).

    The reference reaches into ProcessGroupNCCL's protected broadcastUniqueNCCLID through a reinterpret_cast
    (`HackGroupNCCL`); here rank 0 creates the RCCL unique id and it travels as a plain byte tensor through the
    caller's process group (any backend)."""
    if isinstance(group, LocalTensorParallelGroup):
        comm = C.c_void_p()
        capi.check(capi.lib().ftcf_comm_init_local(group.id.ctypes.data_as(C.POINTER(C.c_uint8)), world_size, rank, device,
                                                   C.byref(comm)))
        return comm
    import torch.distributed as dist
    ids = np.zeros(capi.UNIQUE_ID_BYTES, dtype=np.uint8)
    if os.environ.get("FTCF_TP_EXCHANGE", "") == "host":
        # HOST-EXCHANGE communicator (include/ftcf.h ftcf_comm_init_host_exchange): no RCCL communicator at all -- the
        # bootstrap, the hipIpc handles of the exchange windows, agreements and barriers (and, host staged, the prefill
        # collectives) are all-gathers of host bytes over the caller's process group (gloo: CPU tensors).  What
        # nccl_inherit_utils.cc:25-68 does with the caller's ProcessGroup for the bootstrap, taken one step further; it lets
        # two PROCESSES share one device, i.e. the inter-process path of the in-kernel all-reduce run on a single-GPU box.
        return _HostExchangeComm(group, rank, world_size, device).handle
    if os.environ.get("FTCF_FAKE_TP") == "1":
        # timing aid (bench.py --fake-tp N): ONE process runs rank 0's shard of a TP=N model over a 1-rank communicator --
        # the kernels and collectives of a rank are all launched, only the peers are missing (outputs are meaningless)
        capi.check(capi.lib().ftcf_comm_get_unique_id(ids.ctypes.data_as(C.POINTER(C.c_uint8))))
        comm = C.c_void_p()
        capi.check(capi.lib().ftcf_comm_init(ids.ctypes.data_as(C.POINTER(C.c_uint8)), 1, 0, device, C.byref(comm)))
        return comm
    if rank == 0:
        capi.check(capi.lib().ftcf_comm_get_unique_id(ids.ctypes.data_as(C.POINTER(C.c_uint8))))
    backend = dist.get_backend(group)
    t = torch.from_numpy(ids)
    if backend == "nccl":
        t = t.cuda(device)
    src = dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0
    dist.broadcast(t, src=src, group=group)
    ids = t.cpu().numpy().copy()
    comm = C.c_void_p()
    capi.check(capi.lib().ftcf_comm_init(ids.ctypes.data_as(C.POINTER(C.c_uint8)), world_size, rank, device,
                                         C.byref(comm)))
    return comm


class _HostExchangeComm:
    """Owns the ctypes callback (it must outlive the communicator) of a host-exchange communicator."""
    _alive = []

    def __init__(self, group, rank, world_size, device):
        import torch.distributed as dist
        self.group, self.world = group, world_size

        def allgather(_user, send, recv, nbytes):
            try:
                src = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
                out = torch.empty(nbytes * self.world, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, src, group=self.group)
                C.memmove(recv, out.data_ptr(), nbytes * self.world)
                return 0
            except Exception as e:  # noqa: BLE001  (an exception must not unwind through the C caller)
                print(f"[ftcf] host-exchange all-gather failed: {e!r}", flush=True)
                return 1

        self._cb = capi.HOST_ALLGATHER(allgather)
        self.handle = C.c_void_p()
        capi.check(capi.lib().ftcf_comm_init_host_exchange(world_size, rank, device, self._cb, None, C.byref(self.handle)))
        _HostExchangeComm._alive.append(self)


class GptNeoXOp:
    def __init__(self, comm, rank, head_num, size_per_head, inter_size, layer_num, vocab_size, rotary_embedding_dim,
                 start_id, end_id, tensor_para_size, pipeline_para_size, int8_mode, max_seq_len, use_gptj_residual,
                 weights, int8_weights, scale):
        capi.require_gpu()
        self.rank_ = int(rank)
        st = weights[0].dtype  # GptNeoXOp.cc:46: the dtype of weights[0] selects the engine
        for i, t in enumerate(weights):
            _check_input(t, f"weights[{i}]")
            if t.numel() and t.dtype != st:
                raise RuntimeError("Invalid datatype. All weights must have the same dtype")
        if st not in (torch.float16, torch.float32):  # GptNeoXOp.cc:56-105: FTGptNeoX<float> / FTGptNeoX<half>
            raise RuntimeError("Wrong Tensor type.")
        if st == torch.float32 and int(int8_mode) != 0:
            raise RuntimeError("int8_mode needs half weights (CutlassFpAIntBGemmRunner<half, uint8_t>)")
        self.end_id_ = int(end_id)
        self.vocab_size_ = int(vocab_size)
        self.tensor_para_size_ = int(tensor_para_size)
        self.device_ = weights[0].device.index if weights[0].device.index is not None else torch.cuda.current_device()
        # keep the tensors alive, like GptNeoXOp.h:402-404
        self.weights = list(weights)
        self.int8_weights = list(int8_weights)
        self.scale = list(scale)
        for i, t in enumerate(self.int8_weights):
            _check_input(t, f"int8_weights[{i}]", torch.int8)
        for i, t in enumerate(self.scale):
            _check_input(t, f"scale[{i}]", torch.float16)

        self._comm = None
        if tensor_para_size > 1:
            self._comm = init_tensor_parallel_comm(comm, self.rank_ % tensor_para_size, tensor_para_size, self.device_)

        def ptr_array(ts):
            arr = (C.c_void_p * max(1, len(ts)))()
            for i, t in enumerate(ts):
                arr[i] = t.data_ptr() if t.numel() > 0 else None
            return arr

        self._w_arr, self._q_arr, self._s_arr = ptr_array(self.weights), ptr_array(self.int8_weights), ptr_array(self.scale)
        w = capi.GptNeoXWeights(C.cast(self._w_arr, C.POINTER(C.c_void_p)), len(self.weights),
                                C.cast(self._q_arr, C.POINTER(C.c_void_p)), len(self.int8_weights),
                                C.cast(self._s_arr, C.POINTER(C.c_void_p)), len(self.scale))
        stream = torch.cuda.current_stream(self.device_).cuda_stream  # GptNeoXOp.h:180-185
        cfg = capi.GptNeoXConfig(int(head_num), int(size_per_head), int(inter_size), int(layer_num), int(vocab_size),
                                 int(rotary_embedding_dim), int(start_id), int(end_id), int(tensor_para_size),
                                 self.rank_ % int(tensor_para_size), int(pipeline_para_size), int(int8_mode),
                                 capi.FP32 if st == torch.float32 else capi.FP16, int(bool(use_gptj_residual)),
                                 int(self.device_), stream,
                                 self._comm, 1)
        self._h = C.c_void_p()
        capi.check(capi.lib().ftcf_gptneox_create(C.byref(cfg), C.byref(w), C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                capi.lib().ftcf_gptneox_destroy(self._h)
                self._h = None
            if getattr(self, "_comm", None):
                capi.lib().ftcf_comm_destroy(self._comm)
                self._comm = None
        except Exception:
            pass

    # GptNeoXOp::forward (GptNeoXOp.cc:113-185)
    def forward(self, input_ids, input_lengths, output_len, beam_width=None, top_k=None, top_p=None,
                beam_search_diversity_rate=None, temperature=None, len_penalty=None, repetition_penalty=None,
                random_seed=None, stop_words_list=None, optional_last_tokens=None, return_cum_log_probs=None,
                callback=None, _debug_logits=None):
        _check_input(input_ids, "input_ids", torch.int32)
        if input_ids.dim() != 2:
            raise RuntimeError("input_ids must be a matrix")
        _check_input(input_lengths, "input_lengths", torch.int32)
        if input_lengths.dim() != 1:
            raise RuntimeError("input_lengths must be a vector")
        return_cum_log_probs = int(return_cum_log_probs) if return_cum_log_probs is not None else 0
        if return_cum_log_probs not in (0, 1):
            raise RuntimeError("return_cum_log_probs should be 0 (no return) or 1 (the cumulative log probs of "
                               "generated sequences)")
        beam_width = int(beam_width) if beam_width is not None else 1
        B, S = int(input_ids.size(0)), int(input_ids.size(1))
        total = S + int(output_len)
        dev = input_ids.device
        output_ids = torch.empty((B, beam_width, total), dtype=torch.int32, device=dev)
        sequence_lengths = torch.empty((B, beam_width), dtype=torch.int32, device=dev)
        cum_log_probs = torch.empty((B, beam_width), dtype=torch.float32, device=dev) if return_cum_log_probs else None

        keep = []

        def host(t, np_dtype, name):
            if t is None:
                return None, 0
            if t.is_cuda:
                raise RuntimeError(f"{name} must be a CPU tensor")
            a = np.ascontiguousarray(t.detach().numpy().astype(np_dtype).reshape(-1))
            keep.append(a)
            return a.ctypes.data, int(a.size)

        a = capi.ForwardArgs()
        a.input_ids, a.input_lengths = input_ids.data_ptr(), input_lengths.data_ptr()
        a.batch_size, a.max_input_len, a.output_len, a.beam_width = B, S, int(output_len), beam_width
        a.top_k, a.n_top_k = host(top_k, np.int32, "top_k")
        a.top_p, a.n_top_p = host(top_p, np.float32, "top_p")
        a.beam_search_diversity_rate, a.n_beam_search_diversity_rate = host(
            beam_search_diversity_rate, np.float32, "beam_search_diversity_rate")
        a.temperature, a.n_temperature = host(temperature, np.float32, "temperature")
        a.len_penalty, a.n_len_penalty = host(len_penalty, np.float32, "len_penalty")
        a.repetition_penalty, a.n_repetition_penalty = host(repetition_penalty, np.float32, "repetition_penalty")
        a.random_seed, a.n_random_seed = host(random_seed, np.uint64, "random_seed")
        if stop_words_list is not None:
            _check_input(stop_words_list, "stop_words_list", torch.int32)
            a.stop_words_list, a.stop_words_len = stop_words_list.data_ptr(), int(stop_words_list.size(2))
        if optional_last_tokens is not None:
            _check_input(optional_last_tokens, "optional_last_tokens", torch.int32)
            a.optional_last_tokens = optional_last_tokens.data_ptr()
            a.optional_last_tokens_count = int(optional_last_tokens.size(1))
        a.return_cum_log_probs = return_cum_log_probs
        if callback is not None:
            def _cb(tokens, idxs, batch, beam, _user):
                # pybind_callback_utils.cc:79-103: {"last_tokens": [[...]*beam]*batch, "idxs": ...}
                lt = [[int(tokens[b * beam + w]) for w in range(beam)] for b in range(batch)]
                ix = [[int(idxs[b * beam + w]) for w in range(beam)] for b in range(batch)]
                callback({"last_tokens": lt, "idxs": ix})
            cb = capi.TOKEN_CALLBACK(_cb)
            keep.append(cb)
            a.callback = cb
        a.output_ids, a.sequence_lengths = output_ids.data_ptr(), sequence_lengths.data_ptr()
        a.cum_log_probs = cum_log_probs.data_ptr() if cum_log_probs is not None else None
        if _debug_logits is not None:
            a.debug_logits = _debug_logits.data_ptr()
        capi.check(capi.lib().ftcf_gptneox_forward(self._h, C.byref(a)))
        if return_cum_log_probs:
            return [output_ids, sequence_lengths, cum_log_probs]
        return [output_ids, sequence_lengths]

    # --- extras (not part of the reference surface) ---
    def stats(self):
        s = capi.ForwardStats()
        capi.check(capi.lib().ftcf_gptneox_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in capi.ForwardStats._fields_}

    def set_profiling(self, enabled):
        capi.check(capi.lib().ftcf_gptneox_set_profiling(self._h, int(bool(enabled))))


def symmetric_quantize_last_axis_of_batched_matrix_int8(weight):
    """`libth_common.symmetric_quantize_last_axis_of_batched_matrix_int8` (WeightOnlyQuantOps.cc:230-233,344-349).

    weight: CPU contiguous [K,N] or [E,K,N], fp32 / fp16.  Returns [int8 tensor of the same shape in the engine's
    private gfx950 tile layout, scales [N] (or [E,N]) in the weight dtype]."""
    if weight.is_cuda:
        raise RuntimeError("weight must be a CPU tensor")  # CHECK_CPU
    if not weight.is_contiguous():
        raise RuntimeError("weight must be contiguous")
    if weight.numel() == 0:
        raise RuntimeError("weight should not be empty tensor")
    if weight.dim() not in (2, 3):
        raise RuntimeError("Invalid dim. The dim of weight should be 2 or 3")
    if weight.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise RuntimeError("Invalid datatype. Weight must be FP16 or BF16")
    E = 1 if weight.dim() == 2 else int(weight.size(0))
    K, N = int(weight.size(-2)), int(weight.size(-1))
    q = torch.empty(weight.shape, dtype=torch.int8)
    scales = torch.empty((N,) if weight.dim() == 2 else (E, N), dtype=weight.dtype)
    capi.check(capi.lib().ftcf_symmetric_quantize_int8(
        C.c_void_p(weight.data_ptr()),
        {torch.float16: capi.FP16, torch.bfloat16: capi.BF16, torch.float32: capi.FP32}[weight.dtype], C.c_size_t(E),
        C.c_size_t(K), C.c_size_t(N), C.c_void_p(q.data_ptr()), C.c_void_p(scales.data_ptr())))
    return [q, scales]
