"""Helpers for the -m gpu parity tests: build engine inputs from the reference weight-list contract."""
import ctypes as C

import numpy as np
import torch

from fastertransformer4codefuse_amd import capi
from fastertransformer4codefuse_amd.gptneox_op import GptNeoXOp, symmetric_quantize_last_axis_of_batched_matrix_int8


def dev(a, dtype=torch.float16):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def make_op(cfg, w, int8_mode=0, tp=1, rank=0, comm=None, use_gptj_residual=True, dtype=torch.float16, op_class=None):
    """cfg: dict like tests.helpers ; w: reference-order list of float32 numpy arrays -- the full TP=1 layout, or with
    tp > 1 the shard of `rank` (tests.helpers.shard_weights)."""
    L = cfg["num_layer"]
    H = cfg["head_num"] * cfg["size_per_head"]
    I = cfg["inter_size"]
    shapes = {2: (H, 3 * H // tp), 4: (H // tp, H), 6: (H, I // tp), 8: (I // tp, H)}
    weights, int8_w, scales = [], [None] * (4 * L), [None] * (4 * L)
    for g in range(12):
        for l in range(L):
            a = w[g * L + l]
            if g in shapes and a.size:
                a = a.reshape(shapes[g])
                if int8_mode:
                    q, s = symmetric_quantize_last_axis_of_batched_matrix_int8(torch.from_numpy(a).half().contiguous())
                    qi = {2: 0, 4: 1, 6: 2, 8: 3}[g]
                    int8_w[qi * L + l] = q.cuda()
                    scales[qi * L + l] = s.cuda()
                    weights.append(torch.empty(0, dtype=dtype, device="cuda"))
                    continue
            weights.append(dev(a, dtype) if a.size else torch.empty(0, dtype=dtype, device="cuda"))
    V = cfg["vocab_size"]
    weights += [dev(w[12 * L].reshape(V, H), dtype), dev(w[12 * L + 1], dtype), dev(w[12 * L + 2], dtype),
                dev(w[12 * L + 3].reshape(V, H), dtype)]
    if not int8_mode:
        int8_w, scales = [], []
    # (op_class: the compiled libth_gptneox.GptNeoXOp instead of the ctypes one -- tests/test_gpu_th_modules.py)
    op = (op_class or GptNeoXOp)(comm, rank, cfg["head_num"], cfg["size_per_head"], I, L, V, cfg["rotary_dim"], cfg.get("start_id", 0),
                   cfg["end_id"], tp, 1, int8_mode, 1024, use_gptj_residual, weights, int8_w, scales)
    return op


def run_op(op, input_ids, input_lengths, out_len, V, return_logits=True, **kw):
    ids = torch.from_numpy(np.ascontiguousarray(input_ids, dtype=np.int32)).cuda()
    lens = torch.from_numpy(np.ascontiguousarray(input_lengths, dtype=np.int32)).cuda()
    B = ids.shape[0]
    dbg = torch.zeros((out_len, B, V), dtype=torch.float32, device="cuda") if return_logits else None
    t = lambda v, dt: None if v is None else torch.tensor(v if isinstance(v, (list, tuple)) else [v], dtype=dt)
    outs = op.forward(ids, lens, out_len, 1, t(kw.get("top_k"), torch.int32), t(kw.get("top_p"), torch.float32), None,
                      t(kw.get("temperature"), torch.float32), None, t(kw.get("repetition_penalty"), torch.float32),
                      t(kw.get("random_seed"), torch.int64),
                      None if kw.get("stop_words") is None else torch.from_numpy(
                          np.ascontiguousarray(kw["stop_words"], dtype=np.int32)).cuda(),
                      None if kw.get("optional_last_tokens") is None else torch.from_numpy(
                          np.ascontiguousarray(kw["optional_last_tokens"], dtype=np.int32)).cuda(),
                      kw.get("return_cum_log_probs", 1), kw.get("callback"), _debug_logits=dbg)
    torch.cuda.synchronize()
    res = {"output_ids": outs[0][:, 0, :].cpu().numpy(), "sequence_lengths": outs[1][:, 0].cpu().numpy()}
    if len(outs) > 2:
        res["cum_log_probs"] = outs[2][:, 0].cpu().numpy()
    if return_logits:
        res["logits"] = dbg.cpu().numpy()
    return res


def run_op_beam(op, input_ids, input_lengths, out_len, V, K, return_logits=False, **kw):
    """Beam search through GptNeoXOp.forward: output_ids [B, K, total], sequence_lengths / cum_log_probs [B, K]."""
    ids = torch.from_numpy(np.ascontiguousarray(input_ids, dtype=np.int32)).cuda()
    lens = torch.from_numpy(np.ascontiguousarray(input_lengths, dtype=np.int32)).cuda()
    B = ids.shape[0]
    dbg = torch.zeros((out_len, B * K, V), dtype=torch.float32, device="cuda") if return_logits else None
    t = lambda v, dt: None if v is None else torch.tensor(v if isinstance(v, (list, tuple)) else [v], dtype=dt)
    outs = op.forward(ids, lens, out_len, K, None, None, t(kw.get("beam_search_diversity_rate"), torch.float32),
                      t(kw.get("temperature"), torch.float32), t(kw.get("len_penalty"), torch.float32),
                      t(kw.get("repetition_penalty"), torch.float32), None,
                      None if kw.get("stop_words") is None else torch.from_numpy(
                          np.ascontiguousarray(kw["stop_words"], dtype=np.int32)).cuda(),
                      None if kw.get("optional_last_tokens") is None else torch.from_numpy(
                          np.ascontiguousarray(kw["optional_last_tokens"], dtype=np.int32)).cuda(),
                      1, kw.get("callback"), _debug_logits=dbg)
    torch.cuda.synchronize()
    res = {"output_ids": outs[0].cpu().numpy(), "sequence_lengths": outs[1].cpu().numpy(),
           "cum_log_probs": outs[2].cpu().numpy()}
    if return_logits:
        res["logits"] = dbg.cpu().numpy()
    return res


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
