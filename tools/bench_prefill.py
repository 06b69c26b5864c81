"""Prompt-phase time by prompt length on the CodeFuse-13B shape (synthetic weights, bs = 1): for every length one warm
request, then `reps` requests of (prefill + 2 tokens); prints the engine's own prefill time (HIP events around the context
phase, ftcf_gptneox_get_stats) -- median of the reps -- as one JSON line per length.
Usage: python tools/bench_prefill.py [--lens 17,33,64,65,96,128,192,256,512] [--dtype int8] [--fake-tp N]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastertransformer4codefuse_amd import capi  # noqa: E402
from fastertransformer4codefuse_amd.gptneox_op import GptNeoXOp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lens", default="17,33,48,64,65,96,128,160,192,256,384,512")
    ap.add_argument("--dtype", default="int8")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--fake-tp", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    x = ap.parse_args()
    sys.argv = [sys.argv[0], "--dtype", x.dtype]
    a = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tp, group = 1, None
    if x.fake_tp > 1:
        tp, group = x.fake_tp, object()
        os.environ["FTCF_FAKE_TP"] = "1"
    weights, int8_w, scales = bench.synth_weights(a, tp, dev)
    op = GptNeoXOp(group, 0, a.heads, a.head_dim, a.inter, a.layers, a.vocab, a.rotary, 0, 2, tp, 1,
                   1 if a.dtype == "int8" else 0, 2048, True, weights, int8_w, scales)
    L = capi.lib()
    B = x.batch
    for S in [int(v) for v in x.lens.split(",")]:
        gi = torch.Generator().manual_seed(42 + S)
        ids = torch.randint(3, a.vocab, (B, S), generator=gi, dtype=torch.int32).to(dev)
        lens = torch.full((B,), S, dtype=torch.int32, device=dev)
        out_ids = torch.empty((B, 1, S + 2), dtype=torch.int32, device=dev)
        seq = torch.empty((B, 1), dtype=torch.int32, device=dev)
        top_k = np.array([1], np.int32)
        fa = capi.ForwardArgs()
        fa.input_ids, fa.input_lengths = ids.data_ptr(), lens.data_ptr()
        fa.batch_size, fa.max_input_len, fa.output_len, fa.beam_width = B, S, 2, 1
        fa.top_k, fa.n_top_k = top_k.ctypes.data, 1
        fa.output_ids, fa.sequence_lengths = out_ids.data_ptr(), seq.data_ptr()
        ms = []
        for r in range(x.reps + 1):
            capi.check(L.ftcf_gptneox_forward(op._h, C.byref(fa)))
            torch.cuda.synchronize()
            if r:
                ms.append(op.stats()["prefill_ms"])
        ms.sort()
        print(json.dumps({"prompt_len": S, "batch": B, "dtype": a.dtype, "tp_shard": tp, "prefill_ms": round(ms[len(ms) // 2], 3),
                          "min": round(ms[0], 3), "max": round(ms[-1], 3)}), flush=True)


if __name__ == "__main__":
    main()
