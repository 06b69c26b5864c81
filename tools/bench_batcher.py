"""Throughput of the continuous-batching front end on the CodeFuse-13B-shaped int8 model: N requests of `--prompt-len` tokens
asking for `--new` tokens each, submitted at once to a batcher with `--slots` slots.  Prints one JSON line.  For comparison the
static batch of the same shape through GptNeoXOp.forward (bench.py --batch N) decodes in lockstep from one prefill."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastertransformer4codefuse_amd.batcher import ContinuousBatcher  # noqa: E402
from fastertransformer4codefuse_amd.gptneox_op import GptNeoXOp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--requests", type=int, default=32)
    ap.add_argument("--prompt-len", type=int, default=256)
    ap.add_argument("--new", type=int, default=128)
    ap.add_argument("--page", type=int, default=64)
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--late-prompt", type=int, default=0,
                    help="latency mode: `--slots` - 1 requests decode; after 10 iterations one request with a prompt of this many "
                         "tokens arrives; prints the longest gap between two tokens of a running request (token callback "
                         "timestamps) -- compare FTCF_BATCHER_PREFILL_CHUNK=0 (whole prompts) with the default 512")
    a = ap.parse_args()
    m = argparse.Namespace(layers=a.layers, heads=40, head_dim=128, inter=20480, vocab=100864, rotary=32, dtype="int8")
    dev = torch.device("cuda", 0)
    weights, int8_w, scales = bench.synth_weights(m, 1, dev)
    op = GptNeoXOp(None, 0, m.heads, m.head_dim, m.inter, m.layers, m.vocab, m.rotary, 0, 2, 1, 1, 1, 2048, True, weights,
                   int8_w, scales)
    per_seq = (max(a.prompt_len, a.late_prompt) + a.new + a.page - 1) // a.page
    cb = ContinuousBatcher(op, a.slots, a.page, per_seq * a.slots, max(a.prompt_len, a.late_prompt) + a.new)
    rng = np.random.RandomState(0)
    if a.late_prompt:
        stamps = {}
        cb.set_token_callback(lambda rid, tok, fin: stamps.setdefault(rid, []).append(time.perf_counter()))
        for _ in range(2):  # the first round warms every kernel shape, the second is measured
            stamps.clear()
            first = [cb.submit(rng.randint(3, m.vocab, size=a.prompt_len).tolist(), a.new) for _ in range(a.slots - 1)]
            it, late, t_adm = 0, None, None
            while cb.busy():
                if it == 10:
                    late = cb.submit(rng.randint(3, m.vocab, size=a.late_prompt).tolist(), 8)
                    t_adm = time.perf_counter()
                cb.step()
                it += 1
        gaps = [max(b - x for x, b in zip(stamps[r][:-1], stamps[r][1:])) for r in first]
        steady = sorted(b - x for r in first for x, b in zip(stamps[r][:-1], stamps[r][1:]))
        print(json.dumps({"workload": f"13B-shaped int8, {a.slots - 1} requests decoding ({a.prompt_len}-in), one {a.late_prompt}-token "
                                      f"prompt arrives", "prefill_chunk": os.environ.get("FTCF_BATCHER_PREFILL_CHUNK", "512"),
                          "longest_token_gap_ms": round(max(gaps) * 1e3, 2),
                          "median_token_gap_ms": round(steady[len(steady) // 2] * 1e3, 2),
                          "late_request_first_token_ms": round((stamps[late][0] - t_adm) * 1e3, 2)}))
        return
    # varied lengths: requests leave at different times, slots are refilled from the queue
    reqs = [(rng.randint(3, m.vocab, size=a.prompt_len).tolist(), int(rng.randint(a.new // 2, a.new + 1))) for _ in range(a.requests)]
    cb.submit(reqs[0][0], 4)  # warm-up
    cb.run_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p, n in reqs:
        cb.submit(p, n)
    toks, iters = 0, 0
    while cb.busy():
        toks += len(cb.step())
        iters += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": f"13B-shaped int8 (L={a.layers}), {a.requests} requests x {a.prompt_len}-in / {a.new // 2}..{a.new}-out, "
                                  f"{a.slots} slots, {a.page}-token pages", "generated_tokens": toks, "seconds": round(dt, 3),
                      "tokens_per_sec": round(toks / dt, 1), "iterations": iters, "ms_per_iteration": round(dt / iters * 1e3, 3)}))


if __name__ == "__main__":
    main()
