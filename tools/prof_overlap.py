#!/usr/bin/env python3
"""How much kernel time of a rocprofv3 --kernel-trace (rocpd SQLite) ran CONCURRENTLY: per stream the busy time, and over all
streams the sum of kernel durations against the length of the union of their intervals (sum / union > 1: kernels overlapped).
Only dispatches of the named kernels (substring match, default: the batched decode layer's) are counted.
Usage: python tools/prof_overlap.py <results.db> [name-substring ...]"""
import sqlite3
import sys


def main(db, names):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    rows = list(c.execute(f"select s.{name_col}, d.stream_id, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    rows = [r for r in rows if any(n in r[0] for n in names)]
    per = {}
    for n, st, a, b in rows:
        per.setdefault(st, []).append((a, b))
    tot = sum(b - a for _, _, a, b in rows)
    ev = sorted((a, b) for _, _, a, b in rows)
    union, cur_a, cur_b = 0, None, None
    for a, b in ev:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        union += cur_b - cur_a
    print(f"# {db}: {len(rows)} dispatches of {names}")
    for st, iv in sorted(per.items()):
        print(f"stream {st}: {len(iv)} kernels, busy {sum(b - a for a, b in iv) / 1e6:.3f} ms")
    print(f"sum of kernel durations {tot / 1e6:.3f} ms, union of their intervals {union / 1e6:.3f} ms, concurrency {tot / max(union, 1):.2f}x")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:] or ["k_gemm_smallm_burst", "k_mmha_split", "k_residual_dual_ln", "AllReduce", "allreduce"])
