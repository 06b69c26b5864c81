#!/usr/bin/env python3
"""Summarise the in-kernel stamps of the rows kernel (FTCF_PERSIST_TS=<file> with a 3..16-row request): [NB][L][8 waves][16].
Wave 0 is a workgroup's control wave, waves 1..7 stream.  Usage: rows_timeline.py <file> [layer]"""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
NB, L, W, K = np.frombuffer(raw[:16], dtype=np.int32)
ts = np.frombuffer(raw[16:], dtype=np.int64).reshape(NB, L, W, K).astype(np.float64) / 100.0  # us (100 MHz)
S = ["layer top", "go QKV seen", "QKV pass out", "FFN1 pass out", "go AT seen", "AT unit done", "go FFN2 seen", "FFN2 pass out",
     "go OUT seen", "OUT pass out"]
C = ["layer top", "x' swept, stats, go QKV", "QKV partials in", "q|k|v published", "fq polled, go AT", "mid published",
     "fm polled, go FFN2", "ctx merged + published", "fc polled, go OUT", "p2 published", "p3 published", "x' merged"]
lay = int(sys.argv[2]) if len(sys.argv) > 2 else L // 2
t0 = ts[:, lay, :, 0][ts[:, lay, :, 0] > 0].min()


def fmt(v):
    v = v[v > 0] - t0
    return f"{np.percentile(v, 5):6.1f} {np.median(v):6.1f} {v.max():6.1f}" if v.size else "     -      -      -"


print(f"--- layer {lay}: us since the first wave entered the layer: p5 / median / max over workgroups ---")
print("streamer waves (1..7 pooled)")
for k, n in enumerate(S):
    print(f"  {k:2d} {n:<30} {fmt(ts[:, lay, 1:, k].ravel())}")
print("control wave")
for k, n in enumerate(C):
    print(f"  {k:2d} {n:<30} {fmt(ts[:, lay, 0, k].ravel())}")
# durations inside a streamer wave, median over waves / workgroups / layers 2..L-1
sub = ts[:, 2:, 1:, :]
ok = (sub[..., :10] > 0).all(axis=-1)
d = np.diff(sub[..., :10], axis=-1)[ok]
print("streamer wave, median (p95) us per segment over layers 2..: " + " | ".join(
    f"{S[k + 1]} {np.median(d[:, k]):.1f} ({np.percentile(d[:, k], 95):.1f})" for k in range(9)))
subc = ts[:, 2:, 0, :12]
okc = (subc > 0).all(axis=-1)
dc = np.diff(subc, axis=-1)[okc]
print("control wave, median (p95) us per segment: " + " | ".join(
    f"{C[k + 1]} {np.median(dc[:, k]):.1f} ({np.percentile(dc[:, k], 95):.1f})" for k in range(11)))
if L > 1:
    per = ts[:, 1:, 0, 0] - ts[:, :-1, 0, 0]
    print("per-layer time (control wave layer-top to layer-top), us: median %.2f  min %.2f  max %.2f" % (
        np.median(per), per.min(), per.max()))
    first = ts[:, :, :, 0].copy()
    first[first <= 0] = np.inf
    tops = first.min(axis=(0, 2))
    print("layer period (first wave in to first wave in), us: median %.2f" % np.median(np.diff(tops)))
# ---- who is late?  stamps relative to the layer's first wave, mean over layers 4..L-1 ----
if L > 6:
    sub = ts[:, 4:, :, :]
    first = sub[:, :, :, 0].copy()
    first[first <= 0] = np.inf
    rel = sub - first.min(axis=(0, 2))[None, :, None, None]
    print("streamer stamps by wave index (mean over workgroups and layers): " + " | ".join(
        S[k] + " " + " ".join(f"{rel[:, :, w, k].mean():.1f}" for w in range(1, W)) for k in (2, 3, 5, 7, 9)))
    for k in (2, 3, 5, 7, 9):
        v = rel[:, :, 1:, k].max(axis=2).mean(axis=1)  # [NB]: the workgroup's last wave, mean over layers
        a = rel[:, 0::2, 1:, k].max(axis=2).mean(axis=1)
        b = rel[:, 1::2, 1:, k].max(axis=2).mean(axis=1)
        print(f"{S[k]:<14} (wg's last wave): " + " ".join("xcd%d %5.1f" % (x, np.median(v[x::8])) for x in range(8))
              + f" | p5 {np.percentile(v, 5):.1f} p95 {np.percentile(v, 95):.1f} | persistence {np.corrcoef(a, b)[0, 1]:.2f}"
              + " | slowest wgs: " + " ".join(str(i) for i in np.argsort(v)[-8:]))
    # per layer: the slowest workgroup's stamp minus the median workgroup's (what every hand-off waits for)
    for k in (2, 3, 5, 7, 9):
        m = rel[:, :, 1:, k].max(axis=2)  # [NB][layers]
        print(f"{S[k]:<14} slowest wg - median wg per layer: mean {np.mean(m.max(axis=0) - np.median(m, axis=0)):.1f} us")
