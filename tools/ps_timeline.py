#!/usr/bin/env python3
"""Summarise the in-kernel stamps of the persistent decode kernel (FTCF_PERSIST_TS=<file>): [NB][L][8 waves][16]."""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
NB, L, W, K = np.frombuffer(raw[:16], dtype=np.int32)
ts = np.frombuffer(raw[16:], dtype=np.int64).reshape(NB, L, W, K).astype(np.float64) / 100.0  # us (100 MHz)
names = ["layer top", "x gathered", "LN done (+ctrl prime)", "P1 run end", "P1 epilogue out", "P3 set up", "qkv swept",
         "attn computed", "barrier after attn", "P3 primed (ctrl: ctx swept+primed)", "P3 run end",
         "pieces out + next setup", "x' merged", "attn merged (split-0 wgs)", "ctx swept", "partials swept (split-0 wgs)"]
lay = int(sys.argv[2]) if len(sys.argv) > 2 else L // 2
t0 = ts[:, lay, :, 0].min()
print(f"--- layer {lay}: us since the first wave entered the layer; median over workgroups (max in brackets) per wave ---")
print(f"{'':<36}" + "".join(f"   wave{w:<8d}" for w in range(W)))
for k in range(16 if lay > 0 else 15):
    row = []
    for w in range(W):
        v = ts[:, lay, w, k]
        v = v[v > 0] - t0
        row.append(f"{np.median(v):6.1f} ({v.max():5.1f})" if v.size else "      -       ")
    print(f"{k:2d} {names[k]:<33}" + " ".join(row))
if L > 1:
    per = ts[:, 1:, 0, 0] - ts[:, :-1, 0, 0]
    print("per-layer time (layer-top to layer-top), us: median %.2f  min %.2f  max %.2f" % (np.median(per), per.min(), per.max()))
e = ts[:, 0, :, 15]
if (e > 0).all():
    print("kernel entry -> first layer top (preamble), us: median %.2f max %.2f ; entry skew across workgroups %.2f" % (
        np.median(ts[:, 0, :, 0] - e), (ts[:, 0, :, 0] - e).max(), e.max() - e.min()))
    print("kernel entry (first wg) -> last layer's x' merged (last wg): %.2f us" % (ts[:, L - 1, :, 12].max() - e.min()))
# per XCD (workgroup b runs on XCD b % 8): are the late workgroups always the same ones?  Mean over layers 4 .. L-2 of the
# workgroup's stamp relative to the layer's first wave, median over the XCD's workgroups, for the streams' ends
if len(sys.argv) > 3 and sys.argv[3] == "xcd":
    sub = ts[:, 4:L - 1]
    rel = sub - sub[:, :, :, 0].min(axis=(0, 2))[None, :, None, None]
    for k, w in ((3, 2), (3, 7), (4, 2), (6, 2), (7, 2), (10, 2), (10, 7), (10, 0), (12, 2)):
        v = rel[:, :, w, k].mean(axis=1)  # [NB]
        print(f"{names[k]:<34} wave {w}: " + " ".join("xcd%d %5.1f" % (x, np.median(v[x::8])) for x in range(8))
              + "   | slowest wgs: " + " ".join(str(i) for i in np.argsort(v)[-6:]))
    # is a workgroup's lateness persistent across layers?  correlation of its P3-run-end offset between even and odd layers
    a = rel[:, 0::2, 2, 10].mean(axis=1)
    b = rel[:, 1::2, 2, 10].mean(axis=1)
    print("correlation of a workgroup's P3 run end (wave 2) between even and odd layers: %.2f; spread (p95 - p5) %.2f us" % (
        np.corrcoef(a, b)[0, 1], np.percentile(a, 95) - np.percentile(a, 5)))
    a = rel[:, 0::2, 2, 3].mean(axis=1)
    b = rel[:, 1::2, 2, 3].mean(axis=1)
    print("the same for the P1 run end: %.2f; spread %.2f us" % (np.corrcoef(a, b)[0, 1], np.percentile(a, 95) - np.percentile(a, 5)))
