#!/usr/bin/env python3
"""Summarise the in-kernel stamps of the persistent decode kernel (FTCF_PERSIST_TS=<file>)."""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
NB, L, W, K = np.frombuffer(raw[:16], dtype=np.int32)
ts = np.frombuffer(raw[16:], dtype=np.int64).reshape(NB, L, W, K).astype(np.float64) / 100.0  # us (100 MHz)
t0 = ts[:, 0, 0, 0].min()
ts = ts - t0
names = ["layer top", "x gathered", "LN done/primed", "P1 run end", "P1 signalled", "P1 wait done", "mid staged",
         "attn computed", "ctx wait done", "ctx staged/primed", "P3 run end", "pieces out + next setup", "merge done"]
lay = range(L) if L <= 4 else [0, 1, L // 2, L - 1]
for l in lay:
    print(f"--- layer {l} (us since kernel start; min / median / max over workgroups) ---")
    for w, wn in ((0, "ctrl wave0"), (1, "strm wave2")):
        for k in range(13):
            v = ts[:, l, w, k]
            v = v[v > 0] if k else v
            if v.size:
                print(f"  {wn} {k:2d} {names[k]:<26} {v.min():9.2f} {np.median(v):9.2f} {v.max():9.2f}")
if L > 1:
    per = ts[:, 1:, 0, 0] - ts[:, :-1, 0, 0]
    print("per-layer time (layer-top to layer-top), us: median %.2f  min %.2f  max %.2f" % (np.median(per), per.min(), per.max()))
d = np.diff(ts[:, :, :, :13], axis=3)
print("median phase durations over all layers / workgroups (us):")
for w, wn in ((0, "ctrl"), (1, "strm")):
    print(" ", wn, " ".join(f"{k}->{k+1}:{np.median(d[:, 1:, w, k]):.2f}" for k in range(12)))
