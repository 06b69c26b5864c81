"""Random engine configurations against the oracle (the long form of tests/test_gpu_fuzz.py): heads, head size, inter size, layers,
vocabulary, rotary, batch 1..33, prompt 1..70, int8 / fp16, beams -- logits within 4 % of their range, arg max equal outside a near tie,
beam search replayed exactly on the GPU's logits.  Usage: python tools/fuzz_engine.py <seed> <cases>   (prints OK / BAD per case)"""
import os, sys, itertools, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import oracle as orc
from tests.helpers import random_model, weight_list_to_layers, quantize_layers
from tests import gpu_helpers as gh
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for case in range(ncase):
    dh = int(rng.choice([64, 128]))
    nh = int(rng.choice([1, 2, 3, 4, 5, 8]))
    H = nh * dh
    inter = int(rng.choice([64, 128, 192, 256, 512, 1024])) * int(rng.choice([1, 2, 3]))
    L = int(rng.choice([1, 2, 3]))
    V = int(rng.choice([64, 128, 200, 1000, 2048]))
    V = (V + 7) // 8 * 8
    rot = int(rng.choice([0, 16, 32, dh]))
    B = int(rng.choice([1, 2, 3, 4, 5, 9, 16, 17, 33]))
    S = int(rng.choice([1, 2, 7, 33, 70]))
    out = int(rng.choice([3, 9]))
    int8 = int(rng.choice([0, 1]))
    K = int(rng.choice([1, 1, 1, 2, 3]))
    if K > 1:
        B = min(B, 5)
    cfg = dict(head_num=nh, size_per_head=dh, inter_size=inter, num_layer=L, vocab_size=V, rotary_dim=rot, start_id=0, end_id=2)
    desc = f"case {case}: nh={nh} dh={dh} I={inter} L={L} V={V} rot={rot} B={B} S={S} out={out} int8={int8} K={K}"
    try:
        w = random_model(cfg, seed=case, std=0.05)
        layers, glob = weight_list_to_layers(cfg, w)
        if int8:
            layers = quantize_layers(layers)
        lens = rng.randint(1, S + 1, size=B).astype(np.int32)
        lens[0] = S
        ids = np.full((B, S), 2, dtype=np.int32)
        for b in range(B):
            ids[b, :lens[b]] = rng.randint(3, V, size=lens[b])
        op = gh.make_op(cfg, w, int8_mode=int8)
        m = orc.Model(dict(cfg, fp16=1, int8_mode=int8), layers, glob)
        if K == 1:
            r = gh.run_op(op, ids, lens, out, V, top_k=1)
            o = m.generate(ids, lens, out, return_logits=True)
            ok = True
            for b in range(B):
                for t in range(out):
                    ref = o["logits"][t, b]
                    scale = np.abs(ref).max()
                    if np.abs(r["logits"][t, b] - ref).max() > 0.04 * scale:
                        ok = False; print("  logits off", b, t, np.abs(r["logits"][t, b] - ref).max(), scale)
                        break
                    if o["output_ids"][b, lens[b] + t] == 2:
                        break  # row finished: later logits are not consumed
                    if r["output_ids"][b, lens[b] + t] != o["output_ids"][b, lens[b] + t]:
                        top2 = np.sort(ref)[-2:]
                        if top2[1] - top2[0] > 0.04 * scale:
                            ok = False; print("  token flip w/o tie", b, t)
                        break
                if not ok: break
            path = op.stats()["decode_path"]
        else:
            r = gh.run_op_beam(op, ids, lens, out, V, K, return_logits=True)
            from tests.test_gpu_beam import _replay
            p_ids, p_len, p_cum = _replay(cfg, ids, lens, out, K, r["logits"], orc.BeamParams(B))
            ok = r["output_ids"].tolist() == p_ids.tolist() and np.allclose(r["cum_log_probs"], p_cum, atol=1e-3, rtol=1e-4)
            o = m.generate_beam(ids, lens, out, K)
            agree = (r["output_ids"] == o["output_ids"]).mean()
            if agree < 0.6: ok = False; print("  beam agree", agree)
            path = op.stats()["decode_path"]
        print(("OK  " if ok else "BAD ") + desc + f" path={path}")
        bad += (not ok)
        del op
    except Exception as e:
        print("EXC " + desc + " :: " + str(e)[:300])
        bad += 1
print("bad", bad)
