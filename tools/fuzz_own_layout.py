"""Random shapes and launch grids for the one- / two-row persistent kernel against the oracle, with the own-group layout of its
out-proj / FFN2 stage forced wherever the shape divides (FTCF_PERSIST_OWN=1, FTCF_PERSIST_NB = the grid): hidden sizes 64..1024, grids
of 4..64 workgroups, one and two rows, fp16 / int8.  Prints which layout ran (ftcf_forward_stats.persist_layout) and OK / BAD per case.
Usage: python tools/fuzz_own_layout.py <seed> <cases>"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle as orc
from tests.helpers import random_model, weight_list_to_layers, quantize_layers
from tests import gpu_helpers as gh


def run(seed, ncase, verbose=True):
    """-> (cases that differ from the oracle, cases in which the own-group layout ran)"""
    rng = np.random.RandomState(seed)
    bad, own = 0, 0
    keep = {k: os.environ.get(k) for k in ("FTCF_PERSIST_NB", "FTCF_PERSIST_OWN")}
    try:
        for case in range(ncase):
            b, o = _case(rng, case, verbose)
            bad += b
            own += o
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return bad, own


def _case(rng, case, verbose):
    bad, own = 0, 0
    if True:
        dh = int(rng.choice([64, 128]))
        nh = int(rng.choice([1, 2, 4, 8]))
        H = nh * dh
        inter = int(rng.choice([1, 2, 3, 4])) * H
        L = int(rng.choice([1, 2, 3]))
        V, rot = 256, int(rng.choice([0, 16, 32]))
        B = int(rng.choice([1, 2]))
        S, out = int(rng.choice([1, 9, 40])), 5
        int8 = int(rng.choice([0, 1]))
        NG = H // 16
        nb = int(rng.choice([n for n in (4, 6, 8, 12, 16, 24, 32, 48, 64) if n >= B * nh]))
        os.environ["FTCF_PERSIST_NB"] = str(nb)
        os.environ["FTCF_PERSIST_OWN"] = "1"
        cfg = dict(head_num=nh, size_per_head=dh, inter_size=inter, num_layer=L, vocab_size=V, rotary_dim=rot, start_id=0, end_id=2)
        desc = f"case {case}: H={H} (NG {NG}) I={inter} L={L} B={B} S={S} int8={int8} grid={nb}"
        try:
            w = random_model(cfg, seed=1000 + case, std=0.05)
            layers, glob = weight_list_to_layers(cfg, w)
            if int8:
                layers = quantize_layers(layers)
            lens = rng.randint(1, S + 1, size=B).astype(np.int32)
            lens[0] = S
            ids = np.full((B, S), 2, dtype=np.int32)
            for b in range(B):
                ids[b, :lens[b]] = rng.randint(3, V, size=lens[b])
            op = gh.make_op(cfg, w, int8_mode=int8)
            r = gh.run_op(op, ids, lens, out, V, top_k=1)
            o = orc.Model(dict(cfg, fp16=1, int8_mode=int8), layers, glob).generate(ids, lens, out, return_logits=True)
            st = op.stats()
            ok = True
            for b in range(B):
                for t in range(out):
                    ref = o["logits"][t, b]
                    scale = np.abs(ref).max()
                    if np.abs(r["logits"][t, b] - ref).max() > 0.02 * scale:
                        ok = False
                        print("  logits off", b, t, np.abs(r["logits"][t, b] - ref).max() / scale)
                        break
                    if o["output_ids"][b, lens[b] + t] == 2:
                        break
                    if r["output_ids"][b, lens[b] + t] != o["output_ids"][b, lens[b] + t]:
                        top2 = np.sort(ref)[-2:]
                        ok = ok and top2[1] - top2[0] <= 0.02 * scale
                        break
            own += st["persist_layout"] == 1
            if verbose or not ok:
                print(("OK  " if ok else "BAD ") + desc + f" path={st['decode_path']} layout={st['persist_layout']}")
            bad += not ok
            del op
        except Exception as e:  # noqa: BLE001
            print("EXC " + desc + " :: " + str(e)[:300])
            bad += 1
    return bad, own


if __name__ == "__main__":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    nbad, nown = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, n)
    print("bad", nbad, "| own-group layout ran in", nown, "of", n)
