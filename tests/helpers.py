"""Shared test helpers (oracle model construction from the reference's weight-list contract)."""
import os

import numpy as np

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tiny():
    z = np.load(os.path.join(GOLDEN, "tiny_gptneox_fp32.npz"))
    nh, dh, inter, L, V, rot, start_id, end_id = [int(v) for v in z["cfg"]]
    cfg = dict(head_num=nh, size_per_head=dh, inter_size=inter, num_layer=L, vocab_size=V, rotary_dim=rot,
               start_id=start_id, end_id=end_id)
    w = [z[f"w{i:03d}"].astype(np.float32) for i in range(12 * L + 4)]
    return cfg, w, z


def weight_list_to_layers(cfg, w, tp=1):
    """Reference weight-list contract (GptNeoXOp.h:121-174, codefuse_example.py:244-268):
    [ln1.beta xL, ln1.gamma xL, qkv.kernel xL, qkv.bias xL, out.kernel xL, out.bias xL, ffn1.kernel xL, ffn1.bias xL,
     ffn2.kernel xL, ffn2.bias xL, ln2.beta xL, ln2.gamma xL, wte, final_ln.gamma, final_ln.beta, lm_head]"""
    L = cfg["num_layer"]
    H = cfg["head_num"] * cfg["size_per_head"]
    hl = H // tp
    il = cfg["inter_size"] // tp
    layers = []
    for l in range(L):
        g = lambda k: w[k * L + l]
        layers.append(dict(
            ln1_b=g(0), ln1_g=g(1), qkv_w=g(2).reshape(H, 3 * hl) if g(2).size else None, qkv_b=g(3).reshape(-1),
            out_w=g(4).reshape(hl, H) if g(4).size else None, out_b=g(5) if g(5).size else None,
            ffn1_w=g(6).reshape(H, il) if g(6).size else None, ffn1_b=g(7),
            ffn2_w=g(8).reshape(il, H) if g(8).size else None, ffn2_b=g(9), ln2_b=g(10), ln2_g=g(11)))
    glob = dict(wte=w[12 * L].reshape(cfg["vocab_size"], H), final_ln_g=w[12 * L + 1], final_ln_b=w[12 * L + 2],
                lm_head=w[12 * L + 3].reshape(cfg["vocab_size"], H))
    return layers, glob


def quantize_layers(layers, weight_is_half=True):
    """Adds the weight-only int8 tensors (unprocessed layout) to oracle layer dicts."""
    out = []
    for lay in layers:
        d = dict(lay)
        for k in ("qkv", "out", "ffn1", "ffn2"):
            q, s = orc.symmetric_quantize_int8(lay[k + "_w"], weight_is_half)
            d[k + "_q"], d[k + "_s"] = q, s
        out.append(d)
    return out


def random_model(cfg, seed=0, std=0.05, fp16=True, tp=1, rank=0):
    """Random weights in the reference's weight-list order for a config (full, unsharded)."""
    rng = np.random.RandomState(seed)
    L = cfg["num_layer"]
    H = cfg["head_num"] * cfg["size_per_head"]
    I = cfg["inter_size"]
    V = cfg["vocab_size"]

    def r(*shape, s=std, mean=0.0):
        a = (mean + s * rng.randn(*shape)).astype(np.float32)
        return orc.round_half(a) if fp16 else a

    groups = [[] for _ in range(12)]
    for _ in range(L):
        groups[0].append(r(H, s=0.05))
        groups[1].append(r(H, s=0.05, mean=1.0))
        groups[2].append(r(H, 3 * H))
        groups[3].append(r(3 * H, s=0.05))
        groups[4].append(r(H, H))
        groups[5].append(np.zeros((0,), np.float32))
        groups[6].append(r(H, I))
        groups[7].append(r(I, s=0.05))
        groups[8].append(r(I, H))
        groups[9].append(r(H, s=0.05))
        groups[10].append(r(H, s=0.05))
        groups[11].append(r(H, s=0.05, mean=1.0))
    w = [a for g in groups for a in g]
    w += [r(V, H, s=0.3), r(H, s=0.05, mean=1.0), r(H, s=0.05), r(V, H, s=0.3)]
    return w


def shard_weights(cfg, w, tp, rank):
    """Tensor-parallel shard of a reference-order weight list, as the reference's converter + loader produce it
    (huggingface_convert.py:35-81: QKV/FFN1 columns, out-proj/FFN2 rows, row-split GEMM biases divided by TP)."""
    L = cfg["num_layer"]
    nh, dh = cfg["head_num"], cfg["size_per_head"]
    H, I = nh * dh, cfg["inter_size"]
    hl, il = H // tp, I // tp
    out = []
    for g in range(12):
        for l in range(L):
            a = w[g * L + l]
            if g == 2:
                a = a.reshape(H, 3, H)[:, :, rank * hl:(rank + 1) * hl].reshape(H, 3 * hl)
            elif g == 3:
                a = a.reshape(3, H)[:, rank * hl:(rank + 1) * hl].reshape(3 * hl)
            elif g == 4:
                a = a.reshape(H, H)[rank * hl:(rank + 1) * hl, :]
            elif g == 6:
                a = a.reshape(H, I)[:, rank * il:(rank + 1) * il]
            elif g == 7:
                a = a.reshape(I)[rank * il:(rank + 1) * il]
            elif g == 8:
                a = a.reshape(I, H)[rank * il:(rank + 1) * il, :]
            elif g == 9:
                a = a / tp
            out.append(np.ascontiguousarray(a, dtype=np.float32))
    out += [w[12 * L], w[12 * L + 1], w[12 * L + 2], w[12 * L + 3]]
    return out
