"""Offline tools of the CodeFuse path: HF GPT-NeoX -> `.bin` + config.ini checkpoint, and the weight-only int8
quantiser pass.  Counterparts of the reference's examples/pytorch/codefuse/huggingface_convert.py (:22-206) and
quant_and_save.py (:12-101): same file names, tensor orientation, TP splits and config keys, so a directory written
by either side loads in the other (the int8 `.q.bin` files hold this engine's gfx950 tile layout, which is private
exactly like the reference's CUDA layout -- see DESIGN.md).

    python -m fastertransformer4codefuse_amd.convert hf2ft -i <hf_dir> -o <out_dir> -i_g 2 -weight_data_type fp16 -m_n codefuse
    python -m fastertransformer4codefuse_amd.convert quant --in_dir <out_dir>/2-gpu --out_dir <q_dir> --tensor_para_size 2
"""
import argparse
import configparser
import os
import shutil

import numpy as np

_REPLICATED = ("input_layernorm.weight", "input_layernorm.bias", "post_attention_layernorm.weight",
               "post_attention_layernorm.bias", "final_layernorm.weight", "final_layernorm.bias")
_ROW_SPLIT_BIAS = ("attention.dense.bias", "mlp.dense_4h_to_h.bias")   # replicated, pre-divided by TP
_ROW_SPLIT = ("attention.dense.weight", "mlp.dense_4h_to_h.weight")     # split along K (axis 0)
_COL_SPLIT = ("mlp.dense_h_to_4h.weight", "mlp.dense_h_to_4h.bias")     # split along N (last axis)


def split_and_convert_process(saved_dir, factor, key, args, config, val):
    """Writes one HF parameter (`val` already transposed to [in, out] for matrices) as FT `.bin` file(s).

    QKV is re-ordered from HF's [hidden, heads, 3, head_dim] to [hidden, 3, heads*head_dim] before the column split
    (huggingface_convert.py:56-75); biases of the row-split GEMMs are divided by the TP size (:35-41)."""
    has = lambda names: any(n in key for n in names)
    if has(_REPLICATED):
        val.tofile(f"{saved_dir}/model.{key}.bin")
        return
    if has(_ROW_SPLIT_BIAS):
        (val / factor if factor > 1 else val).tofile(f"{saved_dir}/model.{key}.bin")
        return
    if has(_ROW_SPLIT):
        parts = np.split(val, factor, axis=0)
    elif has(_COL_SPLIT):
        parts = np.split(val, factor, axis=-1)
    elif "attention.query_key_value.bias" in key:
        n_head = config["num_attention_heads"]
        local = val.shape[-1] // 3
        qkv = val.reshape(n_head, 3, local // n_head).transpose(1, 0, 2).reshape(3, local)
        parts = np.split(qkv, factor, axis=-1)
    elif "attention.query_key_value.weight" in key:
        n_head = config["num_attention_heads"]
        hidden, local = val.shape[0], val.shape[-1] // 3
        qkv = val.reshape(hidden, n_head, 3, local // n_head).transpose(0, 2, 1, 3).reshape(hidden, 3, local)
        parts = np.split(qkv, factor, axis=-1)
    else:
        print("[ERROR] cannot find key '{}'".format(key))
        return
    for j, part in enumerate(parts):
        np.ascontiguousarray(part).tofile(f"{saved_dir}/model.{key}.{j}.bin")


def rotary_dim_of(hf_config, head_size):
    """transformers < 5 stores `rotary_pct`; 5.x moved it to rope_parameters.partial_rotary_factor (the reference
    silently skips config.ini there, huggingface_convert.py:104,123-124)."""
    if hf_config.get("rotary_pct") is not None:
        return int(head_size * hf_config["rotary_pct"])
    rp = hf_config.get("rope_parameters") or {}
    if "partial_rotary_factor" in rp:
        return int(head_size * rp["partial_rotary_factor"])
    raise KeyError("neither rotary_pct nor rope_parameters.partial_rotary_factor in the HF config")


def write_config(saved_dir, hf_config, model_name, weight_data_type):
    n_heads = hf_config["num_attention_heads"]
    head_size = hf_config["hidden_size"] // n_heads
    config = configparser.ConfigParser()
    config["gptneox"] = {
        "model_name": model_name, "head_num": str(n_heads), "size_per_head": str(head_size),
        "inter_size": str(hf_config["intermediate_size"]), "num_layer": str(hf_config["num_hidden_layers"]),
        "rotary_embedding": str(rotary_dim_of(hf_config, head_size)), "vocab_size": str(hf_config["vocab_size"]),
        "start_id": str(hf_config["bos_token_id"]), "end_id": str(hf_config["eos_token_id"]),
        "use_gptj_residual": str(int(hf_config["use_parallel_residual"])), "weight_data_type": weight_data_type}
    with open(os.path.join(saved_dir, "config.ini"), "w") as f:
        config.write(f)


def convert_model(model, saved_dir, factor, weight_data_type="fp32", model_name="gptneox"):
    """HF GPTNeoXForCausalLM (already loaded) -> FT checkpoint directory `saved_dir`."""
    os.makedirs(saved_dir, exist_ok=True)
    np_dt = {"fp32": np.float32, "fp16": np.float16}[weight_data_type]
    hf_config = vars(model.config)
    write_config(saved_dir, hf_config, model_name, weight_data_type)
    globals_ = {"gpt_neox.embed_in.weight": "model.wte.bin", "gpt_neox.final_layer_norm.bias":
                "model.final_layernorm.bias.bin", "gpt_neox.final_layer_norm.weight":
                "model.final_layernorm.weight.bin", "embed_out.weight": "model.lm_head.weight.bin",
                "lm_head.weight": "model.lm_head.weight.bin"}  # transformers 5.x renamed embed_out
    for name, param in model.named_parameters():
        array = param.detach().cpu().numpy().astype(np_dt)
        if name in globals_:
            array.tofile(os.path.join(saved_dir, globals_[name]))
        elif "weight" in name or "bias" in name:
            split_and_convert_process(saved_dir, factor, name.replace("gpt_neox.", ""), None, hf_config, array.T)
        else:
            print("skipped", name)
    if hf_config["use_parallel_residual"]:
        # one fused bias for the parallel-residual layer: attention.dense.bias + mlp.dense_4h_to_h.bias (:192-206)
        for l in range(hf_config["num_hidden_layers"]):
            a = np.fromfile(f"{saved_dir}/model.layers.{l}.attention.dense.bias.bin", dtype=np_dt)
            b = np.fromfile(f"{saved_dir}/model.layers.{l}.mlp.dense_4h_to_h.bias.bin", dtype=np_dt)
            (a + b).astype(np_dt).tofile(f"{saved_dir}/model.layers.{l}.mlp.attention.bias.sum.bin")


def quant_and_save(in_dir, out_dir, tensor_para_size, inference_data_type="fp16"):
    """Copies a checkpoint and replaces the four GEMM kernels of every layer/rank by `.q.bin` (int8, engine tile
    layout) + `.s.bin` (scales in the checkpoint's weight dtype)."""
    import torch
    from .gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as quantize
    if os.path.exists(out_dir):
        shutil.rmtree(out_dir)
    shutil.copytree(in_dir, out_dir)
    config = configparser.ConfigParser()
    config.read(os.path.join(in_dir, "config.ini"))
    sec = config["gptneox"]
    head_num, dh, L = int(sec["head_num"]), int(sec["size_per_head"]), int(sec["num_layer"])
    H = head_num * dh
    hl = H // tensor_para_size
    il = (int(sec["inter_size"]) if "inter_size" in sec else 4 * H) // tensor_para_size
    np_dt = {"fp16": np.float16, "fp32": np.float32, "float16": np.float16, "float32": np.float32}[sec["weight_data_type"]]
    t_dt = {"fp16": torch.float16, "fp32": torch.float32}[inference_data_type]
    shapes = {"attention.query_key_value.weight": (H, 3 * hl), "attention.dense.weight": (hl, H),
              "mlp.dense_h_to_4h.weight": (H, il), "mlp.dense_4h_to_h.weight": (il, H)}
    for rk in range(tensor_para_size):
        for fn, shape in shapes.items():
            for li in range(L):
                base = os.path.join(out_dir, f"model.layers.{li}.{fn}.{rk}")
                w = torch.from_numpy(np.fromfile(base + ".bin", dtype=np_dt)).to(t_dt).reshape(shape).contiguous()
                q, s = quantize(w)
                q.numpy().astype(np.int8).tofile(base + ".q.bin")
                s.numpy().astype(np_dt).tofile(base + ".s.bin")
                os.remove(base + ".bin")


def import_cuda_qbin(in_dir, out_dir, tensor_para_size):
    """Re-lays the `.q.bin` files of a checkpoint that was quantised by a CUDA build of the reference (SM75..SM89 layout,
    cutlass_preprocessors.cc:500-539) into this engine's tile layout; scales and every other file are copied unchanged.
    The int8 values themselves are not touched: the result is what `quant` would have produced from the same weights."""
    from . import capi
    import ctypes as C
    if os.path.exists(out_dir):
        shutil.rmtree(out_dir)
    shutil.copytree(in_dir, out_dir)
    config = configparser.ConfigParser()
    config.read(os.path.join(in_dir, "config.ini"))
    sec = config["gptneox"]
    head_num, dh, L = int(sec["head_num"]), int(sec["size_per_head"]), int(sec["num_layer"])
    H = head_num * dh
    hl = H // tensor_para_size
    il = (int(sec["inter_size"]) if "inter_size" in sec else 4 * H) // tensor_para_size
    shapes = {"attention.query_key_value.weight": (H, 3 * hl), "attention.dense.weight": (hl, H),
              "mlp.dense_h_to_4h.weight": (H, il), "mlp.dense_4h_to_h.weight": (il, H)}
    lib = capi.lib()
    i8p = C.POINTER(C.c_int8)
    for rk in range(tensor_para_size):
        for fn, (K, N) in shapes.items():
            for li in range(L):
                base = os.path.join(out_dir, f"model.layers.{li}.{fn}.{rk}")
                q = np.fromfile(base + ".q.bin", dtype=np.int8)
                if q.size != K * N:
                    raise ValueError(f"{base}.q.bin holds {q.size} bytes, expected {K}x{N}")
                rm, tiled = np.empty(K * N, np.int8), np.empty(K * N, np.int8)
                capi.check(lib.ftcf_int8_cuda_sm80_to_rowmajor(q.ctypes.data_as(i8p), K, N, rm.ctypes.data_as(i8p)))
                capi.check(lib.ftcf_int8_rowmajor_to_tiled(rm.ctypes.data_as(i8p), K, N, tiled.ctypes.data_as(i8p)))
                tiled.tofile(base + ".q.bin")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("hf2ft")
    c.add_argument("-saved_dir", "-o", required=True)
    c.add_argument("-in_file", "-i", required=True)
    c.add_argument("-infer_gpu_num", "-i_g", type=int, required=True)
    c.add_argument("-weight_data_type", default="fp32", choices=["fp32", "fp16"])
    c.add_argument("-model_name", "-m_n", required=True)
    q = sub.add_parser("quant")
    q.add_argument("--in_dir", required=True)
    q.add_argument("--out_dir", required=True)
    q.add_argument("--tensor_para_size", type=int, required=True)
    q.add_argument("--inference_data_type", "--data_type", choices=["fp32", "fp16"], default="fp16")
    i = sub.add_parser("import-cuda-qbin", help="re-lay .q.bin files written by a CUDA build (SM75..SM89 layout)")
    i.add_argument("--in_dir", required=True)
    i.add_argument("--out_dir", required=True)
    i.add_argument("--tensor_para_size", type=int, required=True)
    a = ap.parse_args()
    if a.cmd == "import-cuda-qbin":
        import_cuda_qbin(a.in_dir, a.out_dir, a.tensor_para_size)
        return
    if a.cmd == "hf2ft":
        from transformers import GPTNeoXForCausalLM
        out = os.path.join(a.saved_dir, "%d-gpu" % a.infer_gpu_num)
        assert not os.path.exists(out), "target path has exist, please remove %s first." % out
        convert_model(GPTNeoXForCausalLM.from_pretrained(a.in_file), out, a.infer_gpu_num, a.weight_data_type,
                      a.model_name)
    else:
        quant_and_save(a.in_dir, a.out_dir, a.tensor_para_size, a.inference_data_type)


if __name__ == "__main__":
    main()
