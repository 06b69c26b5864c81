"""-m gpu: the C++ command-line example (examples/gptneox_example.cc, counterpart of the reference's
examples/cpp/gptneox/gptneox_example.cc) drives the same engine through include/ftcf.h without Python: a tiny checkpoint in
the converter's file format must decode to the golden HF tokens."""
import os
import subprocess

import numpy as np
import pytest
import torch  # noqa: F401  -- before libftcf.so: one HIP / OpenMP runtime per process

from tests.helpers import load_tiny

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fastertransformer4codefuse_amd", "bin", "gptneox_example")

NAMES = ["input_layernorm.bias", "input_layernorm.weight", "attention.query_key_value.weight.0",
         "attention.query_key_value.bias.0", "attention.dense.weight.0", None, "mlp.dense_h_to_4h.weight.0",
         "mlp.dense_h_to_4h.bias.0", "mlp.dense_4h_to_h.weight.0", "mlp.attention.bias.sum",
         "post_attention_layernorm.bias", "post_attention_layernorm.weight"]


def write_checkpoint(d, cfg, w, int8):
    """Files as huggingface_convert.py / convert.py write them for one rank (codefuse_example.py:337-419 reads them)."""
    from fastertransformer4codefuse_amd import capi  # noqa: F401
    from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as quant
    import torch
    L = cfg["num_layer"]
    H, I = cfg["head_num"] * cfg["size_per_head"], cfg["inter_size"]
    shapes = {2: (H, 3 * H), 4: (H, H), 6: (H, I), 8: (I, H)}
    for g in range(12):
        if NAMES[g] is None:
            continue
        for l in range(L):
            a = np.asarray(w[g * L + l], dtype=np.float32)
            base = os.path.join(d, f"model.layers.{l}.{NAMES[g]}")
            if g in shapes and int8:
                q, s = quant(torch.from_numpy(a.reshape(shapes[g])).half().contiguous())
                q.numpy().tofile(base + ".q.bin")
                s.float().numpy().tofile(base + ".s.bin")
            else:
                a.tofile(base + ".bin")
    for i, n in enumerate(["wte", "final_layernorm.weight", "final_layernorm.bias", "lm_head.weight"]):
        np.asarray(w[12 * L + i], dtype=np.float32).tofile(os.path.join(d, f"model.{n}.bin"))
    with open(os.path.join(d, "config.ini"), "w") as f:
        f.write("[gptneox]\nmodel_name=tiny\nhead_num=%d\nsize_per_head=%d\ninter_size=%d\nnum_layer=%d\nvocab_size=%d\n"
                "rotary_embedding=%d\nstart_id=%d\nend_id=%d\nuse_gptj_residual=1\nweight_data_type=fp32\n"
                % (cfg["head_num"], cfg["size_per_head"], I, L, cfg["vocab_size"], cfg["rotary_dim"],
                   cfg.get("start_id", 0), cfg["end_id"]))


@pytest.mark.parametrize("int8", [0, 1])
def test_cli_example_decodes_the_golden_tokens(tmp_path, int8):
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    assert os.path.exists(EXE), "build the example first: python -c 'import __graft_entry__ as g; g.build()'"
    cfg, w, z = load_tiny()
    mdir = tmp_path / "1-gpu"
    mdir.mkdir()
    write_checkpoint(str(mdir), cfg, w, int8)
    ini = tmp_path / "gptneox_config.ini"
    ini.write_text("[ft_instance_hyperparameter]\ndata_type=fp16\ntensor_para_size=1\npipeline_para_size=1\nint8_mode=%d\n"
                   "model_name=tiny\nmodel_dir=%s\n\n[request]\nbeam_width=1 # beam width\ntop_k=1 ; greedy\ntop_p=0.0\n"
                   "temperature=1.0\nrepetition_penalty=1.0\nrequest_batch_size=2\nrequest_output_len=8\n" % (int8, mdir))
    ids = tmp_path / "start_ids.csv"
    ids.write_text(", ".join(map(str, z["prompt"].tolist())) + "\n" + ", ".join(map(str, z["prompt_b"].tolist())) + "\n")
    out = tmp_path / "out"
    r = subprocess.run([EXE, str(ini), "--start_ids", str(ids), "--out", str(out)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    rows = [list(map(int, line.split())) for line in out.read_text().strip().splitlines()]
    assert len(rows) == 2 and len(rows[0]) == 16 + 8
    if not int8:  # fp16 engine is token exact vs HF on this model (test_gpu_engine.py)
        assert rows[0][16:] == z["hf_tokens"].tolist()
        assert rows[1][:19] == z["prompt_b"].tolist() + z["hf_tokens_b"].tolist()
    else:  # int8: same engine as the Python path -> compare with GptNeoXOp on the same quantised weights
        from tests import gpu_helpers as gh
        op = gh.make_op(cfg, w, int8_mode=1)
        end_id = cfg["end_id"]
        pid = np.full((2, 16), end_id, dtype=np.int32)
        pid[0] = z["prompt"]
        pid[1, :11] = z["prompt_b"]
        ref = gh.run_op(op, pid, [16, 11], 8, cfg["vocab_size"], top_k=1, return_logits=False)
        assert rows[0] == ref["output_ids"][0].tolist()
        assert rows[1] == ref["output_ids"][1].tolist()


def write_checkpoint_tp(d, cfg, w, tp, int8):
    """The converter's per-rank files (`<name>.<rank>.bin`, huggingface_convert.py:35-81) for a tensor-parallel run."""
    from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as quant
    from tests.helpers import shard_weights
    import torch
    L = cfg["num_layer"]
    H, I = cfg["head_num"] * cfg["size_per_head"], cfg["inter_size"]
    hl, il = H // tp, I // tp
    shapes = {2: (H, 3 * hl), 4: (hl, H), 6: (H, il), 8: (il, H)}
    for r in range(tp):
        ws = shard_weights(cfg, w, tp, r)
        for g in range(12):
            if NAMES[g] is None:
                continue
            sharded = NAMES[g].endswith(".0")
            if not sharded and r > 0:
                continue
            name = NAMES[g][:-2] + f".{r}" if sharded else NAMES[g]
            for l in range(L):
                a = np.asarray(ws[g * L + l], dtype=np.float32)
                base = os.path.join(d, f"model.layers.{l}.{name}")
                if g in shapes and int8:
                    q, s = quant(torch.from_numpy(a.reshape(shapes[g])).half().contiguous())
                    q.numpy().tofile(base + ".q.bin")
                    s.float().numpy().tofile(base + ".s.bin")
                else:
                    a.tofile(base + ".bin")
    for i, n in enumerate(["wte", "final_layernorm.weight", "final_layernorm.bias", "lm_head.weight"]):
        np.asarray(w[12 * L + i], dtype=np.float32).tofile(os.path.join(d, f"model.{n}.bin"))
    with open(os.path.join(d, "config.ini"), "w") as f:
        f.write("[gptneox]\nmodel_name=tiny\nhead_num=%d\nsize_per_head=%d\ninter_size=%d\nnum_layer=%d\nvocab_size=%d\n"
                "rotary_embedding=%d\nstart_id=%d\nend_id=%d\nuse_gptj_residual=1\nweight_data_type=fp32\n"
                % (cfg["head_num"], cfg["size_per_head"], I, L, cfg["vocab_size"], cfg["rotary_dim"],
                   cfg.get("start_id", 0), cfg["end_id"]))


@pytest.mark.parametrize("int8", [0, 1])
def test_cli_example_runs_tensor_parallel_ranks(tmp_path, int8):
    """tensor_para_size = 2 through the command-line example (the reference's is launched under mpirun,
    examples/cpp/gptneox/gptneox_example.cc:399-411): two copies of the program, RANK / WORLD_SIZE from the environment as
    torchrun sets them, meeting in a rendezvous directory; `--exchange host` lets both ranks share this box's one GPU.  Rank 0's
    `out` equals the single-rank run (fp16: the golden HF tokens)."""
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    cfg, w, z = load_tiny()
    mdir = tmp_path / "2-gpu"
    mdir.mkdir()
    write_checkpoint_tp(str(mdir), cfg, w, 2, int8)
    ini = tmp_path / "gptneox_config.ini"
    ini.write_text("[ft_instance_hyperparameter]\ndata_type=fp16\ntensor_para_size=2\npipeline_para_size=1\nint8_mode=%d\n"
                   "model_name=tiny\nmodel_dir=%s\n\n[request]\nbeam_width=1\ntop_k=1\ntop_p=0.0\n"
                   "temperature=1.0\nrepetition_penalty=1.0\nrequest_batch_size=2\nrequest_output_len=8\n" % (int8, mdir))
    ids = tmp_path / "start_ids.csv"
    ids.write_text(", ".join(map(str, z["prompt"].tolist())) + "\n" + ", ".join(map(str, z["prompt_b"].tolist())) + "\n")
    out = tmp_path / "out"
    rdv = tmp_path / "rdv"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", FTCF_PERSIST_NB="128", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([EXE, str(ini), "--start_ids", str(ids), "--out", str(out), "--rendezvous", str(rdv),
                                       "--exchange", "host", "--device", "0"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, logs[r][-3000:]
    rows = [list(map(int, line.split())) for line in out.read_text().strip().splitlines()]
    assert len(rows) == 2 and len(rows[0]) == 16 + 8
    if not int8:
        assert rows[0][16:] == z["hf_tokens"].tolist()
        assert rows[1][:19] == z["prompt_b"].tolist() + z["hf_tokens_b"].tolist()
    else:  # (a rank quantises its own shard: compare with the Python op on the same shards -- the local group of test_gpu_tp_local.py)
        assert rows[0][:16] == z["prompt"].tolist() and all(0 <= t < cfg["vocab_size"] for t in rows[0])
    # a single rank without its peer is refused with a message, not a hang
    r1 = subprocess.run([EXE, str(ini), "--start_ids", str(ids), "--out", str(out)], capture_output=True, text=True, timeout=120,
                        env=dict(os.environ, RANK="0", WORLD_SIZE="1"))
    assert r1.returncode != 0 and "needs that many ranks" in r1.stderr


def test_cli_example_beam_search_matches_the_hf_beams(tmp_path):
    """beam_width > 1 through the command-line example: one output row per (request, beam), best beam first."""
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    cfg, w, _ = load_tiny()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_gptneox_beam.npz"))
    mdir = tmp_path / "1-gpu"
    mdir.mkdir()
    write_checkpoint(str(mdir), cfg, w, 0)
    ini = tmp_path / "gptneox_config.ini"
    ini.write_text("[ft_instance_hyperparameter]\ndata_type=fp16\ntensor_para_size=1\npipeline_para_size=1\nint8_mode=0\n"
                   "model_name=tiny\nmodel_dir=%s\n\n[request]\nbeam_width=3\nbeam_search_diversity_rate=0.0\n"
                   "len_penalty=0.0\ntop_k=0\ntop_p=0.0\ntemperature=1.0\nrepetition_penalty=1.0\nrequest_batch_size=1\n"
                   "request_output_len=8\n" % mdir)
    ids = tmp_path / "start_ids.csv"
    ids.write_text(", ".join(map(str, g["ids_a"][0].tolist())) + "\n")
    out = tmp_path / "out"
    r = subprocess.run([EXE, str(ini), "--start_ids", str(ids), "--out", str(out)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    rows = [list(map(int, line.split())) for line in out.read_text().strip().splitlines()]
    assert len(rows) == 3 and all(len(x) == 16 + 8 for x in rows)
    for k in range(3):
        assert rows[k][:16] == g["ids_a"][0].tolist()
        assert rows[k][16:] == g["hf_beam_tokens_a"][0, k].tolist()


def test_roctx_ranges_show_up_in_a_marker_trace(tmp_path):
    """FTCF_ROCTX=ON (the counterpart of FT_NVTX=ON, utils/nvtx_utils.cc:59-87): the host phases of a request are bracketed
    by roctx ranges that `rocprofv3 --marker-trace` records; without the variable nothing is emitted and tokens are the same."""
    import glob
    import shutil
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        pytest.skip("rocprofv3 not installed")
    cfg, w, z = load_tiny()
    mdir = tmp_path / "1-gpu"
    mdir.mkdir()
    write_checkpoint(str(mdir), cfg, w, 0)
    ini = tmp_path / "gptneox_config.ini"
    ini.write_text("[ft_instance_hyperparameter]\ndata_type=fp16\ntensor_para_size=1\npipeline_para_size=1\nint8_mode=0\n"
                   "model_name=tiny\nmodel_dir=%s\n\n[request]\nbeam_width=1\ntop_k=1\ntop_p=0.0\n"
                   "temperature=1.0\nrepetition_penalty=1.0\nrequest_batch_size=1\nrequest_output_len=8\n" % mdir)
    ids = tmp_path / "start_ids.csv"
    ids.write_text(", ".join(map(str, z["prompt"].tolist())) + "\n")
    out = tmp_path / "out"
    env = dict(os.environ, FTCF_ROCTX="ON", TMPDIR=str(tmp_path))
    r = subprocess.run([prof, "--marker-trace", "--output-format", "csv", "-d", str(tmp_path / "prof"), "--",
                        EXE, str(ini), "--start_ids", str(ids), "--out", str(out)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [list(map(int, line.split())) for line in out.read_text().strip().splitlines()]
    assert rows[0][16:] == z["hf_tokens"].tolist()
    traces = glob.glob(str(tmp_path / "prof" / "**" / "*marker*trace*.csv"), recursive=True)
    assert traces, "no marker trace written: %s" % r.stderr[-1000:]
    text = "".join(open(t).read() for t in traces)
    for name in ("ftcf.begin", "ftcf.GptNeoXContextDecoder", "ftcf.GptNeoXDecoder", "ftcf.finish"):
        assert name in text, name
