"""-m gpu: tensor parallelism across two PROCESSES on ONE GPU -- the inter-process path of the in-kernel all-reduce.

tests/test_gpu_tp_local.py runs the tensor-parallel engine with all ranks inside one process (plain device pointers as
exchange windows, one group launch).  What it cannot show is the product's set-up between address spaces
(th_op/gptneox/utils/nccl_inherit_utils.cc:25-68 bootstraps the reference's NCCL communicator; custom_ar_kernels.cu:139-200
is its one-shot all-reduce): exporting the exchange window with hipIpcGetMemHandle, mapping the peers' windows with
hipIpcOpenMemHandle, the hand-shake kernel that proves a granule stored by the PEER's kernel becomes visible to a polling
kernel here, and the persistent decode kernel's tensor-parallel instantiation exchanging x' through those mappings while
the peer's kernel -- another process, another queue -- runs next to it.  RCCL cannot put two ranks on one device, so the
ranks use a HOST-EXCHANGE communicator (include/ftcf.h ftcf_comm_init_host_exchange): every exchange is an all-gather of
host bytes over the test's gloo group.  Each process runs its persistent kernel on half of the compute units
(FTCF_PERSIST_NB=128) so that both are resident together.  Compared with the TP = 1 engine and the CPU oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import load_tiny, quantize_layers, random_model, weight_list_to_layers

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MID = dict(head_num=8, size_per_head=64, inter_size=2048, num_layer=3, vocab_size=2048, rotary_dim=16, start_id=0, end_id=2)


# the CodeFuse-13B shape (BASELINE.json configs 3 / 4) on bench.py's synthetic weights
FULL13B = dict(layers=40, heads=40, head_dim=128, inter=20480, vocab=100864, rotary=32, dtype="int8")


def request_13b():
    import torch
    g = torch.Generator().manual_seed(48)
    return torch.randint(3, FULL13B["vocab"], (1, 192), generator=g, dtype=torch.int32), 6


def requests(cfg, model):
    """name -> (ids, lens, n_out, sampling kwargs); shared by the workers and the checker."""
    if model == "tiny":
        _, _, z = load_tiny()
        ids = np.full((2, 16), cfg["end_id"], dtype=np.int32)
        ids[0] = z["prompt"]
        ids[1, :11] = z["prompt_b"]
        return {"one_row": (ids[:1], [16], 8, dict(top_k=1)), "two_rows": (ids, [16, 11], 8, dict(top_k=1))}
    rng = np.random.RandomState(5)
    S = 40
    ids = rng.randint(3, cfg["vocab_size"], size=(4, S)).astype(np.int32)
    lens = [S, S - 7, S, 9]
    for b, n in enumerate(lens):
        ids[b, n:] = cfg["end_id"]
    return {"one_row": (ids[:1], lens[:1], 6, dict(top_k=1)), "four_rows": (ids, lens, 6, dict(top_k=1))}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RANK_RUNS = {}  # results of a pair of rank processes by (model, weights, environment): tests that need the same run share it


def _run_ranks(tmp_path, model, int8_mode, world=2, extra_env=None, timeout=600):
    key = (model, int8_mode, world, tuple(sorted((extra_env or {}).items())))
    if key not in _RANK_RUNS:
        _RANK_RUNS[key] = _launch_ranks(tmp_path, model, int8_mode, world, extra_env, timeout)
    return _RANK_RUNS[key]


def _launch_ranks(tmp_path, model, int8_mode, world, extra_env, timeout):
    port = str(_free_port())
    env = dict(os.environ, FTCF_PERSIST_NB="128", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tp_process_worker.py"), str(r), str(world), port,
                               model, str(int8_mode), outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [dict(np.load(o)) for o in outs], logs


def _check(ref_tokens, ref_logits, got_tokens, got_logits, lens, what, frac):
    scale = np.abs(ref_logits).max()
    for b in range(ref_tokens.shape[0]):
        for t in range(ref_logits.shape[0]):
            err = np.abs(got_logits[t, b] - ref_logits[t, b]).max() / scale
            assert err <= frac, (what, b, t, err)
            if got_tokens[b, lens[b] + t] != ref_tokens[b, lens[b] + t]:
                top2 = np.sort(ref_logits[t, b])[-2:]
                assert top2[1] - top2[0] <= 2 * frac * scale, (what, b, t, "token flip without a near tie")
                break


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


@pytest.mark.parametrize("model,int8_mode", [("tiny", 0), ("mid", 0), ("mid", 1)])
def test_two_processes_on_one_gpu_exchange_through_ipc_windows(gh, tmp_path, model, int8_mode):
    from oracle import oracle as orc
    if model == "tiny":
        cfg, w, _ = load_tiny()
    else:
        cfg, w = MID, random_model(MID, seed=11)
    res, logs = _run_ranks(tmp_path, model, int8_mode)
    assert not any("exchange windows unavailable" in lg for lg in logs), logs[0][-2000:]
    layers, glob = weight_list_to_layers(cfg, w)
    lay = quantize_layers(layers) if int8_mode else layers
    op1 = gh.make_op(cfg, w, int8_mode=int8_mode)
    frac = 5e-3 if int8_mode == 0 else 2e-2  # (int8: a rank quantises its own row shards -- test_gpu_tp_local.py)
    for name, (ids, lens, n_out, kw) in requests(cfg, model).items():
        B = ids.shape[0]
        # every rank holds the same tokens and bit-identical logits: the in-kernel sums run in rank order on both sides
        assert res[0][name + ".output_ids"].tolist() == res[1][name + ".output_ids"].tolist(), name
        np.testing.assert_array_equal(res[0][name + ".logits"], res[1][name + ".logits"])
        # one or two rows: the persistent kernel's tensor-parallel instantiation, the all-reduce inside the launch through
        # the IPC-mapped windows (decode_path 1); more rows: the general path with host-staged collectives (2)
        want = 1 if B <= 2 else 2
        got = (int(res[0][name + ".decode_path"][0]), int(res[1][name + ".decode_path"][0]))
        if B <= 2 and got == (0, 0) and int(res[0]["attempts"][0]) == 6:
            # six fresh engine pairs in a row fell back to the collective path TOGETHER (the designed reaction to a peer
            # kernel that is not scheduled next to this one: nothing guarantees two processes co-run on one GPU)
            pytest.xfail("the two processes' kernels were not co-scheduled in six attempts: in-kernel path not exercised")
        assert got == (want, want), name
        if model == "mid" and B == 4:
            # the prompt phase's per-layer all-reduce (160 rows x 512: 160 KB) went through the IPC-mapped windows -- the
            # two-shot kernel, no RCCL, no host staging -- in every layer, on both ranks (round 4)
            if int(res[0][name + ".window_allreduces"][0]) == 0 and any("exchange-window all-reduce gave up" in lg for lg in logs):
                pytest.xfail("the two processes' all-reduce kernels were not co-scheduled: the request was replayed on the host-staged path")
            assert int(res[0][name + ".window_allreduces"][0]) >= cfg["num_layer"], res[0][name + ".window_allreduces"]
            assert int(res[1][name + ".window_allreduces"][0]) == int(res[0][name + ".window_allreduces"][0])
        o = orc.Model(dict(cfg, fp16=1, int8_mode=int8_mode), lay, glob).generate(ids, lens, n_out, return_logits=True)
        _check(o["output_ids"], o["logits"], res[0][name + ".output_ids"], res[0][name + ".logits"], lens, f"{model} {name} vs oracle", frac)
        r1 = gh.run_op(op1, ids, lens, n_out, cfg["vocab_size"], **kw)
        _check(r1["output_ids"], r1["logits"], res[0][name + ".output_ids"], res[0][name + ".logits"], lens, f"{model} {name} vs TP=1 engine", frac)


def test_two_processes_overlap_the_all_reduce_of_batched_decode(gh, tmp_path):
    """Row n1 between two PROCESSES: the four-row request's decode steps as two micro-batches on two streams with the layer's
    all-reduce on a third (FTCF_DECODE_OVERLAP=1; engine.hip.h decoder_overlapped; GptNeoXDecoder.cc:342-359 has it in line) --
    both ranks hold the tokens and logits of the un-overlapped pair of processes bit for bit, and the stats say which form ran."""
    (tmp_path / "ov").mkdir()
    (tmp_path / "plain").mkdir()
    ov, _ = _run_ranks(tmp_path / "ov", "mid", 1, extra_env={"FTCF_DECODE_OVERLAP": "1"})
    plain, _ = _run_ranks(tmp_path / "plain", "mid", 1)  # (the default is the in-line form; the run is shared with the test above)
    for r in range(2):
        assert int(ov[r]["four_rows.decode_overlap"][0]) == 1 and int(plain[r]["four_rows.decode_overlap"][0]) == 0
        assert int(ov[r]["one_row.decode_overlap"][0]) == 0
        assert int(ov[r]["four_rows.decode_path"][0]) == 2
        assert ov[r]["four_rows.output_ids"].tolist() == plain[0]["four_rows.output_ids"].tolist()
        np.testing.assert_array_equal(ov[r]["four_rows.logits"], plain[0]["four_rows.logits"])


def test_compiled_module_bootstraps_tensor_parallelism_from_the_callers_process_group(gh, tmp_path):
    """The two ranks build `libth_gptneox.GptNeoXOp` -- the COMPILED pybind11 module -- over their gloo group: the constructor casts the
    Python group to c10d::ProcessGroup, builds the host-exchange communicator (comm_from_group, FTCF_TP_EXCHANGE=host) and every
    exchange of the engine (window hand-shake, agreement words, collectives of the general path) travels through
    HostExchange::allgather on that group (csrc/th_op/th_gptneox.cc:49-97; the reference: th_op/gptneox/utils/nccl_inherit_utils.cc:25-68,
    GptNeoXOp.cc:25-106).  Tokens equal on both ranks, logits bit-identical across ranks, and what the TP = 1 engine produces."""
    cfg, w, _ = load_tiny()
    res, logs = _run_ranks(tmp_path, "tiny", 0, extra_env={"FTCF_TEST_COMPILED_OP": "1"})
    op1 = gh.make_op(cfg, w)
    for name, (ids, lens, n_out, kw) in requests(cfg, "tiny").items():
        assert res[0][name + ".output_ids"].tolist() == res[1][name + ".output_ids"].tolist(), name
        np.testing.assert_array_equal(res[0][name + ".logits"], res[1][name + ".logits"])
        r1 = gh.run_op(op1, ids, lens, n_out, cfg["vocab_size"], **kw)
        _check(r1["output_ids"], r1["logits"], res[0][name + ".output_ids"], res[0][name + ".logits"], lens,
               f"compiled module, {name} vs TP=1 engine", 5e-3)


def test_two_processes_fall_back_together_when_the_windows_are_refused(gh, tmp_path):
    """FTCF_TP_WINDOWS=0 on ONE rank only: the agreement after the hand-shake leaves win_ok false on BOTH, and the request
    runs on the collective path (per-stage launches + one host-staged all-reduce per layer) with the same tokens."""
    cfg, w, _ = load_tiny()
    port = str(_free_port())
    outs = [str(tmp_path / f"r{r}.npz") for r in range(2)]
    procs = []
    for r in range(2):
        env = dict(os.environ, FTCF_PERSIST_NB="128", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", FTCF_TEST_EXPECT_FALLBACK="1")
        if r == 1:
            env["FTCF_TP_WINDOWS"] = "0"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tp_process_worker.py"), str(r), "2", port, "tiny",
                                       "0", outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, logs[r][-3000:]
    res = [dict(np.load(o)) for o in outs]
    assert int(res[0]["one_row.decode_path"][0]) == 0 and int(res[1]["one_row.decode_path"][0]) == 0
    assert res[0]["one_row.output_ids"].tolist() == res[1]["one_row.output_ids"].tolist()
    op1 = gh.make_op(cfg, w)
    ids, lens, n_out, kw = requests(cfg, "tiny")["one_row"]
    r1 = gh.run_op(op1, ids, lens, n_out, cfg["vocab_size"], **kw)
    assert res[0]["one_row.output_ids"].tolist() == r1["output_ids"].tolist()


@pytest.mark.timeout(2400)
def test_13b_tp2_shards_between_two_processes(tmp_path):
    """BASELINE config 4 (CodeFuse-13B int8, TP = 2) as far as ONE GPU can run it: two PROCESSES, each holding its exact 6.9 GB
    shard of the 40-layer model and 128 workgroups of the tensor-parallel persistent kernel, x' exchanged through the
    IPC-mapped windows in every layer of every token (the prompt phase's collectives are host staged) -- against the TP = 1
    engine on the whole model.  Until round 4 the product instantiation had only met 512-hidden models between processes."""
    import argparse
    import sys
    import torch
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    res, logs = _run_ranks(tmp_path, "13b", 1, timeout=2000)
    assert not any("exchange windows unavailable" in lg for lg in logs), logs[0][-2000:]
    assert res[0]["output_ids"].tolist() == res[1]["output_ids"].tolist()
    np.testing.assert_array_equal(res[0]["logits"], res[1]["logits"])  # rank-ordered in-kernel sums: bit-identical on both ranks
    got = (int(res[0]["decode_path"][0]), int(res[1]["decode_path"][0]))
    if got == (0, 0) and int(res[0]["attempts"][0]) == 4:
        pytest.xfail("the two processes' kernels were not co-scheduled in four attempts: in-kernel path not exercised")
    assert got == (1, 1)
    assert int(res[0]["window_allreduces"][0]) >= FULL13B["layers"]  # the 1.9 MB prompt-phase messages: two-shot window kernel
    # row n1 at the 13B shard shape: 16 rows, the decode steps' all-reduce in line and overlapped (two micro-batches of 8 rows on two
    # streams, the reduction on a third): bit-identical on both ranks, and -- unless the two processes' all-reduce kernels were not
    # co-scheduled and the request was replayed on the host-staged path -- every message through the IPC-mapped windows
    for r in range(2):
        ov, ln = res[r]["b16_1.stats"], res[r]["b16_0.stats"]
        assert (int(ov[0]), int(ln[0])) == (1, 0) and int(ov[1]) == 2 and int(ln[1]) == 2, (ov, ln)
        assert res[r]["b16_1.output_ids"].tolist() == res[0]["b16_0.output_ids"].tolist()
        np.testing.assert_array_equal(res[r]["b16_1.logits"], res[0]["b16_0.logits"])
        if not any("exchange-window all-reduce gave up" in lg for lg in logs):
            # prompt phase: one message per layer; two decode steps: one (in line) / two (micro-batches) per layer each
            assert int(ln[2]) >= 3 * FULL13B["layers"] and int(ov[2]) >= 5 * FULL13B["layers"], (ov, ln)
    sys.path.insert(0, ROOT)
    import bench
    from fastertransformer4codefuse_amd.gptneox_op import GptNeoXOp
    a = argparse.Namespace(**FULL13B)
    weights, int8_w, scales = bench.synth_weights(a, 1, torch.device("cuda", 0))
    op1 = GptNeoXOp(None, 0, a.heads, a.head_dim, a.inter, a.layers, a.vocab, a.rotary, 0, 2, 1, 1, 1, 2048, True, weights, int8_w, scales)
    ids, n_out = request_13b()
    S = ids.shape[1]
    lens = torch.full((1,), S, dtype=torch.int32, device="cuda")
    dbg = torch.zeros((n_out, 1, a.vocab), dtype=torch.float32, device="cuda")
    o = op1.forward(ids.cuda(), lens, n_out, 1, torch.tensor([1], dtype=torch.int32), _debug_logits=dbg)
    torch.cuda.synchronize()
    # the shards are exact slices of the TP = 1 model: only the summation order of the row-split GEMMs differs
    _check(o[0][:, 0].cpu().numpy(), dbg.cpu().numpy(), res[0]["output_ids"], res[0]["logits"], [S], "13B TP=2 processes vs TP=1", 5e-3)
