"""CPU, world_size 2 over gloo: the tensor-parallel host logic (sharding, per-layer all-reduce placement, vocabulary
split + all-gather + transpose of the logits, unique-id exchange) reproduces the single-rank result.

The compute runs in the CPU oracle (no GPU here); what is under test is the N>1 *structure* the engine shares with it
(GptNeoXDecoder.cc:342-359, GptNeoX.cc:888-925, nccl_inherit_utils.cc:25-68)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import load_tiny, shard_weights, weight_list_to_layers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    cfg, w, z = load_tiny()
    ws = shard_weights(cfg, w, world, rank)
    layers, glob = weight_list_to_layers(cfg, ws, tp=world)

    def allreduce(buf, n, _ctx):
        a = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(a)
        dist.all_reduce(t)

    def allgather(buf, n_per_rank, _ctx):
        a = np.ctypeslib.as_array(buf, shape=(world * n_per_rank,))
        t = torch.from_numpy(a)
        mine = t[rank * n_per_rank:(rank + 1) * n_per_rank].clone()
        outs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine)
        t.copy_(torch.cat(outs))

    m = orc.Model(dict(cfg, fp16=0, tp_size=world, tp_rank=rank), layers, glob, allreduce=allreduce,
                  allgather=allgather)
    ids = np.full((2, 16), cfg["end_id"], dtype=np.int32)
    ids[0] = z["prompt"]
    ids[1, :11] = z["prompt_b"]
    r = m.generate(ids, [16, 11], 8, return_logits=True)
    # unique-id exchange helper over the caller's (gloo) group: rank 0's bytes reach every rank
    from fastertransformer4codefuse_amd import capi
    uid = np.zeros(capi.UNIQUE_ID_BYTES, dtype=np.uint8)
    if rank == 0:
        uid[:] = (np.arange(capi.UNIQUE_ID_BYTES) * 3 % 251).astype(np.uint8)
    t = torch.from_numpy(uid)
    dist.broadcast(t, src=0)
    q.put((rank, r["output_ids"].tolist(), r["logits"], uid.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tensor_parallel_equals_single_rank():
    from oracle import oracle as orc
    cfg, w, z = load_tiny()
    layers, glob = weight_list_to_layers(cfg, w)
    ids = np.full((2, 16), cfg["end_id"], dtype=np.int32)
    ids[0] = z["prompt"]
    ids[1, :11] = z["prompt_b"]
    ref = orc.Model(dict(cfg, fp16=0), layers, glob).generate(ids, [16, 11], 8, return_logits=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out_ids, logits, uid in res:
        assert out_ids == ref["output_ids"].tolist(), rank
        np.testing.assert_allclose(logits, ref["logits"], atol=5e-4, rtol=1e-4)
        assert uid == [(i * 3) % 251 for i in range(128)]
    assert res[0][1][0][16:] == z["hf_tokens"].tolist()


def test_shards_match_the_reference_converter_digests():
    """shard_weights (the layout the engine expects per rank) agrees with what the reference's converter + loader
    produce for tensor_para_size=2 (sha256 captured in tests/golden/tiny_gptneox_tp2.json)."""
    import hashlib
    import json
    from tests.helpers import GOLDEN
    cfg, w, _ = load_tiny()
    with open(os.path.join(GOLDEN, "tiny_gptneox_tp2.json")) as f:
        gold = json.load(f)
    for r in range(2):
        ws = shard_weights(cfg, w, 2, r)
        for i, (a, g) in enumerate(zip(ws, gold[f"rank{r}"])):
            assert a.size == int(np.prod(g["shape"])), (r, i)
            sha = hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()
            assert sha == g["sha256"], (r, i)


def test_comm_entry_points_fail_loudly_without_a_device():
    from fastertransformer4codefuse_amd import capi
    if capi.device_count() > 0:
        pytest.skip("CPU-only behaviour")
    ids = (C.c_uint8 * capi.UNIQUE_ID_BYTES)()
    comm = C.c_void_p()
    rc = capi.lib().ftcf_comm_init(ids, 2, 0, 0, C.byref(comm))
    assert rc == -5 and not comm.value
