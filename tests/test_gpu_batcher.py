"""-m gpu: continuous batching over the paged K/V cache (include/ftcf.h `ftcf_batcher_*`).

The oracle for a batcher is the engine itself: whatever order requests arrive in, however they share decode steps and pages,
every request must produce what `GptNeoXOp.forward` produces for it alone (greedy: token exact on the tiny model, up to
near ties on the int8 1024-hidden model), and the pool must get all its pages back."""
import numpy as np
import pytest
import torch

from tests.helpers import load_tiny, random_model

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
MID = dict(head_num=8, size_per_head=64, inter_size=2048, num_layer=3, vocab_size=2048, rotary_dim=16, start_id=0, end_id=2)


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


def _alone(gh, op, prompt, n_new, V, end_id):
    """Tokens the engine generates for this prompt by itself (greedy), cut at end_id like the batcher's stream."""
    r = gh.run_op(op, np.asarray(prompt, np.int32)[None, :], [len(prompt)], n_new, V, top_k=1)
    toks = r["output_ids"][0, len(prompt):].tolist()
    out = []
    for t in toks:
        out.append(t)
        if t == end_id:
            break
    return out, r["logits"][:, 0]


def _drain(cb, arrivals):
    """arrivals: {iteration: [(prompt, max_new), ...]} -> {request index: tokens}, iterations used."""
    got, ids, it, k = {}, {}, 0, 0
    pending = dict(arrivals)
    while pending or cb.busy():
        for prompt, max_new in pending.pop(it, []):
            ids[cb.submit(prompt, max_new)] = k
            k += 1
        for rid, tok, fin in cb.step():
            got.setdefault(ids[rid], []).append(tok)
        it += 1
        assert it < 10000
    return got, it


def test_tiny_requests_arriving_over_time_match_the_engine_alone(gh):
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    op = gh.make_op(cfg, w)
    rng = np.random.RandomState(3)
    prompts = [z["prompt"].tolist(), z["prompt_b"].tolist(), z["prompt"][:5].tolist(), z["prompt_1"].tolist(),
               rng.randint(3, V, size=23).tolist(), rng.randint(3, V, size=9).tolist(), z["prompt"][::-1].tolist()]
    new = [8, 6, 12, 5, 10, 3, 7]
    ref = [_alone(gh, op, p, n, V, end_id)[0] for p, n in zip(prompts, new)]
    # 3 slots, 8-token pages: more requests than slots, arrivals while others are decoding
    cb = ContinuousBatcher(op, max_batch=3, page_tokens=8, num_pages=24, max_seq_len=64)
    free0 = cb.status()["free_pages"]
    got, _ = _drain(cb, {0: [(prompts[0], new[0]), (prompts[1], new[1])], 2: [(prompts[2], new[2]), (prompts[3], new[3])],
                         3: [(prompts[4], new[4])], 9: [(prompts[5], new[5]), (prompts[6], new[6])]})
    for i in range(len(prompts)):
        assert got[i] == ref[i], (i, got[i], ref[i])
    assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}  # every page came back
    # the engine is still usable for plain requests afterwards, and the batcher for a second wave
    assert _alone(gh, op, prompts[0], new[0], V, end_id)[0] == ref[0]
    got2, _ = _drain(cb, {0: [(prompts[4], new[4])]})
    assert got2[0] == ref[4]


def test_requests_wait_for_pages_and_all_finish(gh):
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    op = gh.make_op(cfg, w)
    prompts = [z["prompt"].tolist(), z["prompt_b"].tolist(), z["prompt"][:7].tolist(), z["prompt"][3:].tolist()]
    new = [9, 9, 9, 9]
    ref = [_alone(gh, op, p, n, V, end_id)[0] for p, n in zip(prompts, new)]
    # 4 slots but only 7 pages of 8 tokens: a 16 + 9 token request takes 4 of them -> the queue has to wait for pages
    cb = ContinuousBatcher(op, max_batch=4, page_tokens=8, num_pages=7, max_seq_len=32)
    got, iters = _drain(cb, {0: [(p, n) for p, n in zip(prompts, new)]})
    for i in range(4):
        assert got[i] == ref[i], i
    assert iters > 9  # not everybody could run together
    assert cb.status()["free_pages"] == 7
    with pytest.raises(RuntimeError):
        cb.submit(list(range(3, 40)), 4)  # longer than max_seq_len


def test_end_id_stops_a_sequence_and_frees_its_slot(gh):
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V = cfg["vocab_size"]
    ref_tokens = z["hf_tokens"].tolist()
    cfg2 = dict(cfg, end_id=int(ref_tokens[3]))  # the 4th generated token of the golden prompt is now the end token
    op = gh.make_op(cfg2, w)
    alone, _ = _alone(gh, op, z["prompt"].tolist(), 8, V, cfg2["end_id"])
    assert alone[-1] == cfg2["end_id"] and len(alone) <= 4
    cb = ContinuousBatcher(op, max_batch=2, page_tokens=16, num_pages=8, max_seq_len=48)
    got, _ = _drain(cb, {0: [(z["prompt"].tolist(), 8), (z["prompt_b"].tolist(), 8)]})
    assert got[0] == alone
    assert got[1] == _alone(gh, op, z["prompt_b"].tolist(), 8, V, cfg2["end_id"])[0]


@pytest.mark.parametrize("int8_mode,max_batch", [(0, 2), (1, 6), (1, 20)])
def test_mid_model_batches_follow_the_engine(gh, int8_mode, max_batch):
    """1024-hidden model: max_batch 2 (GEMV forms), 6 (burst GEMMs with the in-launch split-K reduction), 20 (tiled GEMM)."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg = MID
    w = random_model(cfg, seed=17)
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    op = gh.make_op(cfg, w, int8_mode=int8_mode)
    rng = np.random.RandomState(9)
    n_req = max_batch + 3
    prompts = [rng.randint(3, V, size=int(rng.randint(4, 60))).tolist() for _ in range(n_req)]
    new = [int(rng.randint(2, 12)) for _ in range(n_req)]
    cb = ContinuousBatcher(op, max_batch=max_batch, page_tokens=16, num_pages=6 * max_batch, max_seq_len=96)
    arrivals = {0: [(prompts[i], new[i]) for i in range(max_batch)], 4: [(prompts[i], new[i]) for i in range(max_batch, n_req)]}
    got, _ = _drain(cb, arrivals)
    for i in range(n_req):
        alone, logits = _alone(gh, op, prompts[i], new[i], V, end_id)
        for t, (a, b) in enumerate(zip(got[i], alone)):
            if a != b:  # the paged attention sums in another order than the split-KV kernel: only a near tie may flip a token
                top2 = np.sort(logits[t])[-2:]
                assert top2[1] - top2[0] <= 1e-2 * np.abs(logits[t]).max(), (i, t)
                break
        else:
            assert len(got[i]) == len(alone)
    assert cb.status()["free_pages"] == 6 * max_batch


def test_sampling_parameters_run_through_the_batcher(gh):
    """top-k / top-p / temperature rows together in one batch: tokens are valid, sequences end, pages come back."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    op = gh.make_op(cfg, w)
    cb = ContinuousBatcher(op, max_batch=4, page_tokens=8, num_pages=16, max_seq_len=40)
    ids = [cb.submit(z["prompt"].tolist(), 6, top_k=8, temperature=0.8, seed=1),
           cb.submit(z["prompt_b"].tolist(), 6, top_k=0, top_p=0.7, seed=2),
           cb.submit(z["prompt"][:5].tolist(), 6, top_k=40, top_p=0.5, temperature=1.3, seed=3),
           cb.submit(z["prompt_1"].tolist(), 6)]
    out = cb.run_all()
    assert sorted(out) == sorted(ids)
    for rid in ids:
        assert 1 <= len(out[rid]) <= 6 and all(0 <= t < cfg["vocab_size"] for t in out[rid])
    # same seeds, same arrival order -> the same tokens again
    ids2 = [cb.submit(z["prompt"].tolist(), 6, top_k=8, temperature=0.8, seed=1),
            cb.submit(z["prompt_b"].tolist(), 6, top_k=0, top_p=0.7, seed=2),
            cb.submit(z["prompt"][:5].tolist(), 6, top_k=40, top_p=0.5, temperature=1.3, seed=3),
            cb.submit(z["prompt_1"].tolist(), 6)]
    out2 = cb.run_all()
    assert [out2[i] for i in ids2] == [out[i] for i in ids]
    assert cb.status()["free_pages"] == 16


def test_cancel_returns_the_pages_and_leaves_the_others_untouched(gh):
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    op = gh.make_op(cfg, w)
    ref, _ = _alone(gh, op, z["prompt_b"].tolist(), 10, V, end_id)
    cb = ContinuousBatcher(op, max_batch=2, page_tokens=8, num_pages=12, max_seq_len=40)
    a = cb.submit(z["prompt"].tolist(), 12)
    b = cb.submit(z["prompt_b"].tolist(), 10)
    c = cb.submit(z["prompt"][:6].tolist(), 5)  # waits: two slots
    got = {}
    for rid, tok, fin in cb.step() + cb.step():
        got.setdefault(rid, []).append(tok)
    assert cb.cancel(a) and cb.cancel(c) and not cb.cancel(12345)
    while cb.busy():
        for rid, tok, fin in cb.step():
            got.setdefault(rid, []).append(tok)
    assert got[b] == ref and c not in got and len(got[a]) == 2
    assert cb.status() == {"waiting": 0, "running": 0, "free_pages": 12}


def _oracle_alone(model, prompt, n_new, end_id, **kw):
    """What `orc.Model.generate` -- the CPU restatement of the reference -- produces for this request by itself, cut where the
    batcher's stream of it ends: at end_id, or after an emitted stop sequence (stop_criteria_kernels.cu:24-83)."""
    from oracle import oracle as orc
    stop = kw.pop("stop_words", None)
    sw = None
    if stop:
        flat = [t for w in stop for t in w]
        sw = np.zeros((1, 2, len(flat)), np.int32)
        sw[0, 0] = flat
        sw[0, 1] = -1
        sw[0, 1, :len(stop)] = np.cumsum([len(w) for w in stop])
    sp = orc.Sampling(1, stop_words=sw, **kw)
    o = model.generate(np.asarray(prompt, np.int32)[None, :], [len(prompt)], n_new, sampling=sp)
    toks = o["output_ids"][0, len(prompt):].tolist()
    hist, out = list(prompt), []
    for t in toks:
        out.append(t)
        hist.append(t)
        if t == end_id or any(len(hist) >= len(w) and hist[-len(w):] == list(w) for w in (stop or [])):
            break
    return out


def test_every_request_matches_the_oracle_with_penalty_and_stop_words(gh):
    """The checker of the batcher is no longer only the engine: every request -- greedy, with a repetition penalty, with stop
    words, with top-k sampling -- is generated by the CPU oracle alone and compared token for token with the stream the
    batcher produced for it while it shared slots, pages and decode steps with the others."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    from oracle import oracle as orc
    from tests.helpers import weight_list_to_layers
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    layers, glob = weight_list_to_layers(cfg, w)
    model = orc.Model(dict(cfg, fp16=1), layers, glob)
    op = gh.make_op(cfg, w)
    # prompts and penalties whose greedy trajectories keep a top-2 margin above 9e-3 of max|logit| in the oracle (checked
    # when the test was written; the fp16 engine is within 5e-3 of it): a near tie would test the tie, not the batcher
    rng, rng2 = np.random.RandomState(17), np.random.RandomState(5)
    r19 = rng.randint(3, V, size=19).tolist()
    r7 = rng.randint(3, V, size=7).tolist()
    xs = [rng2.randint(3, V, size=int(rng2.randint(5, 24))).tolist() for _ in range(4)]
    pb = z["prompt_b"].tolist()
    # stop sequences taken from what the request generates WITHOUT them: one single-token, one two-token sequence
    free0 = _oracle_alone(model, pb, 12, end_id, top_k=1)
    free3 = _oracle_alone(model, pb, 12, end_id, top_k=1, repetition_penalty=1.3)
    reqs = [dict(prompt=pb, n=12, kw=dict(top_k=1), stop_words=[[free0[4]]]),
            dict(prompt=xs[3], n=10, kw=dict(top_k=1, repetition_penalty=1.5)),
            dict(prompt=r19, n=9, kw=dict(top_k=1)),
            dict(prompt=pb, n=12, kw=dict(top_k=1, repetition_penalty=1.3), stop_words=[[7, 8, 9], free3[5:7]]),
            dict(prompt=xs[0], n=8, kw=dict(top_k=1, repetition_penalty=0.8)),
            dict(prompt=r7, n=6, kw=dict(top_k=1, repetition_penalty=0.8))]
    ref = [_oracle_alone(model, r["prompt"], r["n"], end_id, stop_words=r.get("stop_words"), **r["kw"]) for r in reqs]
    assert len(ref[0]) == 5 and len(ref[3]) == 7  # the stop sequences really end those two requests early
    cb = ContinuousBatcher(op, max_batch=3, page_tokens=8, num_pages=24, max_seq_len=64)
    ids = {}
    arrivals = {0: [0, 1], 1: [2], 3: [3, 4], 6: [5]}
    got, it = {}, 0
    while arrivals or cb.busy():
        for k in arrivals.pop(it, []):
            r = reqs[k]
            kw = dict(r["kw"])
            ids[cb.submit(r["prompt"], r["n"], top_k=kw.get("top_k", 0), top_p=kw.get("top_p", 0.0),
                          temperature=kw.get("temperature", 1.0), seed=kw.get("random_seed", 0),
                          repetition_penalty=kw.get("repetition_penalty", 1.0), stop_words=r.get("stop_words"))] = k
        for rid, tok, fin in cb.step():
            got.setdefault(ids[rid], []).append(tok)
        it += 1
        assert it < 1000
    for k in range(len(reqs)):
        assert got[k] == ref[k], (k, got[k], ref[k])
    assert cb.status()["free_pages"] == 24


def test_a_seeded_sampling_request_is_the_same_alone_and_in_a_busy_batcher(gh):
    """The uniform draws depend on the request's seed and draw count only (curand_init(seed, 0, 0) per row in the reference:
    sampling_topk_kernels.cu:32-65), not on the admission row or the slot a request lands in."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V = cfg["vocab_size"]
    op = gh.make_op(cfg, w)
    prompt = z["prompt"].tolist()
    kw = dict(top_k=8, top_p=0.95, temperature=1.1, seed=77)
    cb = ContinuousBatcher(op, max_batch=4, page_tokens=8, num_pages=32, max_seq_len=64)
    rid = cb.submit(prompt, 10, **kw)
    alone = cb.run_all()[rid]
    rng = np.random.RandomState(2)
    for i in range(3):  # occupy slots 0..2 first: the request lands in slot 3 and is admitted as a later row of a batch
        cb.submit(rng.randint(3, V, size=11 + i).tolist(), 14, top_k=1)
    cb.step()
    rid2 = cb.submit(prompt, 10, **kw)
    busy = cb.run_all()[rid2]
    assert busy == alone, (busy, alone)


@pytest.mark.parametrize("chunk", [8, 4])
def test_a_long_prompt_is_admitted_in_chunks_between_decode_steps(gh, monkeypatch, chunk):
    """With a slot running, a prompt longer than the chunk is prefilled alone, chunk by chunk, and the running slots decode one
    token after every chunk but the last (the admission no longer stalls them for the whole prompt): in the step() call that
    admits a 19-token prompt in chunks of 8, the running request receives 1 + 2 tokens.  Every stream still equals what the CPU
    oracle generates for the request alone (the chunked prompt phase is the same arithmetic row by row).  The token callback
    sees every event as it is produced; with chunks of 4 the iteration produces 9 events for event arrays of 6: the rest comes
    with the next call, before a new iteration runs."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    from oracle import oracle as orc
    from tests.helpers import weight_list_to_layers
    monkeypatch.setenv("FTCF_BATCHER_PREFILL_CHUNK", str(chunk))
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    layers, glob = weight_list_to_layers(cfg, w)
    model = orc.Model(dict(cfg, fp16=1), layers, glob)
    op = gh.make_op(cfg, w)
    rng = np.random.RandomState(17)
    r19 = rng.randint(3, V, size=19).tolist()
    r7 = rng.randint(3, V, size=7).tolist()
    pb = z["prompt_b"].tolist()
    reqs = [(pb, 12), (r19, 9), (r7, 6)]
    ref = [_oracle_alone(model, p, n, end_id, top_k=1) for p, n in reqs]
    cb = ContinuousBatcher(op, max_batch=3, page_tokens=8, num_pages=24, max_seq_len=64)
    ids, got, per_step, streamed, returned = {}, {}, [], [], []
    cb.set_token_callback(lambda rid, tok, fin: streamed.append((rid, tok, fin)))
    arrivals = {0: [0], 2: [1, 2]}
    it = 0
    while arrivals or cb.busy():
        for k in arrivals.pop(it, []):
            ids[cb.submit(reqs[k][0], reqs[k][1], top_k=1)] = k
        evs = cb.step()
        returned += evs
        per_step.append([ids[rid] for rid, _, _ in evs])
        for rid, tok, fin in evs:
            got.setdefault(ids[rid], []).append(tok)
        it += 1
        assert it < 1000
    for k in range(len(reqs)):
        assert got[k] == ref[k], (k, got[k], ref[k])
    assert streamed == returned
    if chunk == 8:
        # call 2: one decode step, then the 19-token prompt in chunks of 8, 8, 3 with a decode step after the first two; the
        # 7-token prompt in one piece
        assert per_step[2] == [0, 0, 0, 1, 2], per_step[:4]
    else:
        # chunks of 4, 4, 4, 4, 3 (four decode steps in between), then the 7-token prompt as 4 + 3 with one decode step of the
        # now TWO running requests in between: 9 events for arrays of 6
        assert per_step[2] == [0, 0, 0, 0, 0, 1] and per_step[3] == [0, 1, 2], per_step[:5]
    cb.set_token_callback(None)
    assert cb.status()["free_pages"] == 24


def test_decode_steps_inside_an_admission_use_the_running_slots_sampling_parameters(gh, monkeypatch):
    """A request admitted a moment ago decodes inside the NEXT request's chunked admission: the host-side view of the running
    slots' sampling parameters (largest top_k, any top-p row, temperature) must include it by then.  A seeded top-k / top-p
    request that is admitted in chunks right before another long prompt produces the same tokens as alone."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    monkeypatch.setenv("FTCF_BATCHER_PREFILL_CHUNK", "4")
    cfg, w, z = load_tiny()
    V = cfg["vocab_size"]
    op = gh.make_op(cfg, w)
    rng = np.random.RandomState(23)
    long_a = rng.randint(3, V, size=17).tolist()
    long_b = rng.randint(3, V, size=15).tolist()
    kw = dict(top_k=8, top_p=0.95, temperature=1.1, seed=5)
    cb = ContinuousBatcher(op, max_batch=3, page_tokens=8, num_pages=24, max_seq_len=64)
    rid = cb.submit(long_a, 10, **kw)
    alone = cb.run_all()[rid]
    g0 = cb.submit(z["prompt_b"].tolist(), 14, top_k=1)  # a greedy request is running (largest top_k so far: 1)
    cb.step()
    cb.step()
    rid2 = cb.submit(long_a, 10, **kw)                      # admitted in chunks; then decodes inside long_b's admission
    g1 = cb.submit(long_b, 6, top_k=1)
    out = cb.run_all()
    assert out[rid2] == alone, (out[rid2], alone)
    assert len(out[g1]) <= 6 and g0 in out


@pytest.mark.parametrize("tp,max_batch,overlap", [(2, 3, 0), (2, 6, 0), (4, 3, 0), (2, 6, 1), (2, 19, 1), (4, 5, 1)])
def test_tensor_parallel_batchers_follow_the_single_gpu_engine(gh, monkeypatch, tp, max_batch, overlap):
    """Tensor parallelism inside the batcher (round 4; the reference's serving layer runs TP through its Triton backend,
    triton_backend/gptneox/GptNeoXTritonModelInstance.cc): one batcher per rank over its shard, fed the same requests in the
    same order -- the schedulers take identical decisions, the decode step's per-layer all-reduce and the vocabulary-split LM
    head are the engine's collectives.  The ranks are threads of a local group on this one GPU (tests/test_gpu_tp_local.py);
    every rank must emit the same events, and every request what the TP = 1 engine generates for it alone.
    overlap = 1 (round 5): the decode step's layers as two micro-batches of slots on two streams with the all-reduce on a third
    (FTCF_DECODE_OVERLAP=1; GptNeoXDecoder.cc:342-359 has it in line) -- same events, and the stats say it ran."""
    import threading
    monkeypatch.setenv("FTCF_DECODE_OVERLAP", str(overlap))
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    from fastertransformer4codefuse_amd.gptneox_op import LocalTensorParallelGroup
    from tests.helpers import shard_weights
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    rng = np.random.RandomState(8)
    prompts = [z["prompt"].tolist(), z["prompt_b"].tolist(), z["prompt"][:5].tolist(), rng.randint(3, V, size=21).tolist(),
               z["prompt_1"].tolist(), rng.randint(3, V, size=9).tolist()]
    new = [8, 6, 11, 9, 5, 7]
    op1 = gh.make_op(cfg, w)
    ref = [_alone(gh, op1, p, n, V, end_id)[0] for p, n in zip(prompts, new)]
    # with room for it, a beam request of width 3 rides along (its admission is the engine's own tensor-parallel beam request)
    beam = (prompts[1], 7, 3) if max_batch >= 6 else None
    beam_ref = _beam_alone(gh, op1, beam[0], beam[1], V, beam[2]) if beam else None
    del op1
    arrivals = {0: [0, 1], 2: [2, 3], 3: [4], 7: [5]}
    group = LocalTensorParallelGroup()
    res, err = [None] * tp, []
    beams = [None] * tp
    gate = threading.Barrier(tp)

    def worker(r):
        try:
            op = gh.make_op(cfg, shard_weights(cfg, w, tp, r), tp=tp, rank=r, comm=group)
            cb = ContinuousBatcher(op, max_batch=max_batch, page_tokens=8, num_pages=32 + 8 * max_batch, max_seq_len=64)
            free0 = cb.status()["free_pages"]
            events, ids, it, pending = [], {}, 0, dict(arrivals)
            while pending or cb.busy():
                for k in pending.pop(it, []):
                    ids[cb.submit(prompts[k], new[k])] = k
                if beam and it == 1:
                    brid = cb.submit_beam(beam[0], beam[1], beam[2])
                    ids[brid] = "beam"
                gate.wait(timeout=120)  # (the ranks step together, like the ranks of a serving job)
                events.append([(ids[rid], tok, fin) for rid, tok, fin in cb.step()])
                it += 1
                assert it < 2000
            if beam:
                beams[r] = cb.beam_result(brid)
            assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}
            assert op.stats()["decode_overlap"] == (1 if overlap and max_batch >= 4 else 0)
            res[r] = events
        except BaseException as e:  # noqa: BLE001
            err.append((r, repr(e)))
            gate.abort()

    ths = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(tp)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=400)
    assert not err, err
    assert all(x is not None for x in res), "a rank did not finish (stuck in a collective?)"
    for r in range(1, tp):
        assert res[r] == res[0], f"rank {r} saw different events"
    got = {}
    for evs in res[0]:
        for k, tok, fin in evs:
            got.setdefault(k, []).append(tok)
    for k in range(len(prompts)):
        assert got[k] == ref[k], (k, got[k], ref[k])
    if beam:
        assert got["beam"] == [-1]
        for r in range(tp):
            assert np.array_equal(beams[r][0], beam_ref[0]) and np.array_equal(beams[r][1], beam_ref[1]), (r, beams[r], beam_ref)
            np.testing.assert_allclose(beams[r][2], beam_ref[2], rtol=5e-3, atol=5e-3)


def _beam_alone(gh, op, prompt, n_new, V, K, **kw):
    """What GptNeoXOp.forward returns for this prompt by itself with beam_width K."""
    r = gh.run_op_beam(op, np.asarray(prompt, np.int32)[None, :], [len(prompt)], n_new, V, K, **kw)
    return r["output_ids"][0], r["sequence_lengths"][0], r["cum_log_probs"].reshape(-1)


@pytest.mark.parametrize("page_tokens", [8, 16])
def test_beam_requests_share_the_batcher_with_greedy_ones(gh, page_tokens):
    """Beam search inside the batcher (K consecutive slots, pages shared copy-on-write instead of the reference's cache
    indirection): every hypothesis, its length and its score are what the engine's own beam search returns for the prompt
    alone, while greedy requests come and go around it and stay what they are alone; every page comes back."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    op = gh.make_op(cfg, w)
    rng = np.random.RandomState(5)
    greedy = [z["prompt"].tolist(), rng.randint(3, V, size=11).tolist(), z["prompt_b"].tolist()]
    g_new = [10, 7, 9]
    g_ref = [_alone(gh, op, p, n, V, end_id)[0] for p, n in zip(greedy, g_new)]
    beams = [(z["prompt"].tolist(), 9, 3, {}), (z["prompt_b"][:6].tolist(), 12, 4, dict(beam_search_diversity_rate=[-0.4])),
             (rng.randint(3, V, size=17).tolist(), 6, 2, dict(repetition_penalty=[1.3], temperature=[0.8]))]
    b_ref = [_beam_alone(gh, op, p, n, V, K, **kw) for p, n, K, kw in beams]
    cb = ContinuousBatcher(op, max_batch=6, page_tokens=page_tokens, num_pages=200 // page_tokens + 8, max_seq_len=48)
    free0 = cb.status()["free_pages"]
    ids, bids, got, done = {}, {}, {}, set()

    def scalar(kw, key, default):
        return float(kw[key][0]) if key in kw else default

    def submit_beam(i):
        p, n, K, kw = beams[i]
        bids[cb.submit_beam(p, n, K, scalar(kw, "beam_search_diversity_rate", 0.0), 0.0, scalar(kw, "temperature", 1.0),
                            scalar(kw, "repetition_penalty", 1.0))] = i

    arrivals = {0: [("g", 0), ("b", 0)], 1: [("g", 1)], 3: [("b", 1)], 4: [("g", 2)], 6: [("b", 2)]}
    it = 0
    while arrivals or cb.busy():
        for kind, i in arrivals.pop(it, []):
            if kind == "g":
                ids[cb.submit(greedy[i], g_new[i])] = i
            else:
                submit_beam(i)
        for rid, tok, fin in cb.step():
            if rid in bids:
                assert tok == -1 and fin
                done.add(rid)
            else:
                got.setdefault(ids[rid], []).append(tok)
        it += 1
        assert it < 2000
    for i in range(len(greedy)):
        assert got[i] == g_ref[i], (i, got[i], g_ref[i])
    assert done == set(bids)
    for rid, i in bids.items():
        out, lens, cum = cb.beam_result(rid)
        ref_out, ref_lens, ref_cum = b_ref[i]
        assert out.shape == ref_out.shape, (i, out.shape, ref_out.shape)
        assert np.array_equal(out, ref_out), (i, out, ref_out)
        assert np.array_equal(lens, ref_lens), (i, lens, ref_lens)
        np.testing.assert_allclose(cum, ref_cum, rtol=2e-3, atol=2e-3)
        assert cb.beam_result(rid) is None  # fetched once
    assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}


def test_a_beam_request_with_a_long_prompt_is_admitted_in_chunks(gh, monkeypatch):
    """Round 5: a beam request whose prompt is longer than the chunk, arriving while a greedy request runs, has its prompt phase
    cut into chunks with a decode step of the running slot after every chunk but the last (the engine's chunked context decoder
    takes the beam request's cache tile); its hypotheses, lengths and scores are what the engine's own beam search returns for the
    prompt alone, the greedy request stays what it is alone, and the token callback sees the group's finished event exactly once."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    monkeypatch.setenv("FTCF_BATCHER_PREFILL_CHUNK", "8")
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    op = gh.make_op(cfg, w)
    rng = np.random.RandomState(23)
    greedy, g_new = z["prompt_b"].tolist(), 14
    g_ref = _alone(gh, op, greedy, g_new, V, end_id)[0]
    bp, b_new, K = rng.randint(3, V, size=21).tolist(), 7, 3
    b_ref = _beam_alone(gh, op, bp, b_new, V, K)
    cb = ContinuousBatcher(op, max_batch=4, page_tokens=8, num_pages=40, max_seq_len=48)
    free0 = cb.status()["free_pages"]
    streamed = []
    cb.set_token_callback(lambda rid, tok, fin: streamed.append((rid, tok, fin)))
    gid = cb.submit(greedy, g_new)
    per_step, got, bid = [], [], None
    it = 0
    while it == 0 or cb.busy():
        if it == 1:
            bid = cb.submit_beam(bp, b_new, K, 0.0, 0.0, 1.0, 1.0)
        evs = cb.step()
        per_step.append([(rid == gid) for rid, _, _ in evs])
        got += [tok for rid, tok, _ in evs if rid == gid]
        it += 1
        assert it < 500
    assert got == g_ref, (got, g_ref)
    # the call that admits the 21-token prompt in chunks of 8, 8, 5: one decode step of its own + one after each of the first two chunks
    assert per_step[1] == [True, True, True], per_step[:3]
    out, lens, cum = cb.beam_result(bid)
    # (the chunked prompt phase runs its GEMMs on other row counts than the one-piece one: the same arithmetic per row up to the
    #  split-K form, so a near tie may order two hypotheses differently; the tiny model's are apart)
    assert np.array_equal(out, b_ref[0]), (out, b_ref[0])
    assert np.array_equal(lens, b_ref[1])
    np.testing.assert_allclose(cum, b_ref[2], rtol=5e-3, atol=5e-3)
    assert [e for e in streamed if e[0] == bid] == [(bid, -1, 1)]
    cb.set_token_callback(None)
    assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}


def test_a_cancelled_beam_request_returns_its_pages(gh):
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    op = gh.make_op(cfg, w)
    cb = ContinuousBatcher(op, max_batch=4, page_tokens=8, num_pages=40, max_seq_len=48)
    free0 = cb.status()["free_pages"]
    rid = cb.submit_beam(z["prompt"].tolist(), 20, 4)
    for _ in range(5):
        assert all(r != rid for r, _, _ in cb.step())
    assert cb.status()["free_pages"] < free0
    assert cb.cancel(rid)
    assert cb.status()["free_pages"] == free0 and cb.beam_result(rid) is None
    # the slots serve the next request
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    ref = _alone(gh, op, z["prompt_b"].tolist(), 6, V, end_id)[0]
    got, _ = _drain(cb, {0: [(z["prompt_b"].tolist(), 6)]})
    assert got[0] == ref


@pytest.mark.parametrize("int8_mode", [0, 1])
def test_beams_that_finish_on_end_id_inside_the_batcher(gh, int8_mode):
    """Beams end on end_id at different steps (a finished beam appends nothing, keeps the pages it shares and offers only its end
    token), the group leaves when all of them have or at max_new_tokens -- on the 512-hidden model, fp16 and int8, next to sampling
    neighbours: the hypotheses are the engine's own."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg = dict(MID)
    w = random_model(cfg, seed=21)
    V = cfg["vocab_size"]
    rng = np.random.RandomState(9)
    prompts = [rng.randint(3, V, size=n).tolist() for n in (19, 7, 33)]
    K, n_new = 4, 16
    free = _beam_alone(gh, gh.make_op(cfg, w, int8_mode=int8_mode), prompts[0], n_new, V, K)
    # a token the best hypothesis emits mid-way becomes the end token: beams finish during the search
    cfg2 = dict(cfg, end_id=int(free[0][0, len(prompts[0]) + 5]))
    op = gh.make_op(cfg2, w, int8_mode=int8_mode)
    refs = [_beam_alone(gh, op, p, n_new, V, K) for p in prompts]
    assert any((r[1] < len(p) + n_new).any() for r, p in zip(refs, prompts))  # the case really finishes beams early
    cb = ContinuousBatcher(op, max_batch=10, page_tokens=8, num_pages=120, max_seq_len=64)
    free0 = cb.status()["free_pages"]
    bids = {}
    arrivals = {0: [0], 2: [1], 3: [2]}
    it, done = 0, set()
    noise = 0
    while arrivals or cb.busy():
        for i in arrivals.pop(it, []):
            bids[cb.submit_beam(prompts[i], n_new, K)] = i
        if it in (1, 4):  # sampled neighbours in the slots around the groups
            cb.submit(rng.randint(3, V, size=11).tolist(), 9, top_k=4, top_p=0.9, temperature=0.7, seed=it)
            noise += 1
        for rid, tok, fin in cb.step():
            if rid in bids and fin:
                done.add(rid)
        it += 1
        assert it < 2000
    assert done == set(bids)
    for rid, i in bids.items():
        out, lens, cum = cb.beam_result(rid)
        assert np.array_equal(out, refs[i][0]), (i, out, refs[i][0])
        assert np.array_equal(lens, refs[i][1]), (i, lens, refs[i][1])
        np.testing.assert_allclose(cum, refs[i][2], rtol=5e-3, atol=5e-3)
    assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_mix_of_beam_and_greedy_requests_in_a_tight_pool(gh, seed):
    """Stress of the page accounting: beam groups (their budgets, shared prompt pages, copy-on-write) and greedy requests arrive at
    random into a pool too small for all of them -- admissions wait for pages and for K consecutive slots -- and every request still
    is what the engine produces for it alone; the pool ends full."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    op = gh.make_op(cfg, w)
    rng = np.random.RandomState(100 + seed)
    reqs = []
    for i in range(14):
        n = int(rng.randint(2, 20))
        prompt = rng.randint(3, V, size=n).tolist()
        new = int(rng.randint(2, 11))
        K = int(rng.choice([1, 1, 2, 3, 4]))
        reqs.append((prompt, new, K))
    refs = []
    for prompt, new, K in reqs:
        refs.append(_alone(gh, op, prompt, new, V, end_id)[0] if K == 1 else _beam_alone(gh, op, prompt, new, V, K))
    cb = ContinuousBatcher(op, max_batch=6, page_tokens=8, num_pages=26, max_seq_len=32)
    free0 = cb.status()["free_pages"]
    arrive = sorted((int(rng.randint(0, 25)), i) for i in range(len(reqs)))
    ids, got, done, it = {}, {}, set(), 0
    while arrive or cb.busy():
        while arrive and arrive[0][0] <= it:
            _, i = arrive.pop(0)
            prompt, new, K = reqs[i]
            ids[cb.submit(prompt, new) if K == 1 else cb.submit_beam(prompt, new, K)] = i
        for rid, tok, fin in cb.step():
            i = ids[rid]
            if reqs[i][2] == 1:
                got.setdefault(i, []).append(tok)
            else:
                assert tok == -1 and fin
                done.add(i)
        st = cb.status()
        assert 0 <= st["free_pages"] <= free0
        it += 1
        assert it < 5000
    for i, (prompt, new, K) in enumerate(reqs):
        if K == 1:
            assert got[i] == refs[i], (i, got[i], refs[i])
        else:
            assert i in done
            rid = [r for r, j in ids.items() if j == i][0]
            out, lens, cum = cb.beam_result(rid)
            assert np.array_equal(out, refs[i][0]) and np.array_equal(lens, refs[i][1]), (i, out, refs[i][0])
            # (scores: the engine ran the K rows alone, the batcher among six -- other GEMM forms, other summation orders)
            np.testing.assert_allclose(cum, refs[i][2], rtol=1e-2, atol=1e-2)
    assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}


def _beam_capi(op, prompt, n_new, K, min_length=None, stop_words=None):
    """The engine's beam search through the C ABI's argument block (min_length is not reachable through GptNeoXOp.forward)."""
    import ctypes as C
    from fastertransformer4codefuse_amd import capi
    S = len(prompt)
    ids = torch.tensor([prompt], dtype=torch.int32, device="cuda")
    lens = torch.tensor([S], dtype=torch.int32, device="cuda")
    out_ids = torch.zeros((1, K, S + n_new), dtype=torch.int32, device="cuda")
    seq = torch.zeros((1, K), dtype=torch.int32, device="cuda")
    cum = torch.zeros((1, K), dtype=torch.float32, device="cuda")
    fa = capi.ForwardArgs()
    fa.input_ids, fa.input_lengths = ids.data_ptr(), lens.data_ptr()
    fa.batch_size, fa.max_input_len, fa.output_len, fa.beam_width = 1, S, n_new, K
    keep = []
    if min_length is not None:
        ml = np.array([min_length], dtype=np.int32)
        keep.append(ml)
        fa.min_length, fa.n_min_length = ml.ctypes.data, 1
    if stop_words is not None:
        sw = torch.from_numpy(np.ascontiguousarray(stop_words, dtype=np.int32)).cuda()
        keep.append(sw)
        fa.stop_words_list, fa.stop_words_len = sw.data_ptr(), stop_words.shape[2]
    fa.return_cum_log_probs = 1
    fa.output_ids, fa.sequence_lengths, fa.cum_log_probs = out_ids.data_ptr(), seq.data_ptr(), cum.data_ptr()
    capi.check(capi.lib().ftcf_gptneox_forward(op._h, C.byref(fa)))
    torch.cuda.synchronize()
    return out_ids[0].cpu().numpy(), seq[0].cpu().numpy(), cum[0].cpu().numpy()


def test_beam_requests_with_stop_words_and_min_length_in_the_batcher(gh):
    """ftcf_batcher_submit_beam_ex: min_length holds the end token back, stop words finish a beam along its parent chain -- the
    hypotheses are what the engine's own beam search returns for the same arguments."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    cfg, w, z = load_tiny()
    V = cfg["vocab_size"]
    prompt, K, n_new = z["prompt"].tolist(), 3, 12
    free = _beam_capi(gh.make_op(cfg, w), prompt, n_new, K)
    S = len(prompt)
    # the best hypothesis' 4th new token becomes the end token; its 7th and 8th new tokens a stop sequence
    cfg2 = dict(cfg, end_id=int(free[0][0, S + 3]))
    op = gh.make_op(cfg2, w)
    base = _beam_capi(op, prompt, n_new, K)
    stop_list = [[int(base[0][1, S + 1]), int(base[0][1, S + 2])]]
    sw = np.full((1, 2, 2), -1, np.int32)
    sw[0, 0, :] = stop_list[0]
    sw[0, 1, 0] = 2
    cases = {"plain": {}, "min_length": dict(min_length=8), "stop": dict(stop_words=sw), "both": dict(min_length=6, stop_words=sw)}
    refs = {k: _beam_capi(op, prompt, n_new, K, **kw) for k, kw in cases.items()}
    assert not np.array_equal(refs["plain"][0], refs["min_length"][0])  # the arguments matter on this prompt ...
    assert not np.array_equal(refs["plain"][1], refs["stop"][1]) or not np.array_equal(refs["plain"][0], refs["stop"][0])
    cb = ContinuousBatcher(op, max_batch=7, page_tokens=8, num_pages=80, max_seq_len=48)
    free0 = cb.status()["free_pages"]
    rids = {}
    for name, kw in cases.items():
        rids[cb.submit_beam(prompt, n_new, K, min_length=kw.get("min_length", 0),
                            stop_words=stop_list if "stop_words" in kw else None)] = name
    cb.submit(z["prompt_b"].tolist(), 9)  # a greedy neighbour
    it = 0
    while cb.busy():
        cb.step()
        it += 1
        assert it < 2000
    for rid, name in rids.items():
        out, lens, cum = cb.beam_result(rid)
        assert np.array_equal(out, refs[name][0]), (name, out, refs[name][0])
        assert np.array_equal(lens, refs[name][1]), (name, lens, refs[name][1])
        np.testing.assert_allclose(cum, refs[name][2], rtol=1e-2, atol=1e-2)
    assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}


@pytest.mark.parametrize("model,int8_mode,page_tokens", [("tiny", 0, 32), ("mid", 1, 16), ("mid", 0, 32)])  # (tiny: size_per_head 64, blocks of 32 keys)
def test_the_decode_step_runs_the_rows_kernel_over_paged_kv(gh, monkeypatch, model, int8_mode, page_tokens):
    """Round 5: with tensor_para_size 1, at most 16 slots and page_tokens a multiple of the attention block (16 keys at
    size_per_head 128), the layers of a decode step are ONE launch of the rows kernel reading K/V through the page table
    (csrc/rows_device.hip.h, PAGED).  Requests arriving over time -- greedy, sampled, a beam group with copy-on-write pages -- are what
    the engine produces for each of them alone, and what the batcher produces on its per-GEMM launches (FTCF_BATCHER_ROWS=0)."""
    from fastertransformer4codefuse_amd.batcher import ContinuousBatcher
    if model == "tiny":
        cfg, w, z = load_tiny()
        prompts = [z["prompt"].tolist(), z["prompt_b"].tolist()]
    else:
        cfg = dict(head_num=8, size_per_head=128, inter_size=4096, num_layer=2, vocab_size=2048, rotary_dim=32, start_id=0, end_id=2)
        w = random_model(cfg, seed=5 + int8_mode, std=0.04)
        prompts = []
    V, end_id = cfg["vocab_size"], cfg["end_id"]
    rng = np.random.RandomState(31 + page_tokens)
    prompts += [rng.randint(3, V, size=n).tolist() for n in (23, 9, 40, 17, 5)]
    new = [12, 20, 7, 15, 9, 11, 6][:len(prompts)]
    op = gh.make_op(cfg, w, int8_mode=int8_mode)
    ref = [_alone(gh, op, p, n, V, end_id)[0] for p, n in zip(prompts, new)]
    bp, b_new, K = rng.randint(3, V, size=14).tolist(), 8, 3
    b_ref = _beam_alone(gh, op, bp, b_new, V, K)
    results = {}
    for form in ("1", "0"):
        monkeypatch.setenv("FTCF_BATCHER_ROWS", form)
        cb = ContinuousBatcher(op, max_batch=8, page_tokens=page_tokens, num_pages=48, max_seq_len=64)
        free0 = cb.status()["free_pages"]
        ids, got, bid = {}, {}, None
        arrivals = {0: [0, 1], 1: [2], 2: ["beam"], 4: [3, 4], 7: list(range(5, len(prompts)))}
        it = 0
        while arrivals or cb.busy():
            for k in arrivals.pop(it, []):
                if k == "beam":
                    bid = cb.submit_beam(bp, b_new, K, 0.0, 0.0, 1.0, 1.0)
                else:
                    ids[cb.submit(prompts[k], new[k])] = k
            for rid, tok, fin in cb.step():
                if rid != bid:
                    got.setdefault(ids[rid], []).append(tok)
            if it == 3:
                assert op.stats()["decode_path"] == (3 if form == "1" else 2)  # the layers of the last decode step
            it += 1
            assert it < 2000
        results[form] = (got, cb.beam_result(bid))
        assert cb.status() == {"waiting": 0, "running": 0, "free_pages": free0}
    for form, (got, beam) in results.items():
        for k in range(len(prompts)):
            if got[k] != ref[k]:
                # (another summation order than the engine alone: a flip must be a near tie -- not expected on these models)
                raise AssertionError((form, k, got[k], ref[k]))
        assert np.array_equal(beam[0], b_ref[0]) and np.array_equal(beam[1], b_ref[1]), form
        np.testing.assert_allclose(beam[2], b_ref[2], rtol=5e-3, atol=5e-3)
