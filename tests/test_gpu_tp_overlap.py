"""-m gpu: row n1 -- batched decode under tensor parallelism with the layer's all-reduce OFF the compute streams (engine.hip.h
decoder_overlapped: two micro-batches on two streams, the reductions on a third; GptNeoXDecoder.cc:342-359 has the all-reduce in line).
The ranks are engine instances in threads of this process joined by a local group, as in tests/test_gpu_tp_local.py (batches above two
rows take the general path whatever FTCF_TP_PERSIST says, so this file does not run under that file's two-path fixture)."""
import threading

import numpy as np
import pytest

from tests.helpers import random_model, shard_weights
from tests.test_gpu_tp_local import MID

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


def check_ragged(ref, got, lens, out, what, frac, end_id):
    """check_against for ragged rows: row b's token of step t sits at position lens[b] + t (GptNeoX.cc:776-1048 walks every row
    from its own prompt end); a row is followed until its first token flip (which must be a near tie) or its end_id."""
    scale = np.abs(ref["logits"]).max()
    for b, n in enumerate(lens):
        for t in range(out):
            err = np.abs(got["logits"][t, b] - ref["logits"][t, b]).max() / scale
            assert err <= frac, (what, b, t, err)
            if got["output_ids"][b, n + t] != ref["output_ids"][b, n + t]:
                top2 = np.sort(ref["logits"][t, b])[-2:]
                assert top2[1] - top2[0] <= 2 * frac * scale, (what, b, t, "token flip without a near tie")
                break
            if ref["output_ids"][b, n + t] == end_id:
                break


@pytest.mark.parametrize("rows,int8_mode,tp", [(4, 0, 2), (7, 1, 2), (16, 1, 4), (23, 0, 2), (32, 1, 2)])
def test_batched_decode_with_overlapped_all_reduce_is_bit_identical(gh, monkeypatch, rows, int8_mode, tp):
    """Batched decode under tensor parallelism (4..32 rows on the general path, BASELINE config 5's regime): with
    FTCF_DECODE_OVERLAP=1 the batch is cut in two micro-batches and a micro-batch's per-layer all-reduce runs on the side stream
    under the other micro-batch's GEMMs and attention (GptNeoXDecoder.cc:342-359 has it in line on the compute stream).  A row's
    arithmetic does not depend on which rows share its launches (the burst GEMM's K slices depend on k only), so tokens,
    logits and cum_log_probs must equal the un-overlapped path's bit for bit on every rank -- with ragged prompts, rows that
    finish early and an odd row count (micro-batches of 4 + 3) -- and `decode_overlap` in the stats says which ran.  Above 16
    rows the un-overlapped path is the tiled GEMM over all rows, the micro-batches (12 + 11, 16 + 16) stay on the burst GEMM:
    close, not equal."""
    cfg = MID
    w = random_model(cfg, seed=33)
    rng = np.random.RandomState(rows)
    S = 24
    lens = [int(v) for v in rng.randint(5, S + 1, size=rows)]
    lens[0] = S
    ids = rng.randint(3, cfg["vocab_size"], size=(rows, S)).astype(np.int32)
    for b, n in enumerate(lens):
        ids[b, n:] = cfg["end_id"]
    out = 6
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FTCF_DECODE_OVERLAP", mode)
        from fastertransformer4codefuse_amd.gptneox_op import LocalTensorParallelGroup
        group = LocalTensorParallelGroup()
        res, err = [None] * tp, []

        def worker(r):
            try:
                op = gh.make_op(cfg, shard_weights(cfg, w, tp, r), int8_mode=int8_mode, tp=tp, rank=r, comm=group)
                res[r] = gh.run_op(op, ids, lens, out, cfg["vocab_size"], top_k=1)
                res[r]["stats"] = op.stats()
            except BaseException as e:  # noqa: BLE001
                err.append((r, e))

        ths = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(tp)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not err, err
        got[mode] = res
    for r in range(tp):
        assert got["1"][r]["stats"]["decode_overlap"] == 1 and got["0"][r]["stats"]["decode_overlap"] == 0
        assert got["1"][r]["stats"]["decode_path"] == 2
        if rows <= 16:
            assert got["1"][r]["output_ids"].tolist() == got["0"][r]["output_ids"].tolist()
            np.testing.assert_array_equal(got["1"][r]["logits"], got["0"][r]["logits"])
            np.testing.assert_array_equal(got["1"][r]["cum_log_probs"], got["0"][r]["cum_log_probs"])
        else:
            # above 16 rows the un-overlapped path runs the tiled GEMM on all rows at once, the micro-batches the burst GEMM on
            # <= 16 each: same sums in another order
            assert got["1"][r]["output_ids"].tolist() == got["1"][0]["output_ids"].tolist()
            check_ragged(got["0"][r], got["1"][r], lens, out, f"overlapped decode, {rows} rows vs the un-overlapped path", 5e-3,
                         cfg["end_id"])
    # and the overlapped ranks against the TP = 1 engine (rows that have emitted end_id are not compared any further)
    op1 = gh.make_op(cfg, w, int8_mode=int8_mode)
    r1 = gh.run_op(op1, ids, lens, out, cfg["vocab_size"], top_k=1)
    check_ragged(r1, got["1"][0], lens, out, f"overlapped decode, {rows} rows vs tp1 engine", 5e-3 if int8_mode == 0 else 2e-2,
                 cfg["end_id"])

