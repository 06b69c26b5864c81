"""Continuous batching over a paged K/V cache (include/ftcf.h `ftcf_batcher_*`).

    op = GptNeoXOp(...)                       # fp16 / int8 engine, tensor_para_size 1
    cb = ContinuousBatcher(op, max_batch=8, page_tokens=64, num_pages=512, max_seq_len=2048)
    rid = cb.submit(prompt_ids, max_new_tokens=128)          # greedy; top_k / top_p / temperature / seed /
                                                             # repetition_penalty / stop_words optional
    bid = cb.submit_beam(prompt_ids, 128, beam_width=4)      # beam search: one event (bid, -1, True), then cb.beam_result(bid)
    while cb.busy():
        for request_id, token, finished in cb.step():
            ...

The engine `op` must not run a request of its own while a batcher call is in progress."""
import ctypes as C

import numpy as np

from . import capi


class ContinuousBatcher:
    def __init__(self, op, max_batch, page_tokens, num_pages, max_seq_len):
        self._op = op  # keeps the engine (and its weights) alive
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        capi.check(capi.lib().ftcf_batcher_create(op._h, int(max_batch), int(page_tokens), int(num_pages), int(max_seq_len),
                                                  C.byref(self._h)))
        n = 2 * self.max_batch  # (a chunked admission can produce more events per iteration: the rest comes with the next call)
        self._ids = (C.c_long * n)()
        self._tok = (C.c_int * n)()
        self._fin = (C.c_int * n)()
        self._cb = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                capi.lib().ftcf_batcher_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def submit(self, prompt_ids, max_new_tokens, top_k=0, top_p=0.0, temperature=1.0, seed=0, repetition_penalty=1.0,
               stop_words=None):
        """stop_words: a list of token-id lists (each one a stop sequence), or None.  The request ends after a stop
        sequence has been emitted (stop_criteria_kernels.cu:24-83), like GptNeoXOp.forward with stop_words_list."""
        ids = np.ascontiguousarray(prompt_ids, dtype=np.int32).reshape(-1)
        rid = C.c_long(0)
        sw, sw_len = None, 0
        if stop_words:
            flat = [t for w in stop_words for t in w]
            sw_len = len(flat)
            arr = np.zeros((2, sw_len), dtype=np.int32)  # to_word_list_format (codefuse_example.py:26-53)
            arr[0] = flat
            arr[1] = -1
            arr[1, :len(stop_words)] = np.cumsum([len(w) for w in stop_words])
            sw = np.ascontiguousarray(arr)
        capi.check(capi.lib().ftcf_batcher_submit_ex(
            self._h, ids.ctypes.data_as(C.POINTER(C.c_int)), int(ids.size), int(max_new_tokens), int(top_k), C.c_float(top_p),
            C.c_float(temperature), C.c_float(repetition_penalty), C.c_ulonglong(int(seed)),
            sw.ctypes.data_as(C.POINTER(C.c_int)) if sw is not None else None, int(sw_len), C.byref(rid)))
        return int(rid.value)

    def submit_beam(self, prompt_ids, max_new_tokens, beam_width, beam_search_diversity_rate=0.0, len_penalty=0.0, temperature=1.0,
                    repetition_penalty=1.0, min_length=0, stop_words=None):
        """A beam-search request (GptNeoXOp.forward with beam_width > 1).  step() reports one event for it, (request_id, -1, True),
        when it has finished; beam_result(request_id) then returns what forward returns for one prompt."""
        ids = np.ascontiguousarray(prompt_ids, dtype=np.int32).reshape(-1)
        rid = C.c_long(0)
        sw, sw_len = None, 0
        if stop_words:
            flat = [t for w in stop_words for t in w]
            sw_len = len(flat)
            arr = np.zeros((2, sw_len), dtype=np.int32)  # to_word_list_format (codefuse_example.py:26-53)
            arr[0] = flat
            arr[1] = -1
            arr[1, :len(stop_words)] = np.cumsum([len(w) for w in stop_words])
            sw = np.ascontiguousarray(arr)
        capi.check(capi.lib().ftcf_batcher_submit_beam_ex(
            self._h, ids.ctypes.data_as(C.POINTER(C.c_int)), int(ids.size), int(max_new_tokens), int(beam_width),
            C.c_float(beam_search_diversity_rate), C.c_float(len_penalty), C.c_float(temperature), C.c_float(repetition_penalty),
            int(min_length), sw.ctypes.data_as(C.POINTER(C.c_int)) if sw is not None else None, int(sw_len), C.byref(rid)))
        return int(rid.value)

    def beam_result(self, request_id):
        """(output_ids [beam_width, prompt_len + max_new_tokens], sequence_lengths [beam_width], cum_log_probs [beam_width]) of a
        finished beam request, once; None while it is running or when the id is unknown."""
        k, t = C.c_int(0), C.c_int(0)
        capi.check(capi.lib().ftcf_batcher_beam_result(self._h, C.c_long(int(request_id)), None, None, None, 0, C.byref(k), C.byref(t)))
        if k.value == 0:
            return None
        ids = np.zeros((k.value, t.value), dtype=np.int32)
        lens = np.zeros(k.value, dtype=np.int32)
        cum = np.zeros(k.value, dtype=np.float32)
        capi.check(capi.lib().ftcf_batcher_beam_result(
            self._h, C.c_long(int(request_id)), ids.ctypes.data_as(C.POINTER(C.c_int)), lens.ctypes.data_as(C.POINTER(C.c_int)),
            cum.ctypes.data_as(C.POINTER(C.c_float)), int(ids.size), C.byref(k), C.byref(t)))
        return ids, lens, cum

    def set_token_callback(self, fn):
        """fn(request_id, token, finished) is called from inside step() for every token the moment it is on the host (between
        the chunks of a long admission as well); None unsets.  The events are still returned by step()."""
        if fn is None:
            self._cb = None
            capi.check(capi.lib().ftcf_batcher_set_token_callback(self._h, C.cast(None, capi.BATCHER_TOKEN_CALLBACK), None))
            return
        self._cb = capi.BATCHER_TOKEN_CALLBACK(lambda _user, rid, tok, fin: fn(int(rid), int(tok), bool(fin)))
        capi.check(capi.lib().ftcf_batcher_set_token_callback(self._h, self._cb, None))

    def step(self):
        """One scheduler iteration; returns [(request_id, token, finished), ...] in production order."""
        n = C.c_int(0)
        capi.check(capi.lib().ftcf_batcher_step(self._h, self._ids, self._tok, self._fin, 2 * self.max_batch, C.byref(n)))
        return [(int(self._ids[i]), int(self._tok[i]), bool(self._fin[i])) for i in range(n.value)]

    def status(self):
        w, r, f = C.c_int(0), C.c_int(0), C.c_int(0)
        capi.check(capi.lib().ftcf_batcher_status(self._h, C.byref(w), C.byref(r), C.byref(f)))
        return {"waiting": w.value, "running": r.value, "free_pages": f.value}

    def cancel(self, request_id):
        """Drops a waiting or running request; returns False when the id is unknown or already finished."""
        found = C.c_int(0)
        capi.check(capi.lib().ftcf_batcher_cancel(self._h, C.c_long(int(request_id)), C.byref(found)))
        return bool(found.value)

    def busy(self):
        s = self.status()
        return s["waiting"] > 0 or s["running"] > 0

    def run_all(self, max_iterations=1 << 20):
        """Drains the queue; returns {request_id: [tokens]}."""
        out = {}
        it = 0
        while self.busy():
            for rid, tok, _ in self.step():
                if tok >= 0:  # (a beam request's only event carries -1: its hypotheses come from beam_result)
                    out.setdefault(rid, []).append(tok)
            it += 1
            if it > max_iterations:
                raise RuntimeError("the batcher did not drain")
        return out
