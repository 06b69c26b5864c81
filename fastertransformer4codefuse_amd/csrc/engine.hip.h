// The engine behind GptNeoXOp (models/gptneox/GptNeoX.cc:386-1052): the structure every engine translation unit sees --
// engine.hip defines the request loop and the C ABI, batcher.hip the continuous-batching front end that borrows an engine.
#pragma once
#include "comm.hip.h"

// (engine.hip) GEMM / LM-head dispatch shared with the batcher
void gemm_dispatch(const f16* A, const void* W, const f16* scale, const f16* bias, int act, f16* C, int m, int n, int k, bool int8,
                   hipStream_t s, float* smallm_ws = nullptr, size_t smallm_partial = 0, int num_cu = 256, const int* d_step = nullptr,
                   unsigned* smallm_seq = nullptr, float* tiled_ws = nullptr);
void lm_head_dispatch(const f16* A, const f16* W, float* logits, int m, int n, int k, int ldc, hipStream_t s);

namespace {
// (DenseWeight / LayerWeights and the host-side layer units DecoderSelfAttentionLayer, GptContextAttentionLayer, FfnLayer,
// DynamicDecodeLayer: layers.hip.h)

struct DeviceBuffer {
    void*  ptr = nullptr;
    size_t cap = 0;
    void   reserve(size_t bytes)
    {
        if (bytes > cap) {
            if (ptr) {
                FTCF_HIP_CHECK(hipFree(ptr));
                ptr = nullptr;
                cap = 0;
            }
            FTCF_HIP_CHECK(hipMalloc(&ptr, bytes));
            cap = bytes;
        }
    }
    ~DeviceBuffer()
    {
        if (ptr) {
            (void)hipFree(ptr);
        }
    }
};

// carve helper over one arena
struct Carver {
    char*  base;
    size_t off = 0;
    explicit Carver(void* b): base((char*)b) {}
    template<typename T>
    T* take(size_t n)
    {
        off      = (off + 255) & ~(size_t)255;
        T* p     = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

enum { KIND_LN_GEMV = 0, KIND_SPLITK = 1, KIND_LM_HEAD = 2, KIND_FUSED = 3, KIND_PERSIST = 4, KIND_SMALLM = 5, KIND_COUNT = 6 };

}  // namespace

__global__ void k_transpose_gathered_logits(float* out, const float* in, int tp, int B, int vl);  // (defined below)

struct ftcf_gptneox {
    ftcf_gptneox_config       cfg{};
    int                       H = 0, nhl = 0, hl = 0, il = 0, L = 0, V = 0, vl = 0, dh = 0;
    bool                      int8 = false;
    bool                      fp32 = false;  // FTGptNeoX<float> (GptNeoXOp.cc:56-70): fp32 weights, activations and K/V; general path only
    // `stream` is the engine's own work stream (capturable, unlike the legacy null stream torch usually hands over);
    // it is ordered after `user_stream` at begin() and drained before forward()/finish() return
    hipStream_t               stream = nullptr, user_stream = nullptr;
    // second stream of the batched decode layer: [QKV -> MMHA -> out-proj] on `stream`, [FFN1 -> FFN2] here (fork / join by
    // events; under capture the branch becomes a parallel branch of the token's hipGraph)
    hipStream_t               side = nullptr, side2 = nullptr;  // (side2: the second micro-batch of decoder_overlapped)
    hipEvent_t                ev_fork = nullptr, ev_join = nullptr;
    // batched decode GEMMs: 1 = the attention branch and the FFN branch on two streams, 0 = one stream, the independent GEMMs
    // paired per launch.  Default: two streams at tensor_para_size 1 (the launches are bandwidth bound and fill each other's ramps:
    // 13B int8 bs = 16, 4.31 vs 4.66 ms per step), pairs on a tensor-parallel shard (launch-latency bound: TP 8 shard 1.79 vs 2.36
    // ms, TP 4 2.16 vs 2.62, TP 2 3.13 vs 3.28; `bench.py --fake-tp`).  FTCF_DECODE_BRANCHES overrides.
    int                       decode_branches = 1;
    hipEvent_t                ev_user = nullptr;
    hipEvent_t                tok_ev[2] = {nullptr, nullptr};  // per-token events of the pipelined token loop
    bool                      tp_pair_ar = true;  // batched decode under TP: attn | ffn all-reduced as one message, residual inside the next LN pass
    int                       k1_wpg = 2;  // waves per column group of the QKV launch (0: legacy 4-groups-per-block form)
    std::vector<LayerWeights> layers;
    const f16 *               wte = nullptr, *final_g = nullptr, *final_b = nullptr, *lm_head = nullptr;
    std::vector<void*>        owned;  // tiled fp16 copies (int8_mode == 0)
    void*                     bounce = nullptr;  // FTCF_FP16_RETILE_IN_PLACE: staging of one matrix during create()
    size_t                    bounce_bytes = 0;

    DeviceBuffer arena;
    // decode / state views (valid after plan())
    f16 *x = nullptr, *nrm = nullptr, *nrm2 = nullptr, *qkv = nullptr, *ctx = nullptr, *att = nullptr, *mid = nullptr, *ffn = nullptr;
    f16 *k_cache = nullptr, *v_cache = nullptr;
    f16 *px = nullptr, *pnrm = nullptr, *pnrm2 = nullptr, *pqkv = nullptr, *pctx = nullptr, *patt = nullptr, *pmid = nullptr,
        *pffn = nullptr;
    float *      logits = nullptr, *gather = nullptr, *mmha_ws = nullptr, *rot_table = nullptr;
    unsigned long long* chunk_ws = nullptr;
    int          k3_q = 1;  // chunks per column group of the K3 launch (0: legacy one-workgroup-per-group form)
    void*        samp_ws = nullptr;
    DecodeState* state = nullptr;
    uint8_t *    finished = nullptr, *masked = nullptr;
    int *        seq_len = nullptr, *pad_count = nullptr, *step_ids = nullptr, *d_top_k = nullptr,
        *d_min_length = nullptr;
    float *   cum = nullptr, *d_p_topk = nullptr, *d_p_topp = nullptr, *d_temp = nullptr, *d_rep = nullptr;
    uint64_t *draws = nullptr, *d_seed = nullptr;
    float*    smallm_ws = nullptr;  // split-K partials + tickets of the batched decode GEMM (5..16 rows)
    float*    tiled_ws  = nullptr;  // split-K partial tiles + tickets of the tiled GEMM at 17..768 rows (short prompt phases)
    size_t    smallm_partial = 0, smallm_region = 0;  // (all of the partial sums; one set of four GEMM regions)
    unsigned  smallm_seq = 0;       // launch counter: part of the granule tag of its in-launch reduction
    // beam search (beam_width K > 1; rows = batch * K everywhere above)
    int *  tiled_ids = nullptr, *tiled_len = nullptr, *parent_ids = nullptr, *cache_indir = nullptr;
    void*  beam_ws = nullptr;
    float *d_div = nullptr, *d_lenpen = nullptr;
    int*      h_flags = nullptr;  // pinned
    int       nsplit = 1;
    // persistent decode layers (kernels_persist.hip): on whenever the shape is eligible (FTCF_PERSIST=0: per-stage launches)
    int                 persist = 1, persist_tp = 1, persist_nb = 0, persist_cs1 = 12, persist_cs3 = 10, persist_own = 2;
    int                 decode_overlap_mode = 0;  // FTCF_DECODE_OVERLAP as read when the last request began: 0 off (default), 1 on, 2 "auto" (timed trial, RCCL ranks)
    int                 num_cu = 0;
    PersistPlan         pplan{};
    PersistLayer*       d_players = nullptr;  // device [L]
    char*               ps_tab = nullptr;     // the plan's run / tile tables, built once per request by a launch over no layers
    bool                ps_tab_ready = false;
    unsigned long long *ps_gq = nullptr, *ps_gm = nullptr, *ps_gc = nullptr, *ps_gx = nullptr, *ps_gp = nullptr,
                       *ps_ga = nullptr;
    size_t              ps_slab_n = 0;
    int*                ps_err = nullptr;
    int*                tp_scratch = nullptr;  // device int for the barrier all-reduce of the tensor-parallel windows
    long long*          ps_ts = nullptr;  // FTCF_PERSIST_TS=<file>: in-kernel stamps of the last token
    // persistent decode layers for 3..16 rows (kernels_rows.hip): on whenever the shape is eligible (FTCF_ROWS=0: general path)
    int                 rows = 1, rows_nb = 0, rows_min = -1;  // rows_min < 0: see plan()
    RowsPlan            rplan{};
    char*               rows_ws = nullptr;
    long long*          rows_ts = nullptr;
    std::string         ps_ts_file;

    // profiling
    bool               profiling = false;
    ftcf_forward_stats stats{};
    double             kind_ms[KIND_COUNT]{}, kind_bytes[KIND_COUNT]{};
    long               kind_n[KIND_COUNT]{};
    std::vector<std::tuple<hipEvent_t, hipEvent_t, int, double>> pending;
    std::vector<hipEvent_t>                                       event_pool;

    ~ftcf_gptneox()
    {
        for (void* p : owned) {
            (void)hipFree(p);
        }
        if (ses.graph_exec) {
            (void)hipGraphExecDestroy(ses.graph_exec);
        }
        if (ses.graph_exec_n) {
            (void)hipGraphExecDestroy(ses.graph_exec_n);
        }
        if (tp_scratch) {
            (void)hipFree(tp_scratch);
        }
        if (stream) {
            (void)hipStreamDestroy(stream);
            for (int c = 0; c < 2; c++) {
                if (ov_done[c]) {
                    (void)hipEventDestroy(ov_done[c]);
                    (void)hipEventDestroy(ov_red[c]);
                }
            }
            for (int c = 0; c < 2; c++) {
                if (dv_done[c]) {
                    (void)hipEventDestroy(dv_done[c]);
                    (void)hipEventDestroy(dv_red[c]);
                    (void)hipEventDestroy(dv_fork[c]);
                }
            }
            if (side2) {
                (void)hipStreamDestroy(side2);
            }
            if (side) {
                (void)hipStreamDestroy(side);
                (void)hipEventDestroy(ev_fork);
                (void)hipEventDestroy(ev_join);
            }
        }
        if (ev_user) {
            (void)hipEventDestroy(ev_user);
        }
        for (hipEvent_t e : tok_ev) {
            if (e) {
                (void)hipEventDestroy(e);
            }
        }
        if (h_flags) {
            (void)hipHostFree(h_flags);
        }
        for (auto e : event_pool) {
            (void)hipEventDestroy(e);
        }
    }

    hipEvent_t get_event()
    {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        FTCF_HIP_CHECK(hipEventCreate(&e));
        return e;
    }

    template<typename F>
    void timed(int kind, double bytes, F&& f, hipStream_t on = nullptr)
    {
        if (!profiling) {
            f();
            return;
        }
        hipEvent_t a = get_event(), b = get_event();
        FTCF_HIP_CHECK(hipEventRecord(a, on ? on : stream));
        f();
        FTCF_HIP_CHECK(hipEventRecord(b, on ? on : stream));
        pending.emplace_back(a, b, kind, bytes);
    }
    void drain_events()
    {
        for (auto& t : pending) {
            float ms = 0.f;
            FTCF_HIP_CHECK(hipEventSynchronize(std::get<1>(t)));
            FTCF_HIP_CHECK(hipEventElapsedTime(&ms, std::get<0>(t), std::get<1>(t)));
            kind_ms[std::get<2>(t)] += ms;
            kind_bytes[std::get<2>(t)] += std::get<3>(t);
            kind_n[std::get<2>(t)] += 1;
            event_pool.push_back(std::get<0>(t));
            event_pool.push_back(std::get<1>(t));
        }
        pending.clear();
    }

    // ---- arena planning: everything a request of shape (B, S, total) needs, carved once ----
    // B = rows of the request (batch * beam_width)
    void plan(int B, int S, int total, int K)
    {
        const int s_max = total;
        nsplit          = mmha_pick_nsplit(B, nhl, s_max);
        for (int pass = 0; pass < 2; pass++) {
            Carver c(pass == 0 ? nullptr : arena.ptr);
            const size_t es    = fp32 ? 2 : 1;  // fp32 engine: the same views hold floats
            const size_t cache = (size_t)L * B * nhl * s_max * dh * es;
            k_cache            = c.take<f16>(cache);
            v_cache            = c.take<f16>(cache);
            x                  = c.take<f16>((size_t)B * H * es);
            nrm                = c.take<f16>((size_t)B * H * es);
            nrm2               = c.take<f16>((size_t)B * H * es);
            qkv                = c.take<f16>((size_t)B * 3 * hl * es);
            ctx                = c.take<f16>((size_t)B * hl * es);
            att                = c.take<f16>((size_t)2 * B * H * es);  // [att | ffn]: one message for the layer's all-reduce
            ffn                = att + (size_t)B * H * es;
            mid                = c.take<f16>((size_t)B * il * es);
            logits             = c.take<float>((size_t)B * V);
            gather             = c.take<float>((size_t)B * V);
            mmha_ws            = c.take<float>(mmha_workspace_bytes(B, nhl, dh, nsplit) / 4);
            samp_ws            = c.take<char>(sampling_workspace_bytes(B, V));
            rot_table          = c.take<float>((size_t)B * 256);
            chunk_ws           = c.take<unsigned long long>(chunk_workspace_bytes(H, std::min(B, 4), 8) / 8);
            pplan = PersistPlan{};
            // With tensor parallelism the per-layer all-reduce happens INSIDE the persistent launch, through the ranks'
            // exchange windows (persist_device.hip.h ps_tp_exchange); where the windows are not available (peer mapping or
            // hand-shake failed, FTCF_TP_PERSIST=0) the per-stage launches + RCCL all-reduce stay in charge.
            const int  tpn      = cfg.tensor_para_size;
            const bool tp_local = tpn > 1 && cfg.comm && cfg.comm->local;
            // (L <= 255: the hand-off tags carry the layer in their low byte; tpn <= 8: the exchange-window table of the kernel)
            if (persist && !fp32 && K == 1 && B <= 2 && cfg.use_gptj_residual && L <= 255 && tpn <= PERSIST_MAX_TP
                && (tpn == 1 || (persist_tp && cfg.comm && cfg.comm->win_ok))) {
                // (a local group shares ONE device: every rank gets 1 / world of its compute units)
                const int nb = tp_local ? std::max(1, (persist_nb > 0 ? persist_nb : num_cu) / tpn) : persist_nb;
                pplan = persist_plan(B, H, hl, il, nhl, dh, s_max, int8, num_cu, nb, persist_cs1, persist_cs3, persist_own, L);
                const bool resident = !pplan.ok ? false
                                      : tp_local ? persist_group_resident(pplan, int8, B, dh, num_cu, tpn)
                                                 : persist_resident(pplan, int8, B, dh, num_cu, tpn);
                if (!resident) {
                    pplan = PersistPlan{};  // not every workgroup would be resident: the hand-offs could never complete
                }
            }
            if (pplan.ok) {
                ps_slab_n   = (size_t)B * 3 * hl / 2 + (size_t)B * il / 2 + (size_t)B * hl / 2 + (size_t)B * H / 2
                            + (size_t)pplan.gp_n + (size_t)B * nhl * pplan.nsplit * (dh + 2);
                ps_gq       = c.take<unsigned long long>(ps_slab_n + 8);
                ps_gm       = ps_gq ? ps_gq + (size_t)B * 3 * hl / 2 : nullptr;
                ps_gc       = ps_gq ? ps_gm + (size_t)B * il / 2 : nullptr;
                ps_gx       = ps_gq ? ps_gc + (size_t)B * hl / 2 : nullptr;
                ps_gp       = ps_gq ? ps_gx + (size_t)B * H / 2 : nullptr;
                ps_ga       = ps_gq ? ps_gp + (size_t)pplan.gp_n : nullptr;
                ps_err      = ps_gq ? reinterpret_cast<int*>(ps_gq + ps_slab_n) : nullptr;
                d_players   = c.take<PersistLayer>(L);
                ps_tab      = c.take<char>(persist_table_bytes(pplan) * pplan.NB);
                ps_tab_ready = false;
                ps_ts       = ps_ts_file.empty() ? nullptr : c.take<long long>((size_t)pplan.NB * L * 128);
            }
            // 3..16 rows (and what the one- / two-row kernel does not take): the rows kernel, one launch per token.  With tensor
            // parallelism it would be one launch per layer and the all-reduce of x' between them: a rank's shard of a layer is a few
            // microseconds of HBM time behind five in-kernel hand-offs, and the paired launches of the general path are faster
            // (one rank's shard of TP 8 at bs 16: 2.67 against 1.69 ms per step): tensor_para_size 1 only
            rplan = RowsPlan{};
            rows_ws = nullptr;
            rows_ts = nullptr;
            // One and two rows are the one- / two-row kernel's; where ITS plan declines a request (fp16 weights at two rows, more than
            // 1248 / 3072 tokens of context: its LDS) the rows kernel takes it -- 2.73 ms per token at one row against 3.2 on the
            // per-stage launches, 3.10 at two rows against 3.48 on the general path (13B int8, 1024-in / 512-out) -- but not when
            // that kernel has been switched off (FTCF_PERSIST=0, or after it gave up once): then the per-stage launches run, the path
            // of a tensor-parallel rank
            const int rmin = rows_min >= 1 ? rows_min : (persist ? 1 : 3);
            if (rows && cfg.tensor_para_size == 1 && !pplan.ok && !fp32 && K == 1 && B >= rmin && B <= 16
                && cfg.use_gptj_residual && L <= 255) {
                // (ranks that share ONE device -- a local group's threads, the two-process tests -- must be resident together)
                int nb = rows_nb > 0 ? rows_nb : persist_nb;
                if (tp_local) {
                    nb = std::max(1, (nb > 0 ? nb : num_cu) / tpn);
                }
                rplan = rows_plan(B, H, hl, il, nhl, dh, s_max, int8, num_cu, nb);
                if (rplan.ok && !rows_resident(rplan, int8, dh, num_cu, false)) {
                    rplan = RowsPlan{};
                }
                if (rplan.ok) {
                    rows_ws = c.take<char>(rows_workspace_bytes(rplan, B, H, hl, il, nhl, dh));
                    if (!d_players || !pplan.ok) {
                        d_players = c.take<PersistLayer>(L);
                    }
                    rows_ts = ps_ts_file.empty() ? nullptr : c.take<long long>((size_t)rplan.NB * L * 128);
                }
            }
            // (the four GEMMs of a layer may be in flight together: one region each)
            // (17..SMALLM_MAX_ROWS rows run the same kernel in chunks of 16 rows: sized for one chunk)
            // (a prompt phase of up to SMALLM_MAX_ROWS tokens in all is HBM bound like a decode step: it takes the same kernel)
            const int  bc         = 16;
            const long prefill_m  = S > 1 ? (long)(B / K) * S : 0;
            // (tensor parallel: up to 32 rows as two micro-batches of <= 16, each with its own regions -- decoder_overlapped)
            const bool decode_ws  = B > STAGE_MAX_ROWS && (B <= SMALLM_MAX_ROWS || (cfg.tensor_para_size > 1 && B <= 32));
            const bool prefill_ws = prefill_m > 4 && prefill_m <= SMALLM_MAX_ROWS;
            smallm_partial = gemm_smallm_workspace_bytes(bc, 3 * hl, H, int8) + gemm_smallm_workspace_bytes(bc, il, H, int8)
                             + gemm_smallm_workspace_bytes(bc, H, hl, int8) + gemm_smallm_workspace_bytes(bc, H, il, int8);
            smallm_region  = smallm_partial;
            smallm_partial *= cfg.tensor_para_size > 1 ? 2 : 1;
            smallm_ws = (!fp32 && (decode_ws || prefill_ws)) ? c.take<float>((smallm_partial + gemm_smallm_ticket_bytes()) / 4) : nullptr;
            const bool tiled_rows = prefill_m > 16 || (B > 16 && B <= gemm_tiled_splitk_max_m());
            tiled_ws              = (!fp32 && tiled_rows) ? c.take<float>(gemm_tiled_workspace_bytes() / 4) : nullptr;
            state              = c.take<DecodeState>(1);
            finished           = c.take<uint8_t>(B);
            masked             = c.take<uint8_t>((size_t)B * s_max);
            seq_len            = c.take<int>(B);
            pad_count          = c.take<int>(B);
            step_ids           = c.take<int>((size_t)total * B);
            d_top_k            = c.take<int>(B);
            d_min_length       = c.take<int>(B);
            cum                = c.take<float>(B);
            d_p_topk           = c.take<float>(B);
            d_p_topp           = c.take<float>(B);
            d_temp             = c.take<float>(B);
            d_rep              = c.take<float>(B);
            draws              = c.take<uint64_t>(B);
            d_seed             = c.take<uint64_t>(B);
            if (K > 1) {
                tiled_ids   = c.take<int>((size_t)B * S);
                tiled_len   = c.take<int>(B);
                parent_ids  = c.take<int>((size_t)total * B);
                cache_indir = c.take<int>((size_t)2 * B * s_max);
                beam_ws     = c.take<char>(beam_workspace_bytes(B / K, K));
                d_div       = c.take<float>(B);
                d_lenpen    = c.take<float>(B);
            }
            if (S > 1) {
                const size_t M = (size_t)(B / K) * S;  // beam search prefills one row per request
                px             = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pnrm           = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pnrm2          = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pqkv           = c.take<f16>(M * 3 * hl * (fp32 ? 2 : 1));
                pctx           = c.take<f16>(M * hl * (fp32 ? 2 : 1));
                patt           = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pmid           = c.take<f16>(M * il * (fp32 ? 2 : 1));
                pffn           = c.take<f16>(M * H * (fp32 ? 2 : 1));
            }
            if (pass == 0) {
                arena.reserve(c.off + 4096);
            }
        }
    }

    // ---- FfnLayer / attention projections over M rows (general path) ----
    // ---- host-side layer units (layers.hip.h), bound to this engine's GEMM dispatch -----------------------------------
    DecoderSelfAttentionLayer self_attention_layer;
    GptContextAttentionLayer  context_attention_layer;
    FfnLayer                  ffn_layer;
    DynamicDecodeLayer        dynamic_decode_layer;
    bool                      layers_bound = false;
    void bind_layers()
    {
        if (layers_bound) {
            return;
        }
        // (row-count dispatch of gemm(); `slot` is unused here: the burst kernel's four workspace regions are only in flight
        // together on the two-stream branch form, which names its regions itself)
        GemmFn g = [this](const f16* A, const DenseWeight& w, const f16* bias, int act, f16* C, int m, int n, int k, hipStream_t s,
                          int) { gemm(A, w, bias, act, C, m, n, k, s); };
        self_attention_layer    = DecoderSelfAttentionLayer{g, H, hl};
        context_attention_layer = GptContextAttentionLayer{g, H, hl, nhl, dh, cfg.rotary_embedding_dim};
        ffn_layer               = FfnLayer{g, H, il};
        layers_bound            = true;
    }

    void gemm(const f16* A, const DenseWeight& w, const f16* bias, int act, f16* C, int m, int n, int k, hipStream_t on = nullptr)
    {
        hipStream_t stream = on ? on : this->stream;  // (the layer units pass the stream they were given)
        // 5..SMALLM_MAX_ROWS rows (batched decode steps off the branch form, short prompt phases): the burst kernel, 16 rows
        // per launch.  13B int8 prefill, ms: 17 tokens 12.5 -> 6.6, 33..48: 16.3 -> 9.6 (above that the tiled GEMM is as fast).
        if (smallm_ws && m > 4 && m <= SMALLM_MAX_ROWS && gemm_smallm_workspace_bytes(16, n, k, int8) <= smallm_partial) {
            for (int r0 = 0; r0 < m; r0 += 16) {
                launch_gemm_smallm(A + (size_t)r0 * k, w.kernel, w.scale, bias, act, C + (size_t)r0 * n, smallm_ws, smallm_partial,
                                   std::min(16, m - r0), n, k, int8, num_cu, stream, &state->step, &smallm_seq);
            }
            return;
        }
        if (m > 16) {
            launch_gemm_tiled(A, w.kernel, w.scale, bias, act, C, m, n, k, int8, stream, tiled_ws);
            return;
        }
        gemm_dispatch(A, w.kernel, w.scale, bias, act, C, m, n, k, int8, stream, nullptr, smallm_partial, num_cu, &state->step,
                      &smallm_seq);
    }

    void allreduce(f16* buf, size_t count, hipStream_t on = nullptr)
    {
        if (cfg.tensor_para_size > 1) {
            Range r("ftcf.allreduce");
            hipStream_t st = on ? on : stream;
            FTCF_CHECK_ARG(cfg.comm && (cfg.comm->comm || cfg.comm->local || cfg.comm->hx), "tensor_para_size > 1 needs a communicator");
            if (window_allreduce(cfg.comm, buf, count, st)) {
                stats.window_allreduces++;
                return;  // through the peer-mapped windows: no RCCL call (prompt-phase messages; k_window_allreduce)
            }
            if (cfg.comm->local) {
                local_allreduce(cfg.comm, buf, count, true, st);
                return;
            }
            if (cfg.comm->hx) {
                hx_allreduce(cfg.comm, buf, count, true, st);
                return;
            }
            FTCF_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclFloat16, ncclSum, cfg.comm->comm, st));
        }
    }

    // vocabulary-split LM head (GptNeoX.cc:888-925): rank r has written its [B, V/TP] slice of `gath` ([TP][B][V/TP] fp32);
    // all-gather it and transpose into out [B, V]
    void allgather_logits(float* gath, float* out, int B, hipStream_t st)
    {
        const int tp = cfg.tensor_para_size;
        float*    mine = gath + (size_t)cfg.tensor_para_rank * B * vl;
        if (cfg.comm->local) {
            local_allgather(cfg.comm, gath, (size_t)B * vl, false, st);
        }
        else if (cfg.comm->hx) {
            hx_allgather_device(cfg.comm, gath, (size_t)B * vl, 4, st);
        }
        else {
            FTCF_NCCL_CHECK(ncclAllGather(mine, gath, (size_t)B * vl, ncclFloat32, cfg.comm->comm, st));
        }
        hipLaunchKernelGGL(k_transpose_gathered_logits, dim3(256), dim3(256), 0, st, out, gath, tp, B, vl);
    }

    // ---------------------------------------------------------------------------------------------------------------
    // fp32 instantiation (kernels_fp32.hip): the arena views (x, nrm, qkv, ..., the caches, the prefill buffers) hold floats
    // ---------------------------------------------------------------------------------------------------------------
    static float*       F(f16* p) { return reinterpret_cast<float*>(p); }
    static const float* F(const f16* p) { return reinterpret_cast<const float*>(p); }
    static const float* F(const void* p) { return reinterpret_cast<const float*>(p); }
    void allreduce32(float* buf, size_t count)
    {
        if (cfg.tensor_para_size > 1) {
            Range r("ftcf.allreduce");
            FTCF_CHECK_ARG(cfg.comm && (cfg.comm->comm || cfg.comm->local || cfg.comm->hx), "tensor_para_size > 1 needs a communicator");
            if (cfg.comm->local) {
                local_allreduce(cfg.comm, buf, count, false, stream);
                return;
            }
            if (cfg.comm->hx) {
                hx_allreduce(cfg.comm, buf, count, false, stream);
                return;
            }
            FTCF_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclFloat32, ncclSum, cfg.comm->comm, stream));
        }
    }
    // one layer's GEMMs / residual on M rows, shared by the context phase and the decode step
    // (GptNeoXContextDecoder.cc:283-507, GptNeoXDecoder.cc:245-384 with T = float)
    template<typename Attn>
    void layer32(const LayerWeights& w, float* X, float* N1, float* Q, float* C, float* A, float* MID, float* FF, int M,
                 bool first_or_last_inplace_variant, Attn&& attention)
    {
        launch_layernorm(X, w.ln1_g, w.ln1_b, N1, M, H, 1e-5f, false, stream);
        launch32_gemm(N1, F(w.qkv.kernel), nullptr, 0, Q, M, 3 * hl, H, stream);
        attention();
        launch32_gemm(C, F(w.attn_out.kernel), nullptr, 0, A, M, H, hl, stream);
        if (!cfg.use_gptj_residual) {
            // sequential residual (GptNeoXDecoder.cc:313-331,362-367): h = attn + bias + x ; x' = ffn(LN2(h)) + bias + h
            allreduce32(A, (size_t)M * H);
            launch32_add_bias_residual(A, X, A, F(w.attn_out.bias), M, H, stream);
            launch_layernorm(A, w.ln2_g, w.ln2_b, N1, M, H, 1e-5f, false, stream);
            launch32_gemm(N1, F(w.ffn1.kernel), F(w.ffn1.bias), 1, MID, M, il, H, stream);
            launch32_gemm(MID, F(w.ffn2.kernel), nullptr, 0, FF, M, H, il, stream);
            allreduce32(FF, (size_t)M * H);
            launch32_add_bias_residual(X, FF, A, F(w.ffn2.bias), M, H, stream);
            return;
        }
        launch_layernorm(X, w.ln2_g, w.ln2_b, N1, M, H, 1e-5f, false, stream);
        launch32_gemm(N1, F(w.ffn1.kernel), F(w.ffn1.bias), 1, MID, M, il, H, stream);
        launch32_gemm(MID, F(w.ffn2.kernel), nullptr, 0, FF, M, H, il, stream);
        launch_add_bias_attn_ffn_residual(X, FF, A, X, w.ffn2.bias, M, H, cfg.tensor_para_size,
                                          first_or_last_inplace_variant ? 0 : 1, false, stream);
        allreduce32(X, (size_t)M * H);
    }
    void context_decoder32(int B, int S, const int* input_lengths, int s_max, int tile)
    {
        Range        r("ftcf.GptNeoXContextDecoder");
        const int    M       = B * S;
        const size_t cache_l = (size_t)B * tile * nhl * s_max * dh;
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            // (the context decoder's layer_input == layer_output for every layer with padding removal: the fp32-sum variant)
            layer32(w, F(px), F(pnrm), F(pqkv), F(pctx), F(patt), F(pmid), F(pffn), M, false, [&] {
                launch32_context_attention(F(pqkv), F(w.qkv.bias), input_lengths, F(k_cache) + l * cache_l,
                                           F(v_cache) + l * cache_l, B, S, nhl, dh, cfg.rotary_embedding_dim, s_max, F(pctx),
                                           stream, tile);
            });
        }
    }
    void decoder32(int B, int s_max)
    {
        Range        r("ftcf.GptNeoXDecoder");
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        stats.decode_path    = 2;
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            // layer_input/output alias for 0 < l < L-1 in the reference (GptNeoXDecoder.cc:249-250) -> residual form
            const bool outer = !(l > 0 && l < L - 1);
            layer32(w, F(x), F(nrm), F(qkv), F(ctx), F(att), F(mid), F(ffn), B, outer, [&] {
                Mmha32Params mp{};
                mp.qkv = F(qkv);
                mp.qkv_bias = F(w.qkv.bias);
                mp.k_cache = F(k_cache) + l * cache_l;
                mp.v_cache = F(v_cache) + l * cache_l;
                mp.seq_len = seq_len;
                mp.pad_count = pad_count;
                mp.masked_tokens = masked;
                mp.finished = finished;
                mp.d_step = &state->step;
                mp.B = B;
                mp.nh = nhl;
                mp.dh = dh;
                mp.rot = cfg.rotary_embedding_dim;
                mp.s_max = s_max;
                mp.ctx = F(ctx);
                if (ses.K > 1) {
                    mp.cache_indir   = cache_indir;
                    mp.beam_width    = ses.K;
                    mp.max_input_len = ses.S;
                    mp.indir_plane   = (size_t)B * s_max;
                }
                launch32_mmha(mp, stream);
            });
        }
    }

    // GptNeoXContextDecoder::forward (GptNeoXContextDecoder.cc:283-507), parallel residual only
    // B prompt rows; their K/V go to cache rows b * tile of a cache with B * tile rows (beam search: tile = beam_width)
    // Prompt phase under tensor parallelism with the per-layer all-reduce OVERLAPPED (GptNeoXContextDecoder.cc:462-465 calls
    // ftNcclAllReduceSum on the compute stream, nothing runs under it).  The prompt is cut into two micro-batches -- whole
    // sequences when there are several (their attention is independent), the first and the second half of the tokens of a
    // single sequence (every GEMM / LayerNorm / residual is row wise, and the second half's attention reads the first half's
    // K/V from the cache, where the first half's attention call has put them).  A layer runs micro-batch 0, then 1, on the
    // engine stream; each micro-batch's all-reduce goes to the side stream behind an event, and the NEXT layer's work on that
    // micro-batch waits for it: the reduction of one half runs under the GEMMs of the other.  Same arithmetic per row as
    // context_decoder (the all-reduce sums the same values): results are bit-identical to the un-overlapped path.
    // Chunked prompt phase of ONE sequence (the continuous-batching front end, section 4e of DESIGN.md): the prompt's tokens pass
    // through all layers `prefill_chunk` at a time -- every GEMM / LayerNorm / residual is row wise, a chunk's attention reads
    // the earlier chunks' K/V from the cache -- and `prefill_hook` runs between two chunks (the batcher enqueues one decode step
    // of its running slots there: an admission delays them by one chunk, not by the whole prompt).  The row-wise arithmetic is
    // that of context_decoder; the split-K form of the GEMMs depends on the row count, so results agree to fp16 rounding of
    // the GEMM outputs, not bit for bit.
    int                   prefill_chunk = 0;
    std::function<void()> prefill_hook;
    bool context_decoder_chunked(int S, const int* input_lengths, int s_max, int tile = 1)
    {
        if (!prefill_hook || prefill_chunk <= 0 || S <= prefill_chunk || fp32 || cfg.tensor_para_size != 1 || !cfg.use_gptj_residual
            || !residual_dual_ln_supported(H)) {
            return false;
        }
        Range r("ftcf.GptNeoXContextDecoder.chunked");
        bind_layers();
        const size_t cache_l = (size_t)tile * nhl * s_max * dh;  // (tile = beam_width: the sequence's K/V go to cache row 0 of tile rows)
        for (int s0 = 0; s0 < S; s0 += prefill_chunk) {
            const int s1 = std::min(S, s0 + prefill_chunk), m = s1 - s0;
            f16*      X  = px + (size_t)s0 * H;
            for (int l = 0; l < L; l++) {
                const LayerWeights& w = layers[l];
                launch_residual_dual_ln(X, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, pnrm + (size_t)s0 * H,
                                        pnrm2 + (size_t)s0 * H, m, H, 1e-5f, stream);
                context_attention_layer.forward(pnrm, pqkv, pctx, patt, w, input_lengths, k_cache + l * cache_l, v_cache + l * cache_l, 1,
                                                S, s_max, tile, stream, s0, s1);
                ffn_layer.forward(pnrm2 + (size_t)s0 * H, pmid + (size_t)s0 * il, pffn + (size_t)s0 * H, w, m, stream);
                launch_add_bias_attn_ffn_residual(X, pffn + (size_t)s0 * H, patt + (size_t)s0 * H, X, w.ffn2.bias, m, H, 1, 1, true, stream);
            }
            if (s1 < S) {
                prefill_hook();
            }
        }
        return true;
    }

    hipEvent_t ov_done[2] = {nullptr, nullptr}, ov_red[2] = {nullptr, nullptr};
    int        ov_trial = 0;             // auto mode: 0 the next eligible prompt phase runs plain, 1 overlapped, 2 decided
    float      ov_ms[2] = {0.f, 0.f};    // ... what the two trials took (the slowest rank's time)
    bool       ov_ran = false, ov_eligible = false;  // this request: ran overlapped / counts as a trial
    bool context_decoder_overlapped(int B, int S, const int* input_lengths, int s_max)
    {
        // OPT-IN (FTCF_PREFILL_OVERLAP=1; read per request, the tests flip it): on the one GPU the builder has, one rank's shard of
        // the 1024-token prompt phase takes 19.1 -> 27.6 ms (TP 2) / 10.3 -> 18.3 ms (TP 8) in two micro-batches -- GEMMs of 512
        // rows fill the chip worse than GEMMs of 1024 (profiles/r03_faketp_prefill.txt) -- and what the overlap hides (40 all-reduces
        // of 10 MiB over xGMI) cannot be measured without the peers.  Whoever has the node should measure both.
        // Round 4: DECIDED FROM DATA on the node it runs on.  FTCF_PREFILL_OVERLAP = 0 / 1 forces it; unset or "auto" (the
        // default for ranks joined by RCCL, i.e. a real multi-GPU job): the first eligible prompt phase of at least 512 tokens
        // runs plain and is timed, the second one overlapped, every rank learns the slower rank's times (comm_max) and the
        // engine keeps the faster form; ftcf_forward_stats says what ran and what the two trials took.
        const char* ev  = getenv("FTCF_PREFILL_OVERLAP");
        const bool  aut = (!ev || !strcmp(ev, "auto")) && cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->comm && !cfg.comm->local
                         && !cfg.comm->hx && cfg.comm->world > 1;
        const int   env = (ev && strcmp(ev, "auto")) ? atoi(ev) : 0;
        ov_ran      = false;
        ov_eligible = false;
        if ((!env && !aut) || cfg.tensor_para_size == 1 || !cfg.use_gptj_residual || !residual_dual_ln_supported(H) || !side) {
            return false;
        }
        if (aut) {
            ov_eligible = (long)B * S >= 512 && (B >= 2 || (S / 2) / 64 * 64 >= 64);
            const bool want = ov_eligible && (ov_trial == 1 || (ov_trial == 2 && ov_ms[1] < ov_ms[0]));
            if (!want) {
                return false;
            }
        }
        // micro-batches: rows [r0[c], r1[c]) of the [B * S] row space; sequences [b0, b1) x tokens [s0, s1)
        int b0[2] = {0, 0}, b1[2] = {B, B}, s0[2] = {0, 0}, s1[2] = {S, S};
        if (B >= 2) {
            b1[0] = b0[1] = B / 2;
        }
        else {
            const int cut = (S / 2) / 64 * 64;
            if (cut < 64) {
                return false;  // too short to be worth two micro-batches
            }
            s1[0] = s0[1] = cut;
        }
        Range r("ftcf.GptNeoXContextDecoder.overlapped");
        ov_ran = true;
        bind_layers();
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        for (int c = 0; c < 2; c++) {
            if (!ov_done[c]) {
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&ov_done[c], hipEventDisableTiming));
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&ov_red[c], hipEventDisableTiming));
            }
        }
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            for (int c = 0; c < 2; c++) {
                const size_t row0 = (size_t)b0[c] * S + s0[c];
                const int    m    = (B >= 2) ? (b1[c] - b0[c]) * S : s1[c] - s0[c];
                if (l > 0) {
                    FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ov_red[c], 0));  // this micro-batch's x has been reduced
                }
                f16* X = px + row0 * H;
                launch_residual_dual_ln(X, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, pnrm + row0 * H,
                                        pnrm2 + row0 * H, m, H, 1e-5f, stream);
                if (B >= 2) {  // whole sequences [b0, b1): the attention layer on their rows of every buffer
                    const size_t cb = (size_t)b0[c] * nhl * s_max * dh;
                    context_attention_layer.forward(pnrm + row0 * H, pqkv + row0 * 3 * hl, pctx + row0 * hl, patt + row0 * H, w,
                                                    input_lengths + b0[c], k_cache + l * cache_l + cb, v_cache + l * cache_l + cb,
                                                    b1[c] - b0[c], S, s_max, 1, stream);
                }
                else {  // tokens [s0, s1) of the one sequence: the earlier tokens' K/V are in the cache
                    context_attention_layer.forward(pnrm, pqkv, pctx, patt, w, input_lengths, k_cache + l * cache_l, v_cache + l * cache_l,
                                                    1, S, s_max, 1, stream, s0[c], s1[c]);
                }
                ffn_layer.forward(pnrm2 + row0 * H, pmid + row0 * il, pffn + row0 * H, w, m, stream);
                launch_add_bias_attn_ffn_residual(X, pffn + row0 * H, patt + row0 * H, X, w.ffn2.bias, m, H, cfg.tensor_para_size,
                                                  1, true, stream);
                FTCF_HIP_CHECK(hipEventRecord(ov_done[c], stream));
                FTCF_HIP_CHECK(hipStreamWaitEvent(side, ov_done[c], 0));
                allreduce(X, (size_t)m * H, side);
                FTCF_HIP_CHECK(hipEventRecord(ov_red[c], side));
            }
        }
        for (int c = 0; c < 2; c++) {
            FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ov_red[c], 0));
        }
        return true;
    }

    void context_decoder(int B, int S, const int* input_lengths, int s_max, int tile)
    {
        if (B == 1 && context_decoder_chunked(S, input_lengths, s_max, tile)) {  // (tile > 1: a beam request's prompt)
            return;
        }
        if (tile == 1 && context_decoder_overlapped(B, S, input_lengths, s_max)) {
            return;
        }
        Range r("ftcf.GptNeoXContextDecoder");
        bind_layers();
        const int    M       = B * S;
        const size_t cache_l = (size_t)B * tile * nhl * s_max * dh;
        // parallel-residual layers: both LayerNorms in one pass, fused with the previous layer's residual when no collective
        // sits in between (as in the batched decode path)
        const bool dual = cfg.use_gptj_residual && residual_dual_ln_supported(H);
        const bool tp1  = cfg.tensor_para_size == 1;
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            if (!dual) {
                launch_layernorm(px, w.ln1_g, w.ln1_b, pnrm, M, H, 1e-5f, true, stream);
            }
            else if (l == 0 || !tp1) {
                launch_residual_dual_ln(px, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, pnrm, pnrm2,
                                        M, H, 1e-5f, stream);
            }
            context_attention_layer.forward(pnrm, pqkv, pctx, patt, w, input_lengths, k_cache + l * cache_l, v_cache + l * cache_l, B, S,
                                            s_max, tile, stream);
            if (!cfg.use_gptj_residual) {
                // sequential residual (GptNeoXContextDecoder.cc:401-418,463-470): the TensorParallel layers reduce their
                // own outputs; h = attn + bias + x ; x' = ffn(LN2(h)) + bias + h
                allreduce(patt, (size_t)M * H);
                launch_add_bias_residual(patt, px, patt, w.attn_out.bias, M, H, stream);
                launch_layernorm(patt, w.ln2_g, w.ln2_b, pnrm, M, H, 1e-5f, true, stream);
                ffn_layer.forward(pnrm, pmid, pffn, w, M, stream);
                allreduce(pffn, (size_t)M * H);
                launch_add_bias_residual(px, pffn, patt, w.ffn2.bias, M, H, stream);
                continue;
            }
            if (!dual) {
                launch_layernorm(px, w.ln2_g, w.ln2_b, pnrm, M, H, 1e-5f, true, stream);
            }
            ffn_layer.forward(dual ? pnrm2 : pnrm, pmid, pffn, w, M, stream);
            // layer_input == layer_output for every layer with padding removal -> fp32-sum variant (:311-322,:445-461)
            if (dual && tp1) {
                const LayerWeights* nx = l + 1 < L ? &layers[l + 1] : nullptr;
                launch_residual_dual_ln(px, pffn, patt, w.ffn2.bias, 1, 1, nx ? nx->ln1_g : nullptr, nx ? nx->ln1_b : nullptr,
                                        nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, pnrm, pnrm2, M, H, 1e-5f, stream);
            }
            else {
                launch_add_bias_attn_ffn_residual(px, pffn, patt, px, w.ffn2.bias, M, H, cfg.tensor_para_size, 1, true,
                                                  stream);
            }
            allreduce(px, (size_t)M * H);
        }
    }

    PersistParams persist_params(int B, int s_max)
    {
        PersistParams pp{};
        pp.layers = d_players;
        pp.L = L;
        pp.x_in = x;
        pp.x_out = x;
        pp.gq = ps_gq;
        pp.gm = ps_gm;
        pp.gc = ps_gc;
        pp.gx = ps_gx;
        pp.gp = ps_gp;
        pp.ga = ps_ga;
        pp.err = ps_err;
        pp.H = H;
        pp.Hl = hl;
        pp.Il = il;
        pp.nh = nhl;
        pp.dh = dh;
        pp.rot = cfg.rotary_embedding_dim;
        pp.s_max = s_max;
        pp.B = B;
        pp.tp = cfg.tensor_para_size;
        pp.tp_rank = cfg.tensor_para_rank;
        for (int r = 0; r < PERSIST_MAX_TP; r++) {
            pp.xw[r] = (cfg.tensor_para_size > 1 && cfg.comm && r < (int)cfg.comm->win.size())
                           ? static_cast<unsigned long long*>(cfg.comm->win[r]) : nullptr;
        }
        if (cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->world == 1 && !cfg.comm->win.empty()) {
            // timing aid (bench.py --fake-tp N: ONE rank of a TP = N job without its peers): the rank plays every peer --
            // "slot [rank] of rank r's window" is made to land in slot [r] of its own -- so that the kernel's exchange
            // completes (the sums are meaningless, the work and the waits of a rank are all there)
            for (int r = 0; r < cfg.tensor_para_size && r < PERSIST_MAX_TP; r++) {
                pp.xw[r] = static_cast<unsigned long long*>(cfg.comm->win[0])
                           + (ptrdiff_t)(r - cfg.tensor_para_rank) * ((ptrdiff_t)B * H / 2);
            }
        }
        pp.plan = pplan;
        pp.d_step = &state->step;
        pp.d_stop = &state->all_finished;
        pp.seq_len = seq_len;
        pp.pad_count = pad_count;
        pp.masked_tokens = masked;
        pp.finished = finished;
        pp.rot_table = rot_table;
        pp.eps = 1e-5f;
        pp.ts = ps_ts;
        pp.tab = ps_tab;
        pp.tab_mode = (ps_tab && ps_tab_ready) ? 2 : 0;
        return pp;
    }

    RowsParams rows_params(int B, int s_max)
    {
        RowsParams rp{};
        rp.layers = d_players;
        rp.L = L;
        rp.l_begin = 0;
        rp.l_end = L;
        rp.x_in = x;
        rp.x_out = x;
        rp.M = B;
        rp.H = H;
        rp.Hl = hl;
        rp.Il = il;
        rp.nh = nhl;
        rp.dh = dh;
        rp.rot = cfg.rotary_embedding_dim;
        rp.s_max = s_max;
        rp.tp = cfg.tensor_para_size;
        rp.plan = rplan;
        rows_carve(rp, rows_ws);
        rp.d_step = &state->step;
        rp.d_stop = &state->all_finished;
        rp.seq_len = seq_len;
        rp.pad_count = pad_count;
        rp.input_lengths = ses.sp.input_lengths;
        rp.max_input_len = ses.S;
        rp.finished = finished;
        rp.rot_table = rot_table;
        rp.eps = 1e-5f;
        rp.ts = rows_ts;
        return rp;
    }

    // GptNeoXDecoder::forward (GptNeoXDecoder.cc:245-384)
    void decoder(int B, int s_max)
    {
        Range r("ftcf.GptNeoXDecoder");
        bind_layers();
        const double wbytes  = int8 ? 1.0 : 2.0;
        // (beam search reads K/V through the cache indirection, sequential-residual layers have their own order: general path)
        const bool staged = B <= STAGE_MAX_ROWS && ses.K == 1 && cfg.use_gptj_residual && (dh == 64 || dh == 128);
        stats.decode_path = pplan.ok ? 1 : (staged ? 0 : 2);
        stats.persist_layout = pplan.ok ? pplan.own : 0;
        if (!ses.path_logged) {  // once per request
            ses.path_logged = true;
            FT_LOG_DEBUG(cfg.device, "decoder of this request: %s (rows %d, context %d%s)",
                         pplan.ok ? "persistent layers"
                                  : (rplan.ok ? "persistent layers for up to 16 rows"
                                              : (staged ? "per-stage launches" : "general path (batched GEMMs)")),
                         B, s_max, pplan.ok ? (pplan.uk == 16 ? ", 512 keys per KV split" : ", 256 keys per KV split") : "");
        }
        if (pplan.ok) {
            // all stages of every layer inside persistent launches (kernels_persist.hip); one launch per token when
            // there is no collective between the layers
            PersistParams pp = persist_params(B, s_max);
            // algorithmic bytes of a layer: its four weight matrices + the K/V rows of the current length
            const double layer_bytes = wbytes * ((double)H * 3 * hl + (double)H * il + (double)hl * H + (double)il * H)
                                       + 4.0 * ses.next_step * hl * B;
            pp.l_begin = 0;
            pp.l_end   = L;
            if (cfg.tensor_para_size > 1 && cfg.comm->local) {
                // local group: ONE launch runs every rank (workgroups [r * NB, (r + 1) * NB) = rank r), issued by rank 0
                // between two thread barriers; the other ranks' streams are idle meanwhile
                LocalGroup& g = *cfg.comm->local;
                FTCF_HIP_CHECK(hipStreamSynchronize(stream));  // this rank's inputs (x, rotary table, state) are complete
                g.item[cfg.tensor_para_rank] = &pp;
                g.barrier();
                if (cfg.tensor_para_rank == 0) {
                    PersistGroupParams gp{};
                    for (int r = 0; r < g.world; r++) {
                        gp.p[r] = *static_cast<const PersistParams*>(g.item[r]);
                    }
                    gp.world = g.world;
                    gp.nb    = pplan.NB;
                    launch_decode_persistent_group(gp, int8, stream);
                    FTCF_HIP_CHECK(hipStreamSynchronize(stream));
                }
                g.barrier();
                return;
            }
            timed(KIND_PERSIST, layer_bytes * L, [&] { launch_decode_persistent(pp, int8, stream); });
            return;
        }
        if (rplan.ok) {
            // every stage of every layer inside ONE launch for up to 16 rows (kernels_rows.hip); with tensor parallelism one
            // launch per layer and the all-reduce of its output between them (GptNeoXDecoder.cc:357-359)
            RowsParams   rp          = rows_params(B, s_max);
            const double layer_bytes = wbytes * ((double)H * 3 * hl + (double)H * il + (double)hl * H + (double)il * H)
                                       + 4.0 * ses.next_step * hl * B;
            stats.decode_path = 3;
            if (cfg.tensor_para_size == 1) {
                timed(KIND_PERSIST, layer_bytes * L, [&] { launch_decode_rows(rp, int8, stream); });
                return;
            }
            for (int l = 0; l < L; l++) {
                rp.l_begin = l;
                rp.l_end   = l + 1;
                timed(KIND_PERSIST, layer_bytes, [&] { launch_decode_rows(rp, int8, stream); });
                allreduce(x, (size_t)B * H);
            }
            return;
        }
        if (!staged && decoder_overlapped(B, s_max)) {
            return;
        }
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            // layer_input/output alias for 0 < l < L-1 in the reference (:249-250) -> which residual form it runs
            const int inplace = (l > 0 && l < L - 1) ? 1 : 0;
            if (staged) {
                // Per-stage launches over row groups of <= 4 rows (the GEMV kernels' register budget).  STAGE_MAX_ROWS > 4
                // would replay every stage per group; measured no faster than the batched GEMM path (the m = 4 forms
                // of these kernels stream at half the m = 1 rate), so larger batches take the small-m GEMM below.
                const int ngrp = (B + 3) / 4;
                for (int stage = 0; stage < 3; stage++) {
                    for (int rg = 0; rg < ngrp; rg++) {
                        const int r0 = rg * 4, M = std::min(4, B - r0);
                        stage_launch(stage, l, w, inplace, B, s_max, r0, M, l + rg * L, ngrp == 1);
                    }
                }
            }
            else {
                // general path: both LayerNorms of the layer come from one pass over x, fused with the previous layer's
                // residual when there is no collective in between
                MmhaParams mp = mmha_params(l, w, B, s_max, 0, B, l);
                if (ses.K > 1) {
                    mp.cache_indir   = cache_indir;
                    mp.beam_width    = ses.K;
                    mp.max_input_len = ses.S;
                    mp.indir_plane   = (size_t)B * s_max;
                }
                if (!cfg.use_gptj_residual) {
                    // sequential residual (GptNeoXDecoder.cc:313-331,362-367): h = attn + bias + x ; x' = ffn(LN2(h)) + bias + h
                    launch_layernorm(x, w.ln1_g, w.ln1_b, nrm, B, H, 1e-5f, true, stream);
                    self_attention_layer.forward(nrm, qkv, ctx, att, w, mp, B, stream);
                    allreduce(att, (size_t)B * H);
                    launch_add_bias_residual(att, x, att, w.attn_out.bias, B, H, stream);
                    launch_layernorm(att, w.ln2_g, w.ln2_b, nrm, B, H, 1e-5f, true, stream);
                    ffn_layer.forward(nrm, mid, ffn, w, B, stream);
                    allreduce(ffn, (size_t)B * H);
                    launch_add_bias_residual(x, ffn, att, w.ffn2.bias, B, H, stream);
                    continue;
                }
                const bool dual = residual_dual_ln_supported(H);
                const bool tp1  = cfg.tensor_para_size == 1;
                if (!dual) {
                    launch_layernorm(x, w.ln1_g, w.ln1_b, nrm, B, H, 1e-5f, true, stream);
                    launch_layernorm(x, w.ln2_g, w.ln2_b, nrm2, B, H, 1e-5f, true, stream);
                }
                else if (l == 0 || (!tp1 && !tp_pair_ar)) {
                    launch_residual_dual_ln(x, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, nrm,
                                            nrm2, B, H, 1e-5f, stream);
                }
                if (B <= SMALLM_MAX_ROWS && smallm_ws && decode_branches && side) {
                    // The attention branch [QKV -> MMHA -> out-proj] (78.6 + K/V + 26.2 MB at 13B int8) and the FFN branch
                    // [FFN1 -> FFN2] (2 x 104.9 MB) of a parallel-residual layer are independent: two streams.  Every one
                    // of these launches is a short burst -- the whole matrix requested at once, gone in ~30 us -- whose
                    // ramp-up and drain leave the HBM idle; the other branch's launch fills those gaps.
                    const int    bc = std::min(B, 16);
                    const size_t o_qkv = 0, o_f1 = o_qkv + gemm_smallm_workspace_bytes(bc, 3 * hl, H, int8),
                                 o_out = o_f1 + gemm_smallm_workspace_bytes(bc, il, H, int8),
                                 o_f2  = o_out + gemm_smallm_workspace_bytes(bc, H, hl, int8);
                    auto one = [&](const SmallmDesc& d0, size_t off, hipStream_t s) {
                        // (one launch that keeps the weights in registers and passes the rows 16 at a time through the x tile
                        // was measured: 256 VGPRs, one workgroup per CU -- 8.2 / 8.3 / 13.0 ms at 24 / 32 / 64 rows, i.e.
                        // slower than re-reading the weights per 16 rows except at 64)
                        for (int r0 = 0; r0 < B; r0 += 16) {  // 16 rows per launch (launches of one GEMM are in stream order)
                            SmallmDesc d = d0;
                            d.A          = d0.A + (size_t)r0 * d0.k;
                            d.C          = d0.C + (size_t)r0 * d0.n;
                            const int M  = std::min(16, B - r0);
                            timed(KIND_SMALLM, wbytes * (double)d.n * d.k, [&] {
                                launch_gemm_smallm_group(&d, 1, smallm_ws, smallm_partial, M, int8, s, &state->step, &smallm_seq, off);
                            }, s);
                        }
                    };
                    // the attention layer on the engine stream, the FFN layer on the side stream: the same two layer units, their
                    // GEMMs bound to the burst kernel with one workspace region per GEMM of the layer
                    const size_t offs[4] = {o_qkv, o_f1, o_out, o_f2};
                    GemmFn burst = [&](const f16* A, const DenseWeight& dw, const f16* bias, int act, f16* C, int, int n, int k,
                                       hipStream_t s, int slot) { one(SmallmDesc{A, dw.kernel, dw.scale, bias, act, C, n, k}, offs[slot], s); };
                    const DecoderSelfAttentionLayer attn_b{burst, H, hl};
                    const FfnLayer                  ffn_b{burst, H, il};
                    FTCF_HIP_CHECK(hipEventRecord(ev_fork, stream));
                    FTCF_HIP_CHECK(hipStreamWaitEvent(side, ev_fork, 0));
                    attn_b.forward(nrm, qkv, ctx, att, w, mp, B, stream);
                    ffn_b.forward(nrm2, mid, ffn, w, B, side);
                    FTCF_HIP_CHECK(hipEventRecord(ev_join, side));
                    FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0));
                }
                else if (B <= 16 && smallm_ws) {
                    // independent GEMMs share a launch (a dependent launch costs ~8 us of dispatch latency, most of a layer
                    // at tensor-parallel shard sizes): [QKV, FFN1] -> MMHA -> [out-proj, FFN2]
                    const SmallmDesc p1[2] = {{nrm, w.qkv.kernel, w.qkv.scale, nullptr, 0, qkv, 3 * hl, H},
                                              {nrm2, w.ffn1.kernel, w.ffn1.scale, w.ffn1.bias, 1, mid, il, H}};
                    timed(KIND_SMALLM, wbytes * H * (3.0 * hl + il),
                          [&] { launch_gemm_smallm_group(p1, 2, smallm_ws, smallm_partial, B, int8, stream, &state->step, &smallm_seq); });
                    launch_mmha(mp, stream);
                    const SmallmDesc p3[2] = {{ctx, w.attn_out.kernel, w.attn_out.scale, nullptr, 0, att, H, hl},
                                              {mid, w.ffn2.kernel, w.ffn2.scale, nullptr, 0, ffn, H, il}};
                    timed(KIND_SMALLM, wbytes * H * ((double)hl + il),
                          [&] { launch_gemm_smallm_group(p3, 2, smallm_ws, smallm_partial, B, int8, stream, &state->step, &smallm_seq); });
                }
                else {
                    self_attention_layer.forward(nrm, qkv, ctx, att, w, mp, B, stream);
                    ffn_layer.forward(nrm2, mid, ffn, w, B, stream);
                }
                if (dual && !tp1 && tp_pair_ar) {
                    // Tensor parallel: the reference closes the layer with x / TP + attn + ffn + bias and ONE all-reduce of the sum
                    // (GptNeoXDecoder.cc:342-359, add_residual_kernels.cu:116-152), then the next layer's LayerNorms: three launches
                    // on a path that is bound by the latency of dependent launches.  Here attn | ffn (adjacent in the arena) travel
                    // as one message of twice the size and the residual -- x + attn + ffn + TP x (bias / TP) in fp32, rounded once
                    // -- runs inside the next layer's LayerNorm pass, as at TP = 1: two launches (FTCF_TP_PAIR_AR=0: the former).
                    allreduce(att, (size_t)2 * B * H);
                    const LayerWeights* nx = l + 1 < L ? &layers[l + 1] : nullptr;
                    launch_residual_dual_ln(x, ffn, att, w.ffn2.bias, 1, 1, nx ? nx->ln1_g : nullptr, nx ? nx->ln1_b : nullptr,
                                            nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, nrm, nrm2, B, H, 1e-5f, stream,
                                            cfg.tensor_para_size);
                    continue;
                }
                if (dual && tp1) {
                    const LayerWeights* nx = l + 1 < L ? &layers[l + 1] : nullptr;
                    launch_residual_dual_ln(x, ffn, att, w.ffn2.bias, 1, inplace, nx ? nx->ln1_g : nullptr,
                                            nx ? nx->ln1_b : nullptr, nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr,
                                            nrm, nrm2, B, H, 1e-5f, stream);
                }
                else {
                    launch_add_bias_attn_ffn_residual(x, ffn, att, x, w.ffn2.bias, B, H, cfg.tensor_para_size, inplace,
                                                      true, stream);
                }
            }
            allreduce(x, (size_t)B * H);
        }
    }

    // Batched decode under tensor parallelism with the layer's all-reduce OFF the compute stream (GptNeoXDecoder.cc:342-359 runs it
    // in line).  A parallel-residual layer has ONE reduction, of x' = x + attn + ffn, and everything of the next layer depends on
    // it: the only independent work is another row's.  So the batch is cut in two micro-batches of <= 16 rows that walk the
    // layers on TWO compute streams, independent of each other from the token's embedding to its final LayerNorm; micro-batch c
    // hands x' to the comm stream by event and waits for the reduced x' by event before its next layer, so c's reduction runs
    // under (1 - c)'s GEMMs / attention -- and, this path being bound by the latency of dependent launches rather than by bytes
    // (a TP 8 shard's layer at 16 rows: six launches, 43 us, 39 MB), one micro-batch's launch gaps are filled by the other's
    // kernels.  Each micro-batch reads the layer's weight shard once: for 17..32 rows that is what the chunked GEMMs do anyway, for
    // 4..16 rows it doubles the weight bytes.  All reductions sit on ONE stream in the order c = 0, 1, 0, 1, ...: the window
    // all-reduce's flag words and RCCL see a serial sequence, identical on every rank.  Row-wise arithmetic is that of the loop
    // below (the burst GEMM's K slices do not depend on the row count; the micro-batches' GEMMs have their own split-K regions), so
    // the tokens are bit-identical to the un-overlapped path (tests/test_gpu_tp_overlap.py, test_gpu_tp_process.py).
    // FTCF_DECODE_OVERLAP = 1 switches it on; "auto" (ranks joined by RCCL): the first eligible request's token loop runs plain,
    // the second one overlapped, every rank keeps the slowest rank's ms per step (comm_max in finish()) and the engine stays with
    // the faster form -- what an all-reduce over xGMI hides can only be measured on the node.  The default is OFF since round 6:
    // the eager three-stream form hides 5-15 % of a modelled reduction and loses at 16 rows (profiles/r05_decode_overlap_model.txt),
    // the auto trial compares two different requests, and none of it has run on a multi-GPU node.
    hipEvent_t dv_done[2] = {nullptr, nullptr}, dv_red[2] = {nullptr, nullptr}, dv_fork[2] = {nullptr, nullptr};
    int        dv_trial = 0;
    float      dv_ms[2] = {0.f, 0.f};  // ms per decode step of the two trials (the slowest rank's)
    bool       dv_ran = false, dv_eligible = false;
    bool decode_overlap_shape(int B) const
    {
        return cfg.tensor_para_size > 1 && !fp32 && ses.K == 1 && cfg.use_gptj_residual && residual_dual_ln_supported(H) && side
               && smallm_ws && smallm_partial >= 2 * smallm_region && B >= 4 && B <= 2 * 16 && !pplan.ok && !rplan.ok;
    }
    void decode_overlap_streams()
    {
        if (!side2) {
            FTCF_HIP_CHECK(hipStreamCreateWithFlags(&side2, hipStreamNonBlocking));
        }
        for (int c = 0; c < 2; c++) {
            if (!dv_done[c]) {
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&dv_done[c], hipEventDisableTiming));
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&dv_red[c], hipEventDisableTiming));
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&dv_fork[c], hipEventDisableTiming));
            }
        }
    }
    // (no side effects: step() asks it too -- the three-stream form is enqueued eagerly, not captured into the token's hipGraph:
    // a capture of it aborted inside the runtime on the one-rank RCCL communicator of `bench.py --fake-tp`, round 5)
    bool decode_overlap_wanted(int B, bool* auto_trial = nullptr) const
    {
        // (decode_overlap_mode: FTCF_DECODE_OVERLAP read when a request begins -- 0, the default, 1, or 2 = "auto")
        const bool  aut = decode_overlap_mode == 2 && cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->comm && !cfg.comm->local
                         && !cfg.comm->hx && cfg.comm->world > 1;
        const int   env = decode_overlap_mode == 1 ? 1 : 0;
        if ((!env && !aut) || !decode_overlap_shape(B)) {
            return false;
        }
        if (auto_trial) {
            *auto_trial = aut;
        }
        return !aut || dv_trial == 1 || (dv_trial == 2 && dv_ms[1] < dv_ms[0]);
    }
    bool decoder_overlapped(int B, int s_max)
    {
        bool aut = false;
        const bool want = decode_overlap_wanted(B, &aut);
        dv_eligible = aut;
        if (!want) {
            return false;
        }
        Range r("ftcf.GptNeoXDecoder.overlapped");
        dv_ran = true;
        decode_overlap_streams();
        const double      wbytes = int8 ? 1.0 : 2.0;
        const int         r0[2] = {0, (B + 1) / 2}, r1[2] = {(B + 1) / 2, B};
        const hipStream_t cs[2] = {stream, side2};
        // what is on the engine stream so far (embedding rows, the step's state) happens-before both micro-batches
        FTCF_HIP_CHECK(hipEventRecord(dv_fork[0], stream));
        FTCF_HIP_CHECK(hipStreamWaitEvent(side2, dv_fork[0], 0));
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            const int inplace = (l > 0 && l < L - 1) ? 1 : 0;
            for (int c = 0; c < 2; c++) {
                const int         M  = r1[c] - r0[c];
                const size_t      o  = (size_t)r0[c], wo = (size_t)c * smallm_region;
                const hipStream_t st = cs[c];
                f16*              xr = x + o * H;
                if (l > 0 && !tp_pair_ar) {
                    FTCF_HIP_CHECK(hipStreamWaitEvent(st, dv_red[c], 0));  // this micro-batch's x has been reduced
                }
                // (a micro-batch's attn | ffn rows are adjacent: [2 r0 H, 2 r0 H + M H) and the M H behind it)
                f16* const attc = att + 2 * o * H;
                f16* const ffnc = attc + (size_t)M * H;
                if (l == 0 || !tp_pair_ar) {
                    launch_residual_dual_ln(xr, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, nrm + o * H,
                                            nrm2 + o * H, M, H, 1e-5f, st);
                }
                MmhaParams mp = mmha_params(l, w, B, s_max, r0[c], M, l);
                // [QKV, FFN1] -> MMHA -> [out-proj, FFN2], independent GEMMs in one launch (as the loop below)
                const SmallmDesc p1[2] = {{nrm + o * H, w.qkv.kernel, w.qkv.scale, nullptr, 0, qkv + o * 3 * hl, 3 * hl, H},
                                          {nrm2 + o * H, w.ffn1.kernel, w.ffn1.scale, w.ffn1.bias, 1, mid + o * il, il, H}};
                timed(KIND_SMALLM, wbytes * H * (3.0 * hl + il), [&] {
                    launch_gemm_smallm_group(p1, 2, smallm_ws, smallm_partial, M, int8, st, &state->step, &smallm_seq, wo);
                }, st);
                launch_mmha(mp, st);
                const SmallmDesc p3[2] = {{ctx + o * hl, w.attn_out.kernel, w.attn_out.scale, nullptr, 0, attc, H, hl},
                                          {mid + o * il, w.ffn2.kernel, w.ffn2.scale, nullptr, 0, ffnc, H, il}};
                timed(KIND_SMALLM, wbytes * H * ((double)hl + il), [&] {
                    launch_gemm_smallm_group(p3, 2, smallm_ws, smallm_partial, M, int8, st, &state->step, &smallm_seq, wo);
                }, st);
                if (!tp_pair_ar) {
                    launch_add_bias_attn_ffn_residual(xr, ffnc, attc, xr, w.ffn2.bias, M, H, cfg.tensor_para_size, inplace, true, st);
                }
                FTCF_HIP_CHECK(hipEventRecord(dv_done[c], st));
                FTCF_HIP_CHECK(hipStreamWaitEvent(side, dv_done[c], 0));
                allreduce(tp_pair_ar ? attc : xr, (size_t)(tp_pair_ar ? 2 : 1) * M * H, side);
                FTCF_HIP_CHECK(hipEventRecord(dv_red[c], side));
                if (tp_pair_ar) {
                    // the layer's residual inside the next layer's LayerNorm pass, behind the reduction (general loop below)
                    const LayerWeights* nx = l + 1 < L ? &layers[l + 1] : nullptr;
                    FTCF_HIP_CHECK(hipStreamWaitEvent(st, dv_red[c], 0));
                    launch_residual_dual_ln(xr, ffnc, attc, w.ffn2.bias, 1, 1, nx ? nx->ln1_g : nullptr, nx ? nx->ln1_b : nullptr,
                                            nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, nrm + o * H, nrm2 + o * H, M, H, 1e-5f,
                                            st, cfg.tensor_para_size);
                }
            }
        }
        // join: both micro-batches' last reductions, and the second compute stream itself (its last wait is for dv_red[1] of layer
        // L - 2: nothing of it is left running behind dv_red[1] of the last layer, but a capture wants every fork joined)
        FTCF_HIP_CHECK(hipEventRecord(dv_fork[1], side2));
        FTCF_HIP_CHECK(hipStreamWaitEvent(stream, dv_fork[1], 0));
        for (int c = 0; c < 2; c++) {
            FTCF_HIP_CHECK(hipStreamWaitEvent(stream, dv_red[c], 0));
        }
        return true;
    }

    // Rows up to which the per-stage GEMV launches run (when the persistent kernel is not eligible).  Measured at 13B int8,
    // TP = 1, ms per step, per-stage vs general path (burst GEMMs): B = 1: 3.34 vs 3.61 (and 1.28 vs 1.57 on a TP = 8
    // shard); B = 2: 4.13 vs 3.85; B = 3: 4.64 vs 3.65; B = 4: 5.78 vs 3.73 -- the m = 2..4 forms of the GEMV kernels stream
    // at a fraction of the m = 1 rate.  FTCF_STAGE_MAX_ROWS (<= 4) overrides, the tests use it to keep those forms covered.
    int STAGE_MAX_ROWS = 1;
    // Rows up to which the batched decode GEMMs (and short prompt phases) run the burst kernel, 16 rows per launch (the weights
    // are then read ceil(B / 16) times).  Above 16 rows the alternative is the tiled GEMM in its split-K form (64-row tiles cut
    // along K, four k-steps of weights and activations in flight): 13B int8, ms per decode step, chunked burst vs split-K tiled:
    // bs = 24: 6.8 vs 6.4; 32: 7.4 vs 6.5; 48: 10.2 vs 7.5; 64: 13.6 vs 8.7 -- and prompt phases of 17 / 33 / 64 tokens
    // 7.0 / 10.2 / 13.8 vs 6.0 / 6.2 / 6.5 ms.  (Round 2's tiled GEMM without the split: 13.8 ms at bs = 24, 19.4 at 64.)
    // FTCF_SMALLM_MAX_ROWS overrides (<= 256; the chunked form stays covered by the tests through it).
    int SMALLM_MAX_ROWS = 16;

    // decoder attention of rows [r0, r0 + M) of the batch (KV cache [L][B][nh][s_max][dh]); `salt` makes the granule
    // tags of every launch of a token distinct
    MmhaParams mmha_params(int l, const LayerWeights& w, int B, int s_max, int r0, int M, int salt)
    {
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        const size_t row_kv  = (size_t)nhl * s_max * dh;
        MmhaParams   mp{};
        mp.qkv = qkv + (size_t)r0 * 3 * hl;
        mp.qkv_bias = w.qkv.bias;
        mp.k_cache = k_cache + l * cache_l + r0 * row_kv;
        mp.v_cache = v_cache + l * cache_l + r0 * row_kv;
        mp.seq_len = seq_len + r0;
        mp.pad_count = pad_count + r0;
        mp.masked_tokens = masked + (size_t)r0 * s_max;
        mp.finished = finished + r0;
        mp.d_step = &state->step;
        mp.rot_table = rot_table + (size_t)r0 * (cfg.rotary_embedding_dim / 2) * 2;
        mp.B = M;
        mp.nh = nhl;
        mp.dh = dh;
        mp.rot = cfg.rotary_embedding_dim;
        mp.s_max = s_max;
        mp.ctx = ctx + (size_t)r0 * hl;
        mp.gran = (unsigned long long*)mmha_ws + (size_t)r0 * nhl * nsplit * (dh + 2);
        mp.layer = salt;
        mp.nsplit = nsplit;
        return mp;
    }

    // One of the three launches of a layer for rows [r0, r0 + M), M <= 4:
    //   0: K1  LN1 -> QKV                                  (78.6 MB/TP int8)
    //   1: K2  MMHA  ||  LN2 -> FFN1 + bias + gelu         (attention hidden under 104.9 MB/TP of streaming)
    //   2: K3  [out-proj U FFN2] -> residual               (131 MB/TP)
    void stage_launch(int stage, int l, const LayerWeights& w, int inplace, int B, int s_max, int r0, int M, int salt,
                      bool time_it)
    {
        const double wbytes = int8 ? 1.0 : 2.0;
        f16*         xr     = x + (size_t)r0 * H;
        auto run = [&](int kind, double bytes, auto&& f) {
            if (time_it) {
                timed(kind, bytes, f);
            }
            else {
                f();
            }
        };
        if (stage == 0) {
            LnGemvParams a{};
            a.x = xr;
            a.gamma0 = w.ln1_g;
            a.beta0 = w.ln1_b;
            a.W0 = w.qkv.kernel;
            a.scale0 = w.qkv.scale;
            a.out0 = qkv + (size_t)r0 * 3 * hl;
            a.K = H;
            a.NT0 = 3 * hl / 16;
            a.NT1 = 0;
            a.blocks0 = (a.NT0 + 3) / 4;
            a.blocks1 = 0;
            a.eps = 1e-5f;
            run(KIND_LN_GEMV, wbytes * H * (3.0 * hl), [&] {
                if (k1_wpg > 0) {
                    launch_ln_gemv_group(a, int8, M, k1_wpg, stream);
                }
                else {
                    launch_ln_gemv(a, int8, M, stream);
                }
            });
        }
        else if (stage == 1) {
            MmhaParams   mp = mmha_params(l, w, B, s_max, r0, M, salt);
            LnGemvParams f{};
            f.x = xr;
            f.gamma1 = w.ln2_g;
            f.beta1 = w.ln2_b;
            f.W1 = w.ffn1.kernel;
            f.scale1 = w.ffn1.scale;
            f.bias1 = w.ffn1.bias;
            f.out1 = mid + (size_t)r0 * il;
            f.K = H;
            f.NT0 = 0;
            f.NT1 = il / 16;
            f.blocks0 = 0;
            f.blocks1 = f.NT1 / 2;  // two column groups per workgroup (NT1 is even: local inter is a multiple of 64)
            f.eps = 1e-5f;
            run(KIND_FUSED, wbytes * H * (double)il, [&] { launch_mmha_ln_gemv(mp, f, int8, M, stream); });
        }
        else {
            const int tk = int8 ? TILE_K_I8 : TILE_K_F16;
            if (k3_q > 0) {
                ChunkParams c{};
                c.x_a = ctx + (size_t)r0 * hl;
                c.x_b = mid + (size_t)r0 * il;
                c.W_a = w.attn_out.kernel;
                c.W_b = w.ffn2.kernel;
                c.scale_a = w.attn_out.scale;
                c.scale_b = w.ffn2.scale;
                c.bias = w.ffn2.bias;
                c.x_in = xr;
                c.out = xr;
                c.N = H;
                c.KT_a = hl / tk;
                c.KT_b = il / tk;
                c.Q = k3_q;
                c.T = (c.KT_a + c.KT_b + c.Q - 1) / c.Q;
                c.tp = cfg.tensor_para_size;
                c.inplace_variant = inplace;
                c.gran = chunk_ws;
                c.d_step = &state->step;
                c.salt = salt;
                run(KIND_SPLITK, wbytes * H * ((double)hl + il), [&] { launch_gemv_chunked(c, int8, M, stream); });
            }
            else {
                SplitKParams c{};
                c.x_a = ctx + (size_t)r0 * hl;
                c.x_b = mid + (size_t)r0 * il;
                c.W_a = w.attn_out.kernel;
                c.W_b = w.ffn2.kernel;
                c.scale_a = w.attn_out.scale;
                c.scale_b = w.ffn2.scale;
                c.bias = w.ffn2.bias;
                c.x_in = xr;
                c.out = xr;
                c.N = H;
                c.KT_a = hl / tk;
                c.KT_b = il / tk;
                c.tp = cfg.tensor_para_size;
                c.inplace_variant = inplace;
                plan_splitk(c, int8, M, 10);
                run(KIND_SPLITK, wbytes * H * ((double)hl + il),
                    [&] { launch_gemv_splitk(c, int8, M, EPI_RESIDUAL, stream); });
            }
        }
    }

    // ---- request session: forward() == begin() + step(all) + finish() ----
    struct Session {
        bool              active = false;
        ftcf_forward_args a{};
        SamplingParams    sp{};
        BeamParams        bp{};
        int               B = 0, S = 0, total = 0, s_max = 0;  // B = rows (batch * beam_width)
        int               K = 1, batch = 0;
        int               next_step = 0;  // host mirror of state->step
        int               steps = 0;
        bool              all_finished = false;
        hipEvent_t        e0 = nullptr, e1 = nullptr;
        hipGraphExec_t    graph_exec = nullptr;
        bool              path_logged = false;  // the decoder of this request has been named in the log (FT_LOG_LEVEL=DEBUG)
        hipGraphExec_t    graph_exec_n = nullptr;  // graph_tokens consecutive tokens in one graph (persistent path, no callback)
    } ses;
    bool use_graph = true;
    // drops whatever an unfinished request left behind: the captured graph holds the OLD arena pointers, shapes and sampling
    // flags, and plan() may free that arena -- replaying it for the next request would corrupt memory silently
    void abandon_session()
    {
        if (!ses.active && !ses.graph_exec) {
            return;
        }
        (void)hipStreamSynchronize(stream);
        if (ses.graph_exec) {
            (void)hipGraphExecDestroy(ses.graph_exec);
            ses.graph_exec = nullptr;
        }
        if (ses.graph_exec_n) {
            (void)hipGraphExecDestroy(ses.graph_exec_n);
            ses.graph_exec_n = nullptr;
        }
        if (ses.e0) {
            event_pool.push_back(ses.e0);
            ses.e0 = nullptr;
        }
        if (ses.e1) {
            event_pool.push_back(ses.e1);
            ses.e1 = nullptr;
        }
        drain_events();
        ses.active = false;
    }
    void begin(const ftcf_forward_args& a);
    void enqueue_step(bool with_decoder);
    int  step(int max_steps);
    void finish();
    bool persist_failed = false;  // the persistent kernel gave up on a hand-off during the last request
    bool winar_failed = false;    // ... or the exchange-window all-reduce of the prompt phase did
    int  persist_fail_once = 0;
    void forward(const ftcf_forward_args& a)
    {
        begin(a);
        step(a.output_len);
        try {
            finish();
        }
        catch (const Error&) {
            if (winar_failed && !persist_failed) {
                winar_failed = false;
                FT_LOG_WARNING(cfg.device, "exchange-window all-reduce gave up: replaying the request with the communicator's own "
                                           "all-reduce (it stays there)");
                begin(a);
                step(a.output_len);
                finish();
                return;
            }
            if (!persist_failed) {
                throw;
            }
            // The persistent kernel's hand-offs need every workgroup resident; the plan checks that, but compute units can
            // still be taken away (another process, a masked CU) after the check.  Its spins are bounded and report through
            // a sticky error word instead of hanging the GPU; the request is then replayed from the start on the
            // per-stage / general path (tensor parallel: every rank takes this branch -- finish() agrees on the error
            // word across the ranks) and the engine stays off the persistent path.
            persist_failed = false;  // (finish() has switched off the kernel that gave up: `persist` or `rows`)
            FT_LOG_WARNING(cfg.device, "persistent decode kernel gave up on a hand-off: replaying the request on the per-stage "
                                       "path (this engine stays there)");
            begin(a);
            step(a.output_len);
            finish();
        }
    }
};
