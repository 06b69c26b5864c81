// Host-side engine + C ABI (include/ftcf.h) of the MI355X GPT-NeoX / CodeFuse decoder.
//
// Host classes mirror the reference's layer split (layers are thin: they own no scratch of their own and never
// reMalloc per call -- one arena is planned per request shape):
//   GptNeoX               <- models/gptneox/GptNeoX.cc:386-1052 (generation loop, LM head, dynamic decode, outputs)
//   GptNeoXContextDecoder <- models/gptneox/GptNeoXContextDecoder.cc:223-512 (prefill)
//   GptNeoXDecoder        <- models/gptneox/GptNeoXDecoder.cc:197-389 (one token through L layers)
//   DecoderSelfAttentionLayer / FfnLayer / DynamicDecodeLayer are the launch helpers used by those.
#include "engine.hip.h"

thread_local std::string g_last_error;

extern "C" const char* ftcf_last_error(void)
{
    return g_last_error.c_str();
}
extern "C" int ftcf_version(void)
{
    return FTCF_VERSION;
}
extern "C" int ftcf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}


// ---------------------------------------------------------------------------------------------------------------
// host quantiser entry points (libth_common counterpart)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int ftcf_symmetric_quantize_int8(const void* weight, ftcf_dtype dtype, size_t E, size_t K, size_t N,
                                            int8_t* out_q, void* out_scale)
{
    return guarded([&] { host_symmetric_quantize_int8(weight, (int)dtype, E, K, N, out_q, out_scale); });
}
extern "C" int ftcf_int8_rowmajor_to_tiled(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] { host_int8_rowmajor_to_tiled(q, K, N, out); });
}
extern "C" int ftcf_int8_tiled_to_rowmajor(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] { host_int8_tiled_to_rowmajor(q, K, N, out); });
}
extern "C" int ftcf_int8_cuda_sm80_to_rowmajor(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(q && out && K % 64 == 0 && N % 2 == 0, "SM80 int8 layout needs K % 64 == 0 and N % 2 == 0");
        host_int8_cuda_sm80_to_rowmajor(q, K, N, out);
    });
}
extern "C" int ftcf_int8_rowmajor_to_cuda_sm80(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(q && out && K % 64 == 0 && N % 2 == 0, "SM80 int8 layout needs K % 64 == 0 and N % 2 == 0");
        host_int8_rowmajor_to_cuda_sm80(q, K, N, out);
    });
}
extern "C" int ftcf_fp16_rowmajor_to_tiled(const void* w, size_t K, size_t N, void* out, void* stream)
{
    return guarded([&] {
        require_device();
        launch_fp16_rowmajor_to_tiled((const f16*)w, K, N, (f16*)out, (hipStream_t)stream);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// kernel-level entry points
// ---------------------------------------------------------------------------------------------------------------
static float* abi_gemm_workspace(int m, hipStream_t s);
void gemm_dispatch(const f16* A, const void* W, const f16* scale, const f16* bias, int act, f16* C, int m, int n, int k, bool int8,
                   hipStream_t s, float* smallm_ws, size_t smallm_partial, int num_cu, const int* d_step, unsigned* smallm_seq,
                   float* tiled_ws)
{
    if (m <= 4) {
        SplitKParams p{};
        p.x_a = A;
        p.W_a = W;
        p.scale_a = scale;
        p.bias = bias;
        p.out = C;
        p.N = n;
        p.KT_a = k / (int8 ? TILE_K_I8 : TILE_K_F16);
        p.KT_b = 0;
        p.act = act;
        p.tp = 1;
        plan_splitk(p, int8, m, 4);
        launch_gemv_splitk(p, int8, m, EPI_PLAIN, s);
    }
    else if (m <= 16) {
        launch_gemm_smallm(A, W, scale, bias, act, C, smallm_ws, smallm_partial, m, n, k, int8, num_cu, s, d_step,
                           smallm_seq);
    }
    else {
        launch_gemm_tiled(A, W, scale, bias, act, C, m, n, k, int8, s, tiled_ws);
    }
}

// the stand-alone GEMM entry points carry no workspace argument (the reference's runner takes one from its caller:
// fpA_intB_gemm.h gemm(..., workspace_ptr, workspace_bytes)): one split-K workspace per (device, stream) that has called with
// 17..768 rows, kept for the life of the process
static float* abi_gemm_workspace(int m, hipStream_t s)
{
    if (m <= 16 || m > gemm_tiled_splitk_max_m()) {
        return nullptr;  // (only the split-K form of 17..768 rows uses it)
    }
    static std::mutex                                    mu;
    static std::map<std::pair<int, hipStream_t>, float*> ws;
    int                                                  dev = 0;
    FTCF_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    float*&                     p = ws[{dev, s}];
    if (!p) {
        FTCF_HIP_CHECK(hipMalloc(&p, gemm_tiled_workspace_bytes()));
        // only the tickets need zeros (the partial tiles are written before they are read), and on the CALLER's stream: a
        // null-stream memset is not ordered with a launch on a non-blocking stream
        FTCF_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(p) + gemm_tiled_workspace_bytes() - gemm_tiled_ticket_bytes(), 0,
                                      gemm_tiled_ticket_bytes(), s));
    }
    return p;
}

extern "C" int ftcf_fpA_intB_gemm(const void* A, const int8_t* B, const void* scales, const void* bias, ftcf_act act,
                                  void* C, int m, int n, int k, void* stream)
{
    return guarded([&] {
        require_device();
        FTCF_CHECK_ARG(k % 64 == 0 && n % 16 == 0 && m >= 1, "fpA_intB GEMM needs k % 64 == 0, n % 16 == 0");
        gemm_dispatch((const f16*)A, B, (const f16*)scales, (const f16*)bias, (int)act, (f16*)C, m, n, k, true,
                      (hipStream_t)stream, nullptr, 0, 256, nullptr, nullptr, abi_gemm_workspace(m, (hipStream_t)stream));
    });
}
extern "C" int ftcf_fp16_gemm(const void* A, const void* W, const void* bias, ftcf_act act, void* C, int m, int n,
                              int k, void* stream)
{
    return guarded([&] {
        require_device();
        FTCF_CHECK_ARG(k % 64 == 0 && n % 16 == 0 && m >= 1, "fp16 GEMM needs k % 64 == 0, n % 16 == 0");
        gemm_dispatch((const f16*)A, W, nullptr, (const f16*)bias, (int)act, (f16*)C, m, n, k, false,
                      (hipStream_t)stream, nullptr, 0, 256, nullptr, nullptr, abi_gemm_workspace(m, (hipStream_t)stream));
    });
}
void lm_head_dispatch(const f16* A, const f16* W, float* logits, int m, int n, int k, int ldc, hipStream_t s)
{
    if (m <= 4) {
        launch_lm_head(A, W, logits, m, n, k, ldc, s);
    }
    else {
        launch_gemm_nk_f32out(A, W, logits, m, n, k, ldc, s);
    }
}
extern "C" int ftcf_lm_head(const void* A, const void* W, float* logits, int m, int n, int k, int ldc, void* stream)
{
    return guarded([&] {
        require_device();
        lm_head_dispatch((const f16*)A, (const f16*)W, logits, m, n, k, ldc, (hipStream_t)stream);
    });
}
extern "C" int ftcf_layernorm(const void* x, const void* gamma, const void* beta, void* out, int m, int n, float eps,
                              ftcf_dtype dtype, void* stream)
{
    return guarded([&] {
        require_device();
        launch_layernorm(x, gamma, beta, out, m, n, eps, dtype == FTCF_FP16, (hipStream_t)stream);
    });
}
extern "C" int ftcf_add_bias_attn_ffn_residual(void* out, const void* ffn, const void* attn, const void* in,
                                               const void* bias, int m, int n, int tp, int inplace_variant,
                                               ftcf_dtype dtype, void* stream)
{
    return guarded([&] {
        require_device();
        launch_add_bias_attn_ffn_residual(out, ffn, attn, in, bias, m, n, tp, inplace_variant, dtype == FTCF_FP16,
                                          (hipStream_t)stream);
    });
}
extern "C" size_t ftcf_masked_multihead_attention_workspace(int B, int nh, int dh, int s_max)
{
    return mmha_workspace_bytes(B, nh, dh, mmha_pick_nsplit(B, nh, s_max));
}
extern "C" int ftcf_masked_multihead_attention(const void* qkv, const void* qkv_bias, void* k_cache, void* v_cache,
                                               const int* seq_len, const int* pad_count, const uint8_t* masked_tokens,
                                               const uint8_t* finished, int B, int nh, int dh, int rot, int s_max,
                                               int step, void* ctx, void* workspace, size_t workspace_bytes,
                                               void* stream)
{
    return guarded([&] {
        require_device();
        MmhaParams p{};
        p.qkv = (const f16*)qkv;
        p.qkv_bias = (const f16*)qkv_bias;
        p.k_cache = (f16*)k_cache;
        p.v_cache = (f16*)v_cache;
        p.seq_len = seq_len;
        p.pad_count = pad_count;
        p.masked_tokens = masked_tokens;
        p.finished = finished;
        p.d_step = nullptr;
        p.step = step;
        p.B = B;
        p.nh = nh;
        p.dh = dh;
        p.rot = rot;
        p.s_max = s_max;
        p.ctx = (f16*)ctx;
        p.gran = (unsigned long long*)workspace;
        p.layer = 0;
        p.nsplit = mmha_pick_nsplit(B, nh, s_max);
        FTCF_CHECK_ARG(workspace_bytes >= mmha_workspace_bytes(B, nh, dh, p.nsplit), "MMHA workspace too small");
        // the granule tags must not match anything left from an earlier call
        FTCF_HIP_CHECK(hipMemsetAsync(p.gran, 0, mmha_workspace_bytes(B, nh, dh, p.nsplit), (hipStream_t)stream));
        launch_mmha(p, (hipStream_t)stream);
    });
}
extern "C" int ftcf_context_attention(const void* qkv, const void* qkv_bias, const int* input_lengths, void* k_cache,
                                      void* v_cache, int B, int S, int nh, int dh, int rot, int s_max, void* ctx,
                                      void* stream)
{
    return guarded([&] {
        require_device();
        launch_context_attention((const f16*)qkv, (const f16*)qkv_bias, input_lengths, (f16*)k_cache, (f16*)v_cache, B,
                                 S, nh, dh, rot, s_max, (f16*)ctx, (hipStream_t)stream);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// the engine
// ---------------------------------------------------------------------------------------------------------------

// transposeAxis01 for the TP logits all-gather: [tp][B][vl] -> [B][V] (GptNeoX.cc:913-924)
__global__ void k_transpose_gathered_logits(float* out, const float* in, int tp, int B, int vl)
{
    const size_t total = (size_t)tp * B * vl;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int    j = (int)(i % vl);
        const size_t t = i / vl;
        const int    b = (int)(t % B), r = (int)(t / B);
        out[(size_t)b * tp * vl + (size_t)r * vl + j] = in[i];
    }
}

template<typename T>
static std::vector<T> broadcast_arg(const T* p, int n, int B, T dflt, const char* name)
{
    std::vector<T> v((size_t)B, dflt);
    if (n > 0) {
        FTCF_CHECK_ARG(p != nullptr, std::string(name) + " pointer is NULL");
        FTCF_CHECK_ARG(n == 1 || n == B, std::string(name) + " must have 1 or batch_size entries");
        for (int b = 0; b < B; b++) {
            v[b] = p[n == 1 ? 0 : b];
        }
    }
    return v;
}

void ftcf_gptneox::begin(const ftcf_forward_args& a)
{
    {  // (once per request, not per token)
        const char* m       = getenv("FTCF_DECODE_OVERLAP");
        decode_overlap_mode = !m ? 0 : (!strcmp(m, "auto") ? 2 : (atoi(m) ? 1 : 0));
    }
    Range r("ftcf.begin");
    FT_LOG_TRACE(cfg.device, "begin: batch %d x beam %d, max_input_len %d, output_len %d", a.batch_size, a.beam_width, a.max_input_len,
                 a.output_len);
    const int B = a.batch_size * (a.beam_width > 0 ? a.beam_width : 1);  // rows
    const int S = a.max_input_len, out_len = a.output_len;
    FTCF_CHECK_ARG(a.batch_size >= 1 && S >= 1 && out_len >= 1, "batch_size, max_input_len and output_len must be >= 1");
    FTCF_CHECK_ARG(a.input_ids && a.input_lengths && a.output_ids && a.sequence_lengths, "NULL tensor");
    const int K = a.beam_width, batch = a.batch_size;
    FTCF_CHECK_ARG(K >= 1 && K <= BEAM_MAX_K, "beam_width must be in [1, 64]");
    FTCF_HIP_CHECK(hipSetDevice(cfg.device));
    abandon_session();  // (a request left open -- begin / step without finish, or a step that threw -- must not leak its graph)
    stats.window_allreduces = 0;
    // everything the caller enqueued on its stream (input tensors) happens-before the engine's work
    FTCF_HIP_CHECK(hipEventRecord(ev_user, user_stream));
    FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ev_user, 0));
    const int total = S + out_len;  // max_output_seq_len == max_seq_len == max_cache_seq_len (GptNeoX.cc:520-523)
    const int s_max = total;
    ses.K = K;  // (decoder path selection reads it)
    const int  tpn     = cfg.tensor_para_size;
    const bool want_tp = tpn > 1 && persist && persist_tp && K == 1 && B <= 2 && cfg.use_gptj_residual && tpn <= PERSIST_MAX_TP;
    // message buffers of the RCCL-free prompt-phase all-reduce (k_window_allreduce) behind the granules of the decode exchange
    static const int winar_on = getenv("FTCF_TP_WINAR") ? atoi(getenv("FTCF_TP_WINAR")) : 1;
    static const int winar_mb = getenv("FTCF_TP_WINAR_MB") ? atoi(getenv("FTCF_TP_WINAR_MB")) : 16;
    const bool want_ar = tpn > 1 && tpn <= 8 && winar_on && !fp32 && cfg.comm && !cfg.comm->local && !cfg.comm->ar_failed;
    if (want_tp || want_ar) {
        // (collective: every rank sees the same request shape) room for two rows: [tp][2 * H / 2] granules
        // (two planes by layer parity: persist_device.hip.h ps_tp_exchange), then 16 flags, then four message buffers
        const size_t gran = ((size_t)2 * tpn * H * 8 + 4095) & ~(size_t)4095;
        const size_t cap  = want_ar ? (size_t)std::max(1, winar_mb) << 20 : 0;
        const bool   had  = cfg.comm->win_ok && cfg.comm->win_bytes >= gran + 4096 + 4 * cap;
        comm_ensure_window(cfg.comm, gran + 4096 + 4 * cap, stream);
        if (cfg.comm->win_ok && cfg.comm->win_bytes >= gran + 4096 + 4 * cap && cap > 0 && (!had || cfg.comm->ar_cap != cap)) {
            cfg.comm->ar_flag_off = gran;  // (a fresh window is zero: the call numbers start over)
            cfg.comm->ar_data_off = gran + 4096;
            cfg.comm->ar_cap      = cap;
            if (!had) {
                cfg.comm->ar_seq = 0;
                if (cfg.comm->ar_sync) {
                    FTCF_HIP_CHECK(hipMemsetAsync(cfg.comm->ar_sync, 0, 64, stream));
                }
            }
        }
    }
    plan(B, S, total, K);
    if (want_tp && pplan.ok) {
        // granule tags repeat from request to request: every rank's window is zeroed between two barriers -- nobody is
        // still writing into it from the previous request, nobody writes before it is clean
        if (!tp_scratch) {
            FTCF_HIP_CHECK(hipMalloc((void**)&tp_scratch, 256));
            FTCF_HIP_CHECK(hipMemsetAsync(tp_scratch, 0, 256, stream));
        }
        comm_barrier(cfg.comm, stream, tp_scratch);
        FTCF_HIP_CHECK(hipMemsetAsync(cfg.comm->win[cfg.tensor_para_rank], 0, (size_t)2 * tpn * H * 8, stream));
        comm_barrier(cfg.comm, stream, tp_scratch);
    }

    // ---- runtime args: routing of TopKSamplingLayer.cu:27-77 / TopPSamplingLayer.cu:30-110 ----
    auto top_k = broadcast_arg<int>(a.top_k, a.n_top_k, batch, 0, "top_k");
    auto top_p = broadcast_arg<float>(a.top_p, a.n_top_p, batch, 0.f, "top_p");
    auto temp  = broadcast_arg<float>(a.temperature, a.n_temperature, batch, 1.f, "temperature");
    auto rep   = broadcast_arg<float>(a.repetition_penalty, a.n_repetition_penalty, batch, 1.f, "repetition_penalty");
    auto seed  = broadcast_arg<uint64_t>(a.random_seed, a.n_random_seed, batch, 0, "random_seed");
    auto minl  = broadcast_arg<int>(a.min_length, a.n_min_length, batch, 0, "min_length");
    auto divr  = broadcast_arg<float>(a.beam_search_diversity_rate, a.n_beam_search_diversity_rate, batch, 0.f,
                                      "beam_search_diversity_rate");
    auto lenp  = broadcast_arg<float>(a.len_penalty, a.n_len_penalty, batch, 0.f, "len_penalty");
    std::vector<int>   k_eff(batch);
    std::vector<float> p_topk(batch), p_topp(batch);
    bool               temp_all_one = true, rep_all_default = true, any_min = false;
    for (int b = 0; b < batch; b++) {
        int   k = top_k[b];
        float p = top_p[b];
        FTCF_CHECK_ARG(k >= 0, "top_k must be >= 0");
        if (k == 0 && p == 0.0f) {
            k = 1;
        }
        float pk = p;
        if (k > 0 && pk == 0.0f) {
            pk = 1.0f;
        }
        k_eff[b]  = k > 1024 ? 1024 : k;
        p_topk[b] = pk < 0.f ? 0.f : (pk > 1.f ? 1.f : pk);
        p_topp[b] = p < 0.f ? 0.f : (p > 1.f ? 1.f : p);
        temp_all_one &= (temp[b] == 1.0f);
        rep_all_default &= (rep[b] == 1.0f);
        any_min |= (minl[b] > 0);
    }
    FTCF_HIP_CHECK(hipMemcpyAsync(d_top_k, k_eff.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_p_topk, p_topk.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_p_topp, p_topp.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_temp, temp.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_rep, rep.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_seed, seed.data(), batch * 8, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_min_length, minl.data(), batch * 4, hipMemcpyHostToDevice, stream));
    if (K > 1) {
        FTCF_HIP_CHECK(hipMemcpyAsync(d_div, divr.data(), batch * 4, hipMemcpyHostToDevice, stream));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_lenpen, lenp.data(), batch * 4, hipMemcpyHostToDevice, stream));
    }
    comm_stream_sync(cfg.comm, stream);  // the host vectors die at scope exit

    hipEvent_t e0 = get_event(), e1 = get_event();
    FTCF_HIP_CHECK(hipEventRecord(e0, stream));
    FTCF_HIP_CHECK(hipMemsetAsync(mmha_ws, 0, mmha_workspace_bytes(B, nhl, dh, nsplit), stream));
    FTCF_HIP_CHECK(hipMemsetAsync(chunk_ws, 0, chunk_workspace_bytes(H, std::min(B, 4), 8), stream));
    if (tiled_ws) {  // tickets of the split-K tiles (re-armed by every launch; a failed request may leave one drawn)
        FTCF_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(tiled_ws) + gemm_tiled_workspace_bytes() - gemm_tiled_ticket_bytes(), 0,
                                      gemm_tiled_ticket_bytes(), stream));
    }
    if (smallm_ws) {
        // granules of the batched-decode GEMMs' in-launch reduction: their tags repeat from request to request
        FTCF_HIP_CHECK(hipMemsetAsync(smallm_ws, 0, smallm_partial + gemm_smallm_ticket_bytes(), stream));
    }
    if (pplan.ok || rplan.ok) {
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        std::vector<PersistLayer> pl(L);
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            PersistLayer&       r = pl[l];
            r.ln1_g = w.ln1_g;
            r.ln1_b = w.ln1_b;
            r.ln2_g = w.ln2_g;
            r.ln2_b = w.ln2_b;
            r.w_qkv = w.qkv.kernel;
            r.w_ffn1 = w.ffn1.kernel;
            r.w_out = w.attn_out.kernel;
            r.w_ffn2 = w.ffn2.kernel;
            r.s_qkv = w.qkv.scale;
            r.s_ffn1 = w.ffn1.scale;
            r.s_out = w.attn_out.scale;
            r.s_ffn2 = w.ffn2.scale;
            r.b_qkv = w.qkv.bias;
            r.b_ffn1 = w.ffn1.bias;
            r.b_res = w.ffn2.bias;
            r.k_cache = k_cache + l * cache_l;
            r.v_cache = v_cache + l * cache_l;
        }
        FTCF_HIP_CHECK(hipMemcpyAsync(d_players, pl.data(), sizeof(PersistLayer) * L, hipMemcpyHostToDevice, stream));
        if (rplan.ok) {  // the rows kernel's flags are monotone tags of (step, layer): steps repeat from request to request
            FTCF_HIP_CHECK(hipMemsetAsync(rows_ws, 0, rows_flag_bytes(rplan, B, nhl), stream));
        }
        if (pplan.ok) {
            FTCF_HIP_CHECK(hipMemsetAsync(ps_gq, 0, (ps_slab_n + 8) * 8, stream));
        }
        ps_tab_ready = false;
        if (pplan.ok && ps_tab && !(cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->local)) {
            // the run / tile tables of this plan: one launch over no layers builds and stores them, every token's launch
            // loads them (17 us of table building per launch otherwise)
            PersistParams pp = persist_params(B, s_max);
            pp.l_begin = pp.l_end = 0;
            pp.tab_mode = 1;
            pp.d_stop   = nullptr;  // (the flag still holds the previous request's outcome here)
            launch_decode_persistent(pp, int8, stream);
            ps_tab_ready = true;
        }
        comm_stream_sync(cfg.comm, stream);  // `pl` dies at scope exit
    }
    // beam search: the reference tiles the inputs K times and runs the context phase on all batch * K rows
    // (GptNeoX.cc:560-574, 640-735).  Every beam reads the prompt K/V of beam 0 anyway (the cache indirection starts at 0),
    // so the prompt is prefilled once per request into cache row b * K and its last hidden state is tiled.
    const int* in_ids = a.input_ids;
    const int* in_len = a.input_lengths;
    if (K > 1) {
        launch_tile_inputs(tiled_ids, tiled_len, a.input_ids, a.input_lengths, batch, K, S, stream);
        FTCF_HIP_CHECK(hipMemsetAsync(cache_indir, 0, (size_t)2 * B * s_max * 4, stream));
        FTCF_HIP_CHECK(hipMemsetAsync(parent_ids, 0, (size_t)total * B * 4, stream));
        in_ids = tiled_ids;
        in_len = tiled_len;
    }
    launch_decode_init(finished, seq_len, cum, pad_count, masked, draws, in_len, state, B, S, s_max, stream, K);
    if (B <= 2 && !fp32) {  // the tagged partials of k_lm_head_greedy: a tag is the step, and steps repeat from request to request
        FTCF_HIP_CHECK(hipMemsetAsync(samp_ws, 0, lm_head_greedy_partial_bytes(B), stream));
    }
    if (S > 1 && K > 1 && fp32) {
        launch32_prompt_embedding(F(px), nullptr, F(wte), a.input_ids, batch, S, H, stream);
        launch_tile_prompt_ids(step_ids, a.input_ids, batch, K, S, stream);
        context_decoder32(batch, S, a.input_lengths, s_max, K);
        launch32_gather_last_token(F(x), F(px), a.input_lengths, batch, S, H, stream, K);
    }
    else if (S > 1 && fp32) {
        launch32_prompt_embedding(F(px), step_ids, F(wte), in_ids, B, S, H, stream);
        context_decoder32(B, S, in_len, s_max, 1);
        launch32_gather_last_token(F(x), F(px), in_len, B, S, H, stream);
    }
    else if (S > 1 && K > 1) {
        launch_prompt_embedding(px, nullptr, wte, a.input_ids, batch, S, H, stream);
        launch_tile_prompt_ids(step_ids, a.input_ids, batch, K, S, stream);
        context_decoder(batch, S, a.input_lengths, s_max, K);
        launch_gather_last_token(x, px, a.input_lengths, batch, S, H, stream, K);
    }
    else if (S > 1) {
        launch_prompt_embedding(px, step_ids, wte, in_ids, B, S, H, stream);
        context_decoder(B, S, in_len, s_max, 1);
        launch_gather_last_token(x, px, in_len, B, S, H, stream);
    }
    else {
        FTCF_HIP_CHECK(hipMemcpyAsync(step_ids, in_ids, (size_t)B * 4, hipMemcpyDeviceToDevice, stream));
    }
    FTCF_HIP_CHECK(hipEventRecord(e1, stream));

    SamplingParams sp{};
    sp.logits = logits;
    sp.B = B;
    sp.V = V;
    sp.max_input_len = S;
    sp.total_len = total;
    sp.end_id = cfg.end_id;
    sp.input_lengths = in_len;
    sp.top_k = d_top_k;
    sp.max_top_k = 1;
    sp.any_top_p = 0;
    for (int i = 0; i < batch; i++) {
        sp.max_top_k = std::max(sp.max_top_k, k_eff[i]);
        sp.any_top_p |= (k_eff[i] == 0);
    }
    sp.top_p_topk = d_p_topk;
    sp.top_p_topp = d_p_topp;
    sp.temperature = d_temp;
    sp.repetition_penalty = a.n_repetition_penalty > 0 ? d_rep : nullptr;
    sp.min_length = any_min ? d_min_length : nullptr;
    sp.random_seed = d_seed;
    sp.draw_counter = draws;
    sp.apply_temperature = temp_all_one ? 0 : 1;
    sp.apply_repetition = (a.n_repetition_penalty > 0 && !rep_all_default) ? 1 : 0;
    sp.stop_words = K > 1 ? nullptr : a.stop_words_list;  // (the beam kernel checks them along the parent chain)
    sp.stop_len = a.stop_words_len;
    sp.optional_last_tokens = a.optional_last_tokens;
    sp.optional_count = a.optional_last_tokens_count;
    sp.return_cum_log_probs = a.return_cum_log_probs ? 1 : 0;
    sp.output_ids = step_ids;
    sp.finished = finished;
    sp.seq_len = seq_len;
    sp.cum_log_probs = cum;
    sp.pad_count = pad_count;
    sp.state = state;
    sp.h_flags = h_flags;
    sp.ws = samp_ws;
    if (!fp32 && H % 8 == 0) {  // k_greedy_decode prepares the next token's input itself (no k_step_prologue launch then)
        sp.next_x    = x;
        sp.wte       = wte;
        sp.rot_table = rot_table;
        sp.H         = H;
        sp.rot       = cfg.rotary_embedding_dim;
    }

    BeamParams bp{};
    if (K > 1) {
        bp.logits = logits;
        bp.B = batch;
        bp.K = K;
        bp.V = V;
        bp.max_input_len = S;
        bp.total_len = total;
        bp.end_id = cfg.end_id;
        bp.s_max = s_max;
        bp.input_lengths = in_len;
        bp.temperature = d_temp;
        bp.repetition_penalty = a.n_repetition_penalty > 0 ? d_rep : nullptr;  // penalty type None otherwise
        bp.diversity_rate = d_div;
        bp.len_penalty = d_lenpen;
        bp.min_length = any_min ? d_min_length : nullptr;
        bp.stop_words = a.stop_words_list;
        bp.stop_len = a.stop_words_len;
        bp.optional_last_tokens = a.optional_last_tokens;
        bp.optional_count = a.optional_last_tokens_count;
        bp.output_ids = step_ids;
        bp.parent_ids = parent_ids;
        bp.finished = finished;
        bp.seq_len = seq_len;
        bp.cum_log_probs = cum;
        bp.cache_indir = cache_indir;
        bp.state = state;
        bp.ws = beam_ws;
    }

    ses.active = true;
    ses.a = a;
    ses.sp = sp;
    ses.bp = bp;
    ses.K = K;
    ses.batch = batch;
    ses.B = B;
    ses.S = S;
    ses.total = total;
    ses.s_max = s_max;
    ses.next_step = S;
    ses.steps = 0;
    ses.path_logged = false;
    ses.all_finished = false;
    ses.e0 = e0;
    ses.e1 = e1;
    stats.decode_ms = 0.f;
    dv_ran = dv_eligible = false;
}

// enqueues one iteration of the token loop (GptNeoX.cc:776-1048) on `stream` (and the side stream)
void ftcf_gptneox::enqueue_step(bool with_decoder)
{
    const ftcf_forward_args& a = ses.a;
    const int B = ses.B, S = ses.S, s_max = ses.s_max;
    const int tp = cfg.tensor_para_size;
    if (with_decoder && fp32) {
        launch32_step_prologue(F(x), F(wte), step_ids, &state->step, rot_table, pad_count, B, H, cfg.rotary_embedding_dim,
                               stream);
        decoder32(B, s_max);
    }
    else if (with_decoder) {
            // (an all-greedy batch: the previous token's k_greedy_decode has written this token's embedding row and rotary
            // table already -- unless this is the request's first dynamic-decode step, a one-token prompt)
            if (!(ses.K == 1 && ses.sp.next_x && ses.steps > 0 && dynamic_decode_is_fused(ses.sp, true))) {
                launch_step_prologue(x, wte, step_ids, &state->step, rot_table, pad_count, B, H, cfg.rotary_embedding_dim,
                                     stream, &state->all_finished);
            }
            decoder(B, s_max);
        }
        // final LayerNorm (GptNeoX.cc:854-863) is fused into the LM-head GEMV for m <= 4
        const bool fuse_ln = B <= 4 && !fp32;
        if (!fuse_ln) {
            launch_layernorm(x, final_g, final_b, nrm, B, H, 1e-5f, !fp32, stream);
        }
        auto lm = [&](const f16* Wrows, float* out, int rows, int ld) {
            if (fp32) {  // Wrows counts f16 elements: the caller's row offset is scaled here
                launch32_lm_head(F(nrm), F(lm_head) + (Wrows - lm_head), out, B, rows, H, ld, stream);
            }
            else if (fuse_ln) {
                launch_lm_head(x, Wrows, out, B, rows, H, ld, stream, final_g, final_b, 1e-5f, &state->all_finished);
            }
            else {
                lm_head_dispatch(nrm, Wrows, out, B, rows, H, ld, stream);
            }
        };
        // one GPU, <= 2 rows, an all-greedy step: the LM head launch picks the tokens, closes the step and prepares the next
        // token's input itself (k_lm_head_greedy) -- nothing is launched behind it
        const bool lm_greedy = tp == 1 && fuse_ln && ses.K == 1 && lm_head_greedy_ok(ses.sp, H);
        if (lm_greedy) {
            timed(KIND_LM_HEAD, 2.0 * V * H,
                  [&] { launch_lm_head_greedy(x, lm_head, logits, H, final_g, final_b, 1e-5f, ses.sp, stream); });
        }
        else if (tp == 1) {
            timed(KIND_LM_HEAD, 2.0 * V * H, [&] { lm(lm_head, logits, V, V); });
        }
        else {
            // rank r computes rows [r*vl, (r+1)*vl) of the replicated lm_head (GptNeoX.cc:888-925)
            float* mine = gather + (size_t)cfg.tensor_para_rank * B * vl;
            timed(KIND_LM_HEAD, 2.0 * vl * H,
                  [&] { lm(lm_head + (size_t)cfg.tensor_para_rank * vl * H, mine, vl, vl); });
            allgather_logits(gather, logits, B, stream);
        }
        if (a.debug_logits) {
            FTCF_HIP_CHECK(hipMemcpyAsync(a.debug_logits + (size_t)(ses.next_step - S) * B * V, logits,
                                          (size_t)B * V * 4, hipMemcpyDeviceToDevice, stream));
        }
    if (ses.K > 1) {
        dynamic_decode_layer.forward(ses.bp, ses.sp, stream);
    }
    else if (!lm_greedy) {
        dynamic_decode_layer.forward(ses.sp, stream);
    }
}

// the token loop of GptNeoX<T>::forward (GptNeoX.cc:776-1048); returns the number of iterations executed
int ftcf_gptneox::step(int max_steps)
{
    Range r("ftcf.step");
    FTCF_CHECK_ARG(ses.active, "no request in flight: call ftcf_gptneox_begin first");
    FTCF_HIP_CHECK(hipSetDevice(cfg.device));
    const ftcf_forward_args& a = ses.a;
    const int B = ses.B, S = ses.S, total = ses.total;
    const int tp = cfg.tensor_para_size;
    std::vector<int> h_tokens(B), h_idx(B), h_seq(B);
    hipEvent_t ea = get_event(), eb = get_event();
    FTCF_HIP_CHECK(hipEventRecord(ea, stream));
    // (read per call, not once per process: the tests switch it between engines)
    const int graph_tokens_cfg = getenv("FTCF_GRAPH_TOKENS") ? atoi(getenv("FTCF_GRAPH_TOKENS")) : 8;
    int  done    = 0;
    bool lagging = false;  // the host has not yet seen the flags of the token launched last
    int  lag_slot = 0;     // launches of the pipelined loop so far (two events alternate)
    while (done < max_steps && ses.next_step < total && !ses.all_finished) {
        const int  step         = ses.next_step;
        const bool with_decoder = !(S > 1 && step == S);
        // (with tensor parallelism the step contains RCCL collectives or the exchange windows' spins: not captured -- a capture
        // of the RCCL form aborted inside this image's HIP runtime, profiles/r05_notes.md)
        const bool graph_ok     = use_graph && with_decoder && !profiling && tp == 1 && !a.debug_logits;
        // Several tokens per graph launch (FTCF_GRAPH_TOKENS, default 8): a graph launch costs ~14 us of GPU idle time between
        // two tokens (profiles/r03_notes.md section 6), and every kernel of a persistent-path token returns at once when the
        // device-side "every row has finished" flag is set, so the tokens of a graph behind the request's last one cost a few
        // microseconds each, not a decoder pass.  Persistent decode path, no streaming callback, one GPU.
        const int  graph_tokens = std::max(1, std::min(64, graph_tokens_cfg));
        const bool multi        = graph_ok && graph_tokens > 1 && pplan.ok && ses.K == 1 && !a.callback && tp == 1
                                  && max_steps - done >= graph_tokens && total - step >= graph_tokens;
        int launched = 1;
        if (graph_ok) {
            hipGraphExec_t& ge = multi ? ses.graph_exec_n : ses.graph_exec;
            if (!ge) {
                // capture the regular decode step(s) (all pointers are fixed for the session, the step counter lives
                // on the device) and replay: no per-kernel host launch cost, cross-stream fork/join become edges
                hipGraph_t g = nullptr;
                FTCF_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                try {
                    for (int t = 0; t < (multi ? graph_tokens : 1); t++) {
                        enqueue_step(true);
                    }
                }
                catch (...) {
                    (void)hipStreamEndCapture(stream, &g);
                    if (g) {
                        (void)hipGraphDestroy(g);
                    }
                    throw;
                }
                FTCF_HIP_CHECK(hipStreamEndCapture(stream, &g));
                FTCF_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                FTCF_HIP_CHECK(hipGraphDestroy(g));
            }
            FTCF_HIP_CHECK(hipGraphLaunch(ge, stream));
            launched = multi ? graph_tokens : 1;
        }
        else {
            enqueue_step(with_decoder);
        }
        if (a.debug_logits) {
            // (logits are modified in place by the decode kernels; the tap is taken inside enqueue_step when eager)
        }
        ses.steps += launched;
        ses.next_step += launched;
        done += launched;
        // The reference synchronises once per token here (stop_criteria_kernels.cu:149-156).  Without a streaming
        // callback nothing on the host needs token t before token t+1 is enqueued, so the replayed graph of the next
        // token(s) is launched first and the host only waits for the PREVIOUS launch's event: the GPU never idles for the
        // host round trip (~25 us per token).  `finished` is sticky on the device and the launches behind the last token
        // return at once; the host's counters are set back to the device's when it learns of the end (below).
        if (graph_ok && !a.callback && tp == 1) {  // (tp > 1: every rank must leave the loop at the SAME token -> synchronous)
            hipEvent_t& ev = tok_ev[lag_slot & 1];
            if (!ev) {
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            }
            FTCF_HIP_CHECK(hipEventRecord(ev, stream));
            if (lagging) {
                comm_event_sync(cfg.comm, tok_ev[(lag_slot - 1) & 1]);
                ses.all_finished = h_flags[0] != 0;
            }
            lag_slot++;
            lagging = true;
            continue;
        }
        lagging = false;
        comm_stream_sync(cfg.comm, stream);
        ses.all_finished = h_flags[0] != 0;
        if (a.callback && step + 1 < total && cfg.tensor_para_rank == 0) {
            // pybind_callback_utils.cc:22-103: last token of every row; a row that did not advance reports end_id
            FTCF_HIP_CHECK(hipMemcpy(h_tokens.data(), step_ids + (size_t)step * B, (size_t)B * 4, hipMemcpyDeviceToHost));
            FTCF_HIP_CHECK(hipMemcpy(h_seq.data(), seq_len, (size_t)B * 4, hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++) {
                h_idx[b] = h_seq[b];
                if (h_seq[b] != step) {
                    h_tokens[b] = cfg.end_id;
                }
            }
            a.callback(h_tokens.data(), h_idx.data(), ses.batch, ses.K, a.callback_user);
        }
    }
    FTCF_HIP_CHECK(hipEventRecord(eb, stream));
    comm_event_sync(cfg.comm, eb, "the prefill");
    if (lagging) {
        ses.all_finished = h_flags[0] != 0;
    }
    if (ses.all_finished && ses.K == 1) {
        // launches behind the token that finished the last row did nothing on the device (every kernel of a token returns at
        // once on the flag, k_decode_finish included: the device's step counter stopped): the host's counters follow the
        // device's, so that `steps_done` is the reference's loop count (GptNeoX.cc:776-1048 leaves its loop at that token)
        const int over = ses.next_step - (h_flags[1] + 1);
        if (over > 0) {
            ses.next_step -= over;
            ses.steps -= over;
            done -= over;
        }
    }
    float ms = 0.f;
    FTCF_HIP_CHECK(hipEventElapsedTime(&ms, ea, eb));
    stats.decode_ms += ms;
    event_pool.push_back(ea);
    event_pool.push_back(eb);
    return done;
}

void ftcf_gptneox::finish()
{
    Range r("ftcf.finish");
    FTCF_CHECK_ARG(ses.active, "no request in flight");
    const ftcf_forward_args& a = ses.a;
    // setOutputTensors (GptNeoX.cc:1090-1181)
    if (ses.K > 1) {
        launch_gather_tree_beam(a.output_ids, a.sequence_lengths, step_ids, parent_ids, seq_len, tiled_len, ses.batch,
                                ses.K, ses.S, ses.total, cfg.end_id, stream);
    }
    else {
        launch_gather_tree(a.output_ids, a.sequence_lengths, step_ids, seq_len, a.input_lengths, ses.B, ses.S,
                           ses.total, cfg.end_id, stream);
    }
    if (a.cum_log_probs) {
        FTCF_HIP_CHECK(hipMemcpyAsync(a.cum_log_probs, cum, (size_t)ses.B * 4, hipMemcpyDeviceToDevice, stream));
    }
    int ps_error = 0, smallm_error = 0;
    if (pplan.ok) {
        FTCF_HIP_CHECK(hipMemcpyAsync(&ps_error, ps_err, sizeof(int), hipMemcpyDeviceToHost, stream));
    }
    if (smallm_ws) {
        FTCF_HIP_CHECK(hipMemcpyAsync(&smallm_error, reinterpret_cast<char*>(smallm_ws) + smallm_partial, sizeof(int),
                                      hipMemcpyDeviceToHost, stream));
    }
    int rows_error = 0;
    if (rplan.ok) {
        FTCF_HIP_CHECK(hipMemcpyAsync(&rows_error, rows_ws, sizeof(int), hipMemcpyDeviceToHost, stream));
    }
    comm_stream_sync(cfg.comm, stream);
    float ms = 0.f;
    FTCF_HIP_CHECK(hipEventElapsedTime(&ms, ses.e0, ses.e1));
    stats.prefill_ms   = ms;
    stats.decode_steps = ses.steps;
    if (ov_eligible && ov_trial < 2) {  // a trial of the auto mode: every rank keeps the slowest rank's time
        const int us   = comm_max(cfg.comm, (int)(ms * 1000.f), stream, tp_scratch);
        ov_ms[ov_trial] = us * 1e-3f;
        ov_trial++;
        if (ov_trial == 2) {
            FT_LOG_INFO(cfg.device, "prompt-phase all-reduce overlap (auto): plain %.2f ms, overlapped %.2f ms -> %s", ov_ms[0], ov_ms[1],
                        ov_ms[1] < ov_ms[0] ? "overlapped from now on" : "plain from now on");
        }
    }
    if (dv_eligible && dv_trial < 2 && ses.steps > 0) {  // a trial of the decode overlap's auto mode (engine.hip.h decoder_overlapped)
        const int us = comm_max(cfg.comm, (int)(stats.decode_ms * 1000.f / ses.steps), stream, tp_scratch);
        dv_ms[dv_trial] = us * 1e-3f;
        dv_trial++;
        if (dv_trial == 2) {
            FT_LOG_INFO(cfg.device, "decode all-reduce overlap (auto): plain %.3f ms per step, overlapped %.3f -> %s", dv_ms[0], dv_ms[1],
                        dv_ms[1] < dv_ms[0] ? "overlapped from now on" : "plain from now on");
        }
    }
    stats.decode_overlap            = dv_ran ? 1 : 0;
    stats.decode_step_ms_plain      = dv_ms[0];
    stats.decode_step_ms_overlapped = dv_ms[1];
    stats.prefill_overlap        = ov_ran ? 1 : 0;
    stats.prefill_ms_plain       = ov_ms[0];
    stats.prefill_ms_overlapped  = ov_ms[1];
    event_pool.push_back(ses.e0);
    event_pool.push_back(ses.e1);
    ses.e0 = ses.e1 = nullptr;
    ses.active = false;
    if (ses.graph_exec) {
        (void)hipGraphExecDestroy(ses.graph_exec);
        ses.graph_exec = nullptr;
    }
    if (ses.graph_exec_n) {
        (void)hipGraphExecDestroy(ses.graph_exec_n);
        ses.graph_exec_n = nullptr;
    }
    drain_events();
    if (pplan.ok && ps_ts) {
        std::vector<long long> h((size_t)pplan.NB * L * 128);
        FTCF_HIP_CHECK(hipMemcpy(h.data(), ps_ts, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(ps_ts_file.c_str(), "wb")) {
            const int hdr[4] = {pplan.NB, L, 8, 16};
            fwrite(hdr, 4, 4, f);
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
    if (rplan.ok && rows_ts) {
        std::vector<long long> h((size_t)rplan.NB * L * 128);
        FTCF_HIP_CHECK(hipMemcpy(h.data(), rows_ts, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(ps_ts_file.c_str(), "wb")) {
            const int hdr[4] = {rplan.NB, L, 8, 16};
            fwrite(hdr, 4, 4, f);
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
    if (rplan.ok && persist_fail_once) {  // the same test hook for the rows kernel
        persist_fail_once = 0;
        rows_error        = 99;
    }
    if (rplan.ok && cfg.tensor_para_size > 1) {
        if (!tp_scratch) {
            FTCF_HIP_CHECK(hipMalloc((void**)&tp_scratch, 256));
            FTCF_HIP_CHECK(hipMemsetAsync(tp_scratch, 0, 256, stream));
        }
        rows_error = comm_max(cfg.comm, rows_error, stream, tp_scratch);
    }
    if (pplan.ok && persist_fail_once) {  // test hook (FTCF_PERSIST_FAIL_ONCE=1): pretend the kernel gave up once
        persist_fail_once = 0;
        ps_error          = 99;
    }
    if (pplan.ok && cfg.tensor_para_size > 1) {
        ps_error = comm_max(cfg.comm, ps_error, stream, tp_scratch);  // every rank learns of any rank's failure
    }
    if (cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->ar_seq > 0 && !cfg.comm->ar_failed) {
        // the window all-reduce's sticky give-up word (a peer that never arrived: its bounded waits ran out), agreed on
        int ar_err = 0;
        FTCF_HIP_CHECK(hipMemcpy(&ar_err, cfg.comm->ar_sync + 2, sizeof(int), hipMemcpyDeviceToHost));
        if (!tp_scratch) {
            FTCF_HIP_CHECK(hipMalloc((void**)&tp_scratch, 256));
            FTCF_HIP_CHECK(hipMemsetAsync(tp_scratch, 0, 256, stream));
        }
        ar_err = comm_max(cfg.comm, ar_err, stream, tp_scratch);
        if (ar_err != 0) {
            cfg.comm->ar_failed = true;  // this communicator keeps RCCL (or the emulation) for the prompt phase from now on
            winar_failed        = true;
            throw Error(-2, "exchange-window all-reduce gave up waiting for a peer (its kernel was not running next to this one)");
        }
    }
    if (h_flags[2] != 0) {  // k_lm_head_greedy's finishing workgroup never saw some workgroup's partial (cannot happen: bounded all the same)
        h_flags[2] = 0;
        throw Error(-2, "LM head + greedy decode: the finishing workgroup gave up waiting for the partials of the launch");
    }
    if (smallm_error != 0) {
        throw Error(-2, "batched decode GEMM: a split-K reducer gave up waiting for its sibling workgroups' partial sums");
    }
    if (rows_error != 0) {
        persist_failed = true;
        rows           = 0;  // the next request (and the replay of this one) is planned on the general path
        throw Error(-2, "rows decode kernel gave up waiting for a hand-off (code " + std::to_string(rows_error)
                            + "): not every workgroup was resident");
    }
    if (ps_error != 0) {
        persist_failed = true;
        persist        = 0;  // whoever drives begin / step / finish: the next request is planned off the persistent path
        throw Error(-2, "persistent decode kernel gave up waiting for a hand-off (code " + std::to_string(ps_error)
                            + "): not every workgroup was resident");
    }
}

extern "C" int ftcf_gptneox_create(const ftcf_gptneox_config* cfg, const ftcf_gptneox_weights* w, ftcf_gptneox_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(cfg && w && out, "NULL argument");
        require_device();
        FTCF_CHECK_ARG(cfg->pipeline_para_size == 1, "pipeline_para_size must be 1 (the CodeFuse harness forces it)");
        if (cfg->dtype != FTCF_FP16 && cfg->dtype != FTCF_FP32) {
            throw Error(FTCF_ERR_UNSUPPORTED, "the GPU engine instantiates fp16 and fp32 (as GptNeoXOp.cc:56-105)");
        }
        FTCF_CHECK_ARG(cfg->int8_mode == 0 || cfg->int8_mode == 1, "int8_mode must be 0 or 1");
        if (cfg->dtype == FTCF_FP32 && cfg->int8_mode != 0) {
            throw Error(FTCF_ERR_UNSUPPORTED, "weight-only int8 needs half activations (CutlassFpAIntBGemmRunner<half, uint8_t>)");
        }
        const int tp = cfg->tensor_para_size;
        FTCF_CHECK_ARG(tp >= 1 && cfg->head_num % tp == 0 && cfg->inter_size % tp == 0 && cfg->vocab_size % tp == 0,
                       "head_num, inter_size and vocab_size must be divisible by tensor_para_size");
        const int L = cfg->num_layer;
        FTCF_CHECK_ARG(w->n_weights == 12 * L + 4, "weights must hold 12*L+4 tensors");
        FTCF_CHECK_ARG(L >= 1 && L <= 256, "num_layer must be in 1..256");
        FTCF_HIP_CHECK(hipSetDevice(cfg->device));
        FT_LOG_INFO(cfg->device, "GptNeoX engine on device %d: %d layers, %d heads x %d, inter %d, vocab %d, rotary %d, %s%s, "
                                 "tensor_para %d/%d, %s residual",
                    cfg->device, L, cfg->head_num, cfg->size_per_head, cfg->inter_size, cfg->vocab_size, cfg->rotary_embedding_dim,
                    cfg->dtype == FTCF_FP32 ? "fp32" : "fp16", cfg->int8_mode ? " + int8 weights" : "", cfg->tensor_para_rank, tp,
                    cfg->use_gptj_residual ? "parallel" : "sequential");
        auto e   = std::make_unique<ftcf_gptneox>();
        e->cfg   = *cfg;
        e->L     = L;
        e->dh    = cfg->size_per_head;
        e->H     = cfg->head_num * cfg->size_per_head;
        e->nhl   = cfg->head_num / tp;
        e->hl    = e->nhl * e->dh;
        e->il    = cfg->inter_size / tp;
        e->V     = cfg->vocab_size;
        e->vl    = e->V / tp;
        e->int8  = cfg->int8_mode == 1;
        e->fp32  = cfg->dtype == FTCF_FP32;
        e->user_stream = (hipStream_t)cfg->stream;
        FTCF_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        FTCF_HIP_CHECK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
        FTCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        FTCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
        FTCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_user, hipEventDisableTiming));
        // weights still being uploaded / produced on the caller's stream happen-before the re-tiling below
        FTCF_HIP_CHECK(hipEventRecord(e->ev_user, e->user_stream));
        FTCF_HIP_CHECK(hipStreamWaitEvent(e->stream, e->ev_user, 0));
        if (e->fp32) {
            FTCF_CHECK_ARG(e->dh >= 32 && e->dh <= 256 && e->dh % 2 == 0, "size_per_head must be even, 32..256");
        }
        else {
            // the reference's list (DecoderSelfAttentionLayer.cc:280-282).  64 and 128 run every decode path; the other sizes
            // take the general path (one attention launch per layer): the persistent and the fused per-stage kernels, and the
            // paged attention of the batcher, keep their two tuned lane layouts
            FTCF_CHECK_ARG(mmha_head_size_supported(e->dh), "size_per_head must be one of 32, 48, 64, 80, 96, 128, 144, 160, 192, 224, 256");
            FTCF_CHECK_ARG(e->H % 64 == 0 && e->hl % 64 == 0 && e->il % 64 == 0,
                           "hidden, local hidden and local inter sizes must be multiples of 64");
        }
        if (e->int8) {
            FTCF_CHECK_ARG(w->n_int8_weights == 4 * L && w->n_scales == 4 * L, "int8 lists must hold 4*L tensors");
        }
        auto W = [&](int g, int l) { return w->weights[(size_t)g * L + l]; };
        e->layers.resize(L);
        const int H = e->H, hl = e->hl, il = e->il;
        for (int l = 0; l < L; l++) {
            LayerWeights& lw = e->layers[l];
            lw.ln1_b = (const f16*)W(0, l);
            lw.ln1_g = (const f16*)W(1, l);
            lw.qkv.bias = (const f16*)W(3, l);
            lw.attn_out.bias = (const f16*)W(5, l);
            lw.ffn1.bias = (const f16*)W(7, l);
            lw.ffn2.bias = (const f16*)W(9, l);
            lw.ln2_b = (const f16*)W(10, l);
            lw.ln2_g = (const f16*)W(11, l);
            FTCF_CHECK_ARG(lw.ln1_b && lw.ln1_g && lw.ln2_b && lw.ln2_g && lw.qkv.bias && lw.ffn1.bias && lw.ffn2.bias,
                           "missing layernorm / bias tensor");
            FTCF_CHECK_ARG(cfg->use_gptj_residual || lw.attn_out.bias,
                           "use_gptj_residual == 0 needs the attention output bias (weights[5*L + l])");
            if (e->int8) {
                lw.qkv.kernel = w->int8_weights[0 * L + l];
                lw.attn_out.kernel = w->int8_weights[1 * L + l];
                lw.ffn1.kernel = w->int8_weights[2 * L + l];
                lw.ffn2.kernel = w->int8_weights[3 * L + l];
                lw.qkv.scale = (const f16*)w->scales[0 * L + l];
                lw.attn_out.scale = (const f16*)w->scales[1 * L + l];
                lw.ffn1.scale = (const f16*)w->scales[2 * L + l];
                lw.ffn2.scale = (const f16*)w->scales[3 * L + l];
                FTCF_CHECK_ARG(lw.qkv.kernel && lw.attn_out.kernel && lw.ffn1.kernel && lw.ffn2.kernel && lw.qkv.scale
                                   && lw.attn_out.scale && lw.ffn1.scale && lw.ffn2.scale,
                               "missing int8 kernel / scale tensor");
            }
            else if (e->fp32) {
                // row-major [K, N] fp32 kernels are read in place
                lw.qkv.kernel = W(2, l);
                lw.attn_out.kernel = W(4, l);
                lw.ffn1.kernel = W(6, l);
                lw.ffn2.kernel = W(8, l);
                FTCF_CHECK_ARG(lw.qkv.kernel && lw.attn_out.kernel && lw.ffn1.kernel && lw.ffn2.kernel, "missing fp32 kernel tensor");
            }
            else {
                // re-tile the reference-layout [K, N] fp16 kernels once (the binding keeps the originals alive)
                struct {
                    int          g;
                    size_t       K, N;
                    DenseWeight* d;
                } items[4] = {{2, (size_t)H, (size_t)3 * hl, &lw.qkv},
                              {4, (size_t)hl, (size_t)H, &lw.attn_out},
                              {6, (size_t)H, (size_t)il, &lw.ffn1},
                              {8, (size_t)il, (size_t)H, &lw.ffn2}};
                // FTCF_FP16_RETILE_IN_PLACE=1: the caller's row-major kernels are OVERWRITTEN with the tiled image (through one
                // bounce buffer) instead of being kept next to a tiled copy -- the engine then holds one copy of the fp16 weights
                // (26 GB instead of 52 GB at 13B); the caller must not read those tensors as [K, N] matrices afterwards
                const bool in_place = getenv("FTCF_FP16_RETILE_IN_PLACE") && atoi(getenv("FTCF_FP16_RETILE_IN_PLACE")) != 0;
                for (auto& it : items) {
                    const void* src = W(it.g, l);
                    FTCF_CHECK_ARG(src != nullptr, "missing fp16 kernel tensor");
                    const size_t bytes = it.K * it.N * 2;
                    if (in_place) {
                        if (e->bounce_bytes < bytes) {
                            if (e->bounce) {
                                FTCF_HIP_CHECK(hipStreamSynchronize(e->stream));
                                (void)hipFree(e->bounce);
                            }
                            FTCF_HIP_CHECK(hipMalloc(&e->bounce, bytes));
                            e->bounce_bytes = bytes;
                        }
                        launch_fp16_rowmajor_to_tiled((const f16*)src, it.K, it.N, (f16*)e->bounce, e->stream);
                        FTCF_HIP_CHECK(hipMemcpyAsync(const_cast<void*>(src), e->bounce, bytes, hipMemcpyDeviceToDevice, e->stream));
                        it.d->kernel = src;
                        continue;
                    }
                    void* dst = nullptr;
                    FTCF_HIP_CHECK(hipMalloc(&dst, bytes));
                    e->owned.push_back(dst);
                    launch_fp16_rowmajor_to_tiled((const f16*)src, it.K, it.N, (f16*)dst, e->stream);
                    it.d->kernel = dst;
                }
            }
        }
        e->wte     = (const f16*)w->weights[12 * L];
        e->final_g = (const f16*)w->weights[12 * L + 1];  // GptNeoXOp.h:172-173: slot 12L+1 = gamma
        e->final_b = (const f16*)w->weights[12 * L + 2];
        e->lm_head = (const f16*)w->weights[12 * L + 3];
        FTCF_CHECK_ARG(e->wte && e->final_g && e->final_b && e->lm_head, "missing embedding / final layernorm / lm_head");
        e->k3_q = e->fp32 ? 1 : chunk_pick_q(e->H / 16, (e->hl + e->il) / (e->int8 ? TILE_K_I8 : TILE_K_F16));
        if (const char* m = getenv("FTCF_STAGE_MAX_ROWS")) {
            e->STAGE_MAX_ROWS = std::max(0, std::min(4, atoi(m)));
        }
        if (const char* m = getenv("FTCF_SMALLM_MAX_ROWS")) {
            e->SMALLM_MAX_ROWS = std::max(16, std::min(256, atoi(m)));
        }
        {
            hipDeviceProp_t prop;
            FTCF_HIP_CHECK(hipGetDeviceProperties(&prop, cfg->device));
            e->num_cu = prop.multiProcessorCount;
        }
        if (const char* m = getenv("FTCF_PERSIST")) {
            e->persist = atoi(m);
        }
        if (const char* m = getenv("FTCF_ROWS")) {
            e->rows = atoi(m);
        }
        if (const char* m = getenv("FTCF_ROWS_NB")) {
            e->rows_nb = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_FAIL_ONCE")) {
            e->persist_fail_once = atoi(m);
        }
        if (const char* m = getenv("FTCF_TP_PERSIST")) {
            e->persist_tp = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_TS")) {
            e->ps_ts_file = m;
        }
        if (const char* m = getenv("FTCF_PERSIST_NB")) {
            e->persist_nb = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_CS1")) {
            e->persist_cs1 = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_CS3")) {
            e->persist_cs3 = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_OWN")) {  // 0: P3 as K pieces merged by an owner (the form until round 5), 1: own-group
                                                            // layout wherever the shape divides, 2 (default): where it measured faster
            e->persist_own = atoi(m);
        }
        e->decode_branches = cfg->tensor_para_size == 1 ? 1 : 0;
        if (const char* m = getenv("FTCF_DECODE_BRANCHES")) {
            e->decode_branches = atoi(m);
        }
        e->use_graph = cfg->use_hip_graph != 0;
        if (const char* m = getenv("FTCF_TP_PAIR_AR")) {
            e->tp_pair_ar = atoi(m) != 0;
        }
        if (const char* m = getenv("FTCF_USE_GRAPH")) {
            e->use_graph = atoi(m) != 0;
        }
        if (cfg->tensor_para_size > 1) {
            // scratch word of the communicator helpers (comm_barrier / comm_max / comm_agree): every path of a tensor-parallel
            // engine may reach them (the prompt-phase overlap's trials, the kernels' error words), whatever it planned
            FTCF_HIP_CHECK(hipMalloc((void**)&e->tp_scratch, 256));
            FTCF_HIP_CHECK(hipMemsetAsync(e->tp_scratch, 0, 256, e->stream));
        }
        FTCF_HIP_CHECK(hipHostMalloc((void**)&e->h_flags, 64, hipHostMallocDefault));
        e->h_flags[0] = e->h_flags[1] = e->h_flags[2] = 0;
        FTCF_HIP_CHECK(hipStreamSynchronize(e->stream));
        if (e->bounce) {
            (void)hipFree(e->bounce);
            e->bounce       = nullptr;
            e->bounce_bytes = 0;
        }
        *out = e.release();
    });
}

extern "C" int ftcf_gptneox_forward(ftcf_gptneox_t h, const ftcf_forward_args* args)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h && args, "NULL argument");
        h->forward(*args);
    });
}

extern "C" int ftcf_gptneox_begin(ftcf_gptneox_t h, const ftcf_forward_args* args)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h && args, "NULL argument");
        h->begin(*args);
    });
}
extern "C" int ftcf_gptneox_step(ftcf_gptneox_t h, int max_steps, int* steps_done)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h, "NULL argument");
        const int n = h->step(max_steps);
        if (steps_done) {
            *steps_done = n;
        }
    });
}
extern "C" int ftcf_gptneox_finish(ftcf_gptneox_t h)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h, "NULL argument");
        h->finish();
    });
}

extern "C" int ftcf_gptneox_get_stats(ftcf_gptneox_t h, ftcf_forward_stats* s)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h && s, "NULL argument");
        *s = h->stats;
        // dominant weight-streaming kernel kind
        int best = 0;
        for (int k = 1; k < KIND_COUNT; k++) {
            if (h->kind_ms[k] > h->kind_ms[best]) {
                best = k;
            }
        }
        s->gemv_kind     = best;
        s->gemv_ms_sum   = (float)h->kind_ms[best];
        s->gemv_launches = h->kind_n[best];
        s->gemv_bytes    = h->kind_bytes[best];
    });
}

extern "C" int ftcf_gptneox_set_profiling(ftcf_gptneox_t h, int enabled)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h, "NULL argument");
        h->profiling = enabled != 0;
        for (int k = 0; k < KIND_COUNT; k++) {
            h->kind_ms[k] = h->kind_bytes[k] = 0;
            h->kind_n[k]                     = 0;
        }
    });
}

extern "C" int ftcf_gptneox_destroy(ftcf_gptneox_t h)
{
    return guarded([&] { delete h; });
}
