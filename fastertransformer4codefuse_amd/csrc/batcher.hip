// Continuous batching over a paged K/V cache (include/ftcf.h `ftcf_batcher_*`): split out of engine.hip in round 4.
#include "engine.hip.h"
#include "attn_device.hip.h"  // rotary_coef (the rows kernel's rotary table of a step)

#include <chrono>
#include <set>


// ---------------------------------------------------------------------------------------------------------------
// Continuous batching over a paged K/V cache (no counterpart in the reference, whose serving layer -- the
// Triton backend -- allocates the cache per request, GptNeoX.cc:84-156).
//
// A batcher borrows an engine (weights, streams, kernels) and owns
//   * a K/V POOL of fixed-size pages, [L][page][head][P tokens][dh] fp16, shared by all sequences, with a free list;
//   * `max_batch` SLOTS: page table, length, last token, sampling parameters of the sequence living there;
//   * a queue of waiting requests.
// One iteration (ftcf_batcher_step) = ADMIT waiting requests into free slots while their pages (prompt + max_new_tokens,
// reserved up front: a running sequence never has to be preempted) are available, then ONE decode step for all running
// slots.  Admission runs the prompt through the engine's own context path (ftcf_gptneox_forward with output_len 1: prefill +
// first token sampled by the engine's dynamic decode) and scatters the prompt's K/V into the slot's pages; the decode
// step is the general layer sequence of the engine (§4a: dual LayerNorm, burst / tiled GEMMs, fused residual) with the
// attention replaced by k_mmha_paged, the LM head, and the engine's sampling kernels on per-slot arrays.  A sequence leaves
// when it emits end_id or reaches max_new_tokens; its pages return to the free list at once.
// Scope: parallel-residual models, any tensor_para_size (round 4: one batcher per rank), fp16 / int8 engines; top-k / top-p /
// temperature sampling with repetition penalty, stop words and a token callback; beam search (round 4: BeamGroup below).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_batcher_embed(f16* out, const f16* table, const int* tok, int H)
{
    const int  id  = tok[blockIdx.x];
    const f16* src = table + (size_t)id * H;
    f16*       dst = out + (size_t)blockIdx.x * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
    }
}
// d_tok[b] = the token the sampling kernels just wrote into slot b's history (time-major [max_len][B], position len[b])
__global__ void k_batcher_last_token(int* tok, const int* hist, const int* len, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        tok[b] = hist[(size_t)len[b] * B + b];
    }
}
__global__ void k_batcher_tick(DecodeState* gemm_state)
{
    gemm_state->step = (gemm_state->step + 1) & 0x7ffff;  // part of the burst GEMMs' granule tags (19 bits)
}

// the rows kernel inside the batcher (round 5): its step counter (the hand-off tags are step * 256 + layer + 1) and the rotary table
// of the step -- a slot's new token sits at position len[slot] (decoder_masked_multihead_attention_utils.h:1325-1345 coefficients,
// as k_mmha_paged computes them in place)
__global__ void k_batcher_rows_prep(int* step, float* rot_table, const int* len, int rot)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *step = *step + 1;
    }
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < rot / 2; j += blockDim.x) {
        float cs, sn;
        rotary_coef(j, rot, len[b], cs, sn);
        rot_table[((size_t)b * (rot / 2) + j) * 2]     = cs;
        rot_table[((size_t)b * (rot / 2) + j) * 2 + 1] = sn;
    }
}

// beam groups (see ftcf_batcher::BeamGroup): copy-on-write of K/V pages, every layer of both pools; pairs = {src, dst} page ids
__global__ void k_batcher_copy_pages(f16* kpool, f16* vpool, const int* pairs, size_t page_elems, size_t pool_layer_elems)
{
    f16*         pool = blockIdx.z ? vpool : kpool;
    const int    src = pairs[2 * blockIdx.x], dst = pairs[2 * blockIdx.x + 1];
    const u32x4* sp = reinterpret_cast<const u32x4*>(pool + (size_t)blockIdx.y * pool_layer_elems + (size_t)src * page_elems);
    u32x4*       dp = reinterpret_cast<u32x4*>(pool + (size_t)blockIdx.y * pool_layer_elems + (size_t)dst * page_elems);
    for (size_t i = threadIdx.x; i < page_elems / 8; i += blockDim.x) {
        dp[i] = sp[i];
    }
}
// the sampling kernels' view of `finished`: the rows of beam groups are not theirs (merge = 0: make the view, 1: take the
// sampled rows' new flags back)
__global__ void k_batcher_sampling_view(uint8_t* sfin, uint8_t* fin, const uint8_t* isbeam, int B, int merge)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        if (!merge) {
            sfin[b] = fin[b] | isbeam[b];
        }
        else if (!isbeam[b]) {
            fin[b] = sfin[b];
        }
    }
}
__global__ void k_batcher_set_step(DecodeState* st, int step)
{
    st->step = step;
}
// a group's beams after k_beam_batch: next input token = the word chosen for the beam, one more cached position
__global__ void k_batcher_beam_advance(int* tok, int* len, const int* out_row, int K)
{
    if ((int)threadIdx.x < K) {
        tok[threadIdx.x] = out_row[threadIdx.x];
        len[threadIdx.x] += 1;
    }
}

struct ftcf_batcher {
    struct Request {
        long             id;
        std::vector<int> prompt;
        int              max_new, top_k;
        float            top_p, temperature, repetition_penalty = 1.f;
        uint64_t         seed;
        std::vector<std::vector<int>> stop;  // stop sequences (token ids)
        int              beam_width = 1;     // > 1: a beam group (submit_beam)
        float            diversity = 0.f, len_penalty = 0.f;
        int              min_length = 0;     // beam requests: the end token is held back for this many new tokens
    };
    struct Slot {
        bool             active = false;
        long             id = 0;
        int              len = 0, generated = 0, max_new = 0;
        std::vector<int> pages;
        // the request's own token history (prompt + generated) and stop sequences: the stop criterion
        // (stop_criteria_kernels.cu:24-83: finished AFTER the sequence has been emitted) is the scheduler's, on the host,
        // which sees every token anyway
        std::vector<int>              hist;
        std::vector<std::vector<int>> stop;
        float                         repetition_penalty = 1.f;
        int                           group = -1;  // a beam of the group whose first slot this is (-1: an ordinary sequence)
    };
    // Beam search inside the batcher (round 4; the reference's serving layer runs beam requests through the same decoder:
    // triton_backend/gptneox/GptNeoXTritonModelInstance.cc).  A request of beam width K lives in K CONSECUTIVE slots -- its
    // beams ride in the same batched GEMMs as everybody else's rows -- and is scored by the engine's own beam kernels
    // (launch_beam_search with batch 1: OnlineBeamSearchLayer semantics, kernels_sampling.hip) on per-group state.  The cache
    // indirection of the reference (BaseBeamSearchLayer.cu:30-62: which beam's cache row holds position t) is replaced by what a
    // paged cache is for: after every step beam k's page list becomes a copy of its parent's (pages are reference counted),
    // and only the page the next token will be appended to is copied when it is shared (copy-on-write, all layers, one launch
    // per step); the attention kernel is the ordinary paged one.  The prompt's pages are written once and shared by all beams.
    struct BeamGroup {
        long id = 0;
        int  si = 0, K = 0, n = 0, max_new = 0, generated = 0, budget = 0;
        bool penalised = false;  // repetition_penalty != 1
        bool has_stop = false, has_min = false;
        std::vector<std::vector<int>> lists;  // page list of every beam
    };
    struct BeamResult {
        int                K = 0, total = 0;
        std::vector<int>   ids, lens;
        std::vector<float> cum;
    };
    std::map<int, BeamGroup>   groups;        // by first slot
    std::map<long, BeamResult> beam_results;  // finished beam requests until they are fetched
    std::vector<int>           page_ref;      // references to a page (ordinary sequences: 1)
    ftcf_gptneox* e = nullptr;
    int           max_batch = 0, P = 0, num_pages = 0, max_pages = 0, max_len = 0;
    size_t        pool_layer_elems = 0;
    // device
    f16 *kpool = nullptr, *vpool = nullptr;
    f16 *x = nullptr, *nrm = nullptr, *nrm2 = nullptr, *qkv = nullptr, *ctx = nullptr, *att = nullptr, *mid = nullptr, *ffn = nullptr;
    float*    logits = nullptr;
    float*    gather = nullptr;  // tensor parallel: [TP][max_batch][V / TP] slices of the LM head
    int *     d_pt = nullptr, *d_len = nullptr, *d_tok = nullptr, *d_topk = nullptr, *d_zero = nullptr, *d_prompt = nullptr, *d_plen = nullptr,
        *d_pout = nullptr, *d_pseq = nullptr, *d_pages_tmp = nullptr;
    uint8_t*     d_fin = nullptr;
    // beam groups: the sampling kernels' view of d_fin, which slots are beams, per-group state of the beam kernels
    uint8_t *    d_sfin = nullptr, *d_isbeam = nullptr;
    int *        d_bout = nullptr, *d_bpar = nullptr, *d_bseq = nullptr, *d_bin = nullptr, *d_pairs = nullptr, *d_bres = nullptr,
        *d_bres_len = nullptr;
    float *      d_btemp = nullptr, *d_brep = nullptr, *d_bdiv = nullptr, *d_blen = nullptr;
    int *        d_bsw = nullptr, *d_bmin = nullptr;  // a group's stop words ([2][STOP_LW], the reference's layout) and min_length
    DecodeState* d_bstate = nullptr;
    void*        beam_ws = nullptr;
    float *      d_ptopk = nullptr, *d_ptopp = nullptr, *d_temp = nullptr, *d_cum = nullptr, *d_rep = nullptr;
    int*         d_hist = nullptr;  // [max_len + 1][max_batch] time-major token history of the slots (repetition penalty)
    int *        d_sw = nullptr;    // admission: stop words of the ragged batch, the reference's [n][2][Lw] layout
    uint64_t *   d_seed = nullptr, *d_draws = nullptr;
    DecodeState *d_state = nullptr, *d_gstate = nullptr;
    void*        samp_ws = nullptr;
    float*       smallm_ws = nullptr;
    float*       tiled_ws  = nullptr;  // split-K workspace of the tiled GEMM (decode steps of 17..768 rows)
    // chunked admission: with slots running, a prompt longer than this is prefilled alone, `prefill_chunk` tokens at a time, one
    // decode step of the running slots between two chunks (FTCF_BATCHER_PREFILL_CHUNK; 0 = whole prompts)
    int prefill_chunk = 512;
    // the decode step's layers as ONE launch of the rows kernel (kernels_rows.hip, paged K/V form): tensor_para_size 1, at most 16
    // slots, page_tokens a multiple of the kernel's attention block (FTCF_BATCHER_ROWS=0: the per-GEMM launches)
    RowsPlan      rplan{};
    bool          use_rows = false;
    char*         rows_ws = nullptr;
    PersistLayer* d_rlayers = nullptr;
    float*        d_rot = nullptr;
    int*          d_rstep = nullptr;
    unsigned      rows_steps = 0;
    size_t       smallm_partial = 0, smallm_region = 0;
    unsigned     smallm_seq = 0;
    long         gemm_steps = 0;
    std::vector<void*> owned;
    // host
    std::vector<Slot>   slots;
    std::deque<Request> waiting;
    std::vector<int>    free_pages;
    long                next_id = 1;
    int                 max_prompt = 0;

    template<typename T>
    T* dmalloc(size_t n, bool zero = true)
    {
        void* p = nullptr;
        FTCF_HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        if (zero) {
            FTCF_HIP_CHECK(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
        }
        owned.push_back(p);
        return reinterpret_cast<T*>(p);
    }
    ~ftcf_batcher()
    {
        for (void* p : owned) {
            (void)hipFree(p);
        }
    }

    void init(ftcf_gptneox* eng, int mb, int page_tokens, int pages, int max_seq_len)
    {
        e = eng;
        FTCF_CHECK_ARG(!e->fp32 && e->cfg.use_gptj_residual && (e->dh == 64 || e->dh == 128),
                       "the batcher serves fp16 / int8 engines with parallel residual and size_per_head 64 / 128");
        FTCF_CHECK_ARG(mb >= 1 && mb <= 64 && page_tokens >= 8 && pages >= 1 && max_seq_len >= 2, "bad batcher geometry");
        FTCF_HIP_CHECK(hipSetDevice(e->cfg.device));
        if (const char* c = getenv("FTCF_BATCHER_PREFILL_CHUNK")) {
            prefill_chunk = std::max(0, atoi(c));
        }
        max_batch = mb;
        P         = page_tokens;
        num_pages = pages;
        max_pages = (max_seq_len + P - 1) / P;
        max_len   = max_pages * P;
        max_prompt = max_seq_len - 1;
        FTCF_CHECK_ARG(mmha_paged_smem_bytes(e->dh, max_pages, max_len) <= 64 * 1024, "max_seq_len too large for the paged attention");
        const int    H = e->H, hl = e->hl, il = e->il, L = e->L, V = e->V;
        const size_t B = (size_t)max_batch;
        pool_layer_elems = (size_t)num_pages * e->nhl * P * e->dh;
        kpool = dmalloc<f16>((size_t)L * pool_layer_elems, false);
        vpool = dmalloc<f16>((size_t)L * pool_layer_elems, false);
        x = dmalloc<f16>(B * H);
        nrm = dmalloc<f16>(B * H);
        nrm2 = dmalloc<f16>(B * H);
        qkv = dmalloc<f16>(B * 3 * hl);
        ctx = dmalloc<f16>(B * hl);
        att = dmalloc<f16>(2 * B * H);  // [att | ffn]: one message for a tensor-parallel layer's all-reduce
        ffn = att + B * H;
        mid = dmalloc<f16>(B * il);
        logits = dmalloc<float>(B * V);
        // tensor parallel (every rank runs its own batcher over its shard, fed the same requests in the same order: the
        // schedulers take identical decisions, the decode step's collectives are the engine's): the LM head's [TP][B][V/TP] slices
        gather = e->cfg.tensor_para_size > 1 ? dmalloc<float>(B * V) : nullptr;
        d_pt = dmalloc<int>(B * max_pages);
        d_len = dmalloc<int>(B);
        d_tok = dmalloc<int>(B);
        d_topk = dmalloc<int>(B);
        d_zero = dmalloc<int>(B);
        d_fin = dmalloc<uint8_t>(B);
        d_sfin = dmalloc<uint8_t>(B);
        d_isbeam = dmalloc<uint8_t>(B);
        d_bout = dmalloc<int>(B * (max_len + 2));
        d_bpar = dmalloc<int>(B * (max_len + 2));
        d_bres = dmalloc<int>(B * (max_len + 2));
        d_bres_len = dmalloc<int>(B);
        d_bseq = dmalloc<int>(B);
        d_bin = dmalloc<int>(B);
        d_pairs = dmalloc<int>(2 * B);
        d_btemp = dmalloc<float>(B);
        d_brep = dmalloc<float>(B);
        d_bdiv = dmalloc<float>(B);
        d_blen = dmalloc<float>(B);
        d_bsw = dmalloc<int>(B * 2 * STOP_LW);
        d_bmin = dmalloc<int>(B);
        d_bstate = dmalloc<DecodeState>(B);
        beam_ws = dmalloc<char>(beam_workspace_bytes(1, std::min(max_batch, BEAM_MAX_K)));
        d_ptopk = dmalloc<float>(B);
        d_ptopp = dmalloc<float>(B);
        d_temp = dmalloc<float>(B);
        d_cum = dmalloc<float>(B);
        d_rep = dmalloc<float>(B);
        d_hist = dmalloc<int>((size_t)(max_len + 2) * B);
        d_sw = dmalloc<int>(B * 2 * STOP_LW);
        d_seed = dmalloc<uint64_t>(B);
        d_draws = dmalloc<uint64_t>(B);
        d_state = dmalloc<DecodeState>(1);
        d_gstate = dmalloc<DecodeState>(1);
        d_prompt = dmalloc<int>(B * max_seq_len);
        d_plen = dmalloc<int>(B);
        d_pout = dmalloc<int>(B * (max_seq_len + 1));
        d_pseq = dmalloc<int>(B);
        d_pages_tmp = dmalloc<int>(max_pages);
        samp_ws = dmalloc<char>(sampling_workspace_bytes(max_batch, V), false);
        {
            const int rows_env = getenv("FTCF_BATCHER_ROWS") ? atoi(getenv("FTCF_BATCHER_ROWS")) : 1;
            if (rows_env && e->rows && e->cfg.tensor_para_size == 1 && !e->fp32 && max_batch <= 16 && e->cfg.use_gptj_residual && L <= 255
                && (e->dh == 64 || e->dh == 128) && P % rows_paged_block(e->dh) == 0) {
                rplan = rows_plan(max_batch, H, hl, il, e->nhl, e->dh, 1, e->int8, e->num_cu, 0);
                if (rplan.ok && rows_resident(rplan, e->int8, e->dh, e->num_cu, true)) {
                    rows_ws   = dmalloc<char>(rows_workspace_bytes(rplan, max_batch, H, hl, il, e->nhl, e->dh));
                    d_rlayers = dmalloc<PersistLayer>(L);
                    d_rot     = dmalloc<float>(B * std::max(1, e->cfg.rotary_embedding_dim));
                    d_rstep   = dmalloc<int>(1);
                    std::vector<PersistLayer> pl(L);
                    for (int l = 0; l < L; l++) {
                        const LayerWeights& w = e->layers[l];
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
                        r.k_cache = kpool + (size_t)l * pool_layer_elems;
                        r.v_cache = vpool + (size_t)l * pool_layer_elems;
                    }
                    FTCF_HIP_CHECK(hipMemcpy(d_rlayers, pl.data(), sizeof(PersistLayer) * L, hipMemcpyHostToDevice));
                    use_rows = true;
                }
            }
        }
        // (tensor parallel: up to 32 slots as two micro-batches of <= 16 rows with their own split-K regions -- the decode overlap)
        if (max_batch > 4 && (max_batch <= e->SMALLM_MAX_ROWS || (e->cfg.tensor_para_size > 1 && max_batch <= 32))) {
            const bool i8 = e->int8;
            const int  bc = std::min(max_batch, 16);  // 16 rows per launch
            smallm_partial = gemm_smallm_workspace_bytes(bc, 3 * hl, H, i8) + gemm_smallm_workspace_bytes(bc, il, H, i8)
                             + gemm_smallm_workspace_bytes(bc, H, hl, i8) + gemm_smallm_workspace_bytes(bc, H, il, i8);
            smallm_region  = smallm_partial;
            smallm_partial *= e->cfg.tensor_para_size > 1 ? 2 : 1;
            smallm_ws = dmalloc<float>((smallm_partial + gemm_smallm_ticket_bytes()) / 4 + 1);
        }
        if (max_batch > 16 && max_batch <= gemm_tiled_splitk_max_m()) {
            tiled_ws = dmalloc<float>(gemm_tiled_workspace_bytes() / 4);
        }
        std::vector<uint8_t> fin(max_batch, 1);
        FTCF_HIP_CHECK(hipMemcpy(d_fin, fin.data(), max_batch, hipMemcpyHostToDevice));
        slots.assign(max_batch, Slot{});
        free_pages.resize(num_pages);
        for (int i = 0; i < num_pages; i++) {
            free_pages[i] = num_pages - 1 - i;
        }
        page_ref.assign(num_pages, 0);
    }
    int take_page()
    {
        if (free_pages.empty()) {
            throw Error(-2, "batcher: the page pool is exhausted (a reservation was wrong)");
        }
        const int pg = free_pages.back();
        free_pages.pop_back();
        page_ref[pg] = 1;
        return pg;
    }
    void drop_page(const int pg)
    {
        if (--page_ref[pg] == 0) {
            free_pages.push_back(pg);
        }
    }
    // pages promised to running beam groups but not yet taken (a group takes its pages as its beams diverge: at most
    // budget = K * pages of one full-length sequence)
    int reserved_pages() const
    {
        int r = 0;
        for (const auto& kv : groups) {
            std::set<int> held;
            for (const auto& l : kv.second.lists) {
                held.insert(l.begin(), l.end());
            }
            r += std::max(0, kv.second.budget - (int)held.size());
        }
        return r;
    }

    static constexpr int STOP_LW = 64;  // total stop-word tokens per request (the [2][Lw] word list of the admission)
    long submit(const int* ids, int n, int max_new, int top_k, float top_p, float temperature, uint64_t seed,
                float repetition_penalty = 1.f, const int* stop_words = nullptr, int stop_len = 0)
    {
        FTCF_CHECK_ARG(repetition_penalty > 0.f, "repetition_penalty must be positive");
        FTCF_CHECK_ARG(stop_len >= 0 && stop_len <= STOP_LW && (stop_len == 0 || stop_words), "bad stop word list");
        // (the decode step stages total_len = max_len + 2 history entries: the same bound as launch_dynamic_decode's)
        FTCF_CHECK_ARG(((size_t)max_len + 2) * 8 <= 60 * 1024 || repetition_penalty == 1.f,
                       "max_seq_len too large for the repetition-penalty staging buffer");
        FTCF_CHECK_ARG(ids && n >= 1 && max_new >= 1, "empty prompt or max_new_tokens < 1");
        FTCF_CHECK_ARG(n + max_new <= max_len && n <= max_prompt, "prompt + max_new_tokens exceed the batcher's max_seq_len");
        FTCF_CHECK_ARG((n + max_new + P - 1) / P <= num_pages, "the request needs more pages than the pool has");
        FTCF_CHECK_ARG(top_k >= 0 && top_k <= 1024 && top_p >= 0.f && top_p <= 1.f && temperature > 0.f, "bad sampling parameters");
        for (int i = 0; i < n; i++) {
            FTCF_CHECK_ARG(ids[i] >= 0 && ids[i] < e->V, "token id out of range");
        }
        Request r;
        r.id = next_id++;
        r.prompt.assign(ids, ids + n);
        r.max_new = max_new;
        r.top_k = top_k;
        r.top_p = top_p;
        r.temperature = temperature;
        r.seed = seed;
        r.repetition_penalty = repetition_penalty;
        // to_word_list_format (codefuse_example.py:26-53): [0][..] flat ids, [1][..] cumulative end offsets, -1 padded
        for (int i = 0, start = 0; i < stop_len; i++) {
            const int end = stop_words[stop_len + i];
            if (end < 0) {
                break;
            }
            FTCF_CHECK_ARG(end > start && end <= stop_len, "bad stop word offsets");
            r.stop.emplace_back(stop_words + start, stop_words + end);
            start = end;
        }
        waiting.push_back(std::move(r));
        return waiting.back().id;
    }

    // a beam-search request (GptNeoXOp.forward with beam_width > 1: OnlineBeamSearchLayer): K consecutive slots; one event when
    // it has finished (token = -1), the K hypotheses through beam_result()
    long submit_beam(const int* ids, int n, int max_new, int beam_width, float diversity, float len_penalty, float temperature,
                     float repetition_penalty, int min_length = 0, const int* stop_words = nullptr, int stop_len = 0)
    {
        FTCF_CHECK_ARG(min_length >= 0, "min_length must not be negative");
        FTCF_CHECK_ARG(stop_len >= 0 && stop_len <= STOP_LW && (stop_len == 0 || stop_words), "bad stop word list");
        FTCF_CHECK_ARG(beam_width >= 2 && beam_width <= BEAM_MAX_K && beam_width <= max_batch,
                       "beam_width must be in [2, 64] and fit the batcher's slots");
        FTCF_CHECK_ARG(repetition_penalty > 0.f && temperature > 0.f, "repetition_penalty and temperature must be positive");
        FTCF_CHECK_ARG(((size_t)max_len + 2) * 8 <= 60 * 1024 || repetition_penalty == 1.f,
                       "max_seq_len too large for the repetition-penalty staging buffer");
        FTCF_CHECK_ARG(ids && n >= 1 && max_new >= 1, "empty prompt or max_new_tokens < 1");
        FTCF_CHECK_ARG(n + max_new <= max_len && n <= max_prompt, "prompt + max_new_tokens exceed the batcher's max_seq_len");
        FTCF_CHECK_ARG(beam_width * ((n + max_new + P - 1) / P) <= num_pages, "the request needs more pages than the pool has");
        for (int i = 0; i < n; i++) {
            FTCF_CHECK_ARG(ids[i] >= 0 && ids[i] < e->V, "token id out of range");
        }
        Request r;
        r.id = next_id++;
        r.prompt.assign(ids, ids + n);
        r.max_new = max_new;
        r.top_k = 1;
        r.top_p = 0.f;
        r.temperature = temperature;
        r.seed = 0;
        r.repetition_penalty = repetition_penalty;
        r.beam_width = beam_width;
        r.diversity = diversity;
        r.len_penalty = len_penalty;
        r.min_length = min_length;
        for (int i = 0, start = 0; i < stop_len; i++) {  // to_word_list_format (codefuse_example.py:26-53), as submit()
            const int end = stop_words[stop_len + i];
            if (end < 0) {
                break;
            }
            FTCF_CHECK_ARG(end > start && end <= stop_len, "bad stop word offsets");
            r.stop.emplace_back(stop_words + start, stop_words + end);
            start = end;
        }
        waiting.push_back(std::move(r));
        return waiting.back().id;
    }

    struct Event {
        long id;
        int  token, finished;
    };
    std::vector<Event>* hook_ev = nullptr;  // where the decode steps inside a chunked admission put their events
    std::deque<Event>   outbox;             // events of an iteration that did not fit the caller's arrays
    ftcf_token_callback_fn on_token = nullptr;  // called for every event the moment it exists (inside step())
    void*                  on_token_user = nullptr;
    void emit(std::vector<Event>& ev, const Event& x)
    {
        ev.push_back(x);
        if (on_token) {
            on_token(on_token_user, x.id, x.token, x.finished);
        }
    }

    // the first tokens of an admission reach the callback only when the admission is known to have succeeded: a failed one is
    // rolled back and retried, and a streaming consumer must not see its first tokens twice
    void fire(const std::vector<Event>& ev, const size_t from)
    {
        if (on_token) {
            for (size_t i = from; i < ev.size(); i++) {
                on_token(on_token_user, ev[i].id, ev[i].token, ev[i].finished);
            }
        }
    }

    // stop_criteria_kernels.cu:24-83: does the history END with one of the request's stop sequences?
    static bool hits_stop_word(const Slot& s)
    {
        for (const auto& wd : s.stop) {
            if (!wd.empty() && s.hist.size() >= wd.size() && std::equal(wd.begin(), wd.end(), s.hist.end() - wd.size())) {
                return true;
            }
        }
        return false;
    }
    void release(Slot& s)
    {
        for (int pg : s.pages) {
            drop_page(pg);
        }
        s.pages.clear();
        s.active = false;
    }

    // prompts -> ONE ragged batch through the engine's context path (+ first tokens) -> pages of their slots
    void admit(const std::vector<int>& sis, const std::vector<Request>& rs, std::vector<Event>& ev)
    {
        Range        rg("ftcf.batcher.admit");
        hipStream_t  st = e->stream;
        const int    n  = (int)sis.size();
        int          S  = 0;
        for (const Request& r : rs) {
            S = std::max(S, (int)r.prompt.size());
        }
        std::vector<int>      ids((size_t)n * S, e->cfg.end_id), lens(n), topk(n);
        std::vector<float>    topp(n), temp(n), rep(n);
        bool                  any_stop = false;
        std::vector<int>      sw((size_t)n * 2 * STOP_LW, 0);
        std::vector<uint64_t> seed(n);
        for (int i = 0; i < n; i++) {
            const Request& r = rs[i];
            std::copy(r.prompt.begin(), r.prompt.end(), ids.begin() + (size_t)i * S);
            lens[i] = (int)r.prompt.size();
            topk[i] = r.top_k;
            topp[i] = r.top_p;
            temp[i] = r.temperature;
            rep[i]  = r.repetition_penalty;
            seed[i] = r.seed;
            {  // the request's stop sequences in the reference's word-list layout (the engine samples the FIRST token)
                int* ids_row = sw.data() + (size_t)i * 2 * STOP_LW;
                int* off_row = ids_row + STOP_LW;
                std::fill(off_row, off_row + STOP_LW, -1);
                int pos = 0, k = 0;
                for (const auto& wd : r.stop) {
                    std::copy(wd.begin(), wd.end(), ids_row + pos);
                    pos += (int)wd.size();
                    off_row[k++] = pos;
                    any_stop = true;
                }
            }
            Slot&     s = slots[sis[i]];
            const int need = (lens[i] + r.max_new + P - 1) / P;
            s.pages.clear();
            for (int k = 0; k < need; k++) {
                s.pages.push_back(take_page());
            }
        }
        FTCF_HIP_CHECK(hipMemcpy(d_prompt, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
        FTCF_HIP_CHECK(hipMemcpy(d_plen, lens.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        ftcf_forward_args a{};
        a.input_ids = d_prompt;
        a.input_lengths = d_plen;
        a.batch_size = n;
        a.max_input_len = S;
        a.output_len = 1;
        a.beam_width = 1;
        a.top_k = topk.data();
        a.n_top_k = n;
        a.top_p = topp.data();
        a.n_top_p = n;
        a.temperature = temp.data();
        a.n_temperature = n;
        a.repetition_penalty = rep.data();
        a.n_repetition_penalty = n;
        if (any_stop) {
            FTCF_HIP_CHECK(hipMemcpy(d_sw, sw.data(), sw.size() * 4, hipMemcpyHostToDevice));
            a.stop_words_list = d_sw;
            a.stop_words_len = STOP_LW;
        }
        a.random_seed = seed.data();
        a.n_random_seed = n;
        a.output_ids = d_pout;
        a.sequence_lengths = d_pseq;
        e->forward(a);  // host synchronous: K/V of row i, positions [0, len_i), are in the engine's cache [L][n][nh][S + 1][dh]
        std::vector<int> out((size_t)n * (S + 1));
        FTCF_HIP_CHECK(hipMemcpy(out.data(), d_pout, out.size() * 4, hipMemcpyDeviceToHost));
        const size_t row_kv = (size_t)e->nhl * (S + 1) * e->dh;  // one row of one layer of the engine's cache
        for (int i = 0; i < n; i++) {
            const Request& r  = rs[i];
            const int      si = sis[i], len = lens[i];
            Slot&          s  = slots[si];
            // the engine's output rows are compacted (prompt, then the generated tokens: invokeGatherTree removes the padding)
            const int        first = out[(size_t)i * (S + 1) + len];
            std::vector<int> row(max_pages, 0);
            std::copy(s.pages.begin(), s.pages.end(), row.begin());
            FTCF_HIP_CHECK(hipMemcpyAsync(d_pt + (size_t)si * max_pages, row.data(), (size_t)max_pages * 4, hipMemcpyHostToDevice, st));
            launch_scatter_kv_to_pages(e->k_cache + (size_t)i * row_kv, e->v_cache + (size_t)i * row_kv, kpool, vpool,
                                       d_pt + (size_t)si * max_pages, e->L, e->nhl, e->dh, S + 1, len, P, pool_layer_elems, st,
                                       (size_t)n * row_kv);
            // per-slot state of the decode steps
            const int      keff = (r.top_k == 0 && r.top_p == 0.f) ? 1 : std::min(r.top_k, 1024);  // BaseSamplingLayer: (0, 0) = greedy
            const float    ptk = (r.top_p == 0.f) ? 1.f : r.top_p;
            const uint8_t  zero8 = 0;
            const uint64_t one = 1;
            const float    zf = 0.f;
            FTCF_HIP_CHECK(hipMemcpyAsync(d_len + si, &len, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_tok + si, &first, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_topk + si, &keff, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_ptopk + si, &ptk, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_ptopp + si, &r.top_p, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_temp + si, &r.temperature, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_rep + si, &r.repetition_penalty, 4, hipMemcpyHostToDevice, st));
            // the slot's token history, time-major column si: prompt, then the first token
            s.hist.assign(r.prompt.begin(), r.prompt.end());
            s.hist.push_back(first);
            s.stop = r.stop;
            s.repetition_penalty = r.repetition_penalty;
            FTCF_HIP_CHECK(hipMemcpy2DAsync(d_hist + si, (size_t)max_batch * 4, s.hist.data(), 4, 4, s.hist.size(),
                                            hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_seed + si, &r.seed, 8, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_draws + si, &one, 8, hipMemcpyHostToDevice, st));  // draw 0 went to the first token
            FTCF_HIP_CHECK(hipMemcpyAsync(d_cum + si, &zf, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_fin + si, &zero8, 1, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipStreamSynchronize(st));  // the host temporaries above die here
            s.active = true;
            s.id = r.id;
            s.len = len;
            s.generated = 1;
            s.max_new = r.max_new;
            const int done = (first == e->cfg.end_id || s.generated >= s.max_new || hits_stop_word(s)) ? 1 : 0;
            ev.push_back(Event{r.id, first, done});  // (the token callback fires when the admission has succeeded: step())
            if (done) {
                const uint8_t one8 = 1;
                FTCF_HIP_CHECK(hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice));
                release(s);
            }
        }
    }

    // ---- beam groups ------------------------------------------------------------------------------------------------
    void upload_group_tables(const BeamGroup& g, hipStream_t st)
    {
        std::vector<int> rows((size_t)g.K * max_pages, 0);
        for (int k = 0; k < g.K; k++) {
            std::copy(g.lists[k].begin(), g.lists[k].end(), rows.begin() + (size_t)k * max_pages);
        }
        FTCF_HIP_CHECK(hipMemcpyAsync(d_pt + (size_t)g.si * max_pages, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipStreamSynchronize(st));  // (the host temporary dies here)
    }
    // every unfinished beam gets a page of its OWN for position `pos` (the next append): a fresh one on a page boundary, a copy
    // of the shared one otherwise
    void prepare_append(BeamGroup& g, const int pos, const std::vector<uint8_t>& fin, hipStream_t st)
    {
        std::vector<int> pairs;
        for (int k = 0; k < g.K; k++) {
            if (fin[k]) {
                continue;  // (a finished beam appends nothing: it keeps what it shares)
            }
            std::vector<int>& l = g.lists[k];
            if (pos / P >= (int)l.size()) {
                l.push_back(take_page());
            }
            else if (page_ref[l[pos / P]] > 1) {
                const int old = l[pos / P], pg = take_page();
                pairs.push_back(old);
                pairs.push_back(pg);
                drop_page(old);
                l[pos / P] = pg;
            }
        }
        if (!pairs.empty()) {
            FTCF_HIP_CHECK(hipMemcpyAsync(d_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice, st));
            const size_t page_elems = (size_t)e->nhl * P * e->dh;
            hipLaunchKernelGGL(k_batcher_copy_pages, dim3((unsigned)pairs.size() / 2, e->L, 2), dim3(256), 0, st, kpool, vpool, d_pairs,
                               page_elems, pool_layer_elems);
            FTCF_HIP_CHECK(hipGetLastError());
            FTCF_HIP_CHECK(hipStreamSynchronize(st));
        }
        upload_group_tables(g, st);
    }
    void release_group(const int si)
    {
        auto it = groups.find(si);
        if (it == groups.end()) {
            return;
        }
        BeamGroup&           g = it->second;
        std::vector<uint8_t> ones(g.K, 1), zeros(g.K, 0);
        for (int k = 0; k < g.K; k++) {
            for (int pg : g.lists[k]) {
                drop_page(pg);
            }
            slots[si + k].active = false;
            slots[si + k].group  = -1;
        }
        (void)hipMemcpy(d_fin + si, ones.data(), g.K, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_isbeam + si, zeros.data(), g.K, hipMemcpyHostToDevice);
        groups.erase(it);
    }
    // gatherTree over the group's steps (decoding_kernels.cu:452-583, as GptNeoXOp.forward returns its beams: [K][n + max_new]
    // ids padded with end_id, lengths, cum_log_probs), the result parked until it is fetched, the slots and pages freed
    // fire_now = false (from inside an admission): the event is only recorded, step() fires it once the admission has succeeded
    void finish_group(const int si, std::vector<Event>& ev, const bool fire_now = true)
    {
        BeamGroup&  g = groups.at(si);
        hipStream_t st = e->stream;
        const int   total = g.n + g.max_new;
        launch_gather_tree_beam(d_bres + (size_t)si * (max_len + 2), d_bres_len + si, d_bout + (size_t)si * (max_len + 2),
                                d_bpar + (size_t)si * (max_len + 2), d_bseq + si, d_bin + si, 1, g.K, g.n, total, e->cfg.end_id, st);
        BeamResult r;
        r.K     = g.K;
        r.total = total;
        r.ids.resize((size_t)g.K * total);
        r.lens.resize(g.K);
        r.cum.resize(g.K);
        FTCF_HIP_CHECK(hipMemcpyAsync(r.ids.data(), d_bres + (size_t)si * (max_len + 2), r.ids.size() * 4, hipMemcpyDeviceToHost, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(r.lens.data(), d_bres_len + si, (size_t)g.K * 4, hipMemcpyDeviceToHost, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(r.cum.data(), d_cum + si, (size_t)g.K * 4, hipMemcpyDeviceToHost, st));
        FTCF_HIP_CHECK(hipStreamSynchronize(st));
        const long id    = g.id;
        beam_results[id] = std::move(r);
        while (beam_results.size() > 256) {  // (results nobody fetches do not pile up: the oldest ids go first)
            FT_LOG_WARNING(e->cfg.device, "batcher: the result of beam request %ld was never fetched and is dropped (256 wait at most)",
                           beam_results.begin()->first);
            beam_results.erase(beam_results.begin());
        }
        release_group(si);
        if (fire_now) {
            emit(ev, Event{id, -1, 1});
        }
        else {
            ev.push_back(Event{id, -1, 1});
        }
    }
    // prompt -> the engine's own beam-search request of ONE step (prefill of the K tiled rows, first beam step) -> the group's
    // state; the prompt's K/V are scattered once (beam 0's rows) into pages all beams share
    void admit_beam(const int si, const Request& r, std::vector<Event>& ev)
    {
        Range       rg("ftcf.batcher.admit_beam");
        hipStream_t st = e->stream;
        const int   K = r.beam_width, n = (int)r.prompt.size(), need = (n + r.max_new + P - 1) / P;
        BeamGroup   g;
        g.id      = r.id;
        g.si      = si;
        g.K       = K;
        g.n       = n;
        g.max_new = r.max_new;
        g.budget  = K * need;
        g.penalised = r.repetition_penalty != 1.f;
        FTCF_HIP_CHECK(hipMemcpy(d_prompt, r.prompt.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        FTCF_HIP_CHECK(hipMemcpy(d_plen, &n, 4, hipMemcpyHostToDevice));
        ftcf_forward_args a{};
        a.input_ids = d_prompt;
        a.input_lengths = d_plen;
        a.batch_size = 1;
        a.max_input_len = n;
        a.output_len = 1;
        a.beam_width = K;
        a.temperature = &r.temperature;
        a.n_temperature = 1;
        a.repetition_penalty = &r.repetition_penalty;
        a.n_repetition_penalty = 1;
        a.beam_search_diversity_rate = &r.diversity;
        a.n_beam_search_diversity_rate = 1;
        a.len_penalty = &r.len_penalty;
        a.n_len_penalty = 1;
        a.output_ids = d_pout;
        a.sequence_lengths = d_pseq;
        g.has_min  = r.min_length > 0;
        g.has_stop = !r.stop.empty();
        if (g.has_min) {
            a.min_length = &r.min_length;
            a.n_min_length = 1;
            FTCF_HIP_CHECK(hipMemcpy(d_bmin + si, &r.min_length, 4, hipMemcpyHostToDevice));
        }
        if (g.has_stop) {  // the group's stop words stay on the device for its beam steps (checked along the parent chain there)
            std::vector<int> sw(2 * STOP_LW, 0);
            std::fill(sw.begin() + STOP_LW, sw.end(), -1);
            int pos = 0, k = 0;
            for (const auto& wd : r.stop) {
                std::copy(wd.begin(), wd.end(), sw.begin() + pos);
                pos += (int)wd.size();
                sw[STOP_LW + k++] = pos;
            }
            FTCF_HIP_CHECK(hipMemcpy(d_bsw + (size_t)si * 2 * STOP_LW, sw.data(), sw.size() * 4, hipMemcpyHostToDevice));
            a.stop_words_list = d_bsw + (size_t)si * 2 * STOP_LW;
            a.stop_words_len = STOP_LW;
        }
        e->forward(a);  // host synchronous; the engine's buffers keep the request's state: step_ids / parent_ids [n + 1][K], ...
        // the prompt's pages, shared by every beam
        std::vector<int> shared;
        try {
            for (int i = 0; i < (n + P - 1) / P; i++) {
                shared.push_back(take_page());
            }
        }
        catch (...) {  // (no group owns them yet: step()'s release_group(si) would not find them)
            for (int pg : shared) {
                page_ref[pg] = 1;
                drop_page(pg);
            }
            throw;
        }
        g.lists.assign(K, shared);
        for (int pg : shared) {
            page_ref[pg] = K;
        }
        groups[si] = g;
        BeamGroup& G = groups[si];
        upload_group_tables(G, st);
        const size_t row_kv = (size_t)e->nhl * (n + 1) * e->dh;
        launch_scatter_kv_to_pages(e->k_cache, e->v_cache, kpool, vpool, d_pt + (size_t)si * max_pages, e->L, e->nhl, e->dh, n + 1, n, P,
                                   pool_layer_elems, st, (size_t)K * row_kv);
        const size_t reg = (size_t)si * (max_len + 2);
        FTCF_HIP_CHECK(hipMemcpyAsync(d_bout + reg, e->step_ids, (size_t)(n + 1) * K * 4, hipMemcpyDeviceToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_bpar + reg, e->parent_ids, (size_t)(n + 1) * K * 4, hipMemcpyDeviceToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_bseq + si, e->seq_len, (size_t)K * 4, hipMemcpyDeviceToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_cum + si, e->cum, (size_t)K * 4, hipMemcpyDeviceToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_tok + si, e->step_ids + (size_t)n * K, (size_t)K * 4, hipMemcpyDeviceToDevice, st));
        // (the engine's own `finished` is all ones by now: its one-token request ran into the length criterion)
        std::vector<int>     lens(K, n), first(K);
        std::vector<uint8_t> ones(K, 1), fin(K, 0);
        FTCF_HIP_CHECK(hipMemcpy(first.data(), e->step_ids + (size_t)n * K, (size_t)K * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < K; k++) {
            fin[k] = first[k] == e->cfg.end_id ? 1 : 0;
            for (const auto& wd : r.stop) {  // (stop_criteria_kernels.cu:24-83 on prompt + first token: every beam's parent is the prompt)
                if (!wd.empty() && (int)wd.size() <= n + 1 && wd.back() == first[k]
                    && std::equal(wd.begin(), wd.end() - 1, r.prompt.end() - (wd.size() - 1))) {
                    fin[k] = 1;
                }
            }
        }
        FTCF_HIP_CHECK(hipMemcpyAsync(d_fin + si, fin.data(), (size_t)K, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_len + si, lens.data(), (size_t)K * 4, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_bin + si, lens.data(), (size_t)K * 4, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_isbeam + si, ones.data(), (size_t)K, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_btemp + si, &r.temperature, 4, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_brep + si, &r.repetition_penalty, 4, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_bdiv + si, &r.diversity, 4, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_blen + si, &r.len_penalty, 4, hipMemcpyHostToDevice, st));
        FTCF_HIP_CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < K; k++) {
            Slot& s  = slots[si + k];
            s.active = true;
            s.group  = si;
            s.id     = r.id;
            s.len    = n;
            s.pages.clear();
            s.hist.clear();
            s.stop.clear();
            s.repetition_penalty = 1.f;  // (the beam kernels apply the request's penalty themselves)
        }
        G.generated = 1;
        bool all_fin = true;
        for (int k = 0; k < K; k++) {
            all_fin &= fin[k] != 0;
        }
        if (all_fin || G.generated >= G.max_new) {
            finish_group(si, ev, false);  // (step() fires it when the admission has succeeded)
            return;
        }
        prepare_append(G, n, fin, st);
    }
    // the beam kernels of every running group on this step's logits, BEFORE the sampling kernels (whose end-id mask rewrites the
    // logits of rows that are finished in their view -- the groups' rows)
    void enqueue_beam_steps(hipStream_t st)
    {
        for (auto& kv : groups) {
            BeamGroup&   g   = kv.second;
            const int    si  = g.si, step = g.n + g.generated;
            const size_t reg = (size_t)si * (max_len + 2);
            hipLaunchKernelGGL(k_batcher_set_step, dim3(1), dim3(1), 0, st, d_bstate + si, step);
            BeamParams bp{};
            bp.logits = logits + (size_t)si * e->V;
            bp.B = 1;
            bp.K = g.K;
            bp.V = e->V;
            bp.max_input_len = g.n;
            bp.total_len = max_len + 2;
            bp.end_id = e->cfg.end_id;
            bp.s_max = 0;  // (no cache indirection: the page lists are reordered instead)
            bp.input_lengths = d_bin + si;
            bp.temperature = d_btemp + si;
            bp.repetition_penalty = g.penalised ? d_brep + si : nullptr;  // (NULL: no history staging buffer in LDS)
            bp.diversity_rate = d_bdiv + si;
            bp.len_penalty = d_blen + si;
            bp.min_length = g.has_min ? d_bmin + si : nullptr;
            bp.stop_words = g.has_stop ? d_bsw + (size_t)si * 2 * STOP_LW : nullptr;
            bp.stop_len = STOP_LW;
            bp.output_ids = d_bout + reg;
            bp.parent_ids = d_bpar + reg;
            bp.finished = d_fin + si;
            bp.seq_len = d_bseq + si;
            bp.cum_log_probs = d_cum + si;
            bp.cache_indir = d_zero;
            bp.state = d_bstate + si;
            bp.ws = beam_ws;
            launch_beam_search(bp, st);
        }
    }
    // ... and behind them (k_batcher_last_token writes every slot's input token): the groups' next input tokens and positions
    void enqueue_beam_advance(hipStream_t st)
    {
        for (auto& kv : groups) {
            const BeamGroup& g    = kv.second;
            const int        step = g.n + g.generated;
            hipLaunchKernelGGL(k_batcher_beam_advance, dim3(1), dim3(64), 0, st, d_tok + g.si, d_len + g.si,
                               d_bout + (size_t)g.si * (max_len + 2) + (size_t)step * g.K, g.K);
        }
    }
    // after the step's synchronisation: every beam takes over its parent's pages, finished groups leave
    void advance_groups(std::vector<Event>& ev, const std::vector<uint8_t>& fin_all)
    {
        hipStream_t      st = e->stream;
        std::vector<int> leaders;
        for (auto& kv : groups) {
            leaders.push_back(kv.first);
        }
        for (const int si : leaders) {
            BeamGroup&       g = groups.at(si);
            const int        step = g.n + g.generated;
            std::vector<int> parent(g.K);
            FTCF_HIP_CHECK(hipMemcpy(parent.data(), d_bpar + (size_t)si * (max_len + 2) + (size_t)step * g.K, (size_t)g.K * 4,
                                     hipMemcpyDeviceToHost));
            std::vector<std::vector<int>> nl(g.K);
            for (int k = 0; k < g.K; k++) {
                nl[k] = g.lists[parent[k] % g.K];
                for (int pg : nl[k]) {
                    page_ref[pg]++;
                }
            }
            for (int k = 0; k < g.K; k++) {
                for (int pg : g.lists[k]) {
                    drop_page(pg);
                }
            }
            g.lists = std::move(nl);
            g.generated += 1;
            std::vector<uint8_t> fin(fin_all.begin() + si, fin_all.begin() + si + g.K);
            bool                 all_fin = true;
            for (int k = 0; k < g.K; k++) {
                all_fin &= fin[k] != 0;
                slots[si + k].len += 1;
            }
            if (all_fin || g.generated >= g.max_new) {
                finish_group(si, ev);
            }
            else {
                prepare_append(g, g.n + g.generated - 1, fin, st);
            }
        }
    }

    // Which form the next decode step's layers take under tensor parallelism.  FTCF_DECODE_OVERLAP (read when a request begins: an admission)
    // = 1 switches the overlapped form on, the default is off; "auto" (ranks joined by RCCL): decode steps 4..19 of this batcher run plain, 20..35 overlapped, each timed from
    // enqueue to the step's stream synchronisation on the host; after step 35 every rank learns the slowest rank's two sums
    // (comm_max; the ranks' batchers run the same steps) and the batcher stays with the faster form.
    long   ov_steps = 0;
    double ov_us[2] = {0.0, 0.0};
    int    ov_choice = -1;  // -1 undecided, 0 plain, 1 overlapped
    bool   ov_auto = false, ov_last = false;
    bool decode_overlap_now(int B, bool dual)
    {
        ftcf_gptneox* g  = e;
        ov_auto = g->decode_overlap_mode == 2 && g->cfg.tensor_para_size > 1 && g->cfg.comm && g->cfg.comm->comm && !g->cfg.comm->local
                  && !g->cfg.comm->hx && g->cfg.comm->world > 1;
        const int env = g->decode_overlap_mode == 1 ? 1 : 0;
        ov_last = false;
        if (g->cfg.tensor_para_size == 1 || !dual || !smallm_ws || smallm_partial < 2 * smallm_region || !g->side || B < 4 || B > 32
            || (!env && !ov_auto)) {
            ov_auto = false;
            return false;
        }
        if (ov_auto) {
            ov_last = ov_choice >= 0 ? ov_choice == 1 : (ov_steps >= 20 && ov_steps < 36);
        }
        else {
            ov_last = true;
        }
        return ov_last;
    }
    void decode_overlap_trial(double us, hipStream_t st)
    {
        if (!ov_auto || ov_choice >= 0) {
            return;
        }
        if (ov_steps >= 4 && ov_steps < 36) {
            ov_us[ov_steps >= 20 ? 1 : 0] += us;
        }
        ov_steps++;
        if (ov_steps == 36 && e->tp_scratch) {
            const int a = comm_max(e->cfg.comm, (int)ov_us[0], st, e->tp_scratch), b = comm_max(e->cfg.comm, (int)ov_us[1], st, e->tp_scratch);
            ov_choice = b < a ? 1 : 0;
            e->stats.decode_step_ms_plain      = a * 1e-3f / 16.f;
            e->stats.decode_step_ms_overlapped = b * 1e-3f / 16.f;
            FT_LOG_INFO(e->cfg.device, "batcher: decode all-reduce overlap (auto): plain %.3f ms per step, overlapped %.3f -> %s", a * 1e-3 / 16,
                        b * 1e-3 / 16, ov_choice ? "overlapped from now on" : "plain from now on");
        }
    }

    // one token for every running slot
    void decode(std::vector<Event>& ev)
    {
        Range                  rg("ftcf.batcher.decode");
        hipStream_t            st = e->stream;
        const int              B = max_batch, H = e->H, hl = e->hl, il = e->il, L = e->L, V = e->V;
        const bool             int8 = e->int8;
        const bool             dual = residual_dual_ln_supported(H);
        const int              tp = e->cfg.tensor_para_size;
        const bool             tp1 = tp == 1;  // (tensor parallel: the layer ends with residual + all-reduce, GptNeoXDecoder.cc:357-359)
        e->stats.decode_path = use_rows ? 3 : 2;  // (ftcf_gptneox_get_stats of the borrowed engine: which layers the last decode step ran)
        hipLaunchKernelGGL(k_batcher_embed, dim3(B), dim3(256), 0, st, x, e->wte, d_tok, H);
        if ((gemm_steps++ & 0x3ffff) == 0 && smallm_ws) {  // the tag space of the burst GEMMs wraps: start it clean
            FTCF_HIP_CHECK(hipMemsetAsync(smallm_ws, 0, smallm_partial + gemm_smallm_ticket_bytes(), st));
        }
        hipLaunchKernelGGL(k_batcher_tick, dim3(1), dim3(1), 0, st, d_gstate);
        if (use_rows) {
            // every layer of the step in one launch (rows_device.hip.h, paged K/V): x -> x
            if ((rows_steps++ & 0xfffff) == 0) {  // (the tag space: start it clean now and then)
                FTCF_HIP_CHECK(hipMemsetAsync(rows_ws, 0, rows_flag_bytes(rplan, B, e->nhl), st));
                FTCF_HIP_CHECK(hipMemsetAsync(d_rstep, 0, sizeof(int), st));
            }
            hipLaunchKernelGGL(k_batcher_rows_prep, dim3(B), dim3(64), 0, st, d_rstep, d_rot, d_len, e->cfg.rotary_embedding_dim);
            RowsParams rp{};
            rp.layers = d_rlayers;
            rp.L = L;
            rp.l_begin = 0;
            rp.l_end = L;
            rp.x_in = x;
            rp.x_out = x;
            rp.M = B;
            rp.H = H;
            rp.Hl = hl;
            rp.Il = il;
            rp.nh = e->nhl;
            rp.dh = e->dh;
            rp.rot = e->cfg.rotary_embedding_dim;
            rp.s_max = max_len;
            rp.tp = 1;
            rp.plan = rplan;
            rows_carve(rp, rows_ws);
            rp.d_step = d_rstep;
            rp.d_stop = nullptr;
            rp.seq_len = d_len;
            rp.input_lengths = nullptr;
            rp.max_input_len = 0;
            rp.finished = d_fin;
            rp.rot_table = d_rot;
            rp.eps = 1e-5f;
            rp.page_table = d_pt;
            rp.page_tokens = P;
            rp.max_pages = max_pages;
            launch_decode_rows(rp, int8, st);
        }
        // tensor parallel, 4..32 slots: the layer's all-reduce on the side stream under the other micro-batch's launches (the
        // engine's decoder_overlapped, engine.hip.h, on the paged cache; FTCF_DECODE_OVERLAP = 0 / 1, auto = timed on this node)
        const bool overlap = !use_rows && decode_overlap_now(B, dual);
        const auto t_dec0  = std::chrono::steady_clock::now();
        if (overlap) {
            e->decode_overlap_streams();
            const int         r0[2] = {0, (B + 1) / 2}, r1[2] = {(B + 1) / 2, B};
            const hipStream_t cs[2] = {st, e->side2};
            FTCF_HIP_CHECK(hipEventRecord(e->dv_fork[0], st));
            FTCF_HIP_CHECK(hipStreamWaitEvent(e->side2, e->dv_fork[0], 0));
            for (int l = 0; l < L; l++) {
                const LayerWeights& w = e->layers[l];
                for (int c = 0; c < 2; c++) {
                    const int         M  = r1[c] - r0[c];
                    const size_t      o  = (size_t)r0[c], wo = (size_t)c * smallm_region;
                    const hipStream_t s2 = cs[c];
                    f16*              xr = x + o * H;
                    const bool pair = e->tp_pair_ar;
                    f16* const attc = att + 2 * o * H;  // (a micro-batch's attn | ffn rows adjacent)
                    f16* const ffnc = attc + (size_t)M * H;
                    if (l > 0 && !pair) {
                        FTCF_HIP_CHECK(hipStreamWaitEvent(s2, e->dv_red[c], 0));
                    }
                    if (l == 0 || !pair) {
                        launch_residual_dual_ln(xr, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, nrm + o * H,
                                                nrm2 + o * H, M, H, 1e-5f, s2);
                    }
                    MmhaPagedParams mp{};
                    mp.qkv = qkv + o * 3 * hl;
                    mp.qkv_bias = w.qkv.bias;
                    mp.kpool = kpool + (size_t)l * pool_layer_elems;
                    mp.vpool = vpool + (size_t)l * pool_layer_elems;
                    mp.page_table = d_pt + o * max_pages;
                    mp.len = d_len + o;
                    mp.finished = d_fin + o;
                    mp.B = M;
                    mp.nh = e->nhl;
                    mp.dh = e->dh;
                    mp.rot = e->cfg.rotary_embedding_dim;
                    mp.P = P;
                    mp.max_pages = max_pages;
                    mp.ctx = ctx + o * hl;
                    const SmallmDesc p1[2] = {{nrm + o * H, w.qkv.kernel, w.qkv.scale, nullptr, 0, qkv + o * 3 * hl, 3 * hl, H},
                                              {nrm2 + o * H, w.ffn1.kernel, w.ffn1.scale, w.ffn1.bias, 1, mid + o * il, il, H}};
                    launch_gemm_smallm_group(p1, 2, smallm_ws, smallm_partial, M, int8, s2, &d_gstate->step, &smallm_seq, wo);
                    launch_mmha_paged(mp, max_len, s2);
                    const SmallmDesc p3[2] = {{ctx + o * hl, w.attn_out.kernel, w.attn_out.scale, nullptr, 0, attc, H, hl},
                                              {mid + o * il, w.ffn2.kernel, w.ffn2.scale, nullptr, 0, ffnc, H, il}};
                    launch_gemm_smallm_group(p3, 2, smallm_ws, smallm_partial, M, int8, s2, &d_gstate->step, &smallm_seq, wo);
                    if (!pair) {
                        launch_add_bias_attn_ffn_residual(xr, ffnc, attc, xr, w.ffn2.bias, M, H, tp, (l > 0 && l < L - 1) ? 1 : 0, true, s2);
                    }
                    FTCF_HIP_CHECK(hipEventRecord(e->dv_done[c], s2));
                    FTCF_HIP_CHECK(hipStreamWaitEvent(e->side, e->dv_done[c], 0));
                    e->allreduce(pair ? attc : xr, (size_t)(pair ? 2 : 1) * M * H, e->side);
                    FTCF_HIP_CHECK(hipEventRecord(e->dv_red[c], e->side));
                    if (pair) {
                        const LayerWeights* nx = l + 1 < L ? &e->layers[l + 1] : nullptr;
                        FTCF_HIP_CHECK(hipStreamWaitEvent(s2, e->dv_red[c], 0));
                        launch_residual_dual_ln(xr, ffnc, attc, w.ffn2.bias, 1, 1, nx ? nx->ln1_g : nullptr, nx ? nx->ln1_b : nullptr,
                                                nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, nrm + o * H, nrm2 + o * H, M, H, 1e-5f,
                                                s2, tp);
                    }
                }
            }
            FTCF_HIP_CHECK(hipEventRecord(e->dv_fork[1], e->side2));
            FTCF_HIP_CHECK(hipStreamWaitEvent(st, e->dv_fork[1], 0));
            for (int c = 0; c < 2; c++) {
                FTCF_HIP_CHECK(hipStreamWaitEvent(st, e->dv_red[c], 0));
            }
        }
        for (int l = 0; l < (use_rows || overlap ? 0 : L); l++) {
            const LayerWeights& w = e->layers[l];
            if (!dual) {
                launch_layernorm(x, w.ln1_g, w.ln1_b, nrm, B, H, 1e-5f, true, st);
                launch_layernorm(x, w.ln2_g, w.ln2_b, nrm2, B, H, 1e-5f, true, st);
            }
            else if (l == 0 || (!tp1 && !e->tp_pair_ar)) {
                launch_residual_dual_ln(x, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, nrm, nrm2, B, H,
                                        1e-5f, st);
            }
            MmhaPagedParams mp{};
            mp.qkv = qkv;
            mp.qkv_bias = w.qkv.bias;
            mp.kpool = kpool + (size_t)l * pool_layer_elems;
            mp.vpool = vpool + (size_t)l * pool_layer_elems;
            mp.page_table = d_pt;
            mp.len = d_len;
            mp.finished = d_fin;
            mp.B = B;
            mp.nh = e->nhl;
            mp.dh = e->dh;
            mp.rot = e->cfg.rotary_embedding_dim;
            mp.P = P;
            mp.max_pages = max_pages;
            mp.ctx = ctx;
            if (smallm_ws && e->decode_branches && e->side) {
                // the attention branch and the FFN branch on two streams, as the engine's batched decode
                const int    bc = std::min(B, 16);
                const size_t o_qkv = 0, o_f1 = o_qkv + gemm_smallm_workspace_bytes(bc, 3 * hl, H, int8),
                             o_out = o_f1 + gemm_smallm_workspace_bytes(bc, il, H, int8),
                             o_f2  = o_out + gemm_smallm_workspace_bytes(bc, H, hl, int8);
                auto one = [&](const SmallmDesc& d0, size_t off, hipStream_t s2) {
                    for (int r0 = 0; r0 < B; r0 += 16) {
                        SmallmDesc d = d0;
                        d.A          = d0.A + (size_t)r0 * d0.k;
                        d.C          = d0.C + (size_t)r0 * d0.n;
                        launch_gemm_smallm_group(&d, 1, smallm_ws, smallm_partial, std::min(16, B - r0), int8, s2, &d_gstate->step,
                                                 &smallm_seq, off);
                    }
                };
                const size_t offs[4] = {o_qkv, o_f1, o_out, o_f2};
                GemmFn burst = [&](const f16* A, const DenseWeight& dw, const f16* bias, int act, f16* C, int, int n, int k,
                                   hipStream_t s2, int slot) { one(SmallmDesc{A, dw.kernel, dw.scale, bias, act, C, n, k}, offs[slot], s2); };
                FTCF_HIP_CHECK(hipEventRecord(e->ev_fork, st));
                FTCF_HIP_CHECK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
                DecoderSelfAttentionLayer{burst, H, hl}.forward_paged(nrm, qkv, ctx, att, w, mp, max_len, B, st);
                FfnLayer{burst, H, il}.forward(nrm2, mid, ffn, w, B, e->side);
                FTCF_HIP_CHECK(hipEventRecord(e->ev_join, e->side));
                FTCF_HIP_CHECK(hipStreamWaitEvent(st, e->ev_join, 0));
            }
            else if (smallm_ws && B <= 16) {
                const SmallmDesc p1[2] = {{nrm, w.qkv.kernel, w.qkv.scale, nullptr, 0, qkv, 3 * hl, H},
                                          {nrm2, w.ffn1.kernel, w.ffn1.scale, w.ffn1.bias, 1, mid, il, H}};
                launch_gemm_smallm_group(p1, 2, smallm_ws, smallm_partial, B, int8, st, &d_gstate->step, &smallm_seq);
                launch_mmha_paged(mp, max_len, st);
                const SmallmDesc p3[2] = {{ctx, w.attn_out.kernel, w.attn_out.scale, nullptr, 0, att, H, hl},
                                          {mid, w.ffn2.kernel, w.ffn2.scale, nullptr, 0, ffn, H, il}};
                launch_gemm_smallm_group(p3, 2, smallm_ws, smallm_partial, B, int8, st, &d_gstate->step, &smallm_seq);
            }
            else {
                GemmFn plain = [&](const f16* A, const DenseWeight& dw, const f16* bias, int act, f16* C, int m, int n, int k,
                                   hipStream_t s2, int) {
                    gemm_dispatch(A, dw.kernel, dw.scale, bias, act, C, m, n, k, int8, s2, nullptr, 0, e->num_cu, nullptr, nullptr,
                                  s2 == st ? tiled_ws : nullptr);
                };
                DecoderSelfAttentionLayer{plain, H, hl}.forward_paged(nrm, qkv, ctx, att, w, mp, max_len, B, st);
                FfnLayer{plain, H, il}.forward(nrm2, mid, ffn, w, B, st);
            }
            // (every slot's hidden state is recomputed from its token each step: the residual never aliases across steps,
            // so the fp32-sum variant of the context decoder applies to all layers)
            if (dual && !tp1 && e->tp_pair_ar) {
                // attn | ffn as one all-reduce message, the residual inside the next layer's LayerNorm pass (engine.hip.h decoder)
                e->allreduce(att, (size_t)2 * B * H, st);
                const LayerWeights* nx = l + 1 < L ? &e->layers[l + 1] : nullptr;
                launch_residual_dual_ln(x, ffn, att, w.ffn2.bias, 1, 1, nx ? nx->ln1_g : nullptr, nx ? nx->ln1_b : nullptr,
                                        nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, nrm, nrm2, B, H, 1e-5f, st, tp);
            }
            else if (dual && tp1) {
                const LayerWeights* nx = l + 1 < L ? &e->layers[l + 1] : nullptr;
                launch_residual_dual_ln(x, ffn, att, w.ffn2.bias, 1, (l > 0 && l < L - 1) ? 1 : 0, nx ? nx->ln1_g : nullptr,
                                        nx ? nx->ln1_b : nullptr, nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, nrm, nrm2, B, H,
                                        1e-5f, st);
            }
            else {
                launch_add_bias_attn_ffn_residual(x, ffn, att, x, w.ffn2.bias, B, H, tp, (l > 0 && l < L - 1) ? 1 : 0, true, st);
                e->allreduce(x, (size_t)B * H, st);
            }
        }
        {
            // LM head; tensor parallel: rank r computes rows [r V/TP, (r+1) V/TP) of the replicated lm_head into its slice of
            // `gather`, all-gather + transpose (GptNeoX.cc:888-925, as the engine's own step)
            const int    rows = tp1 ? V : e->vl;
            const f16*   Wr   = tp1 ? e->lm_head : e->lm_head + (size_t)e->cfg.tensor_para_rank * e->vl * H;
            float*       out  = tp1 ? logits : gather + (size_t)e->cfg.tensor_para_rank * B * e->vl;
            if (B <= 4) {
                launch_lm_head(x, Wr, out, B, rows, H, rows, st, e->final_g, e->final_b, 1e-5f);
            }
            else {
                launch_layernorm(x, e->final_g, e->final_b, nrm, B, H, 1e-5f, true, st);
                lm_head_dispatch(nrm, Wr, out, B, rows, H, rows, st);
            }
            if (!tp1) {
                e->allgather_logits(gather, logits, B, st);
            }
        }
        SamplingParams sp{};
        sp.logits = logits;
        sp.B = B;
        sp.V = V;
        sp.max_input_len = 0;
        sp.end_id = e->cfg.end_id;
        sp.input_lengths = d_zero;
        sp.top_k = d_topk;
        sp.top_p_topk = d_ptopk;
        sp.top_p_topp = d_ptopp;
        sp.temperature = d_temp;
        sp.random_seed = d_seed;
        sp.draw_counter = d_draws;
        sp.apply_temperature = host_any_temperature ? 1 : 0;
        sp.apply_repetition = host_any_repetition ? 1 : 0;  // (BaseSamplingLayer.cc:283-313: skipped when every row has 1.0)
        sp.repetition_penalty = d_rep;
        sp.return_cum_log_probs = 1;
        // every slot has its own step: its history is column b of the time-major d_hist, positions [0, len[b]]; the sampled
        // token goes to position len[b] + 1 (the penalty of sampling_penalty_kernels.cu:367-425 reads the whole history)
        sp.output_ids = d_hist;
        sp.row_len = d_len;
        sp.total_len = max_len + 2;
        sp.finished = d_fin;
        sp.seq_len = d_len;     // + 1 per sampled token: the slot's length
        sp.cum_log_probs = d_cum;
        sp.pad_count = d_zero;
        sp.state = d_state;     // step stays 0
        sp.ws = samp_ws;
        sp.max_top_k = host_max_top_k;
        sp.any_top_p = host_any_top_p;
        if (!groups.empty()) {  // the rows of beam groups are "finished" for the sampling kernels
            enqueue_beam_steps(st);
            hipLaunchKernelGGL(k_batcher_sampling_view, dim3(1), dim3(64), 0, st, d_sfin, d_fin, d_isbeam, B, 0);
            sp.finished = d_sfin;
        }
        DynamicDecodeLayer{}.forward(sp, st, false);  // (the stop / length criteria are the scheduler's: no finish step)
        hipLaunchKernelGGL(k_batcher_last_token, dim3(1), dim3(64), 0, st, d_tok, d_hist, d_len, B);
        if (!groups.empty()) {
            hipLaunchKernelGGL(k_batcher_sampling_view, dim3(1), dim3(64), 0, st, d_sfin, d_fin, d_isbeam, B, 1);
            enqueue_beam_advance(st);
        }
        std::vector<int>     tok(B);
        std::vector<uint8_t> fin(B);
        int                  gemm_err = 0;
        FTCF_HIP_CHECK(hipMemcpyAsync(tok.data(), d_tok, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(fin.data(), d_fin, (size_t)B, hipMemcpyDeviceToHost, st));
        if (smallm_ws) {  // sticky flag of the burst GEMMs' in-launch split-K reduction (the engine's finish() reads its own)
            FTCF_HIP_CHECK(hipMemcpyAsync(&gemm_err, reinterpret_cast<char*>(smallm_ws) + smallm_partial, sizeof(int),
                                          hipMemcpyDeviceToHost, st));
        }
        // tensor parallel: the window all-reduce's sticky give-up word of this step's 2 L all-reduces (a peer that never arrived:
        // x is partly reduced and the step's tokens are not to be trusted either), read with the same copies
        int rows_err = 0;
        if (use_rows) {
            FTCF_HIP_CHECK(hipMemcpyAsync(&rows_err, rows_ws, sizeof(int), hipMemcpyDeviceToHost, st));
        }
        int        ar_err  = 0;
        const bool ar_live = e->cfg.tensor_para_size > 1 && e->cfg.comm && e->cfg.comm->ar_sync && e->cfg.comm->ar_seq > 0
                             && !e->cfg.comm->ar_failed;
        if (ar_live) {
            FTCF_HIP_CHECK(hipMemcpyAsync(&ar_err, e->cfg.comm->ar_sync + 2, sizeof(int), hipMemcpyDeviceToHost, st));
        }
        FTCF_HIP_CHECK(hipStreamSynchronize(st));
        decode_overlap_trial(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_dec0).count(), st);
        e->stats.decode_overlap = overlap ? 1 : 0;
        if (ar_live && e->tp_scratch) {
            // (collective: every rank's batcher runs the same step and has called the window all-reduce as often; all of them learn
            //  of any rank's failure and drop the step)
            ar_err = comm_max(e->cfg.comm, ar_err, st, e->tp_scratch);
            if (ar_err != 0) {
                e->cfg.comm->ar_failed = true;  // this communicator keeps RCCL (or the emulation) from now on
                throw Error(-2, "batcher decode: the exchange-window all-reduce gave up waiting for a peer; the step's tokens are dropped");
            }
        }
        if (rows_err != 0) {
            // a hand-off of the rows kernel ran out of its bounded wait (cannot happen with every workgroup resident; bounded all the
            // same): the step's tokens are dropped, the batcher goes on with the per-GEMM launches
            use_rows = false;
            throw Error(-2, "batcher decode: the rows kernel gave up waiting for a hand-off (code " + std::to_string(rows_err) + ")");
        }
        if (gemm_err != 0) {
            // the tokens of this step are not to be trusted: nothing is reported, the slots keep their state (lengths and
            // draw counters advanced on the device: the requests cannot be resumed exactly), the flag is cleared for the caller's
            // next attempt
            FTCF_HIP_CHECK(hipMemsetAsync(smallm_ws, 0, smallm_partial + gemm_smallm_ticket_bytes(), st));
            FTCF_HIP_CHECK(hipStreamSynchronize(st));
            throw Error(-2, "batcher decode: a split-K reducer of the batched GEMM gave up waiting for its sibling workgroups");
        }
        for (int si = 0; si < B; si++) {
            Slot& s = slots[si];
            if (!s.active || s.group >= 0) {
                continue;
            }
            s.len += 1;
            s.generated += 1;
            s.hist.push_back(tok[si]);
            const int done = (fin[si] || s.generated >= s.max_new || hits_stop_word(s)) ? 1 : 0;
            emit(ev, Event{s.id, tok[si], done});
            if (done) {
                if (!fin[si]) {
                    const uint8_t one8 = 1;
                    FTCF_HIP_CHECK(hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice));
                }
                release(s);
            }
        }
        if (!groups.empty()) {
            advance_groups(ev, fin);
        }
    }

    int  host_max_top_k = 1, host_any_top_p = 0;
    bool host_any_temperature = false, host_any_repetition = false;
    std::vector<int>   slot_topk;
    std::vector<float> slot_temp;

    void step(std::vector<Event>& ev)
    {
        FTCF_HIP_CHECK(hipSetDevice(e->cfg.device));
        if (slot_topk.empty()) {
            slot_topk.assign(max_batch, 1);
            slot_temp.assign(max_batch, 1.f);
        }
        // the running slots first (a request admitted in this iteration has its first token already)
        bool any = false;
        for (const Slot& s : slots) {
            any |= s.active;
        }
        if (any) {
            decode(ev);
        }
        // admissions: as many of the queue's head requests as there are free slots and pages, prefilled as ONE ragged batch
        std::vector<int>     sis;
        std::vector<Request> rs;
        int                  pages_left = (int)free_pages.size() - reserved_pages();
        if (!waiting.empty() && waiting.front().beam_width > 1) {
            // a beam request at the head of the queue: K consecutive free slots and the group's whole page budget, admitted alone
            const Request& r    = waiting.front();
            const int      K    = r.beam_width, need = K * (((int)r.prompt.size() + r.max_new + P - 1) / P);
            int            si0  = -1;
            for (int si = 0, run = 0; si < max_batch && si0 < 0; si++) {
                run = slots[si].active ? 0 : run + 1;
                if (run == K) {
                    si0 = si - K + 1;
                }
            }
            if (si0 >= 0 && pages_left >= need) {
                Request rq = std::move(waiting.front());
                waiting.pop_front();
                for (int k = 0; k < K; k++) {
                    slot_topk[si0 + k] = 1;
                    slot_temp[si0 + k] = 1.f;
                }
                // slots are running and the prompt is long: its prompt phase in chunks, a decode step of the running slots after every
                // chunk (as for a sampled request below; the events of those steps are final whatever happens to the admission)
                bool running = false;
                for (const Slot& s : slots) {
                    running |= s.active;
                }
                const bool         chunked = running && prefill_chunk > 0 && (int)rq.prompt.size() > prefill_chunk;
                std::vector<Event> own, between;
                try {
                    if (chunked) {
                        hook_ev          = &between;
                        e->prefill_chunk = prefill_chunk;
                        e->prefill_hook  = [this] {
                            bool live = false;
                            for (const Slot& s : slots) {
                                live |= s.active && s.group < 0;
                            }
                            if (live || !groups.empty()) {
                                refresh_host_flags();
                                decode(*hook_ev);
                            }
                        };
                    }
                    admit_beam(si0, rq, own);
                    e->prefill_hook = nullptr;
                    hook_ev         = nullptr;
                }
                catch (...) {
                    e->prefill_hook = nullptr;
                    hook_ev         = nullptr;
                    (void)hipDeviceSynchronize();
                    (void)hipGetLastError();
                    release_group(si0);
                    beam_results.erase(rq.id);
                    waiting.push_front(std::move(rq));
                    ev.insert(ev.end(), between.begin(), between.end());
                    throw;
                }
                ev.insert(ev.end(), between.begin(), between.end());
                ev.insert(ev.end(), own.begin(), own.end());
                fire(own, 0);  // (the admission's own events reach the callback only now)
            }
            refresh_host_flags();
            return;
        }
        for (int si = 0; si < max_batch && !waiting.empty(); si++) {
            if (slots[si].active) {
                continue;
            }
            const Request& r    = waiting.front();
            if (r.beam_width > 1) {
                break;  // (admitted alone, by the next iteration)
            }
            const int      need = ((int)r.prompt.size() + r.max_new + P - 1) / P;
            if (pages_left < need) {
                break;  // FIFO: nobody overtakes the head of the queue
            }
            pages_left -= need;
            const int keff = (r.top_k == 0 && r.top_p == 0.f) ? 1 : std::min(r.top_k, 1024);
            slot_topk[si]  = keff;
            slot_temp[si]  = r.temperature;
            sis.push_back(si);
            rs.push_back(std::move(waiting.front()));
            waiting.pop_front();
        }
        bool any_long = false;
        for (const Request& r : rs) {
            any_long |= prefill_chunk > 0 && (int)r.prompt.size() > prefill_chunk;
        }
        if (!sis.empty() && any && any_long) {
            // slots are running and a long prompt arrives: one request at a time, its prompt phase in chunks with a decode step of
            // the running slots after every chunk (events of those steps are final whatever happens to the admission)
            while (!sis.empty()) {
                const std::vector<int>     one_si{sis.front()};
                const std::vector<Request> one_r{rs.front()};
                std::vector<Event>         own, between;
                bool                       running = false;
                for (const Slot& s : slots) {
                    running |= s.active;
                }
                try {
                    if (running) {
                        hook_ev          = &between;
                        e->prefill_chunk = prefill_chunk;
                        e->prefill_hook  = [this] {
                            bool live = false;
                            for (const Slot& s : slots) {
                                live |= s.active;
                            }
                            if (live) {
                                refresh_host_flags();
                                decode(*hook_ev);
                            }
                        };
                    }
                    admit(one_si, one_r, own);
                    e->prefill_hook = nullptr;
                    hook_ev         = nullptr;
                }
                catch (...) {
                    e->prefill_hook = nullptr;
                    hook_ev         = nullptr;
                    (void)hipDeviceSynchronize();
                    (void)hipGetLastError();
                    ev.insert(ev.end(), between.begin(), between.end());
                    for (const int si : sis) {  // this request and the ones not yet admitted go back to the queue's head
                        release(slots[si]);
                        const uint8_t one8 = 1;
                        (void)hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice);
                    }
                    for (size_t i = rs.size(); i-- > 0;) {
                        waiting.push_front(std::move(rs[i]));
                    }
                    throw;
                }
                ev.insert(ev.end(), between.begin(), between.end());
                ev.insert(ev.end(), own.begin(), own.end());
                fire(own, 0);
                sis.erase(sis.begin());
                rs.erase(rs.begin());
            }
        }
        if (!sis.empty()) {
            const size_t ev0 = ev.size();
            try {
                admit(sis, rs, ev);
            }
            catch (...) {
                // the whole admission is rolled back: pages to the pool, slots free, the requests back at the head of the
                // queue in their order, their events dropped (the caller sees the exception, not half an admission)
                (void)hipDeviceSynchronize();
                (void)hipGetLastError();
                for (const int si : sis) {
                    release(slots[si]);
                    const uint8_t one8 = 1;
                    (void)hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice);
                }
                for (size_t i = rs.size(); i-- > 0;) {
                    waiting.push_front(std::move(rs[i]));
                }
                ev.resize(ev0);
                throw;
            }
            fire(ev, ev0);
        }
        refresh_host_flags();
    }
    // host view of the running slots' sampling parameters (kernel selection and LDS sizing of the sampling kernels): after every
    // admission, and before a decode step that runs inside an admission (a request admitted a moment ago is already running)
    void refresh_host_flags()
    {
        host_max_top_k = 1;
        host_any_top_p = 0;
        host_any_temperature = false;
        host_any_repetition = false;
        for (int si = 0; si < max_batch; si++) {
            if (slots[si].active) {
                host_any_repetition |= (slots[si].repetition_penalty != 1.f);
                host_max_top_k = std::max(host_max_top_k, slot_topk[si]);
                host_any_top_p |= (slot_topk[si] == 0);
                host_any_temperature |= (slot_temp[si] != 1.f);
            }
        }
    }
};

extern "C" int ftcf_batcher_create(ftcf_gptneox_t engine, int max_batch, int page_tokens, int num_pages, int max_seq_len,
                                   ftcf_batcher_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(engine && out, "NULL argument");
        require_device();
        auto b = std::make_unique<ftcf_batcher>();
        b->init(engine, max_batch, page_tokens, num_pages, max_seq_len);
        *out = b.release();
    });
}
extern "C" int ftcf_batcher_submit(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int top_k,
                                   float top_p, float temperature, unsigned long long seed, long* request_id)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_id, "NULL argument");
        *request_id = b->submit(prompt_ids, prompt_len, max_new_tokens, top_k, top_p, temperature, (uint64_t)seed);
    });
}
extern "C" int ftcf_batcher_submit_ex(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int top_k,
                                      float top_p, float temperature, float repetition_penalty, unsigned long long seed,
                                      const int* stop_words, int stop_len, long* request_id)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_id, "NULL argument");
        *request_id = b->submit(prompt_ids, prompt_len, max_new_tokens, top_k, top_p, temperature, (uint64_t)seed,
                                repetition_penalty, stop_words, stop_len);
    });
}
extern "C" int ftcf_batcher_submit_beam(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int beam_width,
                                        float beam_search_diversity_rate, float len_penalty, float temperature,
                                        float repetition_penalty, long* request_id)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_id, "NULL argument");
        *request_id = b->submit_beam(prompt_ids, prompt_len, max_new_tokens, beam_width, beam_search_diversity_rate, len_penalty,
                                     temperature, repetition_penalty);
    });
}
extern "C" int ftcf_batcher_submit_beam_ex(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int beam_width,
                                           float beam_search_diversity_rate, float len_penalty, float temperature,
                                           float repetition_penalty, int min_length, const int* stop_words, int stop_len,
                                           long* request_id)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_id, "NULL argument");
        *request_id = b->submit_beam(prompt_ids, prompt_len, max_new_tokens, beam_width, beam_search_diversity_rate, len_penalty,
                                     temperature, repetition_penalty, min_length, stop_words, stop_len);
    });
}
extern "C" int ftcf_batcher_beam_result(ftcf_batcher_t b, long request_id, int* output_ids, int* sequence_lengths,
                                        float* cum_log_probs, int capacity, int* beam_width, int* total_len)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && beam_width && total_len, "NULL argument");
        auto it = b->beam_results.find(request_id);
        if (it == b->beam_results.end()) {
            *beam_width = 0;
            *total_len  = 0;
            return;
        }
        const ftcf_batcher::BeamResult& r = it->second;
        *beam_width = r.K;
        *total_len  = r.total;
        if (!output_ids) {
            return;  // (a size query)
        }
        FTCF_CHECK_ARG(capacity >= r.K * r.total && sequence_lengths && cum_log_probs, "beam result: arrays too small");
        std::copy(r.ids.begin(), r.ids.end(), output_ids);
        std::copy(r.lens.begin(), r.lens.end(), sequence_lengths);
        std::copy(r.cum.begin(), r.cum.end(), cum_log_probs);
        b->beam_results.erase(it);
    });
}
extern "C" int ftcf_batcher_step(ftcf_batcher_t b, long* request_ids, int* tokens, int* finished, int capacity, int* n_events)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_ids && tokens && finished && n_events, "NULL argument");
        FTCF_CHECK_ARG(capacity >= 2 * b->max_batch, "event arrays must hold 2 * max_batch entries");
        if (b->outbox.empty()) {  // (else: the rest of the previous iteration's events first)
            std::vector<ftcf_batcher::Event> ev;
            try {
                b->step(ev);
            }
            catch (...) {
                b->outbox.insert(b->outbox.end(), ev.begin(), ev.end());  // tokens of decode steps that did run are not lost
                throw;
            }
            b->outbox.insert(b->outbox.end(), ev.begin(), ev.end());
        }
        int n = 0;
        for (; n < capacity && !b->outbox.empty(); n++) {
            request_ids[n] = b->outbox.front().id;
            tokens[n]      = b->outbox.front().token;
            finished[n]    = b->outbox.front().finished;
            b->outbox.pop_front();
        }
        *n_events = n;
    });
}
extern "C" int ftcf_batcher_set_token_callback(ftcf_batcher_t b, ftcf_token_callback_fn fn, void* user)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b, "NULL argument");
        b->on_token      = fn;
        b->on_token_user = user;
    });
}
extern "C" int ftcf_batcher_status(ftcf_batcher_t b, int* waiting, int* running, int* free_pages)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b, "NULL argument");
        int run = 0;
        for (const auto& s : b->slots) {
            run += s.active ? 1 : 0;
        }
        if (waiting) {
            *waiting = (int)b->waiting.size();
        }
        if (running) {
            *running = run + (b->outbox.empty() ? 0 : 1);  // (events still to be fetched keep the batcher "busy")
        }
        if (free_pages) {
            *free_pages = (int)b->free_pages.size();
        }
    });
}
extern "C" int ftcf_batcher_cancel(ftcf_batcher_t b, long request_id, int* found)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b, "NULL argument");
        int hit = 0;
        for (auto it = b->waiting.begin(); it != b->waiting.end(); ++it) {
            if (it->id == request_id) {
                b->waiting.erase(it);
                hit = 1;
                break;
            }
        }
        for (int si = 0; si < b->max_batch && !hit; si++) {
            ftcf_batcher::Slot& s = b->slots[si];
            if (s.active && s.id == request_id && s.group >= 0) {
                FTCF_HIP_CHECK(hipSetDevice(b->e->cfg.device));
                b->release_group(s.group);
                hit = 1;
                break;
            }
            if (s.active && s.id == request_id) {
                FTCF_HIP_CHECK(hipSetDevice(b->e->cfg.device));
                const uint8_t one8 = 1;
                FTCF_HIP_CHECK(hipMemcpy(b->d_fin + si, &one8, 1, hipMemcpyHostToDevice));
                b->release(s);
                hit = 1;
            }
        }
        if (!hit && b->beam_results.erase(request_id) > 0) {  // a finished beam request whose result was never fetched
            hit = 1;
        }
        if (found) {
            *found = hit;
        }
    });
}
extern "C" int ftcf_batcher_destroy(ftcf_batcher_t b)
{
    return guarded([&] { delete b; });
}
