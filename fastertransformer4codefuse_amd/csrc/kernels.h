// Internal launcher declarations shared by the engine and the C ABI (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "ftcf_common.h"

namespace ftcf {

constexpr int GEMV_SPLITK_MAX_WAVES = 16;
constexpr int EPI_PLAIN             = 0;
constexpr int EPI_RESIDUAL          = 1;

struct LnGemvParams {
    const f16* x;  // [M, K] layer input
    const f16 *gamma0, *beta0, *gamma1, *beta1;
    const void *W0, *W1;          // tiled weights of segment 0 (QKV) / 1 (FFN1)
    const f16 *scale0, *scale1;   // int8 only
    const f16* bias1;
    f16 *out0, *out1;             // [M, NT0*16], [M, NT1*16]
    int   K, NT0, NT1, blocks0, blocks1;
    float eps;
};

struct SplitKParams {
    const f16 *x_a, *x_b;
    const void *W_a, *W_b;
    const f16 *scale_a, *scale_b;
    const f16* bias;
    const f16* x_in;  // residual input (EPI_RESIDUAL)
    const f16* ffn_in;  // EPI_RESIDUAL with a single segment: the FFN output comes from this buffer (two-stream path)
    f16*       out;
    int        N, KT_a, KT_b;
    int        act, tp, inplace_variant;
    int        nwaves, slice_halves;
    int        wave_seg[GEMV_SPLITK_MAX_WAVES], wave_t0[GEMV_SPLITK_MAX_WAVES], wave_nt[GEMV_SPLITK_MAX_WAVES];
};

// K3 in its balanced form: every 16-column group is cut into Q chunks of the concatenated K range [x_a ; x_b]; one
// 2-wave workgroup per (group, chunk), partial sums handed to the chunk-0 workgroup as {tag,value} granules.
struct ChunkParams {
    const f16 *x_a, *x_b;
    const void *W_a, *W_b;
    const f16 *scale_a, *scale_b;
    const f16* bias;
    const f16* x_in;
    f16*       out;
    int        N, KT_a, KT_b, Q, T;  // T = tiles per chunk
    int        tp, inplace_variant;
    unsigned long long* gran;  // [N/16][Q][2][M][16]
    const int* d_step;
    int        step, salt;
};
int    chunk_pick_q(int NT, int KT_total);
size_t chunk_workspace_bytes(int N, int M, int Q);
void   launch_gemv_chunked(const ChunkParams& p, bool int8, int M, hipStream_t s);
void launch_ln_gemv(const LnGemvParams& p, bool int8, int M, hipStream_t s);
// one 16-column group per workgroup of `wpg` waves (balanced form, grid = NT0 + NT1)
void launch_ln_gemv_group(const LnGemvParams& p, bool int8, int M, int wpg, hipStream_t s);
void plan_splitk(SplitKParams& p, bool int8, int M, int max_waves);
void launch_gemv_splitk(const SplitKParams& p, bool int8, int M, int epi, hipStream_t s);
// optional fused LayerNorm of x (gamma != NULL)
void launch_lm_head(const f16* x, const f16* W, float* logits, int M, int n_rows, int K, int ldc, hipStream_t s,
                    const f16* gamma = nullptr, const f16* beta = nullptr, float eps = 1e-5f, const int* d_stop = nullptr);

// ---- MFMA GEMM (prefill / batched decode) : kernels_gemm.hip ----
// C[m,n] = A[m,k] x W(tiled)  (+bias, gelu) ; int8: fused fp32 epilogue ; fp16: half epilogue
// `workspace` (gemm_tiled_workspace_bytes(), zeroed once; NULL = none): lets m <= 256 cut the K extent of its few tiles into
// slices reduced inside the launch.  One workspace serves one stream.
size_t gemm_tiled_workspace_bytes();
size_t gemm_tiled_ticket_bytes();  // the workspace's tail
int    gemm_tiled_splitk_max_m();  // rows up to which launch_gemm_tiled uses the workspace (FTCF_GEMM_SPLITK_MAX_M)
void   launch_gemm_tiled(const f16* A, const void* W, const f16* scale, const f16* bias, int act, f16* C, int m, int n,
                         int k, bool int8, hipStream_t s, float* workspace = nullptr);
// batched decode GEMM for m <= 16 rows (HBM bound forms); with a `workspace` (>= gemm_smallm_workspace_bytes of this
// GEMM) the burst form runs: K slices of 20 tiles, every wave requests its whole slice at once
// at once; `workspace` = [partial_bytes of split-K partial sums][gemm_smallm_ticket_bytes(), zeroed once before the first
// launch]: the reduction runs in the workgroup that takes the last ticket of a column block (no second launch)
size_t gemm_smallm_workspace_bytes(int m, int n, int k, bool int8);
size_t gemm_smallm_ticket_bytes();
struct SmallmDesc {  // C[m, n] = epilogue(A[m, k] x W[k, n]); act 1 = bias + gelu
    const f16*  A;
    const void* W;
    const f16*  scale;
    const f16*  bias;
    int         act;
    f16*        C;
    int         n, k;
};
// one launch for up to two independent GEMMs with the same m (partial_bytes >= the SUM of their workspace bytes)
// d_step: device-resident decode step (part of the launch's granule tag, a replayed hipGraph freezes scalars) or NULL;
// seq: the workspace's launch counter (host).  The granules must be zero when a request begins.
void   launch_gemm_smallm_group(const SmallmDesc* d, int np, float* workspace, size_t partial_bytes, int m, bool int8,
                                hipStream_t s, const int* d_step, unsigned* seq, size_t partial_offset = 0);
void   launch_gemm_smallm(const f16* A, const void* W, const f16* scale, const f16* bias, int act, f16* C, float* workspace,
                          size_t partial_bytes, int m, int n, int k, bool int8, int num_cu, hipStream_t s,
                          const int* d_step = nullptr, unsigned* seq = nullptr);
// logits_f32[m, n] = A[m,k] x W[n,k]^T (row major fp16 weights, m > 4)
void launch_gemm_nk_f32out(const f16* A, const f16* W_nk, float* C, int m, int n, int k, int ldc, hipStream_t s);

// ---- misc : kernels_misc.hip ----
void launch_layernorm(const void* x, const void* gamma, const void* beta, void* out, int m, int n, float eps,
                      bool fp16, hipStream_t s);
// [x += residual of the previous layer;] out1 = LN(x; g1, b1), out2 = LN(x; g2, b2) in one pass (ffn == NULL: no
// residual; g1 == NULL: residual only)
bool residual_dual_ln_supported(int n);
void launch_residual_dual_ln(f16* x, const f16* ffn, const f16* attn, const f16* bias, int tp, int inplace_variant,
                             const f16* g1, const f16* b1, const f16* g2, const f16* b2, f16* out1, f16* out2, int m,
                             int n, float eps, hipStream_t s, int bias_mul = 1);
// out = (bias + a) + b in fp32, rounded once (sequential-residual layers)
void launch_add_bias_residual(f16* out, const f16* a, const f16* b, const f16* bias, int m, int n, hipStream_t s);
void launch_add_bias_attn_ffn_residual(void* out, const void* ffn, const void* attn, const void* in, const void* bias,
                                       int m, int n, int tp, int inplace_variant, bool fp16, hipStream_t s);
void launch_embedding(f16* out, const f16* table, const int* ids, int n_ids, int H, hipStream_t s);
// prompt: ids [B,S] -> out [B*S,H] and time-major output_ids[s*B+b] (gpt_kernels.cu:31-104)
void launch_prompt_embedding(f16* out, int* output_ids, const f16* table, const int* ids, int B, int S, int H,
                             hipStream_t s);
// decode step: embedding of output_ids[(step-1)*B + b] where step is read from device state (decoding_kernels.cu:145-191)
void launch_step_embedding(f16* out, const f16* table, const int* output_ids, const int* d_step, int B, int H,
                           hipStream_t s);
// out[r] = hidden[r / tile][input_lengths[r / tile] - 1] for r < B * tile (beam search: every beam starts from beam 0's state)
void launch_gather_last_token(f16* out, const f16* hidden, const int* input_lengths, int B, int S, int H,
                              hipStream_t s, int tile = 1);
// output_ids[s][b * K + k] = ids[b][s] (time-major ids of the tiled prompt)
void launch_tile_prompt_ids(int* output_ids, const int* ids, int B, int K, int S, hipStream_t s);
void launch_fp16_rowmajor_to_tiled(const f16* w, size_t K, size_t N, f16* out, hipStream_t s);

// ---- attention : kernels_attn.hip ----
struct MmhaParams {
    const f16* qkv;       // [B, 3*Hl]
    const f16* qkv_bias;  // [3*Hl]
    f16 *      k_cache, *v_cache;  // [B, nh, s_max, dh]
    const int* seq_len;            // tlength per row
    const int* pad_count;
    const uint8_t* masked_tokens;  // [B, s_max]
    const uint8_t* finished;
    const int*     d_step;  // device step counter (timestep = step - 1); if NULL `step` is used
    const float*   rot_table;  // optional [B][rot/2][2] {cos, sin} of this step's rotary position
    int            step;
    int            B, nh, dh, rot, s_max;
    f16*           ctx;  // [B, Hl]
    unsigned long long* gran;  // split-KV hand-off slab: [B][nh][nsplit][dh+2] {tag,value} granules (zero at request start)
    int            layer;      // tag salt: unique per launch within a token
    int            nsplit;
    // beam search (standalone launch_mmha only): [2][B][s_max] ping-pong planes of the cache indirection, B = batch * beam
    const int*     cache_indir;
    int            beam_width, max_input_len;
    size_t         indir_plane;  // elements per plane
};
size_t mmha_workspace_bytes(int B, int nh, int dh, int nsplit);
int    mmha_pick_nsplit(int B, int nh, int s_max);
size_t mmha_smem_bytes(int dh, int s_max, int nsplit);
void   launch_mmha(const MmhaParams& p, hipStream_t s);
bool   mmha_head_size_supported(int dh);  // the reference's list (DecoderSelfAttentionLayer.cc:280-282)
// {cos, sin}(pos * 10000^(-2j/rot)), pos = step - 1 - pad_count[b], once per token (decoder_masked_multihead_attention_utils.h:1325-1329)
void   launch_rotary_table(float* table, const int* d_step, const int* pad_count, int B, int rot, hipStream_t s);
// launch_step_embedding + launch_rotary_table in one launch
void   launch_step_prologue(f16* out, const f16* table, const int* output_ids, const int* d_step, float* rot_table,
                            const int* pad_count, int B, int H, int rot, hipStream_t s, const int* d_stop = nullptr);
// (d_stop, here and below: device flag "every row has finished" -- the kernel returns at once when it is set)
void   launch_context_attention(const f16* qkv, const f16* qkv_bias, const int* input_lengths, f16* k_cache,
                                f16* v_cache, int B, int S, int nh, int dh, int rot, int s_max, f16* ctx,
                                hipStream_t s, int cache_row_mult = 1, int s_lo = 0,
                                int s_hi = -1);  // K/V of prompt row b live in cache row b * mult

// paged decoder attention (continuous batching front end): per-slot page tables into a shared K/V pool
struct MmhaPagedParams {
    const f16*     qkv;       // [B, 3*Hl]
    const f16*     qkv_bias;  // [3*Hl]
    f16 *          kpool, *vpool;  // THIS layer's pool: [num_pages][nh][P][dh]
    const int*     page_table;     // [B][max_pages]
    const int*     len;            // [B] cached tokens of the slot (= position of the current token)
    const uint8_t* finished;       // [B] 1: slot empty or finished (skipped)
    int            B, nh, dh, rot, P, max_pages;
    f16*           ctx;  // [B, Hl]
};
size_t mmha_paged_smem_bytes(int dh, int max_pages, int max_len);
void   launch_mmha_paged(const MmhaPagedParams& p, int max_len, hipStream_t s);
void   launch_scatter_kv_to_pages(const f16* kc, const f16* vc, f16* kpool, f16* vpool, const int* pages, int L, int nh, int dh,
                                  int s_max, int S, int P, size_t pool_layer_elems, hipStream_t s, size_t src_layer_elems = 0);

// ---- fused attention + FFN1 weight stream : kernels_fused.hip ----
void launch_mmha_ln_gemv(const MmhaParams& ap, const LnGemvParams& gp, bool int8, int M, hipStream_t s);

// ---- persistent decode layers : kernels_persist.hip ----
// One launch runs layers [l_begin, l_end) of the m <= 2 decode step on ONE resident workgroup per CU; the stages of a
// layer hand their vectors over inside the launch (sc1 stores + counters, {tag,value} granules) and every wave already
// holds the first weight batches of the next stage while the hand-off is in flight.
struct PersistLayer {
    const f16 *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const void *w_qkv, *w_ffn1, *w_out, *w_ffn2;  // engine tile images
    const f16 *s_qkv, *s_ffn1, *s_out, *s_ffn2;   // int8 scales
    const f16 *b_qkv, *b_ffn1, *b_res;            // qkv bias (added by attention), ffn1 bias, residual-epilogue bias
    f16 *      k_cache, *v_cache;                 // [B, nh, s_max, dh] of this layer
};
struct PersistPlan {
    int    ok;        // 0: shape not eligible (caller uses the per-kernel path)
    int    NB;        // workgroups (<= CUs)
    int    nsplit;    // split-KV factor of the attention stage
    int    uk;        // K / V wave-loads per lane of the attention stage (8: <= 256 keys per split, 12: <= 384)
    int    PA, PB;    // K pieces per 16-column group of out-proj / FFN2
    int    RLa, RLb;  // tiles per piece
    int    xs_halves; // LDS x region
    int    e1, e3;    // tile-table entries per wave (P1 / P3)
    int    cs1, cs3;  // stream share of a control wave in 1/16 of a streamer wave's (P1 / P3)
    int    qrot;      // rotation of the QKV column-group split over the workgroups (which ones get the lighter P1 share)
    int    own;       // P3 in the own-group layout (whole column groups per workgroup: one hop at the layer boundary)
    long   gp_n;      // granules of the partial-sum slab (PersistParams::gp)
    size_t smem;
};
struct PersistParams {
    const PersistLayer* layers;  // device array [L]
    int                 L, l_begin, l_end;
    const f16*          x_in;        // [M][H] input of layer l_begin (plain memory)
    f16*                x_out;       // [M][H] output of layer l_end-1
    // granule slabs (pairs of halves unless noted): qkv [M][3Hl/2], mid [M][Il/2], ctx [M][Hl/2], x' [M][H/2],
    // K pieces [NG*(PA+PB)][M*16] (fp32; own-group layout: plan.gp_n granules), attention partials [B][nh][nsplit][dh+2] (fp32); zero at request start
    unsigned long long *gq, *gm, *gc, *gx, *gp, *ga;
    int*                err;              // sticky give-up flag (0 = fine)
    int                 H, Hl, Il, nh, dh, rot, s_max, B, tp;
    PersistPlan         plan;
    const int*          d_step;
    const int*          d_stop;  // optional device flag "every row has finished": the launch returns at once when it is set
    const int*          seq_len;
    const int*          pad_count;
    const uint8_t*      masked_tokens;
    const uint8_t*      finished;
    const float*        rot_table;
    float               eps;
    long long*          ts;          // optional [NB][L][8 waves][16] wall-clock stamps (100 MHz), or NULL
    // tensor parallel (tp > 1): this rank and the exchange windows of ALL ranks as this rank addresses them (own memory
    // for xw[tp_rank], peer mappings otherwise); window layout [tp source ranks][M*H/2] granules of {tag, pair of halves}
    int                 tp_rank;
    unsigned long long* xw[8];
    // the plan's run / tile tables, precomputed once per plan: [NB][persist_table_bytes(plan)] ; tab_mode 0: none (build in
    // the launch), 1: build and store (a launch over no layers), 2: load
    char*               tab;
    int                 tab_mode;
};
constexpr int PERSIST_MAX_TP = 8;
struct PersistGroupParams {  // local group launch: every rank's parameters, nb workgroups each
    PersistParams p[PERSIST_MAX_TP];
    int           world, nb;
};
PersistPlan persist_plan(int B, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8, int num_cu, int force_nb,
                         int cs1, int cs3, int own = 0, int L = 0);
// every workgroup of the plan's grid resident at once on this device?  (also raises the kernel's dynamic-LDS limit there)
bool        persist_resident(const PersistPlan& pl, bool int8, int M, int dh, int num_cu, int tp = 1);
// bytes of the table region one workgroup stores / loads (PersistParams::tab holds NB of them)
size_t      persist_table_bytes(const PersistPlan& pl);
void        launch_decode_persistent(const PersistParams& p, bool int8, hipStream_t s);
// all ranks of a local group in one launch (grid = world * NB): residency of the whole group
bool        persist_group_resident(const PersistPlan& pl, bool int8, int M, int dh, int num_cu, int world);
void        launch_decode_persistent_group(const PersistGroupParams& g, bool int8, hipStream_t s);
// tensor-parallel instantiations (kernels_persist_tp.hip); nullptr when the shape has none
const void* persist_tp_kernel(bool int8, int M, int dh, int uk, bool group, int own = 0);
// own-group instantiations for one GPU (kernels_persist_own.hip)
const void* persist_own_kernel(bool int8, int M, int dh, int uk);

// ---- persistent decode layers for 3..16 rows : kernels_rows.hip ----
// One launch runs layers [l_begin, l_end) of the batched decode step (GptNeoXDecoder.cc:245-384, any B <= 16) on one resident
// 8-wave workgroup per CU.  A layer is five chip-wide streams -- QKV, FFN1, the K/V rows of the attention, FFN2, out-proj --
// in that order, so that every hand-off but the layer boundary travels under the stream that follows its producer.  Weight
// tiles are walked k-major over up to five 16-column groups per workgroup with the rows' MFMA A fragments fetched per k-step
// from L2 (sc1 loads of the producers' write-through stores); hand-offs are flags behind drained write-through stores.
struct RowsPlan {
    int    ok;
    int    NB;            // workgroups (<= CUs)
    int    g1;            // kernel instantiation: column groups per k-step of the QKV pass (4 or 5)
    int    CB, KP2, KP3;  // row-parallel GEMMs: column blocks of RW_G groups, K pieces of FFN2 / out-proj
    int    U;             // attention units of a workgroup at most: whole (row, head) pairs (pair = workgroup + u * NB) + one part
    int    FX;            // ... of a leftover pair: each of the M nh % NB leftover pairs is shared by FX workgroups
    int    CR, cw;        // the layer boundary: column ranges per row (merger = row * CR + range), columns per range
    int    wcum[8];       // K shares of the seven streamer waves: wave s streams [n wcum[s] / wcum[7], n wcum[s + 1] / wcum[7]) of a
                          // workgroup's n k-steps (equal shares; unequal ones measured no better: profiles/r05_notes.md)
    size_t smem;
};
struct RowsParams {
    const PersistLayer* layers;  // device array [L] (k_cache / v_cache: this layer's [B][nh][s_max][dh])
    int                 L, l_begin, l_end;
    const f16*          x_in;   // [M][H] input of layer l_begin (plain memory)
    f16*                x_out;  // [M][H] output of layer l_end - 1
    // the hand-off region: ONE buffer (one descriptor inside the kernel), byte offsets of its parts
    char*               ws;
    unsigned            ws_bytes;
    unsigned            o_fq, o_fm, o_fc, o_f2, o_f3;  // [NB] flags: q|k|v / mid / ctx / FFN2 piece / out-proj piece of a workgroup
    unsigned            o_xs;     // [2][M * CR] 16-byte granules {tag, sum, sum of squares, 0} of a merger's piece of x'
    unsigned            o_xb[2];  // layer inputs handed over inside the launch (layer l reads xb[l & 1]): [M][H] halves
    unsigned            o_qkv, o_mid, o_ctx;  // [M][3 Hl], [M][Il], [M][Hl] halves
    unsigned            o_p2, o_p3;           // fp32 partial sums of FFN2 / out-proj: [KP][M][H]
    unsigned            o_fa, o_pa;           // shared pairs: [NB] flags, [NB][dh + 4] {out, max, sum} of a part
    int*                err;                  // (= ws)
    int                 M, H, Hl, Il, nh, dh, rot, s_max, tp;
    RowsPlan            plan;
    const int*          d_step;
    const int*          d_stop;
    const int*          seq_len;
    const int*          pad_count;
    const int*          input_lengths;  // keys [input_lengths[b], max_input_len) of row b are padding (never attended)
    int                 max_input_len;
    const uint8_t*      finished;
    const float*        rot_table;
    float               eps;
    long long*          ts;  // optional [NB][L][8][16] stamps
    // paged K/V (continuous batching): page_table != NULL -> row b's key t lives in page page_table[b * max_pages + t / P] of
    // the layer's pool [num_pages][nh][P][dh] (k_cache / v_cache of the layer table point at the pools)
    const int*          page_table;
    int                 page_tokens, max_pages;
};
RowsPlan rows_plan(int M, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8, int num_cu, int force_nb);
bool     rows_resident(const RowsPlan& pl, bool int8, int dh, int num_cu, bool paged);  // (the K/V form the caller will launch)
// bytes of the hand-off region (flags first: rows_flag_bytes() of it must be zero when a request begins) and its carve
size_t   rows_workspace_bytes(const RowsPlan& pl, int M, int H, int Hl, int Il, int nh, int dh);
size_t   rows_flag_bytes(const RowsPlan& pl, int M, int nh);
void     rows_carve(RowsParams& p, void* workspace);  // fills the buffer pointers of p (M, H, ..., plan set)
void     launch_decode_rows(const RowsParams& p, bool int8, hipStream_t s);
int      rows_paged_block(int dh);  // page_tokens of a paged K/V cache must be a multiple of this

// ---- fp32 instantiation (FTGptNeoX<float>, GptNeoXOp.cc:56-70) : kernels_fp32.hip ----
struct Mmha32Params {
    const float*   qkv;       // [B, 3*Hl]
    const float*   qkv_bias;  // [3*Hl]
    float *        k_cache, *v_cache;  // [B, nh, s_max, dh]
    const int*     seq_len;
    const int*     pad_count;
    const uint8_t* masked_tokens;  // [B, s_max]
    const uint8_t* finished;
    const int*     d_step;
    int            B, nh, dh, rot, s_max;
    float*         ctx;  // [B, Hl]
    const int*     cache_indir;  // beam search: [2][B][s_max], or NULL
    int            beam_width, max_input_len;
    size_t         indir_plane;
};
void launch32_prompt_embedding(float* out, int* output_ids, const float* table, const int* ids, int B, int S, int H,
                               hipStream_t s);
void launch32_step_prologue(float* out, const float* table, const int* output_ids, const int* d_step, float* rot_table,
                            const int* pad_count, int B, int H, int rot, hipStream_t s);
void launch32_gather_last_token(float* out, const float* hidden, const int* input_lengths, int B, int S, int H,
                                hipStream_t s, int tile = 1);
void launch32_add_bias_residual(float* out, const float* a, const float* b, const float* bias, int m, int n, hipStream_t s);
void launch32_gemm(const float* A, const float* W, const float* bias, int act, float* C, int m, int n, int k, hipStream_t s);
void launch32_lm_head(const float* A, const float* W_nk, float* logits, int m, int n, int k, int ldc, hipStream_t s);
void launch32_context_attention(float* qkv, const float* qkv_bias, const int* input_lengths, float* k_cache, float* v_cache,
                                int B, int S, int nh, int dh, int rot, int s_max, float* ctx, hipStream_t s,
                                int cache_row_mult = 1);
void launch32_mmha(const Mmha32Params& p, hipStream_t s);

// ---- dynamic decode : kernels_sampling.hip ----
struct DecodeState {  // device resident, one per engine
    int step;         // current step (max_input_len .. total-1)
    int all_finished;
    int steps_done;
    int pad;          // ticket counter of k_greedy_decode (zero between launches)
};
struct SamplingParams {
    float*       logits;  // [B, V] fp32 (modified in place)
    int          B, V;
    int          max_input_len, total_len, end_id;
    const int*   input_lengths;
    const int*   top_k;        // device [B] effective k (0 => row belongs to the top-p layer)
    int          max_top_k, any_top_p;  // host view of the same: largest k of the batch, any top-p row (LDS sizing)
    const float* top_p_topk;   // device [B] p used by the top-k layer
    const float* top_p_topp;   // device [B] p used by the top-p layer
    const float* temperature;  // device [B]
    const float* repetition_penalty;  // device [B] or NULL
    const int*   min_length;          // device [B] or NULL
    const uint64_t* random_seed;      // device [B]
    uint64_t*       draw_counter;     // device [B]
    int             apply_temperature, apply_repetition;  // host decided (BaseSamplingLayer.cc:283-313 ALL_OF logic)
    const int*      stop_words;       // device [B,2,stop_len] or NULL
    int             stop_len;
    const int*      optional_last_tokens;  // device [B,M] or NULL
    int             optional_count;
    int             return_cum_log_probs;
    int*            output_ids;  // time-major [total, B]
    uint8_t*        finished;
    int*            seq_len;
    float*          cum_log_probs;
    int*            pad_count;
    DecodeState*    state;
    int*            h_flags;  // pinned host mirror: [0] = all_finished, [1] = step that produced it
    void*           ws;       // workspace (see sampling_workspace_bytes)
    // continuous batching: every row has its OWN step -- row b's token history is output_ids[0 .. row_len[b]] (time-major
    // as always), the new token goes to position row_len[b] + 1.  NULL: the batch-wide state->step
    const int*      row_len;
    // k_greedy_decode only: the NEXT token's prologue (k_step_prologue: embedding row of the token just chosen + the rotary table
    // of the next step) done by the workgroup that finishes the step; next_x NULL: the engine launches k_step_prologue itself
    f16*            next_x;     // [B, H]
    const f16*      wte;        // [V, H]
    float*          rot_table;  // [B, rot / 2, 2]
    int             H, rot;
};
// does launch_dynamic_decode(p, s, finish) take the one-launch path of an all-greedy batch (k_greedy_decode)?
bool dynamic_decode_is_fused(const SamplingParams& p, bool finish);
// beam search (beam_width > 1): OnlineBeamSearchLayer semantics, rows bb = batch * K + beam
constexpr int BEAM_MAX_K = 64;  // online_softmax_beamsearch_kernels.cu:691-695
struct BeamParams {
    float*       logits;  // [B*K, V] fp32 (modified in place)
    int          B, K, V;
    int          max_input_len, total_len, end_id, s_max;
    const int*   input_lengths;       // device [B*K] (tiled)
    const float* temperature;         // device [B]
    const float* repetition_penalty;  // device [B] or NULL
    const float* diversity_rate;      // device [B]
    const float* len_penalty;         // device [B]
    const int*   min_length;          // device [B] or NULL
    const int*   stop_words;          // device [B,2,stop_len] or NULL
    int          stop_len;
    const int*   optional_last_tokens;  // device [B,M] or NULL
    int          optional_count;
    int *        output_ids, *parent_ids;  // time-major [total, B*K]
    uint8_t*     finished;
    int*         seq_len;
    float*       cum_log_probs;
    int*         cache_indir;  // [2][B*K][s_max]: plane (step - max_input_len) % 2 is read, the other one written
    DecodeState* state;
    void*        ws;  // beam_workspace_bytes
};
size_t beam_workspace_bytes(int B, int K);
void   launch_beam_search(const BeamParams& p, hipStream_t s);  // rows + batch kernels (not the finish step)
void   launch_decode_finish(const SamplingParams& p, hipStream_t s);
void   launch_tile_inputs(int* tiled_ids, int* tiled_len, const int* ids, const int* len, int B, int K, int S,
                          hipStream_t s);
void   launch_gather_tree_beam(int* output_ids, int* sequence_lengths, const int* step_ids, const int* parent_ids,
                               const int* seq_len, const int* input_lengths, int B, int K, int max_input_len, int total,
                               int end_id, hipStream_t s);
size_t sampling_workspace_bytes(int B, int V);
void   launch_dynamic_decode(const SamplingParams& p, hipStream_t s, bool finish = true);
// LM head (final LayerNorm fused, as launch_lm_head with gamma) + the all-greedy dynamic decode of the token + the next token's
// prologue in ONE launch (k_lm_head_greedy): one GPU, <= 2 rows, a step launch_dynamic_decode would run as k_greedy_decode.
// The workgroups' partials are tagged with the step: the first lm_head_greedy_partial_bytes(B) bytes of the sampling workspace
// must be zero at the start of a request.
bool   lm_head_greedy_ok(const SamplingParams& p, int K);
size_t lm_head_greedy_partial_bytes(int B);
void   launch_lm_head_greedy(const f16* x, const f16* W, float* logits, int K, const f16* gamma, const f16* beta, float eps,
                             const SamplingParams& p, hipStream_t s);
void   launch_decode_init(uint8_t* finished, int* seq_len, float* cum_log_probs, int* pad_count, uint8_t* masked_tokens,
                          uint64_t* draw_counter, const int* input_lengths, DecodeState* st, int B, int max_input_len,
                          int s_max, hipStream_t s, int beam_width = 1);
void   launch_gather_tree(int* output_ids, int* sequence_lengths, const int* step_ids, const int* seq_len,
                          const int* input_lengths, int B, int max_input_len, int total, int end_id, hipStream_t s);

}  // namespace ftcf
