/*
 * ftcf.h -- C ABI of libftcf.so: the MI355X (gfx950) native engine for the GPT-NeoX / CodeFuse decode path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / pybind types.  The reference's Python
 * extension modules (`libth_gptneox.GptNeoXOp`, `libth_common`) are thin bindings over these entry points
 * (see INTEGRATION.md).  All paths below are relative to the reference tree root.
 *
 * Conventions
 *   - "device pointer" = HIP device memory of the device the handle was created on; "host pointer" = CPU memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - every function returns 0 on success, a negative ftcf_status otherwise; ftcf_last_error() gives the text.
 *     (The reference prints and exit(-1)s on engine errors, th_op/gptneox/GptNeoXOp.h:370-380; the bindings raise.)
 *   - fp16 tensors are IEEE binary16 (`dtype` FTCF_FP16), fp32 tensors `float`.
 */
#ifndef FTCF_H
#define FTCF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTCF_VERSION 100

typedef enum {
    FTCF_OK              = 0,
    FTCF_ERR_INVALID_ARG = -1,
    FTCF_ERR_HIP         = -2,
    FTCF_ERR_UNSUPPORTED = -3,
    FTCF_ERR_COMM        = -4,
    FTCF_ERR_NO_DEVICE   = -5
} ftcf_status;

typedef enum { FTCF_FP32 = 0, FTCF_FP16 = 1, FTCF_BF16 = 2 /* host quantiser input only */ } ftcf_dtype;
typedef enum { FTCF_ACT_NONE = 0, FTCF_ACT_GELU = 1 } ftcf_act;

const char* ftcf_last_error(void);
int         ftcf_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int ftcf_device_count(void);

/* ================================================================================================
 * libth_common counterpart -- host side weight-only quantiser
 *   replaces: th_op/common/WeightOnlyQuantOps.cc:140-233,344-349
 *             (symmetric_quantize_last_axis_of_batched_matrix_int8) and
 *             kernels/cutlass_kernels/cutlass_preprocessors.cc:576-673 (symmetric_quantize) +
 *             :500-539 (preprocess_weights_for_mixed_gemm -- here: the gfx950 tile layout)
 * ================================================================================================ */
/* weight: host [E, K, N] row major (E = 1 for a 2-D matrix), dtype FTCF_FP32, FTCF_FP16 or FTCF_BF16
 * (WeightOnlyQuantOps.cc:149,205).
 * out_q : host int8 [E, K, N] bytes, ENGINE-PRIVATE gfx950 tile layout (opaque, like the reference's).
 * out_scale: host [E, N] in the weight dtype.  Requires K % 64 == 0 and N % 16 == 0. */
int ftcf_symmetric_quantize_int8(const void* weight, ftcf_dtype dtype, size_t E, size_t K, size_t N, int8_t* out_q,
                                 void* out_scale);
/* row-major int8 [K,N] (the reference's "unprocessed" tensor) <-> engine tile layout (host) */
int ftcf_int8_rowmajor_to_tiled(const int8_t* q_rowmajor, size_t K, size_t N, int8_t* q_tiled);
int ftcf_int8_tiled_to_rowmajor(const int8_t* q_tiled, size_t K, size_t N, int8_t* q_rowmajor);
/* int8 [K,N] as a CUDA build of the reference stores it for SM75..SM89 -- what its `.q.bin` files hold
 * (preprocess_weights_for_mixed_gemm, cutlass_preprocessors.cc:500-539; quant_and_save.py:20) -- <-> row major (host).
 * K % 64 == 0, N % 2 == 0.  Importing a CUDA checkpoint = cuda_sm80_to_rowmajor followed by rowmajor_to_tiled. */
int ftcf_int8_cuda_sm80_to_rowmajor(const int8_t* q_cuda, size_t K, size_t N, int8_t* q_rowmajor);
int ftcf_int8_rowmajor_to_cuda_sm80(const int8_t* q_rowmajor, size_t K, size_t N, int8_t* q_cuda);
/* device: fp16 [K,N] row major -> engine fp16 tile layout (out-of-place, K % 32 == 0, N % 16 == 0) */
int ftcf_fp16_rowmajor_to_tiled(const void* w_rowmajor, size_t K, size_t N, void* w_tiled, void* stream);

/* ================================================================================================
 * kernel-level entry points (device pointers).  One per reference `invoke*` / runner on the hot path;
 * used by the parity tests and by the engine itself.
 * ================================================================================================ */
/* CutlassFpAIntBGemmRunner<half,uint8_t>::gemm / gemm_bias_act
 *   (kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm.h:39-106, fpA_intB_gemm_template.h:511-581):
 *   C[m,n] = half( sum_k A[m,k] * half(q[k,n]*scale[n])  (+bias[n], gelu) ), A/C/scale/bias fp16, q tiled int8. */
int ftcf_fpA_intB_gemm(const void* A, const int8_t* B_tiled, const void* scales, const void* bias, ftcf_act act,
                       void* C, int m, int n, int k, void* stream);
/* cublasMMWrapper::Gemm (utils/cublasMMWrapper.cc:94-386) for the fp16 engine: C[m,n] = half(A[m,k] * W), W tiled fp16;
 * optional fused bias+gelu reproduces invokeAddBiasGeluV2 (kernels/activation_kernels.cu:401-426). */
int ftcf_fp16_gemm(const void* A, const void* W_tiled, const void* bias, ftcf_act act, void* C, int m, int n, int k,
                   void* stream);
/* LM head (models/gptneox/GptNeoX.cc:866-912): logits_f32[m, n] = A[m,k] (fp16) x W[n,k]^T (fp16, row major [V,H]) */
int ftcf_lm_head(const void* A, const void* W_nk, float* logits, int m, int n, int k, int ldc, void* stream);
/* invokeGeneralLayerNorm (kernels/layernorm_kernels.cu:1652-1735), fp16 half2 path numerics (:157-286) */
int ftcf_layernorm(const void* x, const void* gamma, const void* beta, void* out, int m, int n, float eps,
                   ftcf_dtype dtype, void* stream);
/* invokeAddBiasAttentionFfnResidual (kernels/add_residual_kernels.cu:116-178) */
int ftcf_add_bias_attn_ffn_residual(void* out, const void* ffn, const void* attn, const void* in, const void* bias,
                                    int m, int n, int tp, int inplace_variant, ftcf_dtype dtype, void* stream);
/* fusedQKV_masked_attention_dispatch (layers/attention_layers/DecoderSelfAttentionLayer.cc:36-146 ->
 * kernels/decoder_masked_multihead_attention/decoder_masked_multihead_attention_template.hpp:1099-1919).
 * qkv [B,3*Hl] fp16; caches are engine private: k_cache/v_cache [B, nh, s_max, dh] fp16. */
int ftcf_masked_multihead_attention(const void* qkv, const void* qkv_bias, void* k_cache, void* v_cache,
                                    const int* seq_len, const int* pad_count, const uint8_t* masked_tokens,
                                    const uint8_t* finished, int B, int nh, int dh, int rot, int s_max, int step,
                                    void* ctx, void* workspace, size_t workspace_bytes, void* stream);
size_t ftcf_masked_multihead_attention_workspace(int B, int nh, int dh, int s_max);
/* GptContextAttentionLayer<T>::forward minus the two projections (layers/attention_layers/
 * GptContextAttentionLayer.cc:142-345): bias + NeoX rotary + cache fill + causal masked softmax(QK^T)V.
 * qkv [B*S, 3*Hl] fp16 (row = b*S+s), ctx [B*S, Hl] fp16. */
int ftcf_context_attention(const void* qkv, const void* qkv_bias, const int* input_lengths, void* k_cache,
                           void* v_cache, int B, int S, int nh, int dh, int rot, int s_max, void* ctx, void* stream);

/* ================================================================================================
 * libth_gptneox counterpart -- the engine behind GptNeoXOp
 *   replaces: th_op/gptneox/GptNeoXOp.h:69-231 (FTGptNeoX ctor), :246-381 (forward),
 *             models/gptneox/GptNeoX.cc:386-1052 (GptNeoX<T>::forward)
 * ================================================================================================ */
typedef struct ftcf_gptneox* ftcf_gptneox_t;
typedef struct ftcf_comm*    ftcf_comm_t;

/* ---- tensor-parallel communicator (utils/nccl_utils.cc:56-435, th_op/gptneox/utils/nccl_inherit_utils.cc:25-68) ---- */
#define FTCF_UNIQUE_ID_BYTES 128
/* rank 0 creates the id, the caller broadcasts the bytes (e.g. torch.distributed), every rank inits. */
int ftcf_comm_get_unique_id(uint8_t id[FTCF_UNIQUE_ID_BYTES]);
int ftcf_comm_init(const uint8_t id[FTCF_UNIQUE_ID_BYTES], int world_size, int rank, int device, ftcf_comm_t* comm);
int ftcf_comm_destroy(ftcf_comm_t comm);
/* LOCAL GROUP (test infrastructure): the ranks of a tensor-parallel job inside ONE process on ONE device, each driven by
 * its own host thread.  Same engine code path as RCCL ranks (sharding, per-layer all-reduce, vocabulary split, in-kernel
 * exchange); the collectives are host-synchronous.  Lets a single-GPU box execute and check tensor_para_size > 1. */
/* HOST-EXCHANGE communicator: one process per rank as with ftcf_comm_init, but every exchange between the ranks -- the
 * bootstrap, the hipIpc handles of the exchange windows, agreements, barriers, and (staged through host memory) the
 * all-reduce / all-gather of device buffers -- travels through ONE callback of the caller: an all-gather of host bytes
 * over the caller's own process group (e.g. torch.distributed with gloo).  This is what nccl_inherit_utils.cc:25-68 does
 * with the caller's ProcessGroup for the bootstrap, taken one step further: no RCCL communicator is created at all, so the
 * ranks may share one device (the inter-PROCESS path of the in-kernel exchange -- IPC-mapped windows, hand-shake, system
 * scope stores -- runs on a single-GPU box), and a box whose RCCL cannot initialise still gets tensor parallelism.  The
 * decode all-reduce is in the persistent kernel's exchange windows as usual; prefill collectives are host staged (slow).
 * allgather(user, send, recv, bytes): recv[r * bytes .. ] = rank r's send; returns 0 on success. */
typedef int (*ftcf_host_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
int ftcf_comm_init_host_exchange(int world_size, int rank, int device, ftcf_host_allgather_fn allgather, void* user,
                                 ftcf_comm_t* comm);
int ftcf_comm_local_unique_id(uint8_t id[FTCF_UNIQUE_ID_BYTES]);
int ftcf_comm_init_local(const uint8_t id[FTCF_UNIQUE_ID_BYTES], int world_size, int rank, int device, ftcf_comm_t* comm);
/* ftNcclAllReduceSum / ftNcclAllGather (in place, fp16 / fp32) exposed for tests */
int ftcf_comm_allreduce_sum(ftcf_comm_t comm, void* buf, size_t count, ftcf_dtype dtype, void* stream);
int ftcf_comm_allgather(ftcf_comm_t comm, void* buf, size_t count_per_rank, ftcf_dtype dtype, void* stream);

typedef struct {
    int head_num, size_per_head, inter_size, num_layer, vocab_size, rotary_embedding_dim;
    int start_id, end_id;
    int tensor_para_size, tensor_para_rank, pipeline_para_size; /* pipeline_para_size must be 1 */
    int int8_mode;                                               /* 0, or 1 = weight only */
    int dtype; /* ftcf_dtype of `weights` = the engine instantiated, as GptNeoXOp.cc:56-105 selects it from weights[0]:
                * FTCF_FP16 (every decode path, int8_mode 0 / 1) or FTCF_FP32 (FTGptNeoX<float>: fp32 weights, activations and
                * K/V cache, general path, int8_mode 0 only -- the validation instantiation, not tuned) */
    int use_gptj_residual;
    int device;        /* HIP device ordinal */
    void* stream;      /* hipStream_t all work is enqueued on (GptNeoXOp.h:180-185) */
    ftcf_comm_t comm;  /* NULL when tensor_para_size == 1 */
    int use_hip_graph; /* 1: capture the per-token step in a hipGraph when possible */
} ftcf_gptneox_config;

/* Weight contract of GptNeoXOp (GptNeoXOp.h:121-174): `weights` = 12*L+4 device pointers in the order
 * [ln1.beta xL, ln1.gamma xL, qkv.kernel xL, qkv.bias xL, attn_out.kernel xL, attn_out.bias xL, ffn1.kernel xL,
 *  ffn1.bias xL, ffn2.kernel xL, ffn2.bias xL, ln2.beta xL, ln2.gamma xL, wte, final_ln.gamma, final_ln.beta, lm_head];
 * kernels are [K, N/TP] row major; a pointer may be NULL where the reference passes an empty tensor.
 * `int8_weights` = 4*L tiled int8 tensors [qkv xL, attn_out xL, ffn1 xL, ffn2 xL], `scales` = 4*L fp16 vectors.
 * The engine keeps the pointers (the binding keeps the tensors alive, GptNeoXOp.h:402-404). */
typedef struct {
    const void* const* weights;
    int                n_weights;
    const void* const* int8_weights;
    int                n_int8_weights;
    const void* const* scales;
    int                n_scales;
} ftcf_gptneox_weights;

/* per-step streaming callback (th_op/gptneox/utils/pybind_callback_utils.cc:22-103): called on rank 0 after every
 * step but the last with host arrays last_tokens[B*beam] and idxs[B*beam]. */
typedef void (*ftcf_token_callback)(const int* last_tokens, const int* idxs, int batch, int beam, void* user);

typedef struct {
    /* inputs (GptNeoXOp.cc:113-185) */
    const int* input_ids;     /* device [B, max_input_len] */
    const int* input_lengths; /* device [B] */
    int        batch_size, max_input_len, output_len, beam_width;
    /* runtime args: host arrays of size 1 or B; n == 0 means "not given" */
    const int*      top_k;                      int n_top_k;
    const float*    top_p;                      int n_top_p;
    const float*    beam_search_diversity_rate; int n_beam_search_diversity_rate;
    const float*    temperature;                int n_temperature;
    const float*    len_penalty;                int n_len_penalty;
    const float*    repetition_penalty;         int n_repetition_penalty;
    const uint64_t* random_seed;                int n_random_seed;
    const int*      min_length;                 int n_min_length; /* not reachable through GptNeoXOp; kept for parity */
    const int* stop_words_list;      /* device [B, 2, stop_words_len] or NULL */
    int        stop_words_len;
    const int* optional_last_tokens; /* device [B, optional_last_tokens_count] (-1 padded) or NULL */
    int        optional_last_tokens_count;
    int        return_cum_log_probs;
    ftcf_token_callback callback;
    void*               callback_user;
    /* outputs (device) */
    int*   output_ids;       /* [B, beam, max_input_len + output_len] */
    int*   sequence_lengths; /* [B, beam] */
    float* cum_log_probs;    /* [B, beam] or NULL */
    /* optional debug taps (device, may be NULL): raw fp32 logits of every step [output_len, B, V] */
    float* debug_logits;
} ftcf_forward_args;

typedef struct {
    float prefill_ms;      /* HIP-event time of the context phase of the last forward */
    float decode_ms;       /* HIP-event time of the token loop of the last forward */
    int   decode_steps;    /* executed loop iterations */
    float gemv_ms_sum;     /* sum over timed weight-streaming launches (only when profiling is enabled) */
    long  gemv_launches;
    double gemv_bytes;     /* algorithmic weight bytes those launches streamed */
    int   gemv_kind;       /* which launch kind those three describe: 0 LN->QKV, 1 out-proj+FFN2, 2 LM head,
                              3 MMHA||FFN1, 4 persistent decode layers, 5 batched-decode burst GEMM pair */
    int   decode_path;     /* decoder of the last request: 0 per-stage launches, 1 persistent layers (one launch per
                              token, or per layer with tensor parallelism), 2 general (batched GEMM) path */
    /* tensor parallel, prompt phase: did the last request run its per-layer all-reduce on the side stream under the other
     * micro-batch's GEMMs (FTCF_PREFILL_OVERLAP = 1, or chosen by the auto mode), and what the auto mode's two timed trials
     * took (plain / overlapped, ms, the slowest rank's; 0 until that trial has run) */
    int   prefill_overlap;
    float prefill_ms_plain, prefill_ms_overlapped;
    /* all-reduces of the last request that went through the peer-mapped exchange windows instead of RCCL (the two-shot kernel
     * for prompt-phase messages; FTCF_TP_WINAR=0 switches it off, FTCF_TP_WINAR_MB sizes its buffers: 16) */
    int   window_allreduces;
    /* tensor parallel, batched decode (4..32 rows on the general path): did the last request run the layer's all-reduce on the
     * side stream under the other micro-batch's launches (FTCF_DECODE_OVERLAP = 1, or chosen by the auto mode), and the auto
     * mode's two trials (ms per decode step, plain / overlapped, the slowest rank's; 0 until that trial has run) */
    int   decode_overlap;
    float decode_step_ms_plain, decode_step_ms_overlapped;
    /* decode_path 1: how the one- / two-row kernel ran its out-proj / FFN2 stage: 0 K pieces merged by an owner (two hops at the
     * layer boundary), 1 own-group layout (one hop; FTCF_PERSIST_OWN) */
    int   persist_layout;
} ftcf_forward_stats;

int ftcf_gptneox_create(const ftcf_gptneox_config* cfg, const ftcf_gptneox_weights* w, ftcf_gptneox_t* out);
int ftcf_gptneox_forward(ftcf_gptneox_t h, const ftcf_forward_args* args);
/* The same request split in three so that a caller can stream or time the token loop:
 * forward(args) == begin(args) [buffers, runtime args, prefill] ; step(output_len) [token loop, stops early when every
 * row finished] ; finish() [gatherTree + outputs].  `args` pointers must stay valid until finish(). */
int ftcf_gptneox_begin(ftcf_gptneox_t h, const ftcf_forward_args* args);
int ftcf_gptneox_step(ftcf_gptneox_t h, int max_steps, int* steps_done);
int ftcf_gptneox_finish(ftcf_gptneox_t h);
int ftcf_gptneox_get_stats(ftcf_gptneox_t h, ftcf_forward_stats* stats);
/* enable HIP-event timing around every weight-streaming launch of the decode loop (bench.py roofline leg) */
int ftcf_gptneox_set_profiling(ftcf_gptneox_t h, int enabled);
int ftcf_gptneox_destroy(ftcf_gptneox_t h);


/* ---- continuous batching over a paged K/V cache (no counterpart in the reference, whose serving layer
 * triton_backend/gptneox/ allocates the cache per request, models/gptneox/GptNeoX.cc:84-156) --------------------------------
 * A batcher borrows an engine (which must outlive it and must not run a request of its own while a batcher call is in
 * progress).  K/V live in `num_pages` pages of `page_tokens` tokens shared by all sequences; up to `max_batch` sequences
 * decode together; waiting requests are admitted, in order, as soon as a slot and the pages for prompt + max_new_tokens are
 * free.  fp16 / int8 engines, parallel residual, any tensor_para_size (one batcher per rank, fed the same requests in the same
 * order); sampling: top_k / top_p / temperature / repetition penalty / stop words, and beam search (ftcf_batcher_submit_beam). */
typedef struct ftcf_batcher* ftcf_batcher_t;
int ftcf_batcher_create(ftcf_gptneox_t engine, int max_batch, int page_tokens, int num_pages, int max_seq_len,
                        ftcf_batcher_t* out);
/* prompt_ids: HOST array.  (top_k, top_p) = (0, 0) is greedy, as in the reference's sampling layer. */
int ftcf_batcher_submit(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int top_k, float top_p,
                        float temperature, unsigned long long seed, long* request_id);
/* The same with the request's repetition penalty (sampling_penalty_kernels.cu:367-425: every token of prompt + output so
 * far, once) and stop words: `stop_words` is a HOST int array [2][stop_len] in the reference's to_word_list_format layout
 * (codefuse_example.py:26-53: row 0 the words' ids back to back, row 1 their cumulative end offsets, -1 padded), or NULL.
 * A request ends AFTER a stop sequence has been emitted (stop_criteria_kernels.cu:24-83): the event of its last token
 * carries finished = 1.  stop_len <= 64. */
int ftcf_batcher_submit_ex(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int top_k, float top_p,
                           float temperature, float repetition_penalty, unsigned long long seed, const int* stop_words,
                           int stop_len, long* request_id);
/* A beam-search request (GptNeoXOp.forward with beam_width > 1: OnlineBeamSearchLayer semantics, the engine's own beam kernels):
 * it occupies `beam_width` consecutive slots, its beams share the prompt's pages and every page they have in common (copy on
 * write).  ftcf_batcher_step reports ONE event for it, when it has finished: token = -1, finished = 1; the hypotheses are
 * then fetched ONCE with ftcf_batcher_beam_result: output_ids [beam_width][total_len] (prompt, then the beam's tokens, end_id
 * padded: total_len = prompt_len + max_new_tokens), sequence_lengths [beam_width], cum_log_probs [beam_width] -- the arrays
 * GptNeoXOp.forward returns.  output_ids == NULL: only *beam_width / *total_len are set (0 / 0: unknown id or still running).
 * 2 <= beam_width <= min(64, max_batch).  Up to 256 finished results wait to be fetched (the oldest are dropped beyond
 * that); ftcf_batcher_cancel of a finished request drops its result. */
int ftcf_batcher_submit_beam(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int beam_width,
                             float beam_search_diversity_rate, float len_penalty, float temperature, float repetition_penalty,
                             long* request_id);
/* The same with min_length (the end token is held back for that many new tokens: beam_search_penalty_kernels.cu:155-169; 0 = none)
 * and stop words in the layout of ftcf_batcher_submit_ex (checked along a beam's parent chain: a beam that has emitted a stop
 * sequence is finished). */
int ftcf_batcher_submit_beam_ex(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int beam_width,
                                float beam_search_diversity_rate, float len_penalty, float temperature, float repetition_penalty,
                                int min_length, const int* stop_words, int stop_len, long* request_id);
int ftcf_batcher_beam_result(ftcf_batcher_t b, long request_id, int* output_ids, int* sequence_lengths, float* cum_log_probs,
                             int capacity, int* beam_width, int* total_len);
/* One scheduler iteration: one decode step for the running sequences, then admissions (prefill + first token).  Returns one
 * event per token produced: request id, token, finished (end_id emitted or max_new_tokens reached).  capacity >= 2 * max_batch.
 * With sequences running, a prompt longer than FTCF_BATCHER_PREFILL_CHUNK tokens (default 512; 0 = never) is admitted alone and
 * prefilled in chunks of that many tokens, with one decode step of the running sequences after every chunk but the last: the
 * iteration then produces several tokens per running sequence.  Events that do not fit `capacity` are returned by the next
 * calls, before a new iteration runs (`running` of ftcf_batcher_status counts 1 for them). */
int ftcf_batcher_step(ftcf_batcher_t b, long* request_ids, int* tokens, int* finished, int capacity, int* n_events);
/* Streaming (the reference's token callback, GptNeoX.cc:362-375, 1023 `token_generated_cb_`, per request here): `fn` is called
 * from inside ftcf_batcher_step, on the calling thread, for every event the moment its token is on the host -- i.e. between
 * the chunks of a long admission as well -- and the same events are returned by the step call afterwards.  NULL unsets. */
typedef void (*ftcf_token_callback_fn)(void* user, long request_id, int token, int finished);
int ftcf_batcher_set_token_callback(ftcf_batcher_t b, ftcf_token_callback_fn fn, void* user);
int ftcf_batcher_status(ftcf_batcher_t b, int* waiting, int* running, int* free_pages);
/* drop a waiting or running request (its pages return to the pool at once); *found = 0 when the id is unknown or already done */
int ftcf_batcher_cancel(ftcf_batcher_t b, long request_id, int* found);
int ftcf_batcher_destroy(ftcf_batcher_t b);

#ifdef __cplusplus
}
#endif
#endif /* FTCF_H */
