/*
 * ftcf_oracle.h -- CPU restatement (ORACLE) of the FasterTransformer4CodeFuse GPT-NeoX hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load this library, and only as the checker.
 *
 * Parity pinning: the restatement is pinned (tests/test_oracle_*.py) against
 *   - the reference's own known-answer vectors for the weight-only quantiser / dequant / GEMM tolerance
 *     (tests/weight_only_quant_ops/th_weight_quant_ops_unit_tests.py, tests/gemm_dequantize/th_gemm_dequantize.py),
 *   - golden vectors generated in the build container by importing the reference's Python
 *     (examples/pytorch/codefuse/{codefuse_example,huggingface_convert}.py) and HF GPTNeoXForCausalLM
 *     (tests/golden/make_golden.py writes the .npz fixtures under tests/golden).
 * The CUDA/C++ path itself cannot be compiled here (needs nvcc, cuBLAS, NCCL, un-vendored CUTLASS).
 *
 * All tensors are float32 arrays on the host.  `fp16 != 0` makes every storage point that the reference
 * keeps in `half` round to IEEE binary16 (round-to-nearest-even) so that the reference's rounding points are
 * reproduced; accumulations stay in fp32/fp64 exactly where the reference accumulates in fp32.
 */
#ifndef FTCF_ORACLE_H
#define FTCF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*orc_allreduce_fn)(float* buf, long n, void* ctx);
/* buf holds [tp][n_per_rank]; on entry only slot `rank` is valid, on exit all slots are. */
typedef void (*orc_allgather_fn)(float* buf, long n_per_rank, void* ctx);

typedef struct {
    int head_num, size_per_head, inter_size, num_layer, vocab_size, rotary_dim;
    int start_id, end_id;
    int tp_size, tp_rank;
    int int8_mode;          /* 0 or 1 (weight-only) */
    int fp16;               /* 1: emulate the half engine, 0: float engine */
    int use_gptj_residual;  /* parallel residual */
    orc_allreduce_fn allreduce;
    orc_allgather_fn allgather;
    void* comm_ctx;
} orc_config;

/* Per-rank shard.  Arrays of num_layer pointers; kernels are [K,N] row-major like the reference's
 * torch tensors (GptNeoXOp.h:121-174).  For int8_mode the *_q (int8 [K,N] row-major, UNPROCESSED layout)
 * and *_s (scale [N]) arrays are used and the fp kernels may be NULL. */
typedef struct {
    const float** ln1_g; const float** ln1_b;
    const float** qkv_w; const float** qkv_b;      /* [H, 3*Hl], [3*Hl] */
    const float** out_w; const float** out_b;      /* [Hl, H], [H] (NULL entry if gptj residual) */
    const float** ffn1_w; const float** ffn1_b;    /* [H, Il], [Il] */
    const float** ffn2_w; const float** ffn2_b;    /* [Il, H], [H] */
    const float** ln2_g; const float** ln2_b;
    const int8_t** qkv_q; const float** qkv_s;
    const int8_t** out_q; const float** out_s;
    const int8_t** ffn1_q; const float** ffn1_s;
    const int8_t** ffn2_q; const float** ffn2_s;
    const float* wte;       /* [V, H] */
    const float* final_ln_g; const float* final_ln_b;
    const float* lm_head;   /* [V, H] */
} orc_weights;

typedef struct {
    const int*   top_k;               /* [B] (already broadcast) */
    const float* top_p;               /* [B] */
    const float* temperature;         /* [B] */
    const float* repetition_penalty;  /* [B] or NULL (= penalty type None) */
    const int*   min_length;          /* [B] or NULL */
    const uint64_t* random_seed;      /* [B] */
    const int*   stop_words;          /* [B,2,stop_len] or NULL */
    int          stop_len;
    const int*   optional_last_tokens;/* [B,M] (-1 padded) or NULL */
    int          optional_count;
    int          return_cum_log_probs;
} orc_sampling;

/* ---- scalar helpers ---- */
int      orc_num_threads(void);
float    orc_round_half(float x);      /* F16C where available, else the software conversion */
float    orc_round_half_soft(float x); /* the software conversion (definition) */
uint16_t orc_float_to_half_bits(float x);
float    orc_half_bits_to_float(uint16_t h);
float    orc_uniform(uint64_t seed, uint64_t row, uint64_t draw);   /* (0,1] */

/* ---- quantiser (cutlass_preprocessors.cc:576-673 symmetric_quantize, INT8_WEIGHT_ONLY) ---- */
void orc_symmetric_quantize_int8(const float* w, int K, int N, int weight_is_half, int8_t* q, float* scale);

/* The CUDA build's int8 weight layout for SM75..SM89 (preprocess_weights_for_mixed_gemm, cutlass_preprocessors.cc:500-539),
 * one function per step so that each can be pinned by the reference's own known-answer tests
 * (tests/weight_only_quant_ops/th_weight_quant_ops_unit_tests.py).  All tensors [K rows, N cols] int8. */
void orc_sm80_permute_rows(const int8_t* in, int K, int N, int8_t* out);          /* :139-201, map 0 1 8 9 2 3 10 11 ... */
void orc_sm80_transpose(const int8_t* in, int K, int N, int8_t* out);             /* :207-348 -> column major [N][K] */
void orc_sm80_interleave_columns(const int8_t* in, int K, int N, int8_t* out);    /* :437-498, 64-row tiles, 2 columns */
void orc_sm80_add_bias_interleave_int8(int8_t* inout, size_t n);                  /* :350-370 */
void orc_sm80_preprocess_int8(const int8_t* row_major, int K, int N, int8_t* out); /* :500-539: all four in order */

/* ---- GEMMs ---- */
/* C[m,n] = A[m,k] * W ; W given either as fp [k,n] or as (q int8 [k,n], scale [n]).
 * act: 0 none, 1 gelu(tanh) ; bias may be NULL ; out_fp32: keep fp32 output (LM head). */
void orc_gemm(const float* A, int m, int k, int n, const float* W, const int8_t* q, const float* scale,
              const float* bias, int act, float* C, int fp16, int out_fp32);
/* LM head: logits[m, n] = A[m,k] * Wt[n,k]^T, fp32 out (GptNeoX.cc:866-912) */
void orc_lm_head(const float* A, int m, int k, int n, const float* Wt, float* C);

/* ---- elementwise / norm ---- */
void orc_layernorm(const float* x, const float* gamma, const float* beta, int m, int n, float eps, float* out, int fp16);
void orc_add_bias_gelu(float* x, const float* bias, int m, int n, int fp16);
void orc_add_bias_attn_ffn_residual(float* out, const float* ffn, const float* attn, const float* in, const float* bias,
                                    int m, int n, int tp, int inplace_variant, int fp16);

/* ---- attention ---- */
/* One decode step of masked MHA for all (b, local heads)  (decoder_masked_multihead_attention_template.hpp:1099-1919).
 * qkv [B, 3*Hl] ; k_cache/v_cache [B, NHl, S_max, Dh] ; seq_len[b] = tlength ; out ctx [B, Hl]. */
void orc_mmha_step(const float* qkv, const float* qkv_bias, float* k_cache, float* v_cache, const int* seq_len,
                   const int* pad_count, const uint8_t* masked_tokens, const uint8_t* finished, int B, int nh, int dh,
                   int rot, int s_max, int step, float* ctx, int fp16);
/* Prefill attention for one layer: qkv [B*S, 3*Hl] (padded rows included, row = b*S+s) -> ctx [B*S, Hl];
 * fills k/v caches for positions < S  (GptContextAttentionLayer.cc:101-393 unfused path). */
void orc_context_attention(const float* qkv, const float* qkv_bias, const int* input_lengths, float* k_cache,
                           float* v_cache, int B, int S, int nh, int dh, int rot, int s_max, float* ctx, int fp16);

/* ---- dynamic decode for one step (DynamicDecodeLayer.cc:192-497, beam_width==1) ---- */
/* logits [B, V] fp32 (modified in place like the reference); output_ids time-major [max_seq, B]. */
void orc_dynamic_decode(float* logits, int B, int V, int step, int max_input_len, const int* input_lengths,
                        const orc_sampling* sp, int end_id, int* output_ids, uint8_t* finished, int* seq_len,
                        float* cum_log_probs, uint64_t* draw_counter, int total_len);

/* ---- whole path (GptNeoX.cc:386-1052) ---- */
/* Returns number of executed decode-loop iterations.  dbg_logits (optional) [out_len][B][V] receives the raw
 * fp32 logits of every step before dynamic decode; dbg_hidden (optional) [B][H] the last decoder output. */
/* ---- beam search (beam_width > 1) ---- */
typedef struct {
    const float* temperature;        /* [B] or NULL (1.0) */
    const float* repetition_penalty; /* [B] or NULL (= penalty type None) */
    const float* diversity_rate;     /* [B] or NULL (0.0) */
    const float* len_penalty;        /* [B] or NULL (0.0) */
    const int*   min_length;         /* [B] or NULL */
    const int*   stop_words;         /* [B,2,stop_len] or NULL */
    int          stop_len;
    const int*   optional_last_tokens; /* [B,M] (-1 padded) or NULL */
    int          optional_count;
} orc_beam_params;
void orc_mmha_step_beam(const float* qkv, const float* qkv_bias, float* k_cache, float* v_cache, const int* seq_len,
                        const int* pad_count, const uint8_t* masked_tokens, const uint8_t* finished, int B, int nh, int dh,
                        int rot, int s_max, int step, float* ctx, int fp16, const int* cache_indir, int beam_width);
void orc_decoder_step_beam(const orc_config* c, const orc_weights* w, const float* x_in, float* k_cache, float* v_cache,
                           const int* seq_len, const int* pad_count, const uint8_t* masked_tokens,
                           const uint8_t* finished, int B, int s_max, int step, float* y, const int* cache_indir,
                           int beam_width);
/* one OnlineBeamSearchLayer step; rows bb = batch * K + beam, output_ids / parent_ids time-major [total][B*K],
 * src / tgt cache indirection [B][K][s_max] */
void orc_beam_search_step(float* logits, int B, int K, int V, int step, int max_input_len, const int* input_lengths,
                          const orc_beam_params* bp, int end_id, int* output_ids, int* parent_ids, uint8_t* finished,
                          int* seq_len, float* cum_log_probs, const int* src_indir, int* tgt_indir, int s_max);
void orc_gather_tree_beam(const int* ids, const int* parents, const int* seq_len, const int* t_len, int B, int K,
                          int max_input_len, int total, int end_id, int* output_ids, int* sequence_lengths);
/* output_ids [B][K][S+out_len], sequence_lengths [B][K], cum_log_probs [B][K] */
int orc_generate_beam(const orc_config* cfg, const orc_weights* w, const int* input_ids, const int* input_lengths, int B,
                      int S, int out_len, int K, const orc_beam_params* bp, int* output_ids, int* sequence_lengths,
                      float* cum_log_probs);

int orc_generate(const orc_config* cfg, const orc_weights* w, const int* input_ids, const int* input_lengths, int B,
                 int S, int out_len, const orc_sampling* sp, int* output_ids, int* sequence_lengths,
                 float* cum_log_probs, float* dbg_logits, float* dbg_hidden);

/* One decoder-stack pass for a single token per row given explicit caches (used by bench cpu_baseline and
 * layer-level parity tests).  x [B,H] in, y [B,H] out. */
void orc_decoder_step(const orc_config* cfg, const orc_weights* w, const float* x, float* k_cache, float* v_cache,
                      const int* seq_len, const int* pad_count, const uint8_t* masked_tokens, const uint8_t* finished,
                      int B, int s_max, int step, float* y);

#ifdef __cplusplus
}
#endif
#endif
