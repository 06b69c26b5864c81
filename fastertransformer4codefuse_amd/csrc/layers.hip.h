// Host-side layer units of the GPT-NeoX decode path: what the reference's layer classes do, as sequences of this library's
// HIP launches on a caller-given stream.  No layer owns memory (the reference's allocateBuffer / freeBuffer per call,
// DecoderSelfAttentionLayer.cc:148-187, FfnLayer.cc:455-530, are the engine's one arena: engine.hip plan()), no layer
// synchronises, and none calls a collective: a parallel-residual layer reduces x' ONCE, after the residual
// (GptNeoXDecoder.cc:342-359), which is the decoder's business.
//
//   DecoderSelfAttentionLayer  <- layers/attention_layers/DecoderSelfAttentionLayer.cc:459-686 (+ TensorParallel wrapper :190-226)
//   GptContextAttentionLayer   <- layers/attention_layers/GptContextAttentionLayer.cc:25-403
//   FfnLayer (GeluFfnLayer)    <- layers/FfnLayer.cc:34-380, TensorParallelGeluFfnLayer.cc:33-63
//   DynamicDecodeLayer         <- layers/DynamicDecodeLayer.cc:192-497 (sampling layers, online beam search)
//
// The fused forms -- the persistent decode kernel (persist_device.hip.h), the per-stage GEMV launches, the grouped burst GEMM
// that runs two layers' independent GEMMs in one launch -- cut ACROSS these units on purpose and stay in engine.hip.
#pragma once
#include <functional>

#include "kernels.h"

namespace ftcf {

struct DenseWeight {  // layers/DenseWeight.h:29-66
    const void* kernel = nullptr;  // tiled (int8 or fp16)
    const f16*  scale  = nullptr;  // weight_only_quant_scale
    const f16*  bias   = nullptr;
};
struct LayerWeights {  // models/gptneox/GptNeoXDecoderLayerWeight.h
    const f16 *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    DenseWeight qkv, attn_out, ffn1, ffn2;
};

// C[m, n] = epilogue(A[m, k] x W) on stream s; `slot` names the workspace region of the burst GEMM this call may use
// (0 QKV, 1 FFN1, 2 out-proj, 3 FFN2: the four GEMMs of a layer may be in flight together on two streams).  The engine
// and the batcher bind it to their dispatch (GEMV / burst / tiled MFMA kernels by row count).
using GemmFn = std::function<void(const f16* A, const DenseWeight& w, const f16* bias, int act, f16* C, int m, int n, int k,
                                  hipStream_t s, int slot)>;

struct DecoderSelfAttentionLayer {
    GemmFn gemm;
    int    H = 0, hl = 0;  // hidden size, local hidden size (heads of this rank x size_per_head)
    // in [m, H] (LayerNorm'd) -> qkv_buf [m, 3 hl] -> masked multi-head attention over the K/V cache (appends this step's
    // key / value; qkv bias and NeoX rotary inside the kernel like the reference's) -> ctx [m, hl] -> out [m, H] (no bias:
    // it rides on the residual kernel)
    void forward(const f16* in, f16* qkv_buf, const f16* ctx_buf, f16* out, const LayerWeights& w, const MmhaParams& mp, int m,
                 hipStream_t s) const
    {
        gemm(in, w.qkv, nullptr, 0, qkv_buf, m, 3 * hl, H, s, 0);
        launch_mmha(mp, s);
        gemm(ctx_buf, w.attn_out, nullptr, 0, out, m, H, hl, s, 2);
    }
    // the same over a paged K/V pool (continuous batching)
    void forward_paged(const f16* in, f16* qkv_buf, const f16* ctx_buf, f16* out, const LayerWeights& w, const MmhaPagedParams& mp,
                       int max_len, int m, hipStream_t s) const
    {
        gemm(in, w.qkv, nullptr, 0, qkv_buf, m, 3 * hl, H, s, 0);
        launch_mmha_paged(mp, max_len, s);
        gemm(ctx_buf, w.attn_out, nullptr, 0, out, m, H, hl, s, 2);
    }
};

struct GptContextAttentionLayer {
    GemmFn gemm;
    int    H = 0, hl = 0, nh = 0, dh = 0, rot = 0;
    // rows = sequences x tokens of (a micro-batch of) the prompt: QKV GEMM, bias + rotary + K/V -> cache, causal attention on
    // MFMA tiles (the reference's UNFUSED_MHA with padding removal), output projection.  [s_lo, s_hi): the token range of a
    // chunked prompt phase whose earlier tokens' K/V are in the cache already (default: the whole prompt)
    void forward(const f16* in, f16* qkv_buf, f16* ctx_buf, f16* out, const LayerWeights& w, const int* input_lengths, f16* k_cache,
                 f16* v_cache, int B, int S, int s_max, int cache_row_mult, hipStream_t s, int s_lo = 0, int s_hi = -1) const
    {
        const int    e   = s_hi < 0 ? S : s_hi;
        const size_t r0  = (B == 1) ? (size_t)s_lo : 0;  // (a token range is only cut out of a single sequence)
        const int    m   = (B == 1) ? e - s_lo : B * S;
        gemm(in + r0 * H, w.qkv, nullptr, 0, qkv_buf + r0 * 3 * hl, m, 3 * hl, H, s, 0);
        launch_context_attention(qkv_buf, w.qkv.bias, input_lengths, k_cache, v_cache, B, S, nh, dh, rot, s_max, ctx_buf, s,
                                 cache_row_mult, s_lo, e);
        gemm(ctx_buf + r0 * hl, w.attn_out, nullptr, 0, out + r0 * H, m, H, hl, s, 2);
    }
};

struct FfnLayer {
    GemmFn gemm;
    int    H = 0, il = 0;  // hidden size, local intermediate size
    // out = gelu_tanh(in W1 + b1) W2 ; bias 2 is deferred to the residual kernel (FfnLayer.cc:203-217 gemm_bias_act)
    void forward(const f16* in, f16* mid_buf, f16* out, const LayerWeights& w, int m, hipStream_t s) const
    {
        gemm(in, w.ffn1, w.ffn1.bias, 1, mid_buf, m, il, H, s, 1);
        gemm(mid_buf, w.ffn2, nullptr, 0, out, m, H, il, s, 3);
    }
};

struct DynamicDecodeLayer {
    // beam_width 1: temperature / penalties / end mask, top-k and top-p sampling, stop criteria (DynamicDecodeLayer.cc:410-497)
    void forward(const SamplingParams& sp, hipStream_t s, bool with_finish = true) const { launch_dynamic_decode(sp, s, with_finish); }
    // beam_width > 1: online beam search + the shared stop criteria (:309-408)
    void forward(const BeamParams& bp, const SamplingParams& sp, hipStream_t s) const
    {
        launch_beam_search(bp, s);
        launch_decode_finish(sp, s);
    }
};

}  // namespace ftcf
