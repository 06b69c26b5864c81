// Attention kernels of the GPT-NeoX path for gfx950 (wave64, LDS staged K/V tiles, shuffle reductions).
//
//  * k_mmha_split / k_mmha_combine : per-token decoder attention, the counterpart of
//      mmha::masked_multihead_attention_kernel
//      (kernels/decoder_masked_multihead_attention/decoder_masked_multihead_attention_template.hpp:1099-1919):
//      QKV bias add, NeoX rotary on q/k, append k/v to the cache, softmax(q K^T / sqrt(dh)) V with padding mask.
//      The reference launches NH*B thread blocks (40 at bs=1) which cannot fill 256 CUs; here the KV range is split
//      over `nsplit` workgroups per (head, row) and the partial (max, sum, out) triples are merged exactly
//      (flash-decoding style) by a tiny second kernel.
//  * k_qkv_bias_rotary_cache + k_context_attention : prefill, the counterpart of add_fusedQKV_bias_transpose_kernel
//      (kernels/unfused_attention_kernels.cu:1326-1484), transpose_4d_batch_major_{k,v}_cache (:1673-1749), the
//      batched QK^T / P.V GEMMs and softmax_kernel (:255-332) of GptContextAttentionLayer.cc:142-345, fused into a
//      causal online-softmax kernel (no S x S score buffers): k_context_attention_mfma (QK^T and PV on MFMA tiles).
//
// Cache layout (engine private): K and V both [B, nh, s_max, dh] fp16, dh contiguous: one wave-load = 1 KiB of
// consecutive keys.  Roofline: HBM (decode: 4*t*dh*nh bytes per layer per row).
#include "attn_device.hip.h"

#include <map>
#include <mutex>

namespace ftcf {

template<int DH, bool BEAMS>
__global__ __launch_bounds__(256) void k_mmha_split(const MmhaParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_last;
    mmha_block<DH, BEAMS>(p, smem, s_last, blockIdx.x, blockIdx.y, blockIdx.z);
}

size_t mmha_workspace_bytes(int B, int nh, int dh, int nsplit)
{
    return (size_t)B * nh * nsplit * (dh + 2) * sizeof(unsigned long long);
}

size_t mmha_smem_bytes(int dh, int s_max, int nsplit)
{
    const int    chunk   = (((s_max + nsplit - 1) / nsplit) + 15) & ~15;
    const size_t partial = (size_t)3 * dh * 2 + (8 + 4 * dh) * 4 + (size_t)chunk * 4;
    const size_t merge   = ((size_t)nsplit * (dh + 2) + nsplit + 1) * 4;
    return std::max(partial, merge);
}

int mmha_pick_nsplit(int B, int nh, int s_max)
{
    static const int wgs = getenv("FTCF_MMHA_WGS") ? atoi(getenv("FTCF_MMHA_WGS")) : 640;
    int want = (wgs + B * nh - 1) / (B * nh);  // aim for >= ~640 workgroups
    int maxs = (s_max + 63) / 64;              // at least 64 keys per split
    int n    = std::max(1, std::min(std::min(want, maxs), 16));  // <= MMHA_MAX_SPLIT: one polling pass per merge
    return n;
}

__global__ void k_rotary_table(float* table, const int* d_step, const int* pad_count, int rot)
{
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < rot / 2) {
        const int pos = (*d_step - 1) - (pad_count ? pad_count[b] : 0);
        float     cs, sn;
        rotary_coef(j, rot, pos, cs, sn);
        table[((size_t)b * (rot / 2) + j) * 2]     = cs;
        table[((size_t)b * (rot / 2) + j) * 2 + 1] = sn;
    }
}

// Token prologue in ONE launch: the embedding row of the previous step's token (decoding_kernels.cu:145-191,
// embeddingLookupPosEncoding without position table) and the rotary {cos, sin} table of this step, per row.  (Two
// launches cost two dispatch latencies of ~5 us for a few KB of work.)
__global__ __launch_bounds__(256) void k_step_prologue(f16* out, const f16* __restrict__ table,
                                                       const int* __restrict__ output_ids, const int* d_step,
                                                       float* rot_table, const int* __restrict__ pad_count, int B, int H,
                                                       int rot, const int* d_stop)
{
    if (d_stop && *d_stop) {
        return;  // every row has finished: a token of a multi-token graph behind the request's last one
    }
    const int b    = blockIdx.x;
    const int step = *d_step;
    if ((int)threadIdx.x < rot / 2) {
        const int pos = (step - 1) - (pad_count ? pad_count[b] : 0);
        float     cs, sn;
        rotary_coef(threadIdx.x, rot, pos, cs, sn);
        rot_table[((size_t)b * (rot / 2) + threadIdx.x) * 2]     = cs;
        rot_table[((size_t)b * (rot / 2) + threadIdx.x) * 2 + 1] = sn;
    }
    const int  id  = output_ids[(size_t)(step - 1) * B + b];
    const f16* src = table + (size_t)id * H;
    f16*       dst = out + (size_t)b * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
    }
}

void launch_step_prologue(f16* out, const f16* table, const int* output_ids, const int* d_step, float* rot_table,
                          const int* pad_count, int B, int H, int rot, hipStream_t s, const int* d_stop)
{
    FTCF_CHECK_ARG(rot / 2 <= 256 && H % 8 == 0, "rotary_embedding_dim must be <= 512 and the hidden size a multiple of 8");
    hipLaunchKernelGGL(k_step_prologue, dim3(B), dim3(256), 0, s, out, table, output_ids, d_step, rot_table, pad_count, B, H,
                       rot, d_stop);
    FTCF_HIP_CHECK(hipGetLastError());
}

void launch_rotary_table(float* table, const int* d_step, const int* pad_count, int B, int rot, hipStream_t s)
{
    if (rot <= 0) {
        return;
    }
    hipLaunchKernelGGL(k_rotary_table, dim3(B), dim3(64 * ((rot / 2 + 63) / 64)), 0, s, table, d_step, pad_count, rot);
    FTCF_HIP_CHECK(hipGetLastError());
}

// the reference's head sizes (DecoderSelfAttentionLayer.cc:280-282)
#define FTCF_HEAD_SIZES(X) X(32) X(48) X(64) X(80) X(96) X(128) X(144) X(160) X(192) X(224) X(256)
bool mmha_head_size_supported(int dh)
{
#define X(D) if (dh == D) { return true; }
    FTCF_HEAD_SIZES(X)
#undef X
    return false;
}

void launch_mmha(const MmhaParams& p, hipStream_t s)
{
    FTCF_CHECK_ARG(mmha_head_size_supported(p.dh), "size_per_head must be one of 32, 48, 64, 80, 96, 128, 144, 160, 192, 224, 256");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh, "rotary_embedding_dim must be even and <= size_per_head");
    const size_t smem = mmha_smem_bytes(p.dh, p.s_max, p.nsplit);
    dim3         grid(p.nh, p.B, p.nsplit);
    FTCF_CHECK_ARG(p.nsplit >= 1 && p.nsplit <= 16 && p.gran != nullptr, "bad split-KV configuration");
    if (p.cache_indir) {  // beam search
        FTCF_CHECK_ARG(p.beam_width > 1 && p.B % p.beam_width == 0, "cache indirection needs rows = batch * beam_width");
    }
    const bool beams = p.cache_indir != nullptr;
#define X(D)                                                                                                           \
    if (p.dh == D) {                                                                                                   \
        if (beams) {                                                                                                   \
            hipLaunchKernelGGL((k_mmha_split<D, true>), grid, dim3(256), smem, s, p);                                  \
        }                                                                                                              \
        else {                                                                                                         \
            hipLaunchKernelGGL((k_mmha_split<D, false>), grid, dim3(256), smem, s, p);                                 \
        }                                                                                                              \
    }
    FTCF_HEAD_SIZES(X)
#undef X
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// prefill
// ---------------------------------------------------------------------------------------------------------------
// grid B*S, block 256: one workgroup per prompt token, all heads.  q is rotated in place inside the qkv buffer; k/v go to
// the caches (zeros for padding rows, as the reference's memset + un-padded scatter leaves them,
// GptContextAttentionLayer.cc:152-172).  The {cos, sin} of the token's position are computed once per workgroup; a thread
// owns element d of a head and, inside the rotary range, its partner d + rot/2.  (The first form launched one 128-thread
// workgroup per (token, head): 41k tiny workgroups per layer, 29 us.)
template<bool VEC>
__global__ __launch_bounds__(256) void k_qkv_bias_rotary_cache(f16* qkv, const f16* __restrict__ qkv_bias,
                                                               const int* __restrict__ input_lengths, f16* k_cache,
                                                               f16* v_cache, int S, int nh, int dh, int rot, int s_max, int crm,
                                                               int s_lo, int ns)
{
    // (token range [s_lo, s_lo + ns) of every sequence: the whole prompt, or one micro-batch of a chunked prompt phase)
    __shared__ float s_cs[128], s_sn[128];
    const int  b = blockIdx.x / ns, s = s_lo + blockIdx.x % ns;
    const int  row = b * S + s;
    const int  hl = nh * dh, half = rot / 2;
    const bool valid = s < input_lengths[b];
    if ((int)threadIdx.x < half) {
        float cs, sn;
        rotary_coef(threadIdx.x, rot, s, cs, sn);  // position = index in the (right padded) row
        s_cs[threadIdx.x] = cs;
        s_sn[threadIdx.x] = sn;
    }
    __syncthreads();
    f16* base = qkv + (size_t)row * 3 * hl;
    if (VEC) {
        // 16-byte pieces: piece c8 of a head holds d = c8*8 .. c8*8+7; inside the rotary range piece c8 < half/8 pairs with
        // piece c8 + half/8 (needs rot % 16 == 0, dh % 8 == 0)
        const int pph = dh / 8, hp = half / 8;
        for (int i = threadIdx.x; i < nh * pph; i += blockDim.x) {
            const int h = i / pph, c8 = i % pph;
            if (c8 >= hp && c8 < 2 * hp) {
                continue;  // written by the thread that owns piece c8 - hp
            }
            const int    e    = h * dh + c8 * 8;
            const size_t cidx = (((size_t)b * crm * nh + h) * s_max + s) * dh + c8 * 8;
            f16x8        q = {}, k = {}, v = {};
            if (valid) {
                q = *reinterpret_cast<const f16x8*>(base + e) + *reinterpret_cast<const f16x8*>(qkv_bias + e);
                k = *reinterpret_cast<const f16x8*>(base + hl + e) + *reinterpret_cast<const f16x8*>(qkv_bias + hl + e);
                v = *reinterpret_cast<const f16x8*>(base + 2 * hl + e) + *reinterpret_cast<const f16x8*>(qkv_bias + 2 * hl + e);
            }
            if (c8 < hp) {
                f16x8 q2 = {}, k2 = {}, v2 = {};
                if (valid) {
                    const int e2 = e + half;
                    q2 = *reinterpret_cast<const f16x8*>(base + e2) + *reinterpret_cast<const f16x8*>(qkv_bias + e2);
                    k2 = *reinterpret_cast<const f16x8*>(base + hl + e2) + *reinterpret_cast<const f16x8*>(qkv_bias + hl + e2);
                    v2 = *reinterpret_cast<const f16x8*>(base + 2 * hl + e2)
                         + *reinterpret_cast<const f16x8*>(qkv_bias + 2 * hl + e2);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        f16 a = q[j], c = q2[j];
                        rotary_apply(a, c, s_cs[c8 * 8 + j], s_sn[c8 * 8 + j]);
                        q[j]  = a;
                        q2[j] = c;
                        f16 ka = k[j], kc2 = k2[j];
                        rotary_apply(ka, kc2, s_cs[c8 * 8 + j], s_sn[c8 * 8 + j]);
                        k[j]  = ka;
                        k2[j] = kc2;
                    }
                }
                *reinterpret_cast<f16x8*>(base + e + half)         = q2;
                *reinterpret_cast<f16x8*>(k_cache + cidx + half)   = k2;
                *reinterpret_cast<f16x8*>(v_cache + cidx + half)   = v2;
            }
            *reinterpret_cast<f16x8*>(base + e)       = q;
            *reinterpret_cast<f16x8*>(k_cache + cidx) = k;
            *reinterpret_cast<f16x8*>(v_cache + cidx) = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < hl; i += blockDim.x) {
        const int h = i / dh, d = i % dh;
        if (d >= half && d < rot) {
            continue;  // written by the thread that owns d - rot/2
        }
        const size_t cidx = (((size_t)b * crm * nh + h) * s_max + s) * dh + d;  // cache row b * crm (beam search: beam 0)
        f16          q = (f16)0.f, k = (f16)0.f, v = (f16)0.f;
        if (valid) {
            q = base[i] + qkv_bias[i];
            k = base[hl + i] + qkv_bias[hl + i];
            v = base[2 * hl + i] + qkv_bias[2 * hl + i];
        }
        if (d < half) {
            f16 q2 = (f16)0.f, k2 = (f16)0.f, v2 = (f16)0.f;
            if (valid) {
                q2 = base[i + half] + qkv_bias[i + half];
                k2 = base[hl + i + half] + qkv_bias[hl + i + half];
                v2 = base[2 * hl + i + half] + qkv_bias[2 * hl + i + half];
                rotary_apply(q, q2, s_cs[d], s_sn[d]);
                rotary_apply(k, k2, s_cs[d], s_sn[d]);
            }
            base[i + half]       = q2;
            k_cache[cidx + half] = k2;
            v_cache[cidx + half] = v2;
        }
        base[i]       = q;
        k_cache[cidx] = k;
        v_cache[cidx] = v;
    }
}

// a * x + b * y with both products rounded on their own (no fused multiply-add): the same value whichever operand pair comes first
__device__ __forceinline__ float sym_mad2(const float a, const float x, const float b, const float y)
{
#pragma clang fp contract(off)
    const float p = a * x;
    const float q = b * y;
    return p + q;
}

// MFMA form of the causal prefill attention: one workgroup = 64 query rows of one (row, head), a wave = 16 of them.
// Per 64-key tile: S = Q K^T on mfma_f32_16x16x32_f16 (K rows are d-contiguous = the B-operand order), online softmax in
// the accumulator layout (a row lives in 16 lanes x 4 key groups), P rounded to half (the reference's softmax output
// type) and turned into the A operand through a wave-private LDS tile, O += P V with V staged TRANSPOSED in LDS (the B
// operand wants 8 consecutive keys of one output dim per lane).  Tiles above the diagonal of a wave are skipped.
// DH: any of the reference's head sizes (a multiple of 16; the d steps of Q K^T are padded with zeros to a multiple of 32).
// KT: keys per tile, 64, or 32 for the sizes above 128 (LDS: K tile + transposed V tile + P tiles <= 64 KB of static LDS).
template<int DH, int KT = 64>
__global__ __launch_bounds__(256, (DH > 192 ? 2 : 3)) void k_context_attention_mfma(const f16* __restrict__ qkv,
                                                                const int* __restrict__ input_lengths,
                                                                const f16* __restrict__ k_cache,
                                                                const f16* __restrict__ v_cache, int S, int nh, int s_max,
                                                                f16* __restrict__ ctx, float qk_scale, int crm, int s_lo,
                                                                int s_hi, int nsplit2 = 0, float* __restrict__ split_ws = nullptr,
                                                                unsigned* __restrict__ split_tk = nullptr)
{
    static_assert(DH % 16 == 0 && (KT == 32 || KT == 64), "head size: a multiple of 16");
    constexpr int ND  = (DH + 31) / 32;  // d steps of Q K^T
    constexpr int DHK = ND * 32;         // K tile row, padded to whole d steps (the tail holds zeros)
    constexpr int NKG = KT / 16;         // key groups of a tile
    constexpr int LDK = DHK + 8;   // sK row (halves): rows start in different banks
    constexpr int LDV = KT + 8;    // sVt row: Vt[d][key]
    constexpr int LDP = KT + 8;    // sP row: P[row][key]
    constexpr int NO  = DH / 16;   // output column groups
    __shared__ __attribute__((aligned(16))) f16 sK[KT * LDK];
    __shared__ __attribute__((aligned(16))) f16 sVt[DH * LDV];
    __shared__ __attribute__((aligned(16))) f16 sP[4][16 * LDP];

    // (s_lo: first query row of a chunked prompt phase; the LAST query block -- the most key tiles -- is dispatched first: the
    // launch's tail is made of the short ones)
    // nsplit2 > 0: the launch's time is the CHAIN of key tiles of its last query block (16 dependent tiles at 1024 tokens, each
    // ~3.6 us with three workgroups per CU) while the chip as a whole has room: the nsplit2 heaviest query blocks are cut in two along
    // their keys -- two workgroups, half the chain each -- and the one that finishes second merges the two partial soft-maxes
    // (deterministic: the merge is symmetric in its operands)
    const int nqb  = (int)gridDim.x - nsplit2;  // query blocks of the launch
    const int qbi  = (int)blockIdx.x < 2 * nsplit2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x - nsplit2;  // 0 = the last block
    const int half = (int)blockIdx.x < 2 * nsplit2 ? (int)(blockIdx.x & 1) : -1;                      // -1: not split
    const int b = blockIdx.z, h = blockIdx.y, q0 = s_lo + (nqb - 1 - qbi) * 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int hl  = nh * DH;
    const int len = input_lengths[b];
    if (q0 >= len) {
        return;  // padded query rows are discarded by the reference
    }
    // Q fragments (A operand): row c of this wave's 16, d = s*32 + g*8 .. +8 (already bias + rotary'd in the qkv buffer)
    f16x8 qf[ND];
    {
        int qrow = q0 + wid * 16 + c;
        qrow     = qrow < S ? qrow : S - 1;
        const f16* qp = qkv + ((size_t)b * S + qrow) * 3 * hl + h * DH + g * 8;
#pragma unroll
        for (int s2 = 0; s2 < ND; s2++) {
            qf[s2] = (s2 * 32 + g * 8 < DH) ? *reinterpret_cast<const f16x8*>(qp + s2 * 32) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    if constexpr (DHK != DH) {  // zero tail of the K tile's rows, once (the staging below rewrites [0, DH) only)
        for (int i = threadIdx.x; i < KT * (DHK - DH) / 8; i += 256) {
            const int r = i / ((DHK - DH) / 8), ch = i % ((DHK - DH) / 8);
            *reinterpret_cast<u32x4*>(&sK[r * LDK + DH + ch * 8]) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    const f16* kc = k_cache + ((size_t)b * crm * nh + h) * s_max * DH;
    const f16* vc = v_cache + ((size_t)b * crm * nh + h) * s_max * DH;
    f32x4      o[NO];
    float      m_run[4], l_run[4];
#pragma unroll
    for (int n = 0; n < NO; n++) {
        o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        m_run[j] = -INFINITY;
        l_run[j] = 0.f;
    }
    const int q_last   = min(min(q0 + 63, len - 1), s_hi - 1);  // last query row of the block whose K/V is in the cache
    const int w_last   = q0 + wid * 16 + 15;  // last query row of this wave
    f16*      sPw      = sP[wid];
    // The K / V rows of a tile travel global memory -> registers -> LDS, and the NEXT tile's rows are requested before this
    // tile's arithmetic starts (round 4: the single-buffered form exposed a memory round trip per tile -- 16 of them for the last
    // query block of a 1024-token prompt; 84.8 -> 68.9 us per 13B layer at 1024 tokens; 57.8 with the rows' reductions on DPP).  K tile: row major; V tile: transposed, a thread
    // moves 2 keys x 8 dims (lanes of a wave spread over the banks).
    constexpr int NKC = (KT * DH / 8 + 255) / 256, NVC = ((KT / 2) * (DH / 8) + 255) / 256;
    u32x4         rk[NKC];
    f16x8         rv[NVC][2];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int u = 0; u < NKC; u++) {
            const int i = threadIdx.x + u * 256;
            if (NKC * 256 == KT * DH / 8 || i < KT * DH / 8) {
                const int r = i / (DH / 8), ch = i % (DH / 8);
                int       kk = k0 + r;
                // (keys above the block's last query row are masked for every row of the block: they re-read that row instead of
                // cache lines a chunked prompt phase has not written yet -- a masked P = 0 times a stale NaN / Inf is NaN)
                kk    = kk < q_last ? kk : q_last;
                rk[u] = *reinterpret_cast<const u32x4*>(kc + (size_t)kk * DH + ch * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < NVC; u++) {
            const int i = threadIdx.x + u * 256;
            if (NVC * 256 == (KT / 2) * (DH / 8) || i < (KT / 2) * (DH / 8)) {
                const int rp = i % (KT / 2), ch = i / (KT / 2);
                int       k1 = k0 + 2 * rp, k2 = k0 + 2 * rp + 1;
                k1           = k1 < q_last ? k1 : q_last;
                k2           = k2 < q_last ? k2 : q_last;
                rv[u][0]     = *reinterpret_cast<const f16x8*>(vc + (size_t)k1 * DH + ch * 8);
                rv[u][1]     = *reinterpret_cast<const f16x8*>(vc + (size_t)k2 * DH + ch * 8);
            }
        }
    };
    int k_begin = 0, k_end = q_last + 1;  // keys [k_begin, k_end) in whole tiles
    if (half >= 0) {
        const int tiles = q_last / KT + 1, cut = (tiles / 2) * KT;
        k_begin = half ? cut : 0;
        k_end   = half ? q_last + 1 : cut;
    }
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        __syncthreads();  // every wave is through with the previous tile
#pragma unroll
        for (int u = 0; u < NKC; u++) {
            const int i = threadIdx.x + u * 256;
            if (NKC * 256 == KT * DH / 8 || i < KT * DH / 8) {
                const int r = i / (DH / 8), ch = i % (DH / 8);
                *reinterpret_cast<u32x4*>(&sK[r * LDK + ch * 8]) = rk[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NVC; u++) {
            const int i = threadIdx.x + u * 256;
            if (NVC * 256 == (KT / 2) * (DH / 8) || i < (KT / 2) * (DH / 8)) {
                const int rp = i % (KT / 2), ch = i / (KT / 2);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    *reinterpret_cast<f16x2*>(&sVt[(ch * 8 + e) * LDV + 2 * rp]) = f16x2{rv[u][0][e], rv[u][1][e]};
                }
            }
        }
        __syncthreads();
        if (k0 + KT < k_end) {
            fetch(k0 + KT);  // in flight under this tile's arithmetic
        }
        if (k0 > w_last) {
            continue;  // above this wave's diagonal (the barriers above are still taken by every wave)
        }
        // ---- S = Q K^T : sc[kg][j] = score of row g*4+j and key k0 + kg*16 + c ----
        f32x4 sc[NKG];
#pragma unroll
        for (int kg = 0; kg < NKG; kg++) {
            sc[kg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < ND; s2++) {
                const f16x8 kb = *reinterpret_cast<const f16x8*>(&sK[(kg * 16 + c) * LDK + s2 * 32 + g * 8]);
                sc[kg]         = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[s2], kb, sc[kg], 0, 0, 0);
            }
        }
        // ---- online softmax per row (mask of gpt_kernels.cu:359-402), P -> LDS as half ----
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int qi = q0 + wid * 16 + g * 4 + j;
            float     sv[NKG];
            float     mt = -INFINITY;
#pragma unroll
            for (int kg = 0; kg < NKG; kg++) {
                const int  key   = k0 + kg * 16 + c;
                const bool valid = (key <= qi) && (qi < len);
                sv[kg]           = valid ? qk_scale * sc[kg][j] : -INFINITY;
                mt               = fmaxf(mt, sv[kg]);
            }
            // (a row's 16 lanes are one DPP row: no LDS crossbar round trips -- __shfl_xor is a ds_bpermute, eight of them per
            // row and tile in a dependent chain)
            mt = fmaxf(mt, dpp_read<0xB1>(mt));
            mt = fmaxf(mt, dpp_read<0x4E>(mt));
            mt = fmaxf(mt, dpp_read<0x141>(mt));
            mt = fmaxf(mt, dpp_read<0x140>(mt));
            const float mn = fmaxf(m_run[j], mt);
            const float al = (m_run[j] == -INFINITY) ? 0.f : __expf(m_run[j] - mn);
            float       ls = 0.f;
#pragma unroll
            for (int kg = 0; kg < NKG; kg++) {
                const float e = (sv[kg] == -INFINITY || mn == -INFINITY) ? 0.f : __expf(sv[kg] - mn);
                ls += e;
                sPw[(g * 4 + j) * LDP + kg * 16 + c] = (f16)e;
            }
            ls = group_sum_dpp<16>(ls);
            l_run[j] = l_run[j] * al + ls;
            m_run[j] = mn;
#pragma unroll
            for (int n = 0; n < NO; n++) {
                o[n][j] *= al;
            }
        }
        // sP[wid] is written and read by this wave only: LDS operations of a wave are ordered, no barrier needed
        // ---- O += P V ----
#pragma unroll
        for (int ks = 0; ks < KT / 32; ks++) {
            const f16x8 pa = *reinterpret_cast<const f16x8*>(&sPw[c * LDP + ks * 32 + g * 8]);
#pragma unroll
            for (int n = 0; n < NO; n++) {
                const f16x8 vb = *reinterpret_cast<const f16x8*>(&sVt[(n * 16 + c) * LDV + ks * 32 + g * 8]);
                o[n]           = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, vb, o[n], 0, 0, 0);
            }
        }
    }
    if (half >= 0) {
        // this half's partial {o (relative to m), m, l} -> workspace in the accumulator layout (write-through: the other workgroup
        // may sit on another XCD), then a ticket; the second arrival adds the first one's partial to its registers
        __shared__ int s_second;
        typedef __attribute__((address_space(1))) unsigned gu32;
        constexpr size_t PW   = (size_t)64 * DH + 128;  // floats per partial
        const size_t     item = ((size_t)b * gridDim.y + h) * nsplit2 + qbi;
        gu32*            mine = (gu32*)(split_ws + (item * 2 + half) * PW);
        const gu32*      othr = (const gu32*)(split_ws + (item * 2 + (1 - half)) * PW);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = wid * 16 + g * 4 + j;
#pragma unroll
            for (int n = 0; n < NO; n++) {
                __hip_atomic_store(mine + (size_t)row * DH + n * 16 + c, __float_as_uint(o[n][j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (c == 0) {
                __hip_atomic_store(mine + (size_t)64 * DH + row * 2, __float_as_uint(m_run[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + (size_t)64 * DH + row * 2 + 1, __float_as_uint(l_run[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {
            s_second = __hip_atomic_fetch_add(split_tk + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u ? 1 : 0;
        }
        __syncthreads();
        if (!s_second) {
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int   row = wid * 16 + g * 4 + j;
            const float m2  = __uint_as_float(__hip_atomic_load(othr + (size_t)64 * DH + row * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const float l2  = __uint_as_float(__hip_atomic_load(othr + (size_t)64 * DH + row * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const float mn  = fmaxf(m_run[j], m2);
            const float a1  = (m_run[j] == -INFINITY) ? 0.f : __expf(m_run[j] - mn);
            const float a2  = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
            // (products rounded on their own: a fused multiply-add would make the sum depend on which half merges)
            l_run[j]        = sym_mad2(l_run[j], a1, l2, a2);
            m_run[j]        = mn;
#pragma unroll
            for (int n = 0; n < NO; n++) {
                const float o2 = __uint_as_float(__hip_atomic_load(othr + (size_t)row * DH + n * 16 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                o[n][j]        = sym_mad2(o[n][j], a1, o2, a2);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int qi = q0 + wid * 16 + g * 4 + j;
        if (qi < len && qi < s_hi) {
            const float inv = 1.f / (l_run[j] + 1e-6f);  // unfused_attention_kernels.cu:322
#pragma unroll
            for (int n = 0; n < NO; n++) {
                ctx[((size_t)b * S + qi) * hl + h * DH + n * 16 + c] = (f16)(o[n][j] * inv);
            }
        }
    }
}

// workspace of the key-split prompt attention: one per (device, stream), kept for the life of the process (as the split-K GEMM's)
constexpr size_t CTX_SPLIT_WS = (size_t)96 << 20;
constexpr size_t CTX_SPLIT_TK = 64 * 1024;  // tickets (unsigned)
static float* ctx_split_workspace(hipStream_t s)
{
    static std::mutex                                    mu;
    static std::map<std::pair<int, hipStream_t>, float*> ws;
    int                                                  dev = 0;
    FTCF_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    float*&                     p = ws[{dev, s}];
    if (!p) {
        FTCF_HIP_CHECK(hipMalloc(&p, CTX_SPLIT_WS + CTX_SPLIT_TK * sizeof(unsigned)));
    }
    return p;
}

void launch_context_attention(const f16* qkv, const f16* qkv_bias, const int* input_lengths, f16* k_cache,
                              f16* v_cache, int B, int S, int nh, int dh, int rot, int s_max, f16* ctx, hipStream_t s,
                              int cache_row_mult, int s_lo, int s_hi)
{
    // [s_lo, s_hi): the query / new-key rows of this call (a micro-batch of a chunked prompt phase: the keys below s_lo are
    // in the cache already); default the whole prompt
    s_hi = s_hi < 0 ? S : s_hi;
    FTCF_CHECK_ARG(s_lo >= 0 && s_lo < s_hi && s_hi <= S, "bad token range");
    const int ns = s_hi - s_lo;
    FTCF_CHECK_ARG(mmha_head_size_supported(dh), "size_per_head must be one of 32, 48, 64, 80, 96, 128, 144, 160, 192, 224, 256");
    FTCF_CHECK_ARG(S <= s_max, "prompt longer than the cache");
    FTCF_CHECK_ARG(rot <= 256, "rotary_embedding_dim must be <= 256");
    if (rot % 16 == 0) {
        hipLaunchKernelGGL(k_qkv_bias_rotary_cache<true>, dim3(B * ns), dim3(256), 0, s, const_cast<f16*>(qkv), qkv_bias,
                           input_lengths, k_cache, v_cache, S, nh, dh, rot, s_max, cache_row_mult, s_lo, ns);
    }
    else {
        hipLaunchKernelGGL(k_qkv_bias_rotary_cache<false>, dim3(B * ns), dim3(256), 0, s, const_cast<f16*>(qkv), qkv_bias,
                           input_lengths, k_cache, v_cache, S, nh, dh, rot, s_max, cache_row_mult, s_lo, ns);
    }
    // qk_scale is computed in T by the reference (GptContextAttentionLayer.cc: `const T qk_scale = (T)(1/sqrtf(dh))`)
    const float qk_scale = (float)(f16)(1.0f / sqrtf((float)dh));
    {
        // key split of the heaviest query blocks (see the kernel): those whose chain is longer than half the longest one, when the
        // launch is long enough to care (>= 8 key tiles), the partials fit the workspace and nobody is capturing a graph (the
        // workspace is allocated on first use).  FTCF_CTX_SPLIT=0 switches it off.
        const int nqb = (ns + 63) / 64;
        int       nsplit2 = 0;
        float*    sws = nullptr;
        unsigned* stk = nullptr;
        {
            const char* e    = getenv("FTCF_CTX_SPLIT");
            const int   KTl  = dh > 128 ? 32 : 64;
            const int   tmax = (s_hi - 1) / KTl + 1;  // key tiles of the last query block
            if ((!e || atoi(e) != 0) && tmax >= 8) {
                // query block qbi (0 = last) sees keys up to s_hi - 1 - 64 * qbi: split while its tiles exceed tmax / 2
                int n2 = 0;
                while (n2 < nqb && (s_hi - 1 - 64 * n2) / KTl + 1 > tmax / 2) {
                    n2++;
                }
                const size_t need = (size_t)B * nh * n2 * 2 * ((size_t)64 * dh + 128) * sizeof(float);
                if (n2 > 0 && need <= CTX_SPLIT_WS && (size_t)B * nh * n2 <= CTX_SPLIT_TK) {
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    (void)hipStreamIsCapturing(s, &cs);
                    if (cs == hipStreamCaptureStatusNone) {
                        sws     = ctx_split_workspace(s);
                        stk     = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(sws) + CTX_SPLIT_WS);
                        nsplit2 = n2;
                        FTCF_HIP_CHECK(hipMemsetAsync(stk, 0, (size_t)B * nh * n2 * sizeof(unsigned), s));
                    }
                }
            }
        }
        dim3 grid(nqb + nsplit2, nh, B);
#define X(D)                                                                                                           \
    if (dh == D) {                                                                                                     \
        hipLaunchKernelGGL((k_context_attention_mfma<D, (D > 128 ? 32 : 64)>), grid, dim3(256), 0, s, qkv, input_lengths,\
                           k_cache, v_cache, S, nh, s_max, ctx, qk_scale, cache_row_mult, s_lo, s_hi, nsplit2, sws, stk);  \
    }
        FTCF_HEAD_SIZES(X)
#undef X
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// Paged decoder attention (the continuous-batching front end, engine.hip `ftcf_batcher`): the K/V of a
// sequence live in fixed-size pages of a pool shared by all sequences, [page][head][P tokens][dh], found through the
// sequence's page table.  Same arithmetic as mmha_partial with one split (attn_device.hip.h): half q/k/v + bias, NeoX rotary
// at position = number of cached tokens, fp32 scores, exp against the row maximum, fp32 PV, one normalisation with the
// reference's +1e-6 (decoder_masked_multihead_attention_template.hpp:1632).  One workgroup per (head, slot); a sequence's
// length is its own (no padding, no masks: a slot's tokens are dense from position 0).
// ---------------------------------------------------------------------------------------------------------------
template<int DH>
__global__ __launch_bounds__(256) void k_mmha_paged(const MmhaPagedParams p)
{
    constexpr int LPK = DH / 8;    // lanes per key/value row (16 B each)
    constexpr int KPI = 64 / LPK;  // rows per wave-load
    extern __shared__ __attribute__((aligned(16))) char smem_pg[];
    __shared__ float s_red[16];
    const int h = blockIdx.x, b = blockIdx.y;
    if (p.finished[b]) {
        return;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sub = lane % LPK, grp = lane / LPK;
    const int tl  = p.len[b];  // cached tokens = position of the current one
    f16*      s_q = reinterpret_cast<f16*>(smem_pg);
    f16*      s_k = s_q + DH;
    f16*      s_v = s_k + DH;
    float*    s_o = reinterpret_cast<float*>(s_v + DH);  // [4][DH]
    int*      s_pt = reinterpret_cast<int*>(s_o + 4 * DH);  // [max_pages]
    float*    s_p = reinterpret_cast<float*>(s_pt + p.max_pages);  // [tl + 1]
    const int hl  = p.nh * DH;
    for (int i = threadIdx.x; i < p.max_pages; i += 256) {
        s_pt[i] = p.page_table[(size_t)b * p.max_pages + i];
    }
    if (threadIdx.x < DH) {
        const int    d    = threadIdx.x;
        const size_t base = (size_t)b * 3 * hl + h * DH + d;
        s_q[d] = p.qkv[base] + p.qkv_bias[h * DH + d];
        s_k[d] = p.qkv[base + hl] + p.qkv_bias[hl + h * DH + d];
        s_v[d] = p.qkv[base + 2 * hl] + p.qkv_bias[2 * hl + h * DH + d];
    }
    __syncthreads();
    if ((int)threadIdx.x < p.rot / 2) {
        float cs, sn;
        rotary_coef(threadIdx.x, p.rot, tl, cs, sn);
        const int j = threadIdx.x, j2 = j + p.rot / 2;
        f16       a = s_q[j], c = s_q[j2];
        rotary_apply(a, c, cs, sn);
        s_q[j]  = a;
        s_q[j2] = c;
        f16 ka = s_k[j], kc = s_k[j2];
        rotary_apply(ka, kc, cs, sn);
        s_k[j]  = ka;
        s_k[j2] = kc;
    }
    __syncthreads();
    const size_t page_elems = (size_t)p.nh * p.P * DH;
    auto row_ptr = [&](const f16* pool, const int t) -> const f16* {
        return pool + (size_t)s_pt[t / p.P] * page_elems + ((size_t)h * p.P + (t % p.P)) * DH;
    };
    if (threadIdx.x < DH) {  // append the current token (its page was allocated by the scheduler)
        f16* kd = const_cast<f16*>(row_ptr(p.kpool, tl));
        f16* vd = const_cast<f16*>(row_ptr(p.vpool, tl));
        kd[threadIdx.x] = s_k[threadIdx.x];
        vd[threadIdx.x] = s_v[threadIdx.x];
    }
    const float inv_sqrt_dh = rsqrtf((float)DH);
    const f16x8 qv          = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
    float       lmax        = -INFINITY;
    for (int t0 = 0; t0 < tl; t0 += 4 * KPI) {
        const int   t  = t0 + wid * KPI + grp;
        const int   tc = t < tl ? t : tl - 1;
        const f16x8 kv = *reinterpret_cast<const f16x8*>(row_ptr(p.kpool, tc) + sub * 8);
        float       a  = 0.f;
        a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
        a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
        a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
        a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
        a              = group_sum(a, LPK) * inv_sqrt_dh;
        if (t < tl && sub == 0) {
            s_p[t] = a;
            lmax   = fmaxf(lmax, a);
        }
    }
    if (wid == 0) {  // current token from LDS
        float a = 0.f;
        if (lane < LPK) {
            const f16x8 kv = *reinterpret_cast<const f16x8*>(s_k + lane * 8);
            const f16x8 q8 = *reinterpret_cast<const f16x8*>(s_q + lane * 8);
            a              = dot2(f16x2{q8[0], q8[1]}, f16x2{kv[0], kv[1]}, a);
            a              = dot2(f16x2{q8[2], q8[3]}, f16x2{kv[2], kv[3]}, a);
            a              = dot2(f16x2{q8[4], q8[5]}, f16x2{kv[4], kv[5]}, a);
            a              = dot2(f16x2{q8[6], q8[7]}, f16x2{kv[6], kv[7]}, a);
        }
        a = wave_sum(a) * inv_sqrt_dh;
        if (lane == 0) {
            s_p[tl] = a;
            lmax    = fmaxf(lmax, a);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) {
        s_red[wid] = lmax;
    }
    __syncthreads();
    const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float       lsum = 0.f;
    for (int i = threadIdx.x; i <= tl; i += 256) {
        const float e = __expf(s_p[i] - m);
        s_p[i]        = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) {
        s_red[4 + wid] = lsum;
    }
    __syncthreads();
    const float inv = 1.f / (((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) + 1.e-6f);
    // PV: lane (grp, sub) accumulates dims sub*8 .. +7 over its keys, then the key groups of a wave and the waves are added
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < tl; t0 += 4 * KPI) {
        const int   t  = t0 + wid * KPI + grp;
        const int   tc = t < tl ? t : tl - 1;
        const f16x8 vv = *reinterpret_cast<const f16x8*>(row_ptr(p.vpool, tc) + sub * 8);
        const float w  = t < tl ? s_p[t] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            o[j] = fmaf(w, (float)vv[j], o[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        for (int off = LPK; off < 64; off <<= 1) {
            o[j] += __shfl_xor(o[j], off, 64);
        }
    }
    if (grp == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s_o[wid * DH + sub * 8 + j] = o[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        const int   d = threadIdx.x;
        const float v = ((s_o[d] + s_o[DH + d]) + (s_o[2 * DH + d] + s_o[3 * DH + d])) + s_p[tl] * (float)s_v[d];
        p.ctx[(size_t)b * hl + h * DH + d] = (f16)(v * inv);
    }
}

size_t mmha_paged_smem_bytes(int dh, int max_pages, int max_len)
{
    return (size_t)3 * dh * 2 + (size_t)4 * dh * 4 + (size_t)max_pages * 4 + (size_t)(max_len + 2) * 4;
}

void launch_mmha_paged(const MmhaPagedParams& p, int max_len, hipStream_t s)
{
    FTCF_CHECK_ARG(p.dh == 64 || p.dh == 128, "size_per_head must be 64 or 128");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh, "rotary_embedding_dim must be even and <= size_per_head");
    const size_t smem = mmha_paged_smem_bytes(p.dh, p.max_pages, max_len);
    FTCF_CHECK_ARG(smem <= 64 * 1024, "paged attention keeps a sequence's scores in LDS: max sequence length <= ~15000");
    dim3 grid(p.nh, p.B);
    if (p.dh == 128) {
        hipLaunchKernelGGL((k_mmha_paged<128>), grid, dim3(256), smem, s, p);
    }
    else {
        hipLaunchKernelGGL((k_mmha_paged<64>), grid, dim3(256), smem, s, p);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// K/V of a freshly prefilled prompt: one row of the engine cache [L][rows][nh][s_max][dh] (kc / vc point at the row in layer 0,
// src_layer_elems = rows * nh * s_max * dh) -> the sequence's pages of every layer
__global__ void k_scatter_kv_to_pages(const f16* __restrict__ kc, const f16* __restrict__ vc, f16* kpool, f16* vpool,
                                      const int* __restrict__ pages, int L, int nh, int dh, int s_max, int S, int P,
                                      size_t pool_layer_elems, size_t src_layer_elems)
{
    // one 16-byte piece per thread: (layer, head, token, piece)
    const int    ppr   = dh / 8;
    const size_t total = (size_t)L * nh * S * ppr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int    pc = (int)(i % ppr);
        size_t       r  = i / ppr;
        const int    t  = (int)(r % S);
        r /= S;
        const int    h = (int)(r % nh), l = (int)(r / nh);
        const size_t src = (size_t)l * src_layer_elems + ((size_t)h * s_max + t) * dh + pc * 8;
        const size_t dst = (size_t)l * pool_layer_elems + ((size_t)pages[t / P] * nh + h) * P * dh + (size_t)(t % P) * dh + pc * 8;
        *reinterpret_cast<u32x4*>(kpool + dst) = *reinterpret_cast<const u32x4*>(kc + src);
        *reinterpret_cast<u32x4*>(vpool + dst) = *reinterpret_cast<const u32x4*>(vc + src);
    }
}
void launch_scatter_kv_to_pages(const f16* kc, const f16* vc, f16* kpool, f16* vpool, const int* pages, int L, int nh, int dh,
                                int s_max, int S, int P, size_t pool_layer_elems, hipStream_t s, size_t src_layer_elems)
{
    if (src_layer_elems == 0) {
        src_layer_elems = (size_t)nh * s_max * dh;  // a one-row cache
    }
    const size_t total = (size_t)L * nh * S * (dh / 8);
    if (total == 0) {
        return;
    }
    hipLaunchKernelGGL(k_scatter_kv_to_pages, dim3((int)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, s, kc, vc,
                       kpool, vpool, pages, L, nh, dh, s_max, S, P, pool_layer_elems, src_layer_elems);
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
