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
//      causal online-softmax kernel (no S x S score buffers).
//
// Cache layout (engine private): K and V both [B, nh, s_max, dh] fp16, dh contiguous: one wave-load = 1 KiB of
// consecutive keys.  Roofline: HBM (decode: 4*t*dh*nh bytes per layer per row).
#include "ftcf_common.h"
#include "kernels.h"

namespace ftcf {

__device__ __forceinline__ float group_sum(float v, int lanes_per_key)
{
    for (int o = lanes_per_key >> 1; o >= 1; o >>= 1) {
        v += __shfl_xor(v, o, 64);
    }
    return v;
}

// rotary coefficient exactly as the reference computes it (decoder_masked_multihead_attention_utils.h:1325-1329):
// inv_freq = t / 10000^(2j/rot) ; {cos, sin}(inv_freq) in fp32
__device__ __forceinline__ void rotary_pair(f16& a, f16& b, int j, int rot, int pos)
{
    const float inv_freq = (float)pos / powf(10000.0f, (float)(2 * j) / (float)rot);
    const float cs = cosf(inv_freq), sn = sinf(inv_freq);
    const float fa = (float)a, fb = (float)b;
    a = (f16)(cs * fa - sn * fb);
    b = (f16)(cs * fb + sn * fa);
}

template<int DH>
__device__ __forceinline__ void mmha_partial(const MmhaParams& p, char* smem, float* wsout, int h, int b, int sp)
{
    constexpr int LPK = DH / 8;    // lanes per key/value row (16 B each)
    constexpr int KPI = 64 / LPK;  // rows per wave-load
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tl    = p.seq_len[b];  // tlength: number of cached keys; the new token goes to index tl
    const int chunk = (((p.s_max + p.nsplit - 1) / p.nsplit) + 15) & ~15;
    const int t_beg = sp * chunk;
    int       t_end = t_beg + chunk;  // exclusive, over positions 0..tl (tl = current token)
    if (t_end > tl + 1) {
        t_end = tl + 1;
    }
    if (t_beg > tl) {  // empty split
        if (threadIdx.x == 0) {
            wsout[DH]     = -INFINITY;
            wsout[DH + 1] = 0.f;
        }
        if (threadIdx.x < DH) {
            wsout[threadIdx.x] = 0.f;
        }
        return;
    }
    const bool owns_cur = (tl >= t_beg && tl < t_end);

    f16*   s_q    = reinterpret_cast<f16*>(smem);       // [DH]
    f16*   s_k    = s_q + DH;                            // [DH] new key
    f16*   s_v    = s_k + DH;                            // [DH] new value
    float* s_red  = reinterpret_cast<float*>(s_v + DH);  // [8 + 4*DH]
    float* s_p    = s_red + 8 + 4 * DH;                  // [chunk]

    const int hl   = p.nh * DH;
    const int step = p.d_step ? *p.d_step : p.step;
    const int pos  = (step - 1) - (p.pad_count ? p.pad_count[b] : 0);  // :1303,:1343-1344
    // ---- q (+bias, rotary); new k/v for the split that owns the current position ----
    if (threadIdx.x < DH) {
        const int    d    = threadIdx.x;
        const size_t base = (size_t)b * 3 * hl + h * DH + d;
        const f16    bq   = p.qkv_bias ? p.qkv_bias[h * DH + d] : (f16)0.f;
        s_q[d]            = p.qkv[base] + bq;
        if (owns_cur) {
            const f16 bk = p.qkv_bias ? p.qkv_bias[hl + h * DH + d] : (f16)0.f;
            const f16 bv = p.qkv_bias ? p.qkv_bias[2 * hl + h * DH + d] : (f16)0.f;
            s_k[d]       = p.qkv[base + hl] + bk;
            s_v[d]       = p.qkv[base + 2 * hl] + bv;
        }
    }
    __syncthreads();
    if (p.rot > 0 && threadIdx.x < p.rot / 2) {
        const int j = threadIdx.x;
        f16       a = s_q[j], c = s_q[j + p.rot / 2];
        rotary_pair(a, c, j, p.rot, pos);
        s_q[j]             = a;
        s_q[j + p.rot / 2] = c;
        if (owns_cur) {
            f16 ka = s_k[j], kc = s_k[j + p.rot / 2];
            rotary_pair(ka, kc, j, p.rot, pos);
            s_k[j]             = ka;
            s_k[j + p.rot / 2] = kc;
        }
    }
    __syncthreads();
    f16* kc = p.k_cache + ((size_t)b * p.nh + h) * p.s_max * DH;
    f16* vc = p.v_cache + ((size_t)b * p.nh + h) * p.s_max * DH;
    if (owns_cur && threadIdx.x < DH) {  // append to the cache (:1397, :1837)
        kc[(size_t)tl * DH + threadIdx.x] = s_k[threadIdx.x];
        vc[(size_t)tl * DH + threadIdx.x] = s_v[threadIdx.x];
    }

    const float inv_sqrt_dh = rsqrtf((float)DH);  // DecoderSelfAttentionLayer.cc:118 with q_scaling 1
    const int   sub = lane % LPK, grp = lane / LPK;
    const f16x8 qv  = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
    const int   t_cached_end = owns_cur ? tl : t_end;  // cached keys of this split: [t_beg, t_cached_end)
    const uint8_t* mask = p.masked_tokens ? p.masked_tokens + (size_t)b * p.s_max : nullptr;

    // ---- phase 1: qk for the cached keys (fp32 accumulate, MMHA_USE_FP32_ACUM_FOR_FMA) ----
    float lmax = -INFINITY;
    constexpr int U = 4;
    for (int t0 = t_beg + wid * KPI; t0 < t_cached_end; t0 += 4 * KPI * U) {
        u32x4 kr[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int t = t0 + u * 4 * KPI + grp;
            t     = t < t_cached_end ? t : t_cached_end - 1;
            kr[u] = *reinterpret_cast<const u32x4*>(kc + (size_t)t * DH + sub * 8);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int   t  = t0 + u * 4 * KPI + grp;
            const f16x8 kv = __builtin_bit_cast(f16x8, kr[u]);
            float       a  = 0.f;
            a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
            a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
            a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
            a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
            a              = group_sum(a, LPK) * inv_sqrt_dh;
            if (t < t_cached_end && sub == 0) {
                const bool m = mask && mask[t];
                s_p[t - t_beg] = m ? -INFINITY : a;  // masked keys get probability 0 (:1570,:1610-1622)
                if (!m) {
                    lmax = fmaxf(lmax, a);
                }
            }
        }
    }
    if (owns_cur && wid == 0) {  // current token from LDS (:1407-1437)
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
            s_p[tl - t_beg] = a;
            lmax            = fmaxf(lmax, a);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) {
        s_red[wid] = lmax;
    }
    __syncthreads();
    const float m_loc = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    // ---- phase 2: exp, local sum ----
    float lsum = 0.f;
    for (int i = threadIdx.x; i < t_end - t_beg; i += 256) {
        const float e = (s_p[i] == -INFINITY) ? 0.f : __expf(s_p[i] - m_loc);
        s_p[i]        = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();  // s_red reuse + s_p visible
    if (lane == 0) {
        s_red[4 + wid] = lsum;
    }
    // ---- phase 3: P.V (fp32 accumulate) ----
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        acc[j] = 0.f;
    }
    for (int t0 = t_beg + wid * KPI; t0 < t_cached_end; t0 += 4 * KPI * U) {
        u32x4 vr[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int t = t0 + u * 4 * KPI + grp;
            t     = t < t_cached_end ? t : t_cached_end - 1;
            vr[u] = *reinterpret_cast<const u32x4*>(vc + (size_t)t * DH + sub * 8);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int   t  = t0 + u * 4 * KPI + grp;
            const float pt = (t < t_cached_end) ? s_p[t - t_beg] : 0.f;
            const f16x8 vv = __builtin_bit_cast(f16x8, vr[u]);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                acc[j] = fmaf(pt, (float)vv[j], acc[j]);
            }
        }
    }
    if (owns_cur && wid == 0 && grp == 0) {
        const float pt = s_p[tl - t_beg];
        const f16x8 vv = *reinterpret_cast<const f16x8*>(s_v + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = fmaf(pt, (float)vv[j], acc[j]);
        }
    }
    // fold the KPI row groups of the wave, then the 4 waves
#pragma unroll
    for (int j = 0; j < 8; j++) {
        for (int o = LPK; o < 64; o <<= 1) {
            acc[j] += __shfl_xor(acc[j], o, 64);
        }
    }
    float* s_o = s_red + 8;  // [4][DH]
    if (grp == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s_o[wid * DH + sub * 8 + j] = acc[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        const int d        = threadIdx.x;
        wsout[d]           = (s_o[d] + s_o[DH + d]) + (s_o[2 * DH + d] + s_o[3 * DH + d]);
    }
    if (threadIdx.x == 0) {
        wsout[DH]     = m_loc;
        wsout[DH + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }
}

// One launch: every split workgroup publishes its (max, sum, out[DH]) partial, takes a ticket, and the LAST arriver
// of each (row, head) merges the partials in split order (deterministic) -- the in-launch hand-off recipe of
// cdna_hip_programming.md G16: plain stores -> per-wave vmcnt(0) -> barrier -> one-lane agent release (+ asm vmcnt(0))
// -> relaxed agent ticket ; consumer: one-lane agent acquire -> barrier -> plain loads.  Placement independent.
template<int DH>
__global__ __launch_bounds__(256) void k_mmha_split(const MmhaParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_last;
    const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    if (p.finished && p.finished[b]) {
        return;  // :1176 (ctx of a finished row is never consumed); uniform for all splits of the row
    }
    float* wsout = p.ws + (((size_t)b * p.nh + h) * p.nsplit + sp) * (DH + 2);
    mmha_partial<DH>(p, smem, wsout, h, b, sp);
    if (p.nsplit == 1) {
        __syncthreads();
        if (threadIdx.x < DH) {
            const float inv = 1.f / (wsout[DH + 1] + 1.e-6f);  // :1632
            p.ctx[(size_t)b * p.nh * DH + h * DH + threadIdx.x] = (f16)(wsout[threadIdx.x] * inv);
        }
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* counter = p.counters + (size_t)b * p.nh + h;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last      = (t == p.nsplit - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) {
        return;
    }
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    const float* ws = p.ws + ((size_t)b * p.nh + h) * p.nsplit * (DH + 2);
    float*       sw = reinterpret_cast<float*>(smem);  // [nsplit] weights, then [1] denominator
    if (threadIdx.x < 64) {
        float ms = -INFINITY, ls = 0.f;
        // nsplit <= 64 (mmha_pick_nsplit caps at 32)
        if ((int)threadIdx.x < p.nsplit) {
            ms = ws[threadIdx.x * (DH + 2) + DH];
            ls = ws[threadIdx.x * (DH + 2) + DH + 1];
        }
        const float m = wave_max(ms);
        const float w = (ms == -INFINITY) ? 0.f : __expf(ms - m);
        if ((int)threadIdx.x < p.nsplit) {
            sw[threadIdx.x] = w;
        }
        // fixed-order sum over splits
        float L = 0.f;
        for (int s2 = 0; s2 < p.nsplit; s2++) {
            L += __shfl(w * ls, s2, 64);
        }
        if (threadIdx.x == 0) {
            sw[p.nsplit] = L;
        }
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        const int d = threadIdx.x;
        float     o = 0.f;
#pragma unroll 4
        for (int s2 = 0; s2 < p.nsplit; s2++) {
            o += sw[s2] * ws[s2 * (DH + 2) + d];
        }
        const float inv = 1.f / (sw[p.nsplit] + 1.e-6f);  // :1632
        p.ctx[(size_t)b * p.nh * DH + h * DH + d] = (f16)(o * inv);
    }
}

size_t mmha_workspace_bytes(int B, int nh, int dh, int nsplit)
{
    // partials + one arrival counter per (row, head); the counters must be zero before the first launch and are
    // re-zeroed by the last arriver
    return (((size_t)B * nh * nsplit * (dh + 2) * sizeof(float) + 255) & ~(size_t)255) + (size_t)B * nh * sizeof(int);
}

int* mmha_counters(float* ws, int B, int nh, int dh, int nsplit)
{
    size_t off = (size_t)B * nh * nsplit * (dh + 2) * sizeof(float);
    off        = (off + 255) & ~(size_t)255;
    return reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + off);
}

int mmha_pick_nsplit(int B, int nh, int s_max)
{
    int want = (768 + B * nh - 1) / (B * nh);  // aim for >= ~768 workgroups
    int maxs = (s_max + 63) / 64;              // at least 64 keys per split
    int n    = std::max(1, std::min(std::min(want, maxs), 32));
    return n;
}

void launch_mmha(const MmhaParams& p, hipStream_t s)
{
    FTCF_CHECK_ARG(p.dh == 64 || p.dh == 128, "size_per_head must be 64 or 128");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh, "rotary_embedding_dim must be even and <= size_per_head");
    const int    chunk = (((p.s_max + p.nsplit - 1) / p.nsplit) + 15) & ~15;
    const size_t smem  = (size_t)3 * p.dh * 2 + (8 + 4 * p.dh) * 4 + (size_t)chunk * 4;
    dim3         grid(p.nh, p.B, p.nsplit);
    FTCF_CHECK_ARG(p.nsplit <= 64 && (p.nsplit == 1 || p.counters != nullptr), "bad split-KV configuration");
    if (p.dh == 128) {
        hipLaunchKernelGGL(k_mmha_split<128>, grid, dim3(256), smem, s, p);
    }
    else {
        hipLaunchKernelGGL(k_mmha_split<64>, grid, dim3(256), smem, s, p);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// prefill
// ---------------------------------------------------------------------------------------------------------------
// grid (B*S, nh), block dh.  q is rotated in place inside the qkv buffer; k/v go to the caches (zeros for padding
// rows, as the reference's memset + un-padded scatter leaves them, GptContextAttentionLayer.cc:152-172).
__global__ void k_qkv_bias_rotary_cache(f16* qkv, const f16* __restrict__ qkv_bias, const int* __restrict__ input_lengths,
                                        f16* k_cache, f16* v_cache, int S, int nh, int dh, int rot, int s_max)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16*      sq = reinterpret_cast<f16*>(smem);
    f16*      sk = sq + dh;
    const int row = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
    const int b = row / S, s = row % S;
    const int hl = nh * dh;
    const bool valid = s < input_lengths[b];
    f16*   base = qkv + (size_t)row * 3 * hl + h * dh + d;
    f16    q = (f16)0.f, k = (f16)0.f, v = (f16)0.f;
    if (valid) {
        q = base[0] + qkv_bias[h * dh + d];
        k = base[hl] + qkv_bias[hl + h * dh + d];
        v = base[2 * hl] + qkv_bias[2 * hl + h * dh + d];
    }
    sq[d] = q;
    sk[d] = k;
    __syncthreads();
    if (valid && d < rot / 2) {
        f16 a = sq[d], c = sq[d + rot / 2];
        rotary_pair(a, c, d, rot, s);  // position = index in the (right padded) row
        sq[d]           = a;
        sq[d + rot / 2] = c;
        f16 ka = sk[d], kc2 = sk[d + rot / 2];
        rotary_pair(ka, kc2, d, rot, s);
        sk[d]           = ka;
        sk[d + rot / 2] = kc2;
    }
    __syncthreads();
    base[0] = sq[d];
    const size_t cidx = (((size_t)b * nh + h) * s_max + s) * dh + d;
    k_cache[cidx]     = sk[d];
    v_cache[cidx]     = v;
}

// Causal attention with online softmax.  grid (ceil(S/16), nh, B), 256 threads: each wave owns 4 query rows,
// K/V tiles of 64 keys are staged in LDS and shared by the 16 rows of the block.
// QK: lane = key (no cross-lane reduction); PV: lane = 2 output dims (DH=128) / 1 dim (DH=64).
template<int DH>
__global__ __launch_bounds__(256) void k_context_attention(const f16* __restrict__ qkv, const int* __restrict__ input_lengths,
                                                           const f16* __restrict__ k_cache, const f16* __restrict__ v_cache,
                                                           int S, int nh, int s_max, f16* __restrict__ ctx, float qk_scale)
{
    constexpr int KT  = 64;       // keys per tile
    constexpr int LDK = DH + 8;   // padded LDS row (halves)
    constexpr int DPL = DH / 64;  // output dims per lane
    __shared__ __attribute__((aligned(16))) f16   sK[KT * LDK];
    __shared__ __attribute__((aligned(16))) f16   sV[KT * DH];
    __shared__ __attribute__((aligned(16))) f16   sQ[16 * DH];
    __shared__ __attribute__((aligned(16))) float sP[4][KT][4];  // [wave][key][row]

    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int hl = nh * DH;
    const int len = input_lengths[b];
    if (q0 >= len) {
        return;  // padded query rows are discarded by the reference
    }
    // stage the 16 query rows (already bias+rotary'd in place in the qkv buffer)
    for (int i = threadIdx.x; i < 16 * DH / 8; i += 256) {
        const int r = i / (DH / 8), ch = i % (DH / 8);
        int       qi = q0 + r;
        qi           = qi < S ? qi : S - 1;
        *reinterpret_cast<f16x8*>(&sQ[r * DH + ch * 8]) =
            *reinterpret_cast<const f16x8*>(qkv + ((size_t)b * S + qi) * 3 * hl + h * DH + ch * 8);
    }
    const f16* kc = k_cache + ((size_t)b * nh + h) * s_max * DH;
    const f16* vc = v_cache + ((size_t)b * nh + h) * s_max * DH;

    float m_run[4], l_run[4], o[4][DPL];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        m_run[r] = -INFINITY;
        l_run[r] = 0.f;
#pragma unroll
        for (int j = 0; j < DPL; j++) {
            o[r][j] = 0.f;
        }
    }
    const int q_last = min(q0 + 15, len - 1);
    for (int k0 = 0; k0 <= q_last; k0 += KT) {
        __syncthreads();
        for (int i = threadIdx.x; i < KT * DH / 8; i += 256) {
            const int r = i / (DH / 8), ch = i % (DH / 8);
            int       kk = k0 + r;
            kk           = kk < S ? kk : S - 1;
            const u32x4 kv = *reinterpret_cast<const u32x4*>(kc + (size_t)kk * DH + ch * 8);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(vc + (size_t)kk * DH + ch * 8);
            *reinterpret_cast<u32x4*>(&sK[r * LDK + ch * 8]) = kv;
            *reinterpret_cast<u32x4*>(&sV[r * DH + ch * 8])  = vv;
        }
        __syncthreads();
        // ---- scores: lane = key ----
        float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ch = 0; ch < DH / 8; ch++) {
            const f16x8 kv = *reinterpret_cast<const f16x8*>(&sK[lane * LDK + ch * 8]);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const f16x8 qv = *reinterpret_cast<const f16x8*>(&sQ[(wid * 4 + r) * DH + ch * 8]);
                float       a  = sc[r];
                a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
                a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
                a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
                a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
                sc[r]          = a;
            }
        }
        const int key = k0 + lane;
        float     pr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int  qi    = q0 + wid * 4 + r;
            const bool valid = (key <= qi) && (qi < len);  // mask of gpt_kernels.cu:359-402
            const float s    = valid ? qk_scale * sc[r] : -INFINITY;
            const float mt   = wave_max(s);
            const float mn   = fmaxf(m_run[r], mt);
            const float al   = (m_run[r] == -INFINITY) ? 0.f : __expf(m_run[r] - mn);
            const float e    = (s == -INFINITY || mn == -INFINITY) ? 0.f : __expf(s - mn);
            l_run[r]         = l_run[r] * al + wave_sum(e);
            m_run[r]         = mn;
            pr[r]            = e;
#pragma unroll
            for (int j = 0; j < DPL; j++) {
                o[r][j] *= al;
            }
        }
        *reinterpret_cast<f32x4*>(&sP[wid][lane][0]) = f32x4{pr[0], pr[1], pr[2], pr[3]};
        // sP[wid] is written and read by the same wave only: LDS ops of a wave are ordered, no barrier needed
        // ---- P.V: lane = output dims ----
        const int kmax = min(KT, q_last - k0 + 1);
        for (int kk = 0; kk < kmax; kk++) {
            const f32x4 pk = *reinterpret_cast<const f32x4*>(&sP[wid][kk][0]);
#pragma unroll
            for (int j = 0; j < DPL; j++) {
                const float vv = (float)sV[kk * DH + lane * DPL + j];
                o[0][j]        = fmaf(pk[0], vv, o[0][j]);
                o[1][j]        = fmaf(pk[1], vv, o[1][j]);
                o[2][j]        = fmaf(pk[2], vv, o[2][j]);
                o[3][j]        = fmaf(pk[3], vv, o[3][j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int qi = q0 + wid * 4 + r;
        if (qi < len && qi < S) {
            const float inv = 1.f / (l_run[r] + 1e-6f);  // unfused_attention_kernels.cu:322
#pragma unroll
            for (int j = 0; j < DPL; j++) {
                ctx[((size_t)b * S + qi) * hl + h * DH + lane * DPL + j] = (f16)(o[r][j] * inv);
            }
        }
    }
}

void launch_context_attention(const f16* qkv, const f16* qkv_bias, const int* input_lengths, f16* k_cache,
                              f16* v_cache, int B, int S, int nh, int dh, int rot, int s_max, f16* ctx, hipStream_t s)
{
    FTCF_CHECK_ARG(dh == 64 || dh == 128, "size_per_head must be 64 or 128");
    FTCF_CHECK_ARG(S <= s_max, "prompt longer than the cache");
    hipLaunchKernelGGL(k_qkv_bias_rotary_cache, dim3(B * S, nh), dim3(dh), (size_t)2 * dh * 2, s, const_cast<f16*>(qkv),
                       qkv_bias, input_lengths, k_cache, v_cache, S, nh, dh, rot, s_max);
    // qk_scale is computed in T by the reference (GptContextAttentionLayer.cc: `const T qk_scale = (T)(1/sqrtf(dh))`)
    const float qk_scale = (float)(f16)(1.0f / sqrtf((float)dh));
    dim3        grid((S + 15) / 16, nh, B);
    if (dh == 128) {
        hipLaunchKernelGGL(k_context_attention<128>, grid, dim3(256), 0, s, qkv, input_lengths, k_cache, v_cache, S, nh,
                           s_max, ctx, qk_scale);
    }
    else {
        hipLaunchKernelGGL(k_context_attention<64>, grid, dim3(256), 0, s, qkv, input_lengths, k_cache, v_cache, S, nh,
                           s_max, ctx, qk_scale);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
