// Dynamic decode (beam_width == 1) for gfx950: penalties, end mask, softmax, top-k / top-p sampling, stop criteria.
// Counterpart of DynamicDecodeLayer<float>::forward (layers/DynamicDecodeLayer.cc:192-497),
// BaseSamplingLayer<T>::forward (layers/sampling_layers/BaseSamplingLayer.cc:255-359),
// TopKSamplingLayer::runSampling (TopKSamplingLayer.cu:184-257), TopPSamplingLayer::runSampling and the kernels
// kernels/sampling_penalty_kernels.cu:117-147,367-425,485-520, kernels/sampling_topk_kernels.cu:67-311,
// kernels/sampling_topp_kernels.cu:802-1000,1296-1345, kernels/select_optional_last_tokens.cu:22-85,
// kernels/stop_criteria_kernels.cu:24-158, kernels/decoding_kernels.cu:26-65,452-583.
//
// All per-request state (step counter, finished flags, sequence lengths) lives on the device so the per-token step
// is a fixed launch sequence (hipGraph friendly); the host only reads a pinned "all finished" word.
#include "ftcf_common.h"
#include "kernels.h"
#include "attn_device.hip.h"  // rotary_coef (the next token's prologue inside k_greedy_decode)
#include "lm_head_device.hip.h"  // k_lm_head_greedy

#include <mutex>

namespace ftcf {

constexpr int TOPK_BLOCKS = 8;     // blocks per row in stage 1 (the reference's BLOCKS_PER_BEAM_)
constexpr int TOPK_MAX    = 1024;  // TopKSamplingLayer.cu: setup_topk_runtime_args<1024>

__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
// identical to oracle/ftcf_oracle.c orc_uniform: (0,1].  The reference draws from curand XORWOW
// (sampling_topk_kernels.cu:32-65,283): only the distribution is reproducible, greedy is unaffected.  Like the reference's
// curand_init(seed, 0, 0) per row, the stream depends on the row's SEED and draw count only, never on where the row sits
// in a batch (`row` is 0 at every call site): a request samples the same tokens alone, in any batch, in any batcher slot.
__device__ __forceinline__ float ftcf_uniform(uint64_t seed, uint64_t row, uint64_t draw)
{
    const uint64_t z = splitmix64(seed ^ splitmix64(row * 0x632be59bd9b4e019ULL + draw));
    const uint32_t r = (uint32_t)(z >> 40);
    return (float)(r + 1u) * (1.0f / 16777216.0f);
}

struct VI {
    float v;
    int   i;
};
__device__ __forceinline__ bool better(float v, int i, float bv, int bi)
{
    return (v > bv) || (v == bv && i < bi);
}
__device__ __forceinline__ VI wave_best(VI x)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(x.v, o, 64);
        const int   oi = __shfl_xor(x.i, o, 64);
        if (better(ov, oi, x.v, x.i)) {
            x.v = ov;
            x.i = oi;
        }
    }
    return x;
}
// block arg-best; result valid in all threads.  red: 2*nw words of LDS.
__device__ __forceinline__ VI block_best(VI x, float* redv, int* redi)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    x = wave_best(x);
    if (lane == 0) {
        redv[wid] = x.v;
        redi[wid] = x.i;
    }
    __syncthreads();
    VI r{redv[0], redi[0]};
    for (int w = 1; w < nw; w++) {
        if (better(redv[w], redi[w], r.v, r.i)) {
            r.v = redv[w];
            r.i = redi[w];
        }
    }
    __syncthreads();
    return r;
}

// ---- step 1: penalties, masks, softmax (one block per row) -----------------------------------------------------
__global__ __launch_bounds__(1024) void k_decode_prep(const SamplingParams p)
{
    if (p.state->all_finished) {
        return;  // a token of a multi-token graph behind the request's last one (engine.hip step()): nothing to do
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[64];
    const int        b    = blockIdx.x;
    const int        V    = p.V;
    float*           l    = p.logits + (size_t)b * V;
    const int        step = p.row_len ? p.row_len[b] + 1 : p.state->step;
    const int        tid = threadIdx.x, nt = blockDim.x;

    // K15 select_optional_last_tokens: first generated step only (DynamicDecodeLayer.cc:250-267)
    if (p.optional_last_tokens && step == p.max_input_len) {
        uint32_t* bits = reinterpret_cast<uint32_t*>(smem);
        const int words = (V + 31) / 32;
        for (int i = tid; i < words; i += nt) {
            bits[i] = 0u;
        }
        __syncthreads();
        for (int j = tid; j < p.optional_count; j += nt) {
            const int t = p.optional_last_tokens[(size_t)b * p.optional_count + j];
            if (t >= 0 && t < V) {
                atomicOr(&bits[t >> 5], 1u << (t & 31));
            }
        }
        __syncthreads();
        for (int i = tid; i < V; i += nt) {
            if (!((bits[i >> 5] >> (i & 31)) & 1u)) {
                l[i] = -INFINITY;
            }
        }
        __syncthreads();
    }
    if (p.apply_temperature) {  // sampling_penalty_kernels.cu:117-147
        const float   inv = 1.0f / (p.temperature[b] + 1e-6f);
        constexpr int TU  = 16;  // loads in flight per thread (a plain read-modify-write loop is ~99 dependent round trips: 25 us)
        for (int i0 = tid; i0 < V; i0 += nt * TU) {
            float t[TU];
#pragma unroll
            for (int u = 0; u < TU; u++) {
                const int i = i0 + u * nt;
                t[u]        = i < V ? l[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < TU; u++) {
                const int i = i0 + u * nt;
                if (i < V) {
                    l[i] = t[u] * inv;
                }
            }
        }
        __syncthreads();
    }
    if (p.apply_repetition && step > 1) {  // sampling_penalty_kernels.cu:367-425
        float*      newv = reinterpret_cast<float*>(smem);
        int*        idx  = reinterpret_cast<int*>(newv + p.total_len);
        const float pen  = p.repetition_penalty[b];
        const int   in_len = p.input_lengths[b];
        for (int t = tid; t < step; t += nt) {
            if (t >= in_len && t < p.max_input_len) {
                idx[t] = -1;
                continue;
            }
            const int   id = p.output_ids[(size_t)t * p.B + b];
            const float lg = l[id];
            idx[t]         = id;
            newv[t]        = lg < 0.0f ? lg * pen : lg / pen;
        }
        __syncthreads();
        for (int t = tid; t < step; t += nt) {
            if (idx[t] >= 0) {
                l[idx[t]] = newv[t];
            }
        }
        __syncthreads();
    }
    if (p.min_length && tid == 0) {  // sampling_penalty_kernels.cu:485-520
        if (p.seq_len[b] + 1 - p.max_input_len < p.min_length[b]) {
            l[p.end_id] = -FLT_MAX;
        }
    }
    __syncthreads();
    const bool fin      = p.finished[b];
    const bool is_topp  = p.top_k[b] == 0;
    // (top-k rows with cum_log_probs: the reference runs the same softmax over the whole row and samples from the
    // probabilities -- here the row's max and sum of exponentials fall out of the stage-1 slice scans and only the k
    // candidates are turned into probabilities, in k_sample: no extra pass over the vocabulary)
    const bool need_sm  = is_topp;
    if (fin) {  // addBiasEndMask (sampling_topk_kernels.cu:67-110)
        for (int i = tid; i < V; i += nt) {
            l[i] = (i == p.end_id) ? FLT_MAX : -FLT_MAX;
        }
        __syncthreads();
    }
    if (need_sm) {  // addBiasSoftMax (sampling_topp_kernels.cu:1296-1345)
        // One workgroup per row, three passes (max ; sum of exp ; exp / sum).  Every pass keeps 16 loads per thread in
        // flight (a plain `for (i = tid; i < V; i += nt)` loop is ~99 DEPENDENT round trips per pass at V = 100864: 62 us per
        // token with return_cum_log_probs, which the reference harness always sets -- codefuse_example.py:745) and the
        // un-normalised exponentials are not written back in between.  Element -> thread assignment and summation order
        // are those of the plain loop: bit-identical probabilities.
        constexpr int UNR = 16;
        float         mx  = -FLT_MAX;
        for (int i0 = tid; i0 < V; i0 += nt * UNR) {
            float t[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int i = i0 + u * nt;
                t[u]        = i < V ? l[i] : -FLT_MAX;
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                mx = fmaxf(mx, t[u]);
            }
        }
        mx = wave_max(mx);
        if ((tid & 63) == 0) {
            red[tid >> 6] = mx;
        }
        __syncthreads();
        mx = red[0];
        for (int w = 1; w < (nt >> 6); w++) {
            mx = fmaxf(mx, red[w]);
        }
        __syncthreads();
        float sum = 0.f;
        for (int i0 = tid; i0 < V; i0 += nt * UNR) {
            float t[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int i = i0 + u * nt;
                t[u]        = i < V ? l[i] : -FLT_MAX;
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                if (i0 + u * nt < V) {
                    sum += __expf(t[u] - mx);
                }
            }
        }
        sum = wave_sum(sum);
        if ((tid & 63) == 0) {
            red[tid >> 6] = sum;
        }
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < (nt >> 6); w++) {
            tot += red[w];
        }
        const float den = tot + 1e-6f;
        for (int i0 = tid; i0 < V; i0 += nt * UNR) {
            float t[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int i = i0 + u * nt;
                t[u]        = i < V ? l[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int i = i0 + u * nt;
                if (i < V) {
                    l[i] = __expf(t[u] - mx) / den;
                }
            }
        }
    }
}

// ---- step 2: stage-1 top-k: every block extracts the k best of its contiguous vocabulary slice ------------------
// The k best of a slice as a SET (stage 2 sorts the union of the slices' sets): the k-th largest value is found by a radix
// select over the order-preserving integer image of the floats -- four histogram passes of 8 bits over the slice in LDS
// -- then everything above it is emitted through an LDS counter, and of the elements EQUAL to it the ones with the lowest
// indices (the tie rule of `better`: value desc, index asc) by a block-wide prefix count.  Cost is independent of k.
// (The first version extracted one maximum per pass over the slice: 373 us per token at the reference harness's default
// top_k = 50, codefuse_example.py:799 -- 14 % of a 13B decode step.)
constexpr int STAGE1_MAXE = 60;  // elements per thread of a stage-1 slice (registers): V <= 8 x 256 x 60 = 122880
constexpr int TOPP_K = 128;  // candidates per slice kept for the rows of the top-p layer (see k_sample)

__device__ __forceinline__ unsigned fkey(const float v)  // v1 > v2 <=> fkey(v1) > fkey(v2) (no NaNs among logits)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// exclusive prefix sum of one int per thread over the 256-thread block; `tot` = block total.  red: 4 ints of LDS.
__device__ __forceinline__ int block_excl_scan(const int v, int* red, int& tot)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int       inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) {
            inc += t;
        }
    }
    if (lane == 63) {
        red[wid] = inc;
    }
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wid; w++) {
        base += red[w];
    }
    tot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return base + inc - v;
}

// a candidate {value, id} of a slice: plain stores, or write-through ones when the reader is a workgroup of the SAME launch
template<bool WT>
__device__ __forceinline__ void put_cand(float* ov, int* oi, const int pos, const float v, const int id)
{
    if constexpr (WT) {
        typedef __attribute__((address_space(1))) unsigned gu32;
        __hip_atomic_store((gu32*)(ov + pos), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((gu32*)(oi + pos), (unsigned)id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else {
        ov[pos] = v;
        oi[pos] = id;
    }
}
// The `ke` (>= 2) best of the n elements of a slice as a SET -> ov / oi [0, ke): the block's threads hold element
// threadIdx.x + 256 j in vals[j] (j < ne); l: the slice in memory (only read when more elements equal the threshold than there
// are places; mask_end: element end_local reads as -FLT_MAX there, as it does in vals).  See k_topk_stage1.
template<int MAXE, bool WT>
__device__ __forceinline__ void slice_select(const float (&vals)[MAXE], const int ne, const int n, const int ke, const float* l,
                                             const int i0, float* ov, int* oi, int (*hist8)[257], int* hist, int* s_sel, int* redi,
                                             const bool mask_end, const int end_local)
{
    // ---- radix select: key of the ke-th largest element ----
    unsigned prefix = 0u, mask = 0u;
    int      need = ke;
    for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            hist8[q][threadIdx.x] = 0;
        }
        __syncthreads();
        // (eight private copies of the histogram, picked by lane: the sign / exponent bytes of a row's logits fall into a
        // handful of bins)
#pragma unroll
        for (int j = 0; j < MAXE; j++) {
            const unsigned key = fkey(vals[j]);
            if (j < ne && (key & mask) == prefix) {
                atomicAdd(&hist8[threadIdx.x & 7][(key >> shift) & 255u], 1);
            }
        }
        __syncthreads();
        {
            int t = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                t += hist8[q][threadIdx.x];
            }
            hist[threadIdx.x] = t;
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // bins 4 * lane .. 4 * lane + 3; suffix sums from the top bin down
            const int lane = threadIdx.x;
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int mine = h0 + h1 + h2 + h3;
            int       above = mine;  // inclusive suffix over lanes >= lane
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(above, o, 64);
                if (lane + o < 64) {
                    above += t;
                }
            }
            above -= mine;  // elements in bins of higher lanes
            // the lane whose bins contain the need-th element (counted from the top)
            if (above < need && above + mine >= need) {
                int acc = above;
                int bin = 4 * lane + 3;
                const int hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int q = 3; q >= 0; q--) {
                    if (acc + hh[q] >= need) {
                        bin = 4 * lane + q;
                        break;
                    }
                    acc += hh[q];
                }
                s_sel[0] = bin;
                s_sel[1] = need - acc;
            }
        }
        __syncthreads();
        prefix |= (unsigned)s_sel[0] << shift;
        mask |= 255u << shift;
        need = s_sel[1];
        __syncthreads();
    }
    // prefix = key of the ke-th largest; `need` of the elements equal to it belong to the set, (ke - need) are greater
    if (threadIdx.x == 0) {
        s_sel[2] = 0;
    }
    __syncthreads();
    int ties = 0;
#pragma unroll
    for (int j = 0; j < MAXE; j++) {
        if (j < ne) {
            const unsigned key = fkey(vals[j]);
            if (key > prefix) {
                const int pos = atomicAdd(&s_sel[2], 1);
                put_cand<WT>(ov, oi, pos, vals[j], i0 + (int)threadIdx.x + 256 * j);
            }
            else if (key == prefix) {
                ties++;
            }
        }
    }
    int       tot  = 0;
    int       rank = block_excl_scan(ties, redi, tot);
    const int tie0 = ke - need;
    if (tot == need) {  // every element equal to the threshold belongs to the set (the usual case: exactly one)
#pragma unroll
        for (int j = 0; j < MAXE; j++) {
            if (j < ne && fkey(vals[j]) == prefix) {
                put_cand<WT>(ov, oi, tie0 + rank, vals[j], i0 + (int)threadIdx.x + 256 * j);
                rank++;
            }
        }
        return;
    }
    // more equal elements than places: the ones with the lowest indices (the tie rule of `better`) -- index order needs a
    // contiguous chunk per thread
    const int per = (n + 255) / 256;
    const int c0 = threadIdx.x * per, c1 = min(n, c0 + per);
    ties = 0;
    auto elem = [&](const int i) { return (mask_end && i == end_local) ? -FLT_MAX : l[i]; };
    for (int i = c0; i < c1; i++) {
        ties += fkey(elem(i)) == prefix ? 1 : 0;
    }
    rank = block_excl_scan(ties, redi, tot);
    for (int i = c0; i < c1 && rank < need; i++) {
        if (fkey(elem(i)) == prefix) {
            put_cand<WT>(ov, oi, tie0 + rank, elem(i), i0 + i);
            rank++;
        }
    }
}

__global__ __launch_bounds__(256) void k_topk_stage1(const SamplingParams p, float* cand_v, int* cand_i, int slice)
{
    if (p.state->all_finished) {
        return;  // a token of a multi-token graph behind the request's last one (engine.hip step()): nothing to do
    }
    __shared__ float redv[4];
    __shared__ int   redi[4];
    __shared__ int   hist[256];
    __shared__ int   hist8[8][257];
    __shared__ int   s_sel[4];  // [0] chosen bin, [1] still needed inside it, [2] emit counter
    const int        b = blockIdx.y, blk = blockIdx.x;
    int              k = p.top_k[b];
    const bool       topp = k == 0;
    if (topp) {
        k = TOPP_K;  // top-p rows: the head of the sorted order comes from these candidates
    }
    if (p.finished[b]) {
        return;
    }
    const int    V  = p.V;
    const int    i0 = blk * slice;
    const int    n  = max(0, min(slice, V - i0));
    const float* l  = p.logits + (size_t)b * V + i0;
    float*       ov = cand_v + ((size_t)b * TOPK_BLOCKS + blk) * TOPK_MAX;
    int*         oi = cand_i + ((size_t)b * TOPK_BLOCKS + blk) * TOPK_MAX;
    // the thread's elements i = tid + 256 j live in REGISTERS from here on (all loads in flight at once; the arg max, the
    // soft-max statistics, the four select passes and the emission read them there -- an earlier form staged the slice in LDS
    // and every pass serialised on its LDS read -> atomic dependency)
    float     vals[STAGE1_MAXE];
    const int ne = n > (int)threadIdx.x ? (n - (int)threadIdx.x + 255) / 256 : 0;
#pragma unroll
    for (int j = 0; j < STAGE1_MAXE; j++) {
        vals[j] = (j < ne) ? l[threadIdx.x + 256 * j] : -INFINITY;
    }
    VI    best{-INFINITY, 0x7fffffff};
    float lmax = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < STAGE1_MAXE; j++) {
        if (j < ne) {
            const float v = vals[j];
            const int   i = threadIdx.x + 256 * j;
            lmax          = fmaxf(lmax, v);
            if (best.i == 0x7fffffff || better(v, i, best.v, best.i)) {
                best.v = v;
                best.i = i;
            }
        }
    }
    __syncthreads();
    if (p.return_cum_log_probs && !topp) {  // softmax statistics of the slice: {max, sum of exp(v - max)}
        lmax = wave_max(lmax);
        if ((threadIdx.x & 63) == 0) {
            redv[threadIdx.x >> 6] = lmax;
        }
        __syncthreads();
        const float m = fmaxf(fmaxf(redv[0], redv[1]), fmaxf(redv[2], redv[3]));
        __syncthreads();
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < STAGE1_MAXE; j++) {
            if (j < ne) {
                se += __expf(vals[j] - m);
            }
        }
        se = wave_sum(se);
        if ((threadIdx.x & 63) == 0) {
            redv[threadIdx.x >> 6] = se;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* st = reinterpret_cast<float*>(cand_i + (size_t)p.B * TOPK_BLOCKS * TOPK_MAX) + ((size_t)b * TOPK_BLOCKS + blk) * 2;
            st[0]     = n > 0 ? m : -FLT_MAX;
            st[1]     = n > 0 ? ((redv[0] + redv[1]) + (redv[2] + redv[3])) : 0.f;
        }
        __syncthreads();
    }
    const int ke = k < n ? k : n;  // candidates this slice can supply
    for (int i = ke + threadIdx.x; i < k; i += 256) {
        ov[i] = -INFINITY;
        oi[i] = -1;
    }
    if (ke == 0) {
        return;
    }
    if (ke == 1) {  // greedy / k = 1: the arg max came with the load pass
        const VI r = block_best(best, redv, redi);
        if (threadIdx.x == 0) {
            ov[0] = r.v;
            oi[0] = i0 + r.i;
        }
        return;
    }
    slice_select<STAGE1_MAXE, false>(vals, ne, n, ke, l, i0, ov, oi, hist8, hist, s_sel, redi, false, -1);
}

// ---- step 3: merge + sample (one block per row) -----------------------------------------------------------------
// bitonic sort of n2 (a power of two) {value, id} pairs in LDS, best first (value desc, id asc)
__device__ __forceinline__ void bitonic_sort_best_first(float* v, int* id, const int n2)
{
    for (int sz = 2; sz <= n2; sz <<= 1) {
        for (int st = sz >> 1; st > 0; st >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < n2 / 2; t += blockDim.x) {
                const int  lo = ((t / st) * 2) * st + (t % st), hi = lo + st;
                const bool desc = ((lo & sz) == 0);  // this block of the network sorts best-first
                const bool hi_better = better(v[hi], id[hi], v[lo], id[lo]);
                if (hi_better == desc) {
                    const float tv = v[lo];
                    const int   ti = id[lo];
                    v[lo]  = v[hi];
                    id[lo] = id[hi];
                    v[hi]  = tv;
                    id[hi] = ti;
                }
            }
        }
    }
    __syncthreads();
}

// the same network over n2 pairs in GLOBAL memory (one workgroup): strides >= CH run on global memory, every run of
// strides < CH on CH-element chunks staged through LDS (lv / li: CH entries each)
constexpr int SORT_CH = 4096;
__device__ __forceinline__ void bitonic_sort_global(float* v, int* id, const int n2, float* lv, int* li)
{
    for (int sz = 2; sz <= n2; sz <<= 1) {
        int st = sz >> 1;
        for (; st >= SORT_CH; st >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < n2 / 2; t += blockDim.x) {
                const int  lo = ((t / st) * 2) * st + (t % st), hi = lo + st;
                const bool desc = ((lo & sz) == 0);
                const float vl = v[lo], vh = v[hi];
                const int   il = id[lo], ih = id[hi];
                if (better(vh, ih, vl, il) == desc) {
                    v[lo]  = vh;
                    id[lo] = ih;
                    v[hi]  = vl;
                    id[hi] = il;
                }
            }
        }
        if (st == 0) {
            continue;
        }
        __syncthreads();
        const int ch = n2 < SORT_CH ? n2 : SORT_CH;
        for (int c0 = 0; c0 < n2; c0 += ch) {
            for (int i = threadIdx.x; i < ch; i += blockDim.x) {
                lv[i] = v[c0 + i];
                li[i] = id[c0 + i];
            }
            for (int s2 = st; s2 > 0; s2 >>= 1) {
                __syncthreads();
                for (int t = threadIdx.x; t < ch / 2; t += blockDim.x) {
                    const int  lo = ((t / s2) * 2) * s2 + (t % s2), hi = lo + s2;
                    const bool desc = (((c0 + lo) & sz) == 0);
                    if (better(lv[hi], li[hi], lv[lo], li[lo]) == desc) {
                        const float tv = lv[lo];
                        const int   ti = li[lo];
                        lv[lo] = lv[hi];
                        li[lo] = li[hi];
                        lv[hi] = tv;
                        li[hi] = ti;
                    }
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < ch; i += blockDim.x) {
                v[c0 + i]  = lv[i];
                id[c0 + i] = li[i];
            }
            __syncthreads();
        }
    }
    __syncthreads();
}

// The kc best of the n2 candidates sv / si (LDS) -> sv / si [0, kc) sorted best first: radix select of the kc-th best (as stage 1,
// over the LDS copy), the winners compacted into the k2 = 2^ceil(log2 kc) slots tv / ti, THOSE sorted (k = 50: 21 network stages
// instead of 45 over 512 pairs)
// (h8: eight private copies of the histogram, picked by lane -- for unions of thousands of candidates, whose sign / exponent bytes
//  fall into a handful of bins and would serialise on one LDS word; NULL: the single histogram)
__device__ __forceinline__ void union_topk(float* sv, int* si, const int n2, const int kc, const int k2, float* tv, int* ti, int* hist,
                                           int* s_sel, int* redi, int (*h8)[257] = nullptr)
{
    __syncthreads();
    unsigned prefix = 0u, mask = 0u;
    int      need = kc;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[threadIdx.x] = 0;
        if (h8) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                h8[q][threadIdx.x] = 0;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < n2; c += 256) {
            const unsigned key = fkey(sv[c]);
            if ((key & mask) == prefix) {
                atomicAdd(h8 ? &h8[threadIdx.x & 7][(key >> shift) & 255u] : &hist[(key >> shift) & 255u], 1);
            }
        }
        __syncthreads();
        if (h8) {
            int t = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                t += h8[q][threadIdx.x];
            }
            hist[threadIdx.x] = t;
            __syncthreads();
        }
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int mine = h0 + h1 + h2 + h3;
            int       above = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(above, o, 64);
                if (lane + o < 64) {
                    above += t;
                }
            }
            above -= mine;
            if (above < need && above + mine >= need) {
                int       acc = above, bin = 4 * lane + 3;
                const int hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int q = 3; q >= 0; q--) {
                    if (acc + hh[q] >= need) {
                        bin = 4 * lane + q;
                        break;
                    }
                    acc += hh[q];
                }
                s_sel[0] = bin;
                s_sel[1] = need - acc;
            }
        }
        __syncthreads();
        prefix |= (unsigned)s_sel[0] << shift;
        mask |= 255u << shift;
        need = s_sel[1];
        __syncthreads();
    }
    // `need` of the entries equal to the threshold belong to the set: the ones with the smallest ids (the rule of `better`)
    if (threadIdx.x == 0) {
        s_sel[2] = 0;
    }
    for (int c = threadIdx.x; c < k2; c += 256) {
        tv[c] = -INFINITY;
        ti[c] = 0x7fffffff;
    }
    __syncthreads();
    // (the k-th best itself always equals the threshold: the usual case is exactly `need` such entries, all taken)
    int nties = 0;
    for (int c = threadIdx.x; c < n2; c += 256) {
        nties += fkey(sv[c]) == prefix ? 1 : 0;
    }
    int       tie_total = 0;
    (void)block_excl_scan(nties, redi, tie_total);
    const bool all_ties = tie_total == need;
    for (int c = threadIdx.x; c < n2; c += 256) {
        const unsigned key = fkey(sv[c]);
        bool           take = key > prefix || (all_ties && key == prefix);
        if (!all_ties && key == prefix) {
            int rank = 0;  // entries with the same value and a smaller id: real duplicates at the threshold, rare
            for (int d = 0; d < n2; d++) {
                rank += (fkey(sv[d]) == prefix && (si[d] < si[c] || (si[d] == si[c] && d < c))) ? 1 : 0;
            }
            take = rank < need;
        }
        if (take) {
            const int pos = atomicAdd(&s_sel[2], 1);
            tv[pos]       = sv[c];
            ti[pos]       = si[c];
        }
    }
    bitonic_sort_best_first(tv, ti, k2);
    for (int c = threadIdx.x; c < kc; c += 256) {
        sv[c] = tv[c];
        si[c] = ti[c];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_sample(const SamplingParams p, float* cand_v, int* cand_i)
{
    if (p.state->all_finished) {
        return;  // a token of a multi-token graph behind the request's last one (engine.hip step()): nothing to do
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float redv[4];
    __shared__ int   redi[4];
    __shared__ float s_rnd;
    __shared__ int   s_pick[2];
    const int        b = blockIdx.x;
    const int        V = p.V;
    const int        step = p.row_len ? p.row_len[b] + 1 : p.state->step;
    int*             out_id = p.output_ids + (size_t)step * p.B + b;
    if (p.finished[b]) {
        if (threadIdx.x == 0) {
            *out_id = p.end_id;  // sampling_topk_kernels.cu:239-242 ; top-p: arg max of the end mask
        }
        return;
    }
    const int k = p.top_k[b];
    float*    l = p.logits + (size_t)b * V;
    // the union of the slices' candidate sets, sorted best first: [0, n2) in LDS
    const int kc = k > 0 ? k : TOPP_K;
    int       n2 = 1;
    while (n2 < TOPK_BLOCKS * kc) {
        n2 <<= 1;
    }
    float* sv = reinterpret_cast<float*>(smem);    // [n2] values
    int*   si = reinterpret_cast<int*>(sv + n2);   // [n2] ids
    {
        const float* cv = cand_v + (size_t)b * TOPK_BLOCKS * TOPK_MAX;
        const int*   ci = cand_i + (size_t)b * TOPK_BLOCKS * TOPK_MAX;
        for (int c = threadIdx.x; c < n2; c += 256) {
            const int blk = c / kc, j = c % kc;
            float     v   = -INFINITY;
            int       id  = 0x7fffffff;
            if (c < TOPK_BLOCKS * kc && ci[blk * TOPK_MAX + j] >= 0) {
                v  = cv[blk * TOPK_MAX + j];
                id = ci[blk * TOPK_MAX + j];
            }
            sv[c] = v;
            si[c] = id;
        }
        int k2 = 1;
        while (k2 < kc) {
            k2 <<= 1;
        }
        if (k > 0 && kc > 1 && n2 >= 4 * k2) {
            // top-k rows need the k best of the union, not all of it in order: radix select of the k-th best (as stage 1, over
            // the LDS copy), the winners compacted into k2 = 2^ceil(log2 k) slots, THOSE sorted (k = 50: 21 network stages
            // instead of 45 over 512 pairs)
            __shared__ int hist[256];
            __shared__ int s_sel[4];
            union_topk(sv, si, n2, kc, k2, reinterpret_cast<float*>(si + n2), reinterpret_cast<int*>(si + n2) + k2, hist, s_sel, redi);
        }
        else if (kc > 1) {
            bitonic_sort_best_first(sv, si, n2);
        }
        else {  // one candidate per slice: the best of 8
            __syncthreads();
            VI best{-INFINITY, 0x7fffffff};
            if (threadIdx.x < TOPK_BLOCKS) {
                best.v = sv[threadIdx.x];
                best.i = si[threadIdx.x];
            }
            const VI r = block_best(best, redv, redi);
            if (threadIdx.x == 0) {
                sv[0] = r.v;
                si[0] = r.i;
            }
            __syncthreads();
        }
    }
    if (k > 0) {
        // ---- top-k layer (sampling_topk_kernels.cu:210-311): sv / si [0, k) are the row's k best ----
        if (threadIdx.x == 0) {
            const float smax = sv[0];
            float       ssum = 0.f;
            float       row_max = 0.f, row_den = 1.f;
            if (p.return_cum_log_probs) {  // addBiasSoftMax of the row (sampling_topp_kernels.cu:1296-1345) from the slice statistics
                const float* st = reinterpret_cast<const float*>(cand_i + (size_t)p.B * TOPK_BLOCKS * TOPK_MAX) + (size_t)b * TOPK_BLOCKS * 2;
                row_max         = -FLT_MAX;
                for (int q = 0; q < TOPK_BLOCKS; q++) {
                    row_max = fmaxf(row_max, st[2 * q]);
                }
                float tot = 0.f;
                for (int q = 0; q < TOPK_BLOCKS; q++) {
                    tot += st[2 * q + 1] * __expf(st[2 * q] - row_max);
                }
                row_den = tot + 1e-6f;
            }
            for (int i = 0; i < k; i++) {
                float u = sv[i];
                if (!p.return_cum_log_probs) {
                    u = __expf(u - smax);  // :271-275
                }
                else {
                    u = __expf(u - row_max) / row_den;  // the probability the reference's in-place softmax leaves there
                }
                sv[i] = u;
                ssum += u;
            }
            const float u01 = ftcf_uniform(p.random_seed[b], 0, p.draw_counter[b]);
            p.draw_counter[b] += 1;
            float rnd  = u01 * p.top_p_topk[b] * ssum;  // :283
            int   pick = k - 1;
            for (int i = 0; i < k; i++) {
                rnd -= sv[i];
                if (rnd <= 0.0f || i == k - 1) {
                    pick = i;
                    break;
                }
            }
            int id = si[pick];
            if (id == 0x7fffffff || id < 0) {
                id = 0;
            }
            *out_id = id;
            if (p.return_cum_log_probs && p.cum_log_probs) {
                p.cum_log_probs[b] += logf(sv[pick]);
            }
            p.seq_len[b] += 1;  // :305-308
            p.finished[b] = (id == p.end_id);
        }
    }
    else {
        // ---- top-p layer (sampling_topp_kernels.cu:802-1000): probabilities are in `l` ----
        // The reference sorts the whole row and walks it until rand * p <= cumulative probability.  Here the walk runs over
        // the sorted candidates (TOPP_K per slice): every candidate ABOVE the largest value a slice may have left out
        // (v_cut = max over the slices that supplied all TOPP_K of their smallest candidate) is at its exact place of the
        // full order, so the same fp32 additions happen in the same order.  Only a walk that gets past v_cut -- a nearly
        // flat distribution -- falls back to sorting the whole row in the workspace (bitonic network, below) and walking it from the
        // start.
        const float thr = p.top_p_topp[b];
        if (threadIdx.x == 0) {
            const float u01 = ftcf_uniform(p.random_seed[b], 0, p.draw_counter[b]);
            p.draw_counter[b] += 1;
            s_rnd = u01 * thr;
        }
        float v_cut = -INFINITY;
        {
            const float* cv = cand_v + (size_t)b * TOPK_BLOCKS * TOPK_MAX;
            const int*   ci = cand_i + (size_t)b * TOPK_BLOCKS * TOPK_MAX;
            for (int q = 0; q < TOPK_BLOCKS; q++) {
                if (ci[q * TOPK_MAX + TOPP_K - 1] >= 0) {  // the slice had at least TOPP_K elements: more may follow
                    float mn = INFINITY;
                    for (int j = threadIdx.x; j < TOPP_K; j += 256) {
                        mn = fminf(mn, cv[q * TOPK_MAX + j]);
                    }
                    mn = -wave_max(-mn);
                    if ((threadIdx.x & 63) == 0) {
                        redv[threadIdx.x >> 6] = mn;
                    }
                    __syncthreads();
                    v_cut = fmaxf(v_cut, fminf(fminf(redv[0], redv[1]), fminf(redv[2], redv[3])));
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float cum = 0.f;
            int   i   = 0;
            s_pick[0] = -1;
            for (; i < TOPK_BLOCKS * TOPP_K; i++) {
                if (si[i] == 0x7fffffff || !(sv[i] > v_cut)) {
                    break;  // past the part of the order the candidates are known to hold
                }
                cum += sv[i];
                if (s_rnd <= cum) {
                    s_pick[0] = i;
                    break;
                }
            }
            s_pick[1] = i;       // candidates consumed without reaching the threshold (when s_pick[0] < 0)
            redv[0]   = cum;
        }
        __syncthreads();
        int   id = 0;
        float pr = 0.f;
        if (s_pick[0] >= 0) {
            id = si[s_pick[0]];
            pr = sv[s_pick[0]];
        }
        else {
            // The walk left the part of the order the candidates hold (a flat distribution: > ~1000 tokens under top_p): sort
            // the WHOLE row, as the reference always does (one segmented radix sort, sampling_topp_kernels.cu:1015-1100), and
            // walk it from the start -- the same additions in the same order.  Bitonic network over the row's pairs in the
            // workspace by this one workgroup (short strides through LDS); the first version extracted one maximum per pass over V
            // (seconds per token in this regime).
            int NV = 1;
            while (NV < V) {
                NV <<= 1;
            }
            float* fv = reinterpret_cast<float*>(cand_i + (size_t)p.B * TOPK_BLOCKS * TOPK_MAX) + (size_t)p.B * TOPK_BLOCKS * 2
                        + (size_t)b * 2 * NV;
            int*   fi = reinterpret_cast<int*>(fv + NV);
            __syncthreads();
            for (int i = threadIdx.x; i < NV; i += 256) {
                fv[i] = i < V ? l[i] : -INFINITY;
                fi[i] = i < V ? i : 0x7fffffff;
            }
            bitonic_sort_global(fv, fi, NV, sv, reinterpret_cast<int*>(sv + SORT_CH));
            // sequential fp32 walk by thread 0 over chunks staged through LDS
            float cum = 0.f;
            int   sel = V - 1;
            for (int c0 = 0; c0 < V; c0 += 1024) {
                __syncthreads();
                for (int i = threadIdx.x; i < 1024; i += 256) {
                    sv[i] = c0 + i < V ? fv[c0 + i] : 0.f;
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    s_pick[0] = -1;
                    const int m = min(1024, V - c0);
                    for (int i = 0; i < m; i++) {
                        cum += sv[i];
                        if (s_rnd <= cum) {
                            s_pick[0] = c0 + i;
                            break;
                        }
                    }
                    redv[0] = cum;
                }
                __syncthreads();
                cum = redv[0];
                if (s_pick[0] >= 0) {
                    sel = s_pick[0];
                    break;
                }
            }
            id = fi[sel];
            pr = fv[sel];
        }
        if (threadIdx.x == 0) {
            *out_id = id;
            if (p.return_cum_log_probs && p.cum_log_probs) {
                p.cum_log_probs[b] += logf(pr);
            }
            p.seq_len[b] += 1;
            p.finished[b] = (id == p.end_id);
        }
    }
}

// ---- step 4: stop words, length criterion, bookkeeping (single block) -----------------------------------------
// DEFER_HOST: the caller writes the pinned host flags itself, as the launch's last stores (decode_publish_host)
template<bool DEFER_HOST = false>
__device__ __forceinline__ int decode_finish_body(const SamplingParams& p, const int step, const int steps_done = -1)
{
    __shared__ int s_all;
    if (threadIdx.x == 0) {
        s_all = 1;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
        if (p.stop_words) {  // stop_criteria_kernels.cu:24-83
            const int* words = p.stop_words + (size_t)b * 2 * p.stop_len;
            const int* offs  = words + p.stop_len;
            for (int id = 0; id < p.stop_len; id++) {
                if (offs[id] < 0) {
                    continue;
                }
                const int item_end = offs[id], item_start = id > 0 ? offs[id - 1] : 0;
                const int item_size = item_end - item_start;
                bool      stop      = false;
                if (step + 1 >= item_size) {
                    stop = true;
                    for (int t = item_size - 1; t >= 0; t--) {
                        const int prev = p.output_ids[(size_t)(step - (item_size - 1) + t) * p.B + b];
                        if (prev != words[item_start + t]) {
                            stop = false;
                            break;
                        }
                    }
                }
                if (stop) {
                    p.finished[b] = 1;
                }
            }
        }
        if (step >= p.total_len) {  // length_criterion (stop_criteria_kernels.cu:106-158), limit = total_len
            p.finished[b] = 1;
        }
        if (!p.finished[b]) {
            atomicAnd(&s_all, 0);
        }
        if (step == p.max_input_len) {  // invokeUpdatePaddingCount (gpt_kernels.cu:981-1033)
            p.pad_count[b] += p.max_input_len - p.input_lengths[b];
        }
    }
    __syncthreads();
    const int all = s_all;
    if (threadIdx.x == 0) {
        p.state->all_finished = all;
        p.state->steps_done   = (steps_done >= 0 ? steps_done : p.state->steps_done) + 1;
        if constexpr (!DEFER_HOST) {
            p.h_flags[1] = step;
            __threadfence_system();
            p.h_flags[0] = all;
        }
        p.state->step = step + 1;
    }
    return all;
}
// (the host reads the flags behind an event of the stream, never while the launch runs: engine.hip step())
__device__ __forceinline__ void decode_publish_host(const SamplingParams& p, const int step, const int all)
{
    if (threadIdx.x == 0) {
        p.h_flags[1] = step;
        p.h_flags[0] = all;
    }
}

__global__ void k_decode_finish(const SamplingParams p)
{
    if (p.state->all_finished) {
        return;  // a token of a multi-token graph behind the request's last one (engine.hip step()): nothing to do
    }
    decode_finish_body(p, p.state->step);
}

// ---------------------------------------------------------------------------------------------------------------------
// The tail of an all-greedy step, run by the ONE workgroup (256 threads) that drew the launch's last ticket (k_greedy_decode,
// k_lm_head_greedy): every row's token from the slices' partials part[B][nsl][4] = {best value, its id, slice max, slice sum of
// exponentials} exactly as k_sample does for k = 1 (sampling_topk_kernels.cu:210-311: probability of the best token under the
// row's soft-max, the uniform draw consumed, cum_log_probs, sequence length, finished), the step's bookkeeping
// (decode_finish_body) and the NEXT token's prologue (k_step_prologue: decoding_kernels.cu:145-191 embedding lookup + the
// step's rotary table).  Everything here is a chain of dependent memory round trips on the token's critical path (this tail
// was 10 of k_greedy_decode's 17 us): a row's state is requested together with its partials, the embedding row as soon as the
// token is known, the pinned host flags are the last stores.
// ---------------------------------------------------------------------------------------------------------------------
// TAGGED: the partials are granules {tag, value} (part: [B][nsl][4] 8-byte words, the value in the low half) written by
// workgroups that may still be running: a batch of partials is re-read until every tag is `tag` (bounded: a give-up sets
// p.h_flags[2] and the step is finished on what is there).
// The tail of a step behind the picks, by the workgroup (256 threads) that made them: the NEXT token's prologue (k_step_prologue:
// decoding_kernels.cu:145-191 embedding lookup + the step's rotary table), the step's bookkeeping (decode_finish_body) and the
// host flags.  s_ids: the tokens of rows 0..7 in LDS (the others are re-read from output_ids).
__device__ __forceinline__ void decode_step_tail(const SamplingParams& p, const int step, const int steps_done, const int* s_ids,
                                                 const bool by_wave)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();  // (thread 0's updates are this workgroup's own: a workgroup-scope barrier orders them, no agent-scope fence)
    if (p.next_x && by_wave) {
        // (many rows: one wave per row, a row's pieces requested together -- sixteen rows copied one after the other by the whole
        // workgroup were sixteen dependent memory round trips, 30 us of this tail)
        constexpr int NP = 8;  // 16-byte pieces per lane and pass: 8 KiB of a row
        for (int row = wid; row < p.B; row += 4) {
            const int  id  = row < 8 ? s_ids[row] : p.output_ids[(size_t)step * p.B + row];
            const f16* src = p.wte + (size_t)id * p.H;
            f16*       dst = p.next_x + (size_t)row * p.H;
            for (int i0 = 0; i0 < p.H; i0 += 64 * 8 * NP) {
                u32x4 v[NP];
#pragma unroll
                for (int u = 0; u < NP; u++) {
                    const int i = i0 + (u * 64 + lane) * 8;
                    v[u]        = i < p.H ? *reinterpret_cast<const u32x4*>(src + i) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < NP; u++) {
                    const int i = i0 + (u * 64 + lane) * 8;
                    if (i < p.H) {
                        *reinterpret_cast<u32x4*>(dst + i) = v[u];
                    }
                }
            }
        }
    }
    else if (p.next_x) {
        // the next token's embedding rows: requested now, under the bookkeeping below
        for (int row = 0; row < p.B; row++) {
            const int  id  = row < 8 ? s_ids[row] : p.output_ids[(size_t)step * p.B + row];
            const f16* src = p.wte + (size_t)id * p.H;
            f16*       dst = p.next_x + (size_t)row * p.H;
            for (int i = threadIdx.x * 8; i < p.H; i += blockDim.x * 8) {
                *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
            }
        }
    }
    const int all = decode_finish_body<true>(p, step, steps_done);  // (thread 0's value is the one that is used)
    if (p.next_x) {
        __syncthreads();  // (the padding counts of this step: decode_finish_body)
        const int nstep = step + 1;
        for (int row = 0; row < p.B; row++) {
            if ((int)threadIdx.x < p.rot / 2) {
                const int pos = (nstep - 1) - (p.pad_count ? p.pad_count[row] : 0);
                float     cs, sn;
                rotary_coef(threadIdx.x, p.rot, pos, cs, sn);
                p.rot_table[((size_t)row * (p.rot / 2) + threadIdx.x) * 2]     = cs;
                p.rot_table[((size_t)row * (p.rot / 2) + threadIdx.x) * 2 + 1] = sn;
            }
        }
    }
    decode_publish_host(p, step, all);
}

constexpr int GREEDY_MAXQ = 4;  // partials of a row a thread requests together
constexpr int GREEDY_SPINS = 1 << 20;
template<bool TAGGED = false>
__device__ __forceinline__ void greedy_finish(const SamplingParams& p, const float* part, const int nsl, const int step,
                                              const unsigned tag = 0u)
{
    __shared__ float x_v[2][4], x_m[2][4], x_s[2][4];
    __shared__ int   x_i[2][4];
    __shared__ int   s_ids[8];
    typedef __attribute__((address_space(1))) unsigned gu32;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int       steps_done = 0;
    if (threadIdx.x == 0) {
        steps_done = p.state->steps_done;
    }
    int picked = 0;
    // batches of more than four rows (k_greedy_decode's 32 slices per row): ONE WAVE per row, four rows at a time -- no workgroup
    // barrier per row (16 rows took 70 us one after the other; the same picks, the same arithmetic per row)
    const bool by_wave = !TAGGED && nsl <= 64 && p.B > 4;
    if (by_wave) {
        for (int row = wid; row < p.B; row += 4) {
            int* out_id = p.output_ids + (size_t)step * p.B + row;
            if (p.finished[row]) {
                if (lane == 0) {
                    *out_id = p.end_id;  // sampling_topk_kernels.cu:239-242
                    if (row < 8) {
                        s_ids[row] = p.end_id;
                    }
                }
                continue;
            }
            uint64_t draws = 0;
            float    cum   = 0.f;
            int      slen  = 0;
            if (lane == 0) {
                draws = p.draw_counter[row];
                slen  = p.seq_len[row];
                if (p.return_cum_log_probs && p.cum_log_probs) {
                    cum = p.cum_log_probs[row];
                }
            }
            const gu32* q = (const gu32*)(part + (size_t)row * nsl * 4);
            VI          cand{-INFINITY, 0x7fffffff};
            float       mq = -FLT_MAX, sq = 0.f;
            if (lane < nsl) {
                cand.v = __uint_as_float(__hip_atomic_load(q + (size_t)lane * 4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                cand.i = (int)__hip_atomic_load(q + (size_t)lane * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mq     = __uint_as_float(__hip_atomic_load(q + (size_t)lane * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                sq     = __uint_as_float(__hip_atomic_load(q + (size_t)lane * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            const VI    r       = wave_best(cand);
            const float row_max = wave_max(mq);
            const float tot     = wave_sum(sq * __expf(mq - row_max));  // (lanes without a slice add 0 * exp(.) = 0)
            if (lane == 0) {
                float prob = 1.f;
                if (p.return_cum_log_probs) {
                    prob = __expf(r.v - row_max) / (tot + 1e-6f);
                }
                p.draw_counter[row] = draws + 1;
                int id = r.i;
                if (id == 0x7fffffff || id < 0) {
                    id = 0;
                }
                *out_id = id;
                if (p.return_cum_log_probs && p.cum_log_probs) {
                    p.cum_log_probs[row] = cum + logf(prob);
                }
                p.seq_len[row]  = slen + 1;
                p.finished[row] = (id == p.end_id);
                if (row < 8) {
                    s_ids[row] = id;
                }
            }
        }
    }
    for (int row = 0; row < (by_wave ? 0 : p.B); row++) {
        int* out_id = p.output_ids + (size_t)step * p.B + row;
        if (p.finished[row]) {
            if (threadIdx.x == 0) {
                *out_id = p.end_id;  // sampling_topk_kernels.cu:239-242
                if (row < 8) {
                    s_ids[row] = p.end_id;
                }
            }
            continue;
        }
        // thread 0's view of the row's state travels with the partials
        uint64_t draws = 0;
        float    cum   = 0.f;
        int      slen  = 0;
        if (threadIdx.x == 0) {
            draws = p.draw_counter[row];
            slen  = p.seq_len[row];
            if (p.return_cum_log_probs && p.cum_log_probs) {
                cum = p.cum_log_probs[row];
            }
        }
        const gu32* q = (const gu32*)(part + (size_t)row * nsl * 4);
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        const gu64* qg = (const gu64*)(reinterpret_cast<const unsigned long long*>(part) + (size_t)row * nsl * 4);
        VI    cand{-INFINITY, 0x7fffffff};
        float mq = -FLT_MAX, sq = 0.f;
#pragma unroll 1
        for (int j0 = 0; j0 < nsl; j0 += 256 * GREEDY_MAXQ) {  // (GREEDY_MAXQ partials of a thread travel together)
            unsigned pv[GREEDY_MAXQ][4];
            if constexpr (TAGGED) {
                int spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < GREEDY_MAXQ; j++) {
                        const int sl = j0 + threadIdx.x + 256 * j;
                        const int sc = sl < nsl ? sl : nsl - 1;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const unsigned long long g = __hip_atomic_load(qg + (size_t)sc * 4 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            pv[j][e] = (unsigned)g;
                            ok &= (unsigned)(g >> 32) == tag;
                        }
                    }
                    if (__syncthreads_and(ok ? 1 : 0)) {
                        break;
                    }
                    if (++spins > GREEDY_SPINS) {
                        if (threadIdx.x == 0) {
                            p.h_flags[2] = 1;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            else {
#pragma unroll
                for (int j = 0; j < GREEDY_MAXQ; j++) {
                    const int sl = j0 + threadIdx.x + 256 * j;
                    const int sc = sl < nsl ? sl : nsl - 1;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        pv[j][e] = __hip_atomic_load(q + (size_t)sc * 4 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < GREEDY_MAXQ; j++) {
                if (j0 + (int)threadIdx.x + 256 * j < nsl) {
                    const float v = __uint_as_float(pv[j][0]);
                    const int   i = (int)pv[j][1];
                    if (better(v, i, cand.v, cand.i)) {
                        cand.v = v;
                        cand.i = i;
                    }
                    const float m2 = __uint_as_float(pv[j][2]), s2 = __uint_as_float(pv[j][3]);
                    const float mn = fmaxf(mq, m2);
                    sq             = sq * __expf(mq - mn) + s2 * __expf(m2 - mn);
                    mq             = mn;
                }
            }
        }
        const VI    wb = wave_best(cand);
        const float wm = wave_max(mq);
        const float ws = wave_sum(sq * __expf(mq - wm));  // (threads without a slice add 0 * exp(.) = 0)
        const int   pb = (picked++) & 1;                  // (two sets of slots: one barrier per row)
        if (lane == 0) {
            x_v[pb][wid] = wb.v;
            x_i[pb][wid] = wb.i;
            x_m[pb][wid] = wm;
            x_s[pb][wid] = ws;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            VI    r{x_v[pb][0], x_i[pb][0]};
            float row_max = x_m[pb][0];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                if (better(x_v[pb][w], x_i[pb][w], r.v, r.i)) {
                    r.v = x_v[pb][w];
                    r.i = x_i[pb][w];
                }
                row_max = fmaxf(row_max, x_m[pb][w]);
            }
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                tot += x_s[pb][w] * __expf(x_m[pb][w] - row_max);
            }
            float prob = 1.f;
            if (p.return_cum_log_probs) {
                prob = __expf(r.v - row_max) / (tot + 1e-6f);
            }
            p.draw_counter[row] = draws + 1;  // (the top-k layer draws its uniform number for k = 1 too: sampling_topk_kernels.cu:283)
            int id = r.i;
            if (id == 0x7fffffff || id < 0) {
                id = 0;
            }
            *out_id = id;
            if (p.return_cum_log_probs && p.cum_log_probs) {
                p.cum_log_probs[row] = cum + logf(prob);
            }
            p.seq_len[row]  = slen + 1;  // :305-308
            p.finished[row] = (id == p.end_id);
            if (row < 8) {
                s_ids[row] = id;
            }
        }
    }
    decode_step_tail(p, step, steps_done, s_ids, by_wave);
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole dynamic-decode step of an ALL-GREEDY batch in ONE launch (round 4).  The general pipeline is four launches
// (k_decode_prep 4.7 us, k_topk_stage1 11.8, k_sample 7.5, k_decode_finish 4.2 at V = 100864: 28 us of a 2.66 ms token, and 2 %
// of a TP = 8 rank's token); when every row has top_k = 1 and nothing touches the logits but the min-length mask -- the
// reference's default request, and the headline's -- the same results come out of one: 32 slices per row keep their logits in
// registers for the arg max (`better`: highest value, lowest id) and the soft-max statistics return_cum_log_probs needs
// (sampling_topp_kernels.cu:1296-1345 addBiasSoftMax), write-through partials + a ticket, and the workgroup that draws the
// last ticket picks every row's token exactly as k_sample does for k = 1 (sampling_topk_kernels.cu:210-311: probability of the
// best token under the row's soft-max, the uniform draw consumed, cum_log_probs, sequence length, finished) and then runs the
// step's bookkeeping (decode_finish_body: stop words, length criterion, padding counts, step counter).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GREEDY_SLICES = 32;
constexpr int GREEDY_MAXE   = 16;  // logits per thread: V <= 32 * 256 * 16 = 131072

__global__ __launch_bounds__(256) void k_greedy_decode(const SamplingParams p, float* part)
{
    if (p.state->all_finished) {
        return;
    }
    const int step0 = p.state->step;  // (same cache line as the flag: the finishing workgroup does not pay a round trip for it)
    __shared__ float redv[4];
    __shared__ int   redi[4];
    __shared__ int   s_last;
    const int        b = blockIdx.y, blk = blockIdx.x, V = p.V;
    const int        slice = (V + GREEDY_SLICES - 1) / GREEDY_SLICES;
    if (!p.finished[b]) {
        const int    i0 = blk * slice;
        const int    n  = max(0, min(slice, V - i0));
        const float* l  = p.logits + (size_t)b * V + i0;
        // min_length (sampling_penalty_kernels.cu:485-520): end_id cannot be chosen yet
        const bool mask_end = p.min_length && (p.seq_len[b] + 1 - p.max_input_len < p.min_length[b]);
        float      vals[GREEDY_MAXE];
        const int  ne = n > (int)threadIdx.x ? (n - (int)threadIdx.x + 255) / 256 : 0;
#pragma unroll
        for (int j = 0; j < GREEDY_MAXE; j++) {
            vals[j] = (j < ne) ? l[threadIdx.x + 256 * j] : -INFINITY;
        }
        VI    best{-INFINITY, 0x7fffffff};
        float lmax = -FLT_MAX;
#pragma unroll
        for (int j = 0; j < GREEDY_MAXE; j++) {
            if (j < ne) {
                const int i = threadIdx.x + 256 * j;
                if (mask_end && i0 + i == p.end_id) {
                    vals[j] = -FLT_MAX;
                }
                const float v = vals[j];
                lmax          = fmaxf(lmax, v);
                if (best.i == 0x7fffffff || better(v, i, best.v, best.i)) {
                    best.v = v;
                    best.i = i;
                }
            }
        }
        lmax = wave_max(lmax);
        if ((threadIdx.x & 63) == 0) {
            redv[threadIdx.x >> 6] = lmax;
        }
        __syncthreads();
        const float m = fmaxf(fmaxf(redv[0], redv[1]), fmaxf(redv[2], redv[3]));
        __syncthreads();
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < GREEDY_MAXE; j++) {
            if (j < ne) {
                se += __expf(vals[j] - m);
            }
        }
        se = wave_sum(se);
        if ((threadIdx.x & 63) == 0) {
            redv[threadIdx.x >> 6] = se;
        }
        __syncthreads();
        const float sum = (redv[0] + redv[1]) + (redv[2] + redv[3]);
        __syncthreads();
        const VI r = block_best(best, redv, redi);
        if (threadIdx.x == 0) {
            // {best value, its id, slice max, slice sum of exponentials}: write-through stores (the reader may sit on another XCD)
            typedef __attribute__((address_space(1))) unsigned gu32;
            gu32* o = (gu32*)(part + ((size_t)b * GREEDY_SLICES + blk) * 4);
            __hip_atomic_store(o + 0, __float_as_uint(n > 0 ? r.v : -INFINITY), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 1, (unsigned)(n > 0 ? i0 + r.i : 0x7fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 2, __float_as_uint(n > 0 ? m : -FLT_MAX), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 3, __float_as_uint(n > 0 ? sum : 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- ticket: the last workgroup of the launch finishes the step ----
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's partial has been acknowledged (write-through)
        const int total = GREEDY_SLICES * p.B;
        const int t     = __hip_atomic_fetch_add(&p.state->pad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last          = (t == total - 1) ? 1 : 0;
        if (s_last) {
            __hip_atomic_store(&p.state->pad, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next token
        }
    }
    __syncthreads();
    if (!s_last) {
        return;
    }
    greedy_finish(p, part, GREEDY_SLICES, step0);
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole dynamic-decode step of a plain TOP-K batch in ONE launch (round 5): the reference harness's default request
// (examples/pytorch/codefuse/codefuse_example.py:799-806: top_k = 50, top_p 0, temperature 1, no penalty) took k_decode_prep +
// k_topk_stage1 + k_sample + k_decode_finish + the next token's k_step_prologue behind the LM head -- 61 us of a 2.7 ms token
// where the all-greedy step takes 12.  Same structure as k_greedy_decode: 32 slices per row keep their logits in registers (min_length
// mask, soft-max statistics for return_cum_log_probs, the radix select of the slice's k best: slice_select), write-through
// candidates + a ticket, and the workgroup that draws the last ticket merges every row's 32 candidate sets (union_topk), turns
// the k best into probabilities and draws exactly as k_sample does (sampling_topk_kernels.cu:210-311), then runs the step's tail
// (decode_step_tail).  Rows: top_k in [1, 64], no top-p row, temperature 1, no repetition penalty, no optional-token list,
// at most TKD_MAXB rows (the finishing workgroup takes them one after the other).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TKD_SLICES = 32;
constexpr int TKD_MAXE   = 16;  // logits per thread: V <= 32 * 256 * 16 = 131072
constexpr int TKD_MAXK   = 64;
constexpr int TKD_MAXB   = 4;

__global__ __launch_bounds__(256) void k_topk_decode(const SamplingParams p, float* ws)
{
    if (p.state->all_finished) {
        return;
    }
    const int step0 = p.state->step;
    __shared__ float redv[4];
    __shared__ int   redi[4];
    __shared__ int   hist[256];
    __shared__ int   hist8[8][257];
    __shared__ int   s_sel[4];
    __shared__ int   s_last;
    typedef __attribute__((address_space(1))) unsigned gu32;
    const int b = blockIdx.y, blk = blockIdx.x, V = p.V;
    // candidates [B][SLICES][MAXK] values, then ids, then {max, sum of exponentials} per slice
    float* cand_v = ws;
    int*   cand_i = reinterpret_cast<int*>(ws + (size_t)p.B * TKD_SLICES * TKD_MAXK);
    float* stats  = ws + (size_t)2 * p.B * TKD_SLICES * TKD_MAXK;
    const int slice = (V + TKD_SLICES - 1) / TKD_SLICES;
    if (!p.finished[b]) {
        const int    k  = p.top_k[b];
        const int    i0 = blk * slice;
        const int    n  = max(0, min(slice, V - i0));
        const float* l  = p.logits + (size_t)b * V + i0;
        float*       ov = cand_v + ((size_t)b * TKD_SLICES + blk) * TKD_MAXK;
        int*         oi = cand_i + ((size_t)b * TKD_SLICES + blk) * TKD_MAXK;
        // min_length (sampling_penalty_kernels.cu:485-520): end_id cannot be chosen yet
        const bool mask_end = p.min_length && (p.seq_len[b] + 1 - p.max_input_len < p.min_length[b]);
        float      vals[TKD_MAXE];
        const int  ne = n > (int)threadIdx.x ? (n - (int)threadIdx.x + 255) / 256 : 0;
#pragma unroll
        for (int j = 0; j < TKD_MAXE; j++) {
            vals[j] = (j < ne) ? l[threadIdx.x + 256 * j] : -INFINITY;
        }
        VI    best{-INFINITY, 0x7fffffff};
        float lmax = -FLT_MAX;
#pragma unroll
        for (int j = 0; j < TKD_MAXE; j++) {
            if (j < ne) {
                const int i = threadIdx.x + 256 * j;
                if (mask_end && i0 + i == p.end_id) {
                    vals[j] = -FLT_MAX;
                }
                const float v = vals[j];
                lmax          = fmaxf(lmax, v);
                if (best.i == 0x7fffffff || better(v, i, best.v, best.i)) {
                    best.v = v;
                    best.i = i;
                }
            }
        }
        if (p.return_cum_log_probs) {  // soft-max statistics of the slice: {max, sum of exp(v - max)}
            lmax = wave_max(lmax);
            if ((threadIdx.x & 63) == 0) {
                redv[threadIdx.x >> 6] = lmax;
            }
            __syncthreads();
            const float m = fmaxf(fmaxf(redv[0], redv[1]), fmaxf(redv[2], redv[3]));
            __syncthreads();
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < TKD_MAXE; j++) {
                if (j < ne) {
                    se += __expf(vals[j] - m);
                }
            }
            se = wave_sum(se);
            if ((threadIdx.x & 63) == 0) {
                redv[threadIdx.x >> 6] = se;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                gu32* st = (gu32*)(stats + ((size_t)b * TKD_SLICES + blk) * 2);
                __hip_atomic_store(st + 0, __float_as_uint(n > 0 ? m : -FLT_MAX), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(st + 1, __float_as_uint(n > 0 ? ((redv[0] + redv[1]) + (redv[2] + redv[3])) : 0.f), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        const int ke = k < n ? k : n;  // candidates this slice can supply
        for (int i = ke + threadIdx.x; i < k; i += 256) {
            put_cand<true>(ov, oi, i, -INFINITY, -1);
        }
        if (ke == 1) {
            const VI r = block_best(best, redv, redi);
            if (threadIdx.x == 0) {
                put_cand<true>(ov, oi, 0, r.v, i0 + r.i);
            }
        }
        else if (ke > 1) {
            slice_select<TKD_MAXE, true>(vals, ne, n, ke, l, i0, ov, oi, hist8, hist, s_sel, redi, mask_end, p.end_id - i0);
        }
    }
    // ---- ticket: the last workgroup of the launch finishes the step ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's candidates have been acknowledged (write-through)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = TKD_SLICES * p.B;
        const int t     = __hip_atomic_fetch_add(&p.state->pad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last          = (t == total - 1) ? 1 : 0;
        if (s_last) {
            __hip_atomic_store(&p.state->pad, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next token
        }
    }
    __syncthreads();
    if (!s_last) {
        return;
    }
    // ---- the finishing workgroup: every row's pick ----
    __shared__ float sv[TKD_SLICES * TKD_MAXK];
    __shared__ int   si[TKD_SLICES * TKD_MAXK];
    __shared__ float tv[TKD_MAXK];
    __shared__ int   ti[TKD_MAXK];
    __shared__ int   s_ids[8];
    int steps_done = 0;
    if (threadIdx.x == 0) {
        steps_done = p.state->steps_done;
    }
    for (int row = 0; row < p.B; row++) {
        int* out_id = p.output_ids + (size_t)step0 * p.B + row;
        if (p.finished[row]) {
            if (threadIdx.x == 0) {
                *out_id = p.end_id;  // sampling_topk_kernels.cu:239-242
                if (row < 8) {
                    s_ids[row] = p.end_id;
                }
            }
            continue;
        }
        const int k = p.top_k[row];
        int       n2 = 1;
        while (n2 < TKD_SLICES * k) {
            n2 <<= 1;
        }
        int k2 = 1;
        while (k2 < k) {
            k2 <<= 1;
        }
        __syncthreads();  // (the previous row's thread 0 has read sv / si)
        {
            const gu32* cv = (const gu32*)(cand_v + (size_t)row * TKD_SLICES * TKD_MAXK);
            const gu32* ci = (const gu32*)(cand_i + (size_t)row * TKD_SLICES * TKD_MAXK);
            // (all of a thread's candidates requested together: one round trip, not sixteen)
            constexpr int NC = TKD_SLICES * TKD_MAXK / 256;
            unsigned      rv[NC], ri[NC];
#pragma unroll
            for (int j = 0; j < NC; j++) {
                const int c  = threadIdx.x + 256 * j;
                const int cc = c < TKD_SLICES * k ? c : 0;
                const int q = cc / k, e = cc - q * k;
                ri[j] = __hip_atomic_load(ci + q * TKD_MAXK + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                rv[j] = __hip_atomic_load(cv + q * TKD_MAXK + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < NC; j++) {
                const int c = threadIdx.x + 256 * j;
                if (c < n2) {
                    const bool ok = c < TKD_SLICES * k && (int)ri[j] >= 0;
                    sv[c]         = ok ? __uint_as_float(rv[j]) : -INFINITY;
                    si[c]         = ok ? (int)ri[j] : 0x7fffffff;
                }
            }
        }
        if (k > 1 && n2 >= 4 * k2) {
            union_topk(sv, si, n2, k, k2, tv, ti, hist, s_sel, redi, hist8);
        }
        else if (k > 1) {
            bitonic_sort_best_first(sv, si, n2);
        }
        else {  // one candidate per slice: the best of 32
            __syncthreads();
            VI best{-INFINITY, 0x7fffffff};
            if (threadIdx.x < TKD_SLICES) {
                best.v = sv[threadIdx.x];
                best.i = si[threadIdx.x];
            }
            const VI r = block_best(best, redv, redi);
            if (threadIdx.x == 0) {
                sv[0] = r.v;
                si[0] = r.i;
            }
            __syncthreads();
        }
        // ---- top-k layer (sampling_topk_kernels.cu:210-311): sv / si [0, k) are the row's k best ----
        if (threadIdx.x == 0) {
            const float smax = sv[0];
            float       ssum = 0.f;
            float       row_max = 0.f, row_den = 1.f;
            if (p.return_cum_log_probs) {  // addBiasSoftMax of the row (sampling_topp_kernels.cu:1296-1345) from the slice statistics
                const gu32* st = (const gu32*)(stats + (size_t)row * TKD_SLICES * 2);
                float       sm[TKD_SLICES], ss[TKD_SLICES];
                row_max = -FLT_MAX;
                for (int q = 0; q < TKD_SLICES; q++) {
                    sm[q]   = __uint_as_float(__hip_atomic_load(st + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    ss[q]   = __uint_as_float(__hip_atomic_load(st + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    row_max = fmaxf(row_max, sm[q]);
                }
                float tot = 0.f;
                for (int q = 0; q < TKD_SLICES; q++) {
                    tot += ss[q] * __expf(sm[q] - row_max);
                }
                row_den = tot + 1e-6f;
            }
            for (int i = 0; i < k; i++) {
                float u = sv[i];
                if (!p.return_cum_log_probs) {
                    u = __expf(u - smax);  // :271-275
                }
                else {
                    u = __expf(u - row_max) / row_den;  // the probability the reference's in-place softmax leaves there
                }
                sv[i] = u;
                ssum += u;
            }
            const float u01 = ftcf_uniform(p.random_seed[row], 0, p.draw_counter[row]);
            p.draw_counter[row] += 1;
            float rnd  = u01 * p.top_p_topk[row] * ssum;  // :283
            int   pick = k - 1;
            for (int i = 0; i < k; i++) {
                rnd -= sv[i];
                if (rnd <= 0.0f || i == k - 1) {
                    pick = i;
                    break;
                }
            }
            int id = si[pick];
            if (id == 0x7fffffff || id < 0) {
                id = 0;
            }
            *out_id = id;
            if (p.return_cum_log_probs && p.cum_log_probs) {
                p.cum_log_probs[row] += logf(sv[pick]);
            }
            p.seq_len[row] += 1;  // :305-308
            p.finished[row] = (id == p.end_id);
            if (row < 8) {
                s_ids[row] = id;
            }
        }
    }
    decode_step_tail(p, step0, steps_done, s_ids, false);
}

// ---------------------------------------------------------------------------------------------------------------------
// LM head + the all-greedy dynamic decode of the token in ONE launch (one GPU, one or two rows: the headline's token).  The wave
// that produces a logit keeps the arg max and the soft-max statistics of its vocabulary rows as it goes (wave-uniform
// registers: the logits never come back from memory), the workgroup publishes ONE partial per row as granules {step tag,
// value}, and the workgroup dispatched LAST runs greedy_finish on them, re-reading until every tag is this token's.  Against
// k_lm_head + k_greedy_decode: one launch boundary, the re-read of the logits and the slices' reductions leave the token's
// critical path.  The logits are still stored (debug taps, tests).  (A first version elected the finisher with two levels of
// tickets: 192 -> 182 us per launch against 160 + 14.7 for the two kernels -- 2048 arrivals inside the launch's last
// microseconds cost more than the boundary they replace.)
// ---------------------------------------------------------------------------------------------------------------------
template<int M>
__global__ __launch_bounds__(256, 8) void k_lm_head_greedy(const f16* __restrict__ x, const f16* __restrict__ W,
                                                        float* __restrict__ logits, const int K,
                                                        const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                        const float eps, const SamplingParams p, unsigned long long* part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.state->all_finished) {
        return;  // every row has finished: a token of a multi-token graph behind the request's last one
    }
    const int step0 = p.state->step;
    __shared__ float w_v[4][M], w_m[4][M], w_s[4][M];
    __shared__ int   w_i[4][M];
    f16*   xs  = reinterpret_cast<f16*>(smem);  // [M][K]
    float* red = reinterpret_cast<float*>(smem + (size_t)M * K * 2);
    // min_length (sampling_penalty_kernels.cu:485-520): end_id cannot be chosen yet
    bool mask_end[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
        mask_end[m] = p.min_length && (p.seq_len[m] + 1 - p.max_input_len < p.min_length[m]);
    }
    lm_head_stage_x<M>(x, K, gamma, beta, eps, xs, red);
    VI    best[M];
    float vmax[M], ssum[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
        best[m] = VI{-INFINITY, 0x7fffffff};
        vmax[m] = -FLT_MAX;
        ssum[m] = 0.f;
    }
    lm_head_rows<M>(W, logits, p.V, K, p.V, xs, [&](const int m, const int row, float v) {
#pragma unroll
        for (int mm = 0; mm < M; mm++) {  // (compile-time indices keep the statistics in registers)
            if (mm == m) {
                if (mask_end[mm] && row == p.end_id) {
                    v = -FLT_MAX;
                }
                if (best[mm].i == 0x7fffffff || better(v, row, best[mm].v, best[mm].i)) {
                    best[mm].v = v;
                    best[mm].i = row;
                }
                const float mn = fmaxf(vmax[mm], v);
                ssum[mm]       = ssum[mm] * __expf(vmax[mm] - mn) + __expf(v - mn);
                vmax[mm]       = mn;
            }
        }
    });
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            w_v[wid][m] = best[m].v;
            w_i[wid][m] = best[m].i;
            w_m[wid][m] = vmax[m];
            w_s[wid][m] = ssum[m];
        }
    }
    __syncthreads();
    const int      nb  = gridDim.x;
    const unsigned tag = (unsigned)step0 + 1u;
    if (threadIdx.x == 0) {
        typedef __attribute__((address_space(1))) unsigned long long gu64;
#pragma unroll
        for (int m = 0; m < M; m++) {
            VI    r{w_v[0][m], w_i[0][m]};
            float mx = w_m[0][m];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                if (better(w_v[w][m], w_i[w][m], r.v, r.i)) {
                    r.v = w_v[w][m];
                    r.i = w_i[w][m];
                }
                mx = fmaxf(mx, w_m[w][m]);
            }
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                sum += w_s[w][m] * __expf(w_m[w][m] - mx);
            }
            // granules {tag, value}, write-through (the reader sits on another XCD); a workgroup without a vocabulary row leaves
            // the neutral partial {-inf, no id, -FLT_MAX, 0}
            gu64*                    o  = (gu64*)(part + ((size_t)m * nb + blockIdx.x) * 4);
            const unsigned long long hi = (unsigned long long)tag << 32;
            __hip_atomic_store(o + 0, hi | __float_as_uint(r.v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 1, hi | (unsigned)r.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 2, hi | __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 3, hi | __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if ((int)blockIdx.x != nb - 1) {
        return;  // nothing waits for these stores
    }
    // the workgroup dispatched last closes the step: no ticket (2048 arrivals at a few addresses inside the launch's last
    // microseconds cost more than the launch boundary they replace: measured), it re-reads the partials until they carry this
    // token's tag
    greedy_finish<true>(p, reinterpret_cast<const float*>(part), nb, step0, tag);
}

size_t lm_head_greedy_partial_bytes(int B)
{
    return (size_t)B * 2048 * 4 * sizeof(unsigned long long);  // granules [B][<= 2048 workgroups][4]
}

static bool dynamic_decode_is_greedy_fused(const SamplingParams& p, bool finish);
bool lm_head_greedy_ok(const SamplingParams& p, int K)
{
    const int on = getenv("FTCF_LM_GREEDY") ? atoi(getenv("FTCF_LM_GREEDY")) : 1;  // (read per call: the tests switch the forms)
    // (one or two rows -- the persistent kernel's batch sizes: the kernel is held to 64 VGPRs for eight waves per SIMD, and the
    // LM head's own loop needs 68 / 76 at three / four rows)
    return on && p.B <= 2 && K % 8 == 0 && dynamic_decode_is_greedy_fused(p, true) && p.next_x
           && lm_head_greedy_partial_bytes(p.B) <= sampling_workspace_bytes(p.B, p.V);
}

void launch_lm_head_greedy(const f16* x, const f16* W, float* logits, int K, const f16* gamma, const f16* beta, float eps,
                           const SamplingParams& p, hipStream_t s)
{
    FTCF_CHECK_ARG(lm_head_greedy_ok(p, K), "LM head + greedy decode: not an eligible step");
    const size_t smem = (size_t)p.B * (K + 8) * 2 + 64;
    int          grid = (p.V + 15) / 16;
    if (grid > 2048) {
        grid = 2048;
    }
    unsigned long long* part = reinterpret_cast<unsigned long long*>(p.ws);
#define FTCF_LMG(MM)                                                                                                   \
    hipLaunchKernelGGL((k_lm_head_greedy<MM>), dim3(grid), dim3(256), smem, s, x, W, logits, K, gamma, beta, eps, p, part)
    if (p.B == 1) {
        FTCF_LMG(1);
    }
    else {
        FTCF_LMG(2);
    }
#undef FTCF_LMG
    FTCF_HIP_CHECK(hipGetLastError());
}

void launch_decode_finish(const SamplingParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(k_decode_finish, dim3(1), dim3(64), 0, s, p);
    FTCF_HIP_CHECK(hipGetLastError());
}

size_t sampling_workspace_bytes(int B, int V)
{
    (void)V;
    // candidate values + ids of the stage-1 slices, then {max, sum of exp} per slice
    size_t nv = 1;
    while (nv < (size_t)V) {
        nv <<= 1;
    }
    // ... then, per row, the {value, id} pairs of the whole row padded to a power of two (full sort of a top-p row)
    return (size_t)B * TOPK_BLOCKS * TOPK_MAX * (sizeof(float) + sizeof(int)) + (size_t)B * TOPK_BLOCKS * 2 * sizeof(float)
           + (size_t)B * 2 * nv * sizeof(float);
}

// the one-launch step of a plain top-k batch (k_topk_decode)
static bool dynamic_decode_is_topk_fused(const SamplingParams& p, bool finish)
{
    const int on = getenv("FTCF_TOPK_FUSED") ? atoi(getenv("FTCF_TOPK_FUSED")) : 1;  // (read per call: the tests switch the forms)
    return on && finish && p.max_top_k >= 2 && p.max_top_k <= TKD_MAXK && !p.any_top_p && !p.apply_temperature
           && !p.apply_repetition && !p.optional_last_tokens && !p.row_len && p.V <= TKD_SLICES * 256 * TKD_MAXE && p.B <= TKD_MAXB
           && p.rot / 2 <= 256
           && (size_t)p.B * TKD_SLICES * (TKD_MAXK * 8 + 8) <= sampling_workspace_bytes(p.B, p.V);
}

// the one-launch step of an all-greedy batch (k_greedy_decode; k_lm_head_greedy where the LM head launch takes it too)
static bool dynamic_decode_is_greedy_fused(const SamplingParams& p, bool finish)
{
    const int greedy_on = getenv("FTCF_GREEDY_FUSED") ? atoi(getenv("FTCF_GREEDY_FUSED")) : 1;  // (read per call: the tests switch the forms)
    return greedy_on && finish && p.max_top_k == 1 && !p.any_top_p && !p.apply_temperature && !p.apply_repetition && !p.optional_last_tokens
           && !p.row_len && p.V <= GREEDY_SLICES * 256 * GREEDY_MAXE && p.B <= 1024 && p.rot / 2 <= 256
           && (size_t)p.B * GREEDY_SLICES * 16 <= sampling_workspace_bytes(p.B, p.V);
}

bool dynamic_decode_is_fused(const SamplingParams& p, bool finish)
{
    return dynamic_decode_is_topk_fused(p, finish) || dynamic_decode_is_greedy_fused(p, finish);
}

void launch_dynamic_decode(const SamplingParams& p, hipStream_t s, bool finish)
{
    float* cand_v = reinterpret_cast<float*>(p.ws);
    int*   cand_i = reinterpret_cast<int*>(cand_v + (size_t)p.B * TOPK_BLOCKS * TOPK_MAX);
    // every row greedy, nothing but the min-length mask touches the logits, the step's bookkeeping follows: one launch
    if (dynamic_decode_is_topk_fused(p, finish)) {
        hipLaunchKernelGGL(k_topk_decode, dim3(TKD_SLICES, p.B), dim3(256), 0, s, p, cand_v);
        FTCF_HIP_CHECK(hipGetLastError());
        return;
    }
    if (dynamic_decode_is_greedy_fused(p, finish)) {
        hipLaunchKernelGGL(k_greedy_decode, dim3(GREEDY_SLICES, p.B), dim3(256), 0, s, p, cand_v);
        FTCF_HIP_CHECK(hipGetLastError());
        return;
    }
    size_t prep_smem = 0;
    if (p.optional_last_tokens) {
        prep_smem = std::max(prep_smem, (size_t)((p.V + 31) / 32) * 4);
    }
    if (p.apply_repetition) {
        prep_smem = std::max(prep_smem, (size_t)p.total_len * 8);
    }
    FTCF_CHECK_ARG(prep_smem <= 60 * 1024, "sequence too long for the repetition-penalty staging buffer");
    hipLaunchKernelGGL(k_decode_prep, dim3(p.B), dim3(1024), prep_smem, s, p);
    const int slice = (p.V + TOPK_BLOCKS - 1) / TOPK_BLOCKS;
    FTCF_CHECK_ARG(slice <= 256 * STAGE1_MAXE, "vocabulary too large: a stage-1 slice holds 256 x 60 logits in registers (V <= 122880)");
    hipLaunchKernelGGL(k_topk_stage1, dim3(TOPK_BLOCKS, p.B), dim3(256), 0, s, p, cand_v, cand_i, slice);
    // stage 2 sorts the union of the slices' candidate sets in LDS: the next power of two above 8 * k pairs
    int kc = std::max(1, std::min(p.max_top_k, TOPK_MAX));
    if (p.any_top_p) {
        kc = std::max(kc, TOPP_K);
    }
    size_t n2 = 1;
    while (n2 < (size_t)TOPK_BLOCKS * kc) {
        n2 <<= 1;
    }
    size_t k2h = 1;
    while (k2h < (size_t)kc) {
        k2h <<= 1;
    }
    const size_t sample_smem = std::max(n2 * 8 + k2h * 8, p.any_top_p ? (size_t)SORT_CH * 8 : (size_t)0);
    if (sample_smem > 48 * 1024) {  // per device: the attribute is per device (and cheap to set again)
        static std::mutex mu;
        static bool       done[64] = {};
        int               dev = 0;
        FTCF_HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev & 63]) {
            FTCF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sample), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)((size_t)TOPK_BLOCKS * TOPK_MAX * 8 + (size_t)TOPK_MAX * 8)));
            done[dev & 63] = true;
        }
    }
    hipLaunchKernelGGL(k_sample, dim3(p.B), dim3(256), sample_smem, s, p, cand_v, cand_i);
    if (finish) {  // (the continuous-batching front end keeps its own per-slot bookkeeping on the host)
        hipLaunchKernelGGL(k_decode_finish, dim3(1), dim3(64), 0, s, p);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// invokeDecodingInitialize (decoding_kernels.cu:26-65) + invokeMaskPaddingTokens (gpt_kernels.cu:1035-1082)
__global__ void k_decode_init(uint8_t* finished, int* seq_len, float* cum_log_probs, int* pad_count,
                              uint8_t* masked_tokens, uint64_t* draw_counter, const int* input_lengths, DecodeState* st,
                              int B, int max_input_len, int s_max, int beam_width)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        finished[b]      = 0;
        seq_len[b]       = max_input_len - 1;
        cum_log_probs[b] = (b % beam_width == 0) ? 0.f : -1e20f;  // beams > 0 start behind (decoding_kernels.cu:40-43)
        pad_count[b]     = 0;
        draw_counter[b]  = 0;
        if (b == 0) {
            st->step         = max_input_len;
            st->all_finished = 0;
            st->steps_done   = 0;
            st->pad          = 0;  // (the ticket of k_greedy_decode)
        }
    }
    const int len = input_lengths[b];
    for (int s = threadIdx.x; s < s_max; s += blockDim.x) {
        masked_tokens[(size_t)b * s_max + s] = (s >= len && s < max_input_len) ? 1 : 0;
    }
}

void launch_decode_init(uint8_t* finished, int* seq_len, float* cum_log_probs, int* pad_count, uint8_t* masked_tokens,
                        uint64_t* draw_counter, const int* input_lengths, DecodeState* st, int B, int max_input_len,
                        int s_max, hipStream_t s, int beam_width)
{
    hipLaunchKernelGGL(k_decode_init, dim3(B), dim3(256), 0, s, finished, seq_len, cum_log_probs, pad_count,
                       masked_tokens, draw_counter, input_lengths, st, B, max_input_len, s_max, beam_width);
    FTCF_HIP_CHECK(hipGetLastError());
}

// gatherTree with beam_width 1 and no prompts (decoding_kernels.cu:452-583) + the [time,batch] -> [batch,time] transpose
__global__ void k_gather_tree(int* output_ids, int* sequence_lengths, const int* step_ids, const int* seq_len,
                              const int* input_lengths, int B, int max_input_len, int total, int end_id)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) {
        return;
    }
    const int tmp_len = seq_len[b] + 1;  // max_sequence_length_final_step = 1
    sequence_lengths[b] = tmp_len;
    const int max_len = tmp_len;
    const int msl     = max_len < total ? max_len : total;
    int*      beams   = output_ids + (size_t)b * total;
    for (int t = 0; t < total; t++) {
        beams[t] = 0;
    }
    if (msl <= 0) {
        return;
    }
    const int in_len  = input_lengths[b];
    const int pad_off = max_input_len - in_len;
    beams[msl - 1 - pad_off] = step_ids[(size_t)(msl - 1) * B + b];
    for (int level = msl - 2; level >= 0; level--) {
        if (level >= in_len && level < max_input_len) {
            continue;
        }
        const int tgt = level >= max_input_len ? level - pad_off : level;
        beams[tgt]    = step_ids[(size_t)level * B + b];
    }
    for (int index = max_len - pad_off; index < total; index++) {
        beams[index] = end_id;
    }
    bool fin = false;
    for (int t = max_input_len; t < msl; t++) {
        if (fin) {
            beams[t] = end_id;
        }
        else if (beams[t] == end_id) {
            fin = true;
        }
    }
}

void launch_gather_tree(int* output_ids, int* sequence_lengths, const int* step_ids, const int* seq_len,
                        const int* input_lengths, int B, int max_input_len, int total, int end_id, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_tree, dim3(B), dim3(64), 0, s, output_ids, sequence_lengths, step_ids, seq_len,
                       input_lengths, B, max_input_len, total, end_id);
    FTCF_HIP_CHECK(hipGetLastError());
}


// ================================================================================================================
// beam search (beam_width K > 1): OnlineBeamSearchLayer without BeamHypotheses
//   layers/beam_search_layers/BaseBeamSearchLayer.cu:30-62,191-280, OnlineBeamSearchLayer.cu:25-166,
//   kernels/beam_search_penalty_kernels.cu:89-262, kernels/online_softmax_beamsearch_kernels.cu:100-365,650-701.
// Rows bb = batch * K + beam.  Three launches per token: rows (penalties + per-row top K of log-softmax + cum), batch
// (K best of the K*K candidates, state / parents / cache-indirection update, stop words), then k_decode_finish.
// ================================================================================================================
__global__ __launch_bounds__(1024) void k_beam_rows(const BeamParams p, float* cand_v, int* cand_i, int* snap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[64];
    __shared__ int   redi[64];
    __shared__ int   s_chosen[BEAM_MAX_K];
    __shared__ int   s_cnt;
    const int        bb = blockIdx.x, K = p.K, V = p.V;
    const int        b = bb / K, k = bb % K, BK = p.B * K;
    float*           l    = p.logits + (size_t)bb * V;
    const int        step = p.state->step;
    const int        tid = threadIdx.x, nt = blockDim.x;

    if (p.optional_last_tokens && step == p.max_input_len) {  // select_optional_last_tokens.cu:22-85
        uint32_t* bits  = reinterpret_cast<uint32_t*>(smem);
        const int words = (V + 31) / 32;
        for (int i = tid; i < words; i += nt) {
            bits[i] = 0u;
        }
        __syncthreads();
        for (int j = tid; j < p.optional_count; j += nt) {
            const int t = p.optional_last_tokens[(size_t)b * p.optional_count + j];
            if (t >= 0 && t < V) {
                atomicOr(&bits[t >> 5], 1u << (t & 31));
            }
        }
        __syncthreads();
        for (int i = tid; i < V; i += nt) {
            if (!((bits[i >> 5] >> (i & 31)) & 1u)) {
                l[i] = -INFINITY;
            }
        }
        __syncthreads();
    }
    const float temperature = p.temperature[b];
    if (temperature != 1.0f) {  // beam_search_penalty_kernels.cu:171-262
        const float inv = 1.0f / (temperature + 1e-6f);
        for (int i = tid; i < V; i += nt) {
            l[i] *= inv;
        }
        __syncthreads();
    }
    if (p.repetition_penalty && step > 0 && p.repetition_penalty[b] != 1.0f) {  // :89-153: history of THIS beam
        float*      newv = reinterpret_cast<float*>(smem);
        int*        idx  = reinterpret_cast<int*>(newv + p.total_len);
        const float pen  = p.repetition_penalty[b];
        if (tid == 0) {  // the walk along the parent chain is serial
            const int in_len = p.input_lengths[bb];
            int       cnt    = 0;
            idx[cnt++]       = p.output_ids[(size_t)(step - 1) * BK + bb];
            int parent       = k;
            for (int i = step - 2; i >= 0; i--) {
                if (i >= in_len && i < p.max_input_len) {
                    continue;
                }
                parent     = p.parent_ids[(size_t)i * BK + b * K + parent];
                idx[cnt++] = p.output_ids[(size_t)i * BK + b * K + parent];
            }
            s_cnt = cnt;
        }
        __syncthreads();
        const int cnt = s_cnt;
        for (int c = tid; c < cnt; c += nt) {
            const float lg = l[idx[c]];
            newv[c]        = lg > 0.0f ? lg / pen : lg * pen;
        }
        __syncthreads();
        for (int c = tid; c < cnt; c += nt) {
            l[idx[c]] = newv[c];
        }
        __syncthreads();
    }
    if (p.min_length && tid == 0) {  // :155-169
        if (step - p.max_input_len < p.min_length[b] && p.seq_len[bb] + 1 - p.max_input_len < p.min_length[b]) {
            l[p.end_id] = -FLT_MAX;
        }
    }
    __syncthreads();
    float* cv  = cand_v + (size_t)bb * K;
    int*   ci  = cand_i + (size_t)bb * K;
    const float cum = p.cum_log_probs[bb];
    if (tid == 0) {  // pre-update length / finished of every row for the batch kernel (its workgroups overwrite them)
        snap[bb]      = p.seq_len[bb];
        snap[BK + bb] = p.finished[bb];
    }
    if (p.finished[bb]) {  // a finished beam offers its end token at cum + 0 and nothing else (:296-365)
        for (int i = tid; i < K; i += nt) {
            const int j = (i == 0) ? p.end_id : (i - 1 < p.end_id ? i - 1 : i);
            cv[i]       = (i == 0) ? cum : -INFINITY;
            ci[i]       = j + bb * V;
        }
        return;
    }
    float mx = -FLT_MAX;
    for (int i = tid; i < V; i += nt) {
        mx = fmaxf(mx, l[i]);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) {
        red[tid >> 6] = mx;
    }
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < (nt >> 6); w++) {
        mx = fmaxf(mx, red[w]);
    }
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < V; i += nt) {
        sum += expf(l[i] - mx);
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) {
        red[tid >> 6] = sum;
    }
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (nt >> 6); w++) {
        tot += red[w];
    }
    __syncthreads();
    const float logd = logf(tot);
    // K rounds of block arg-best (ties: lower token id), skipping the tokens already taken
    for (int r = 0; r < K; r++) {
        VI best{-INFINITY, 0x7fffffff};
        for (int i = tid; i < V; i += nt) {
            const float v = l[i];
            if (better(v, i, best.v, best.i)) {
                bool taken = false;
                for (int c = 0; c < r; c++) {
                    taken |= (s_chosen[c] == i);
                }
                if (!taken) {
                    best.v = v;
                    best.i = i;
                }
            }
        }
        best = block_best(best, red, redi);
        if (tid == 0) {
            s_chosen[r] = best.i;
            cv[r]       = (best.v - mx - logd) + cum;
            ci[r]       = best.i + bb * V;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_beam_batch(const BeamParams p, const float* cand_v, const int* cand_i,
                                                    const int* snap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[8];
    __shared__ int   redi[8];
    __shared__ int   s_parent[BEAM_MAX_K], s_word[BEAM_MAX_K], s_seq[BEAM_MAX_K], s_fin[BEAM_MAX_K];
    __shared__ float s_cum[BEAM_MAX_K];
    const int        b = blockIdx.x, K = p.K, V = p.V, BK = p.B * K, KK = K * K;
    const int        tid = threadIdx.x, nt = blockDim.x;
    const int        step = p.state->step;
    float*           sv   = reinterpret_cast<float*>(smem);  // [K*K] penalised scores
    uint8_t*         tk   = reinterpret_cast<uint8_t*>(sv + KK);  // [K*K] taken
    const float*     cy   = cand_v + (size_t)b * KK;
    const int*       cx   = cand_i + (size_t)b * KK;
    const float      len_pen = p.len_penalty[b], diversity = p.diversity_rate[b];
    // batch_topk_kernel (online_softmax_beamsearch_kernels.cu:100-262, no beam hypotheses).  NB it indexes finished /
    // sequence_lengths by the BATCH id, restated as written (only matters with len_penalty != 0).
    const int* old_seq = snap;
    const int* old_fin = snap + BK;
    const int  length  = old_fin[b] ? old_seq[b] : old_seq[b] + 1;
    for (int e = tid; e < KK; e += nt) {
        float v = cy[e];
        if (len_pen != 0.0f && length != 1) {
            v = v / powf((float)length, len_pen);
        }
        v += diversity * (float)(e % K);
        sv[e] = v;
        tk[e] = 0;
    }
    __syncthreads();
    for (int r = 0; r < K; r++) {
        VI best{-INFINITY, 0x7fffffff};
        for (int e = tid; e < KK; e += nt) {
            if (!tk[e] && better(sv[e], e, best.v, best.i)) {
                best.v = sv[e];
                best.i = e;
            }
        }
        best = block_best(best, red, redi);
        if (tid == 0) {
            const int e  = best.i;
            tk[e]        = 1;
            const int z  = cx[e];
            const int pk = (z / V) % K;
            s_parent[r]  = pk;
            s_word[r]    = z % V;
            s_cum[r]     = cy[e];
            // update_kernel (OnlineBeamSearchLayer.cu:25-58): lengths follow the parent beam
            const int pb = b * K + pk;
            s_seq[r]     = old_fin[pb] ? old_seq[pb] : old_seq[pb] + 1;
            s_fin[r]     = (z % V) == p.end_id;
        }
        __syncthreads();
    }
    if (tid < K) {
        const int bb                         = b * K + tid;
        p.seq_len[bb]                        = s_seq[tid];
        p.finished[bb]                       = (uint8_t)s_fin[tid];
        p.parent_ids[(size_t)step * BK + bb] = s_parent[tid];
        p.output_ids[(size_t)step * BK + bb] = s_word[tid];
        p.cum_log_probs[bb]                  = s_cum[tid];
    }
    // update_indir_cache_kernel (BaseBeamSearchLayer.cu:30-62): rows that just finished keep their stale entries
    const size_t plane = (size_t)BK * p.s_max;
    const int*   src   = p.cache_indir + (size_t)((step - p.max_input_len) & 1) * plane;
    int*         tgt   = p.cache_indir + (size_t)(1 - ((step - p.max_input_len) & 1)) * plane;
    const int    nts   = step + 1 < p.s_max ? step + 1 : p.s_max;
    for (int i = tid; i < K * nts; i += nt) {
        const int kk = i / nts, t = i % nts;
        if (!s_fin[kk]) {
            tgt[((size_t)b * K + kk) * p.s_max + t] = (t == step) ? kk : src[((size_t)b * K + s_parent[kk]) * p.s_max + t];
        }
    }
    __threadfence_block();
    __syncthreads();
    if (p.stop_words && tid < K) {  // stop_criteria_kernels.cu:24-83 along the parent chain
        const int  bb    = b * K + tid;
        const int* words = p.stop_words + (size_t)b * 2 * p.stop_len;
        const int* offs  = words + p.stop_len;
        for (int id = 0; id < p.stop_len; id++) {
            if (offs[id] < 0) {
                continue;
            }
            const int item_end = offs[id], item_start = id > 0 ? offs[id - 1] : 0, item_size = item_end - item_start;
            bool      stop = false;
            if (step + 1 >= item_size) {
                stop       = true;
                int parent = tid;
                for (int t = item_size - 1; t >= 0; t--) {
                    const int ts  = step - (item_size - 1) + t;
                    const int tok = p.output_ids[(size_t)ts * BK + b * K + parent];
                    if (tok != words[item_start + t]) {
                        stop = false;
                        break;
                    }
                    parent = p.parent_ids[(size_t)ts * BK + b * K + parent];
                }
            }
            if (stop) {
                p.finished[bb] = 1;
            }
        }
    }
}

size_t beam_workspace_bytes(int B, int K)
{
    return (size_t)B * K * K * (sizeof(float) + sizeof(int)) + (size_t)B * K * 2 * sizeof(int);
}

void launch_beam_search(const BeamParams& p, hipStream_t s)
{
    FTCF_CHECK_ARG(p.K >= 2 && p.K <= BEAM_MAX_K, "beam_width must be in [2, 64]");
    float* cand_v = reinterpret_cast<float*>(p.ws);
    int*   cand_i = reinterpret_cast<int*>(cand_v + (size_t)p.B * p.K * p.K);
    int*   snap   = cand_i + (size_t)p.B * p.K * p.K;
    size_t smem   = 0;
    if (p.optional_last_tokens) {
        smem = std::max(smem, (size_t)((p.V + 31) / 32) * 4);
    }
    if (p.repetition_penalty) {
        smem = std::max(smem, (size_t)p.total_len * 8);
    }
    FTCF_CHECK_ARG(smem <= 60 * 1024, "sequence too long for the repetition-penalty staging buffer");
    hipLaunchKernelGGL(k_beam_rows, dim3(p.B * p.K), dim3(1024), smem, s, p, cand_v, cand_i, snap);
    hipLaunchKernelGGL(k_beam_batch, dim3(p.B), dim3(256), (size_t)p.K * p.K * 5, s, p, cand_v, cand_i, snap);
    FTCF_HIP_CHECK(hipGetLastError());
}

// invokeTileGptInputs (gpt_kernels.cu:632-667)
__global__ void k_tile_inputs(int* tiled_ids, int* tiled_len, const int* ids, const int* len, int K, int S)
{
    const int bb = blockIdx.x, b = bb / K;
    if (threadIdx.x == 0) {
        tiled_len[bb] = len[b];
    }
    for (int s2 = threadIdx.x; s2 < S; s2 += blockDim.x) {
        tiled_ids[(size_t)bb * S + s2] = ids[(size_t)b * S + s2];
    }
}

void launch_tile_inputs(int* tiled_ids, int* tiled_len, const int* ids, const int* len, int B, int K, int S, hipStream_t s)
{
    hipLaunchKernelGGL(k_tile_inputs, dim3(B * K), dim3(256), 0, s, tiled_ids, tiled_len, ids, len, K, S);
    FTCF_HIP_CHECK(hipGetLastError());
}

// gatherTree with parents (decoding_kernels.cu:452-583) + the [time, batch*beam] -> [batch, beam, time] transpose
__global__ void k_gather_tree_beam(int* output_ids, int* sequence_lengths, const int* step_ids, const int* parent_ids,
                                   const int* seq_len, const int* input_lengths, int B, int K, int max_input_len,
                                   int total, int end_id)
{
    const int bb = blockIdx.x, b = bb / K, BK = B * K;
    if (threadIdx.x != 0) {
        return;
    }
    int max_len = -1;
    for (int j = 0; j < K; j++) {
        const int tmp_len = seq_len[b * K + j] + 1;  // max_sequence_length_final_step = 1
        max_len           = tmp_len > max_len ? tmp_len : max_len;
    }
    sequence_lengths[bb] = seq_len[bb] + 1;
    const int msl        = max_len < total ? max_len : total;
    int*      beams      = output_ids + (size_t)bb * total;
    for (int t = 0; t < total; t++) {
        beams[t] = 0;
    }
    if (msl <= 0) {
        return;
    }
    const int in_len  = input_lengths[bb];
    const int pad_off = max_input_len - in_len;
    beams[msl - 1 - pad_off] = step_ids[(size_t)(msl - 1) * BK + bb];
    int parent               = parent_ids[(size_t)(msl - 1) * BK + bb] % K;
    for (int level = msl - 2; level >= 0; level--) {
        if (level >= in_len && level < max_input_len) {
            continue;
        }
        const int tgt = level >= max_input_len ? level - pad_off : level;
        beams[tgt]    = step_ids[(size_t)level * BK + b * K + parent];
        parent        = parent_ids[(size_t)level * BK + b * K + parent] % K;
    }
    for (int index = max_len - pad_off; index < total; index++) {
        beams[index] = end_id;
    }
    bool fin = false;
    for (int t = max_input_len; t < msl; t++) {
        if (fin) {
            beams[t] = end_id;
        }
        else if (beams[t] == end_id) {
            fin = true;
        }
    }
}

void launch_gather_tree_beam(int* output_ids, int* sequence_lengths, const int* step_ids, const int* parent_ids,
                             const int* seq_len, const int* input_lengths, int B, int K, int max_input_len, int total,
                             int end_id, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_tree_beam, dim3(B * K), dim3(64), 0, s, output_ids, sequence_lengths, step_ids, parent_ids,
                       seq_len, input_lengths, B, K, max_input_len, total, end_id);
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
