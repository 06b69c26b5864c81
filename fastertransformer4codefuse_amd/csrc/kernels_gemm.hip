// MFMA GEMM over the engine's tiled weights: the prefill / batched (m > 4) counterpart of the GEMV kernels.
//
// Replaces CutlassFpAIntBGemmRunner<half,uint8_t>::gemm / gemm_bias_act
// (kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:45-197, mainloop
// cutlass_extensions/.../gemm/threadblock/dq_mma_multistage.h) and the cuBLAS fp16 GEMMs
// (utils/cublasMMWrapper.cc:94-386) for GptContextAttentionLayer.cc:101-141,350-393 and FfnLayer.cc:172-372.
//
// Shape: weights are the B operand of v_mfma_f32_16x16x32_f16 straight from global memory -- a weight tile IS one
// wave-load and one lane's 16 bytes ARE that lane's B fragment(s) (int8: two fragments after the in-register
// dequant, k order (lane>>4)*16 + c*8 + j; fp16: one fragment, k order (lane>>4)*8 + j).  Each wave owns 16 output
// columns and keeps its dequantised B fragments in registers across all row groups of the block tile; activations
// are staged through LDS and shared by the waves of the workgroup.  Roofline: MFMA for m >= 512, HBM below.
//
// Forms of k_gemm_tiled, by row count (launch_gemm_tiled): 17..384 (int8) or 768 (fp16) rows with a workspace -- 32 / 48 / 64 / 128-row tiles cut along K
// into ~256 workgroups, a ring of four k-steps of weight tiles and A-tile registers, the slices reduced inside the launch
// (SPLITK, D = 4); more rows -- 64- or 128-row tiles, ring of two k-steps (D = 2) when the k-steps divide, with the A
// fragments prefetched (PF) where one workgroup per CU runs anyway and held to 128 VGPRs (OCC2) otherwise; the plain loop
// (D = 0) for odd k-step counts and without a workspace.  5..16 rows: k_gemm_smallm_burst below.
#include "ftcf_common.h"
#include "kernels.h"

namespace ftcf {

constexpr int GEMM_KSTEP = 64;
// split-K form of the tiled GEMM (short prompt phases): partial tiles [tile][slice] of 64 x 256 fp32 + one ticket per tile
constexpr size_t GEMM_SPLITK_WS    = (size_t)48 << 20;
constexpr size_t GEMM_SPLITK_TILES = 4096;
// rows up to which the split-K form runs, and rows above which its tiles are 128 rows high instead of 64 (a weight fragment is then
// dequantised / read from LDS for twice the MFMAs and K is cut twice as often for the same number of workgroups).  13B prompt phase,
// ms, 64-row split-K up to 320 rows and the un-split 128-row form above (round 4) -> this (profiles/r05_prefill_sweep.txt):
//   fp16 weights  224: 15.2 -> 12.2   256: 15.7 -> 12.5   320: 17.9 -> 13.9   384: 18.7 -> 14.7   512: 22.4 -> 19.9   640: 25.7 -> 23.4   768: 29.8 -> 28.6
//   int8 weights  352: 14.8 -> 13.0   384: 15.0 -> 13.2   (128-row tiles at <= 320 rows: 160: 8.4 -> 10.3, 288: 12.2 -> 12.7; at 448+: no gain)
constexpr int    GEMM_SPLITK_MAX_M_I8 = 384, GEMM_SPLITK_MAX_M_F16 = 768;
constexpr int    GEMM_SK128_MIN_M_I8 = 320, GEMM_SK128_MIN_M_F16 = 192;

// one 16-byte weight fragment against the 16 rows of x in LDS (xr: this lane's row lane&15, k group lane>>4)
template<bool INT8>
__device__ __forceinline__ void consume_tile_raw(const u32x4 w, const f16* xr, const f16x2 scale2, f32x4& acc)
{
    if constexpr (INT8) {
        f16x2 d[8];
        dequant4(w.x, scale2, d[0], d[1]);
        dequant4(w.y, scale2, d[2], d[3]);
        dequant4(w.z, scale2, d[4], d[5]);
        dequant4(w.w, scale2, d[6], d[7]);
        const f16x8 b0 = {d[0][0], d[0][1], d[1][0], d[1][1], d[2][0], d[2][1], d[3][0], d[3][1]};
        const f16x8 b1 = {d[4][0], d[4][1], d[5][0], d[5][1], d[6][0], d[6][1], d[7][0], d[7][1]};
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(xr);
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(xr + 8);
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc, 0, 0, 0);
    }
    else {
        const f16x8 b0 = __builtin_bit_cast(f16x8, w);
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(xr);
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
    }
}
// A tile row in LDS: 64 k + 32 B pad, and the eight 16-byte k chunks of a row stored in the order the MFMA A fragments are
// read -- chunk position p holds k chunk (p & 3) * 2 + (p >> 2) for int8 weights (the B tile interleaves k in 16s), p for
// fp16 -- so that lane (c, g) reads positions g and 4 + g: with this row stride the 16 lanes of every ds_read_b128 phase
// fall on 64 distinct banks (a plain [k] order with 16 B pad was 2-way conflicted: SQ_LDS_BANK_CONFLICT 44 %).
constexpr int GEMM_LDA   = GEMM_KSTEP + 16;

// RG row groups (16 rows each) x NG column groups (16 columns each) per wave: an A fragment read from LDS feeds NG MFMAs
// and a dequantised B fragment RG of them.  NG = 1 reads 1 KiB of LDS per MFMA and is LDS-bandwidth bound (22 % of the
// MFMA peak at m = 1024); NG = 2 halves that.  Block tile: (RG*16) x (WAVES * NG * 16); the waves of a workgroup share the A
// tile in LDS, so WAVES = 8 halves the number of workgroups re-reading A from L2 (prefill is L2-traffic bound: at m = 1024
// a 128 x 128 tile moves 1.2 GB of A and 0.6 GB of weights through the L2 for the QKV GEMM).
//
// SPLITK (short prompt phases, 17..320 rows: HBM bound work on too few tiles to fill 256 CUs -- n = 5120 at 64 < m <= 128 is
// 40 workgroups): the K extent of a block tile is cut into KS slices, one workgroup each (all on the tile's XCD).  Every
// workgroup stores its fp32 accumulators in MFMA fragment order (16 B per lane, one coalesced wave store per fragment) and takes
// a ticket; the one that takes the tile's last ticket adds the KS partial tiles IN SLICE ORDER (deterministic) and applies the
// epilogue.  The partials are write-through (sc1) stores whose acknowledgement every wave awaits (s_waitcnt vmcnt(0)) before the
// workgroup takes its ticket, and the last workgroup reads them back with sc1 loads (the XCDs' L2s are not coherent with each
// other): no agent-scope fence, i.e. no whole-L2 write-back / invalidate per workgroup; it re-arms the ticket.
// PF: all (of two row groups') A fragments of a k-step are read from LDS before the dequantisation; OCC2: register budget of two
// workgroups per CU (128 VGPRs at 8 waves)
template<bool INT8, int RG, int NG, int WAVES, bool NT_W = true, bool XCD = false, bool SPLITK = false, int D = 0, bool PF = false,
         bool OCC2 = false>
__global__ __launch_bounds__(64 * WAVES, OCC2 ? WAVES / 2 : 1) void k_gemm_tiled(const f16* __restrict__ A, const void* __restrict__ W,
                                                    const f16* __restrict__ scale, const f16* __restrict__ bias,
                                                    int act, f16* __restrict__ C, int m, int n, int k, int gx = 0,
                                                    int gy = 0, int KS = 1, int ks_per = 0, f32x4* __restrict__ ws = nullptr,
                                                    unsigned* __restrict__ tickets = nullptr)
{
    constexpr int BM   = RG * 16;
    constexpr int NTHR = 64 * WAVES;
    __shared__ __attribute__((aligned(16))) f16 As[2][BM * GEMM_LDA];  // double buffered: one barrier per k-step

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    int bx = blockIdx.x, by = blockIdx.y, kz = 0;
    if constexpr (XCD) {
        // 1-D grid; workgroup b runs on XCD b % 8.  The gy row blocks of one weight panel go to ONE XCD, in consecutive
        // slots (they run together and share the panel through that XCD's L2 instead of fetching it 8 times over the fabric)
        const int b = blockIdx.x, c = b & 7, sl = b >> 3;
        if constexpr (SPLITK) {
            const int rem = sl % (gy * KS);
            bx            = (sl / (gy * KS)) * 8 + c;
            by            = rem % gy;
            kz            = rem / gy;
        }
        else {
            bx = (sl / gy) * 8 + c;
            by = sl % gy;
        }
        if (bx >= gx) {
            return;
        }
    }
    const int m0 = by * BM;
    const int nt0 = (bx * WAVES + wid) * NG;  // first column group of this wave
    const int NT = n / 16;
    const int  ks0    = SPLITK ? kz * ks_per : 0;  // first k-step of this workgroup's slice
    const int  ksteps = SPLITK ? min(ks_per, k / GEMM_KSTEP - ks0) : k / GEMM_KSTEP;

    // B stream pointers (column groups past the end re-read the last one; their results are dropped)
    const int    KT = INT8 ? k / TILE_K_I8 : k / TILE_K_F16;
    const u32x4* wp[NG];
    f16x2        scale2[NG];
#pragma unroll
    for (int j = 0; j < NG; j++) {
        const int nt = nt0 + j < NT ? nt0 + j : NT - 1;
        wp[j]        = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(W) + ((size_t)nt * KT * 64 + lane) * 16);
        scale2[j]    = f16x2{(f16)1.f, (f16)1.f};
        if constexpr (INT8) {
            const f16 sc = scale[nt * 16 + c];
            scale2[j]    = f16x2{sc, sc};
        }
    }

    // A staging: BM x 64 halves, 8 halves (16 B) per thread-chunk
    constexpr int CHUNKS = BM * GEMM_KSTEP / 8;             // 16-byte chunks per stage
    constexpr int CPT    = (CHUNKS + NTHR - 1) / NTHR;            // chunks per thread
    u32x4         areg[CPT];
    auto load_a_into = [&](u32x4 (&areg)[CPT], int ks) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int ch = threadIdx.x + i * NTHR;
            if (CHUNKS % NTHR == 0 || ch < CHUNKS) {  // (a branch here costs hipcc its count of the loads in flight)
                int row = m0 + ch / 8;
                row     = row < m ? row : m - 1;
                areg[i] = *reinterpret_cast<const u32x4*>(A + (size_t)row * k + (size_t)(ks0 + ks) * GEMM_KSTEP + (ch % 8) * 8);
            }
        }
    };
    auto load_a = [&](int ks) { load_a_into(areg, ks); };
    auto store_a_from = [&](const u32x4 (&areg)[CPT], int buf) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int ch = threadIdx.x + i * NTHR;
            if (CHUNKS % NTHR == 0 || ch < CHUNKS) {  // (a branch here costs hipcc its count of the loads in flight)
                const int kc = ch % 8;                                             // k chunk of this piece
                const int pc = INT8 ? (((kc & 1) << 2) | (kc >> 1)) : kc;          // its position in the row
                *reinterpret_cast<u32x4*>(&As[buf][(ch / 8) * GEMM_LDA + pc * 8]) = areg[i];
            }
        }
    };
    auto store_a = [&](int buf) { store_a_from(areg, buf); };

    f32x4 acc[RG][NG];
#pragma unroll
    for (int r = 0; r < RG; r++) {
#pragma unroll
        for (int j = 0; j < NG; j++) {
            acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // weight tiles: two k-steps in flight per wave (the first reader of a tile misses to HBM: ~900+ cycles, more than one
    // k-step lasts) -- a third one costs the fourth wave per SIMD (134 vs 124 VGPRs), measured slower; requests are clamped,
    // never conditional
    typedef u32x4 BTile[NG][2];  // int8: [0] only ; fp16: two 32-k tiles per 64-k step
    BTile         B0, B1;
    auto load_b = [&](BTile& br, int ks) {
        ks = ks0 + (ks < ksteps ? ks : ksteps - 1);
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if constexpr (INT8) {
                br[j][0] = NT_W ? __builtin_nontemporal_load(wp[j] + (size_t)ks * 64) : wp[j][(size_t)ks * 64];
            }
            else {
                br[j][0] = NT_W ? __builtin_nontemporal_load(wp[j] + (size_t)(2 * ks) * 64) : wp[j][(size_t)(2 * ks) * 64];
                br[j][1] = NT_W ? __builtin_nontemporal_load(wp[j] + (size_t)(2 * ks + 1) * 64) : wp[j][(size_t)(2 * ks + 1) * 64];
            }
        }
    };
    auto dequant_b = [&](const BTile& br, f16x8 (&bf)[NG][2]) {
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if constexpr (INT8) {
                f16x2 d[8];
                dequant4(br[j][0].x, scale2[j], d[0], d[1]);
                dequant4(br[j][0].y, scale2[j], d[2], d[3]);
                dequant4(br[j][0].z, scale2[j], d[4], d[5]);
                dequant4(br[j][0].w, scale2[j], d[6], d[7]);
                bf[j][0] = f16x8{d[0][0], d[0][1], d[1][0], d[1][1], d[2][0], d[2][1], d[3][0], d[3][1]};
                bf[j][1] = f16x8{d[4][0], d[4][1], d[5][0], d[5][1], d[6][0], d[6][1], d[7][0], d[7][1]};
            }
            else {
                bf[j][0] = __builtin_bit_cast(f16x8, br[j][0]);
                bf[j][1] = __builtin_bit_cast(f16x8, br[j][1]);
            }
        }
    };
    auto compute = [&](const BTile& br, int ks) {
        // A fragment k offsets must follow the B fragment's k order (see file header)
        const int  koff0 = g * 8;
        const int  koff1 = 32 + g * 8;
        const f16* as    = As[ks & 1];
        if constexpr (PF) {
            // all A fragments of the step requested up front (their LDS latency passes under the dequantisation), then the MFMAs
            // back to back with the two k halves of an accumulator RG * NG instructions apart (a dependent MFMA stalls its wave)
            constexpr int PR = RG % 2 == 0 ? 2 : 1;  // row groups whose fragments are in flight together
            f16x8         af[PR][2];
#pragma unroll
            for (int r = 0; r < PR; r++) {
                af[r][0] = *reinterpret_cast<const f16x8*>(&as[(r * 16 + c) * GEMM_LDA + koff0]);
                af[r][1] = *reinterpret_cast<const f16x8*>(&as[(r * 16 + c) * GEMM_LDA + koff1]);
            }
            f16x8 bf[NG][2];
            dequant_b(br, bf);
#pragma unroll
            for (int r0 = 0; r0 < RG; r0 += PR) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
#pragma unroll
                    for (int r = 0; r < PR; r++) {
#pragma unroll
                        for (int j = 0; j < NG; j++) {
                            acc[r0 + r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[r][h], bf[j][h], acc[r0 + r][j], 0, 0, 0);
                        }
                    }
                }
                if (r0 + PR < RG) {
#pragma unroll
                    for (int r = 0; r < PR; r++) {
                        af[r][0] = *reinterpret_cast<const f16x8*>(&as[((r0 + PR + r) * 16 + c) * GEMM_LDA + koff0]);
                        af[r][1] = *reinterpret_cast<const f16x8*>(&as[((r0 + PR + r) * 16 + c) * GEMM_LDA + koff1]);
                    }
                }
            }
        }
        else {
            f16x8 bf[NG][2];
            dequant_b(br, bf);
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(&as[(r * 16 + c) * GEMM_LDA + koff0]);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(&as[(r * 16 + c) * GEMM_LDA + koff1]);
#pragma unroll
                for (int j = 0; j < NG; j++) {
                    acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bf[j][0], acc[r][j], 0, 0, 0);
                    acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bf[j][1], acc[r][j], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (D > 0) {
        // deep form (short prompt phases: one or two workgroups per CU, each k-step a dependent round trip otherwise): weight
        // tiles AND the A tile's register stage D k-steps ahead; stage s lives in ring slot s % D (static: the loop is unrolled D
        // times), the LDS double buffer as in the plain form
        static_assert(D % 2 == 0, "the LDS buffer of a stage is its ring slot's parity");
        BTile Bq[D];
        u32x4 aq[D][CPT];
#pragma unroll
        for (int d = 0; d < D; d++) {
            load_a_into(aq[d], d < ksteps ? d : ksteps - 1);
            load_b(Bq[d], d);
        }
        store_a_from(aq[0], 0);
        load_a_into(aq[0], D < ksteps ? D : ksteps - 1);
        __syncthreads();
        for (int ks = 0; ks < ksteps; ks += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {  // ksteps % D == 0 (the launcher sees to it): no step is conditional -- a branch
                                           // around a step makes hipcc wait for ALL outstanding loads at the loop head
                compute(Bq[d], d);
                store_a_from(aq[(d + 1) % D], (d + 1) & 1);  // stage ks + d + 1
                load_a_into(aq[(d + 1) % D], ks + d + 1 + D < ksteps ? ks + d + 1 + D : ksteps - 1);
                __syncthreads();
                load_b(Bq[d], ks + d + D);
                // (hipcc's scheduler otherwise sinks the requests towards their uses, D steps later: the ring drains)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    else {
        // at entry As[ks & 1] holds stage ks (its barrier has been passed) and areg stage ks + 1
        auto step = [&](const BTile& br, int ks) {
            compute(br, ks);
            store_a((ks + 1) & 1);  // the other buffer: its readers (stage ks - 1) finished before the previous barrier
            load_a(ks + 2 < ksteps ? ks + 2 : ksteps - 1);
            __syncthreads();
        };
        load_a(0);
        load_b(B0, 0);
        load_b(B1, 1);
        store_a(0);
        load_a(1 < ksteps ? 1 : 0);
        __syncthreads();
        for (int ks = 0; ks < ksteps; ks += 2) {
            step(B0, ks);
            load_b(B0, ks + 2);
            if (ks + 1 < ksteps) {
                step(B1, ks + 1);
            }
            load_b(B1, ks + 3);
        }
    }
    if constexpr (SPLITK) {
        if (KS > 1) {
            constexpr int FR   = RG * NG * 64;  // f32x4 per wave
            const int     tile = bx * gy + by;
            // partial tiles travel as write-through (sc1) 8-byte stores and are read back with sc1 loads: no whole-L2 write-back /
            // invalidate per workgroup (an agent-scope fence costs that: 30 ms for the 13B prompt phase at 128 tokens against 16)
            typedef unsigned long long u64;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            u64* P = reinterpret_cast<u64*>(ws + ((size_t)(tile * KS + kz) * WAVES + wid) * FR) + lane;
#pragma unroll
            for (int r = 0; r < RG; r++) {
#pragma unroll
                for (int j = 0; j < NG; j++) {
                    const f32x4 a = acc[r][j];
                    __hip_atomic_store(P + ((r * NG + j) * 2 + 0) * 64, __builtin_bit_cast(u64, f32x2{a[0], a[1]}), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(P + ((r * NG + j) * 2 + 1) * 64, __builtin_bit_cast(u64, f32x2{a[2], a[3]}), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __shared__ int last;
            // Every wave waits for the acknowledgement of ITS OWN write-through stores (vmcnt counts stores on gfx950) before the
            // barrier: a workgroup-scope release alone compiles to no wait at all, and wave 0's ticket could then reach the L2
            // ahead of the other waves' partials.  No L2 write-back is needed: the stores are sc1 (written through) already.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned t = __hip_atomic_fetch_add(&tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last             = t == (unsigned)(KS - 1);
                if (last) {
                    __hip_atomic_store(&tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
                }
            }
            __syncthreads();
            if (!last) {
                return;
            }
            // slice by slice, all of a slice's fragments requested together (one round trip per slice, not per fragment)
            const u64* Q = reinterpret_cast<const u64*>(ws + ((size_t)(tile * KS) * WAVES + wid) * FR) + lane;
#pragma unroll
            for (int r = 0; r < RG; r++) {
#pragma unroll
                for (int j = 0; j < NG; j++) {
                    acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            constexpr int ZB = RG <= 4 ? 1 : 2;  // slices requested together (32 / 64 VGPRs each)
            for (int z0 = 0; z0 < KS; z0 += ZB) {
                u64 v[ZB][RG][NG][2];
#pragma unroll
                for (int i = 0; i < ZB; i++) {
                    if (z0 + i < KS) {
#pragma unroll
                        for (int r = 0; r < RG; r++) {
#pragma unroll
                            for (int j = 0; j < NG; j++) {
                                const u64* q  = Q + (size_t)(z0 + i) * WAVES * FR * 2 + (r * NG + j) * 2 * 64;
                                v[i][r][j][0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                v[i][r][j][1] = __hip_atomic_load(q + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < ZB; i++) {
                    if (z0 + i < KS) {
#pragma unroll
                        for (int r = 0; r < RG; r++) {
#pragma unroll
                            for (int j = 0; j < NG; j++) {
                                const f32x2 lo = __builtin_bit_cast(f32x2, v[i][r][j][0]);
                                const f32x2 hi = __builtin_bit_cast(f32x2, v[i][r][j][1]);
                                acc[r][j] += f32x4{lo[0], lo[1], hi[0], hi[1]};
                            }
                        }
                    }
                }
            }
        }
    }
    // C/D layout of mfma 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int j = 0; j < NG; j++) {
        if (nt0 + j >= NT) {
            continue;
        }
        const int   col = (nt0 + j) * 16 + c;
        const float bv  = bias ? (float)bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < RG; r++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = m0 + r * 16 + g * 4 + q;
                if (row < m) {
                    float v = acc[r][j][q];
                    f16   h;
                    if constexpr (INT8) {  // fused fp32 epilogue (epilogue_helpers.h:52-62)
                        v += bv;
                        if (act == 1) {
                            v = gelu_f32(v);
                        }
                        h = (f16)v;
                    }
                    else {  // cuBLAS rounds to half; bias/gelu follow in half (activation_kernels.cu:401-426)
                        h = (f16)v;
                        if (act == 1) {
                            h = gelu_f16(bias ? (f16)(h + bias[col]) : h);
                        }
                        else if (bias) {
                            h = h + bias[col];
                        }
                    }
                    C[(size_t)row * n + col] = h;
                }
            }
        }
    }
}

size_t gemm_tiled_ticket_bytes()
{
    return GEMM_SPLITK_TILES * sizeof(unsigned);
}
size_t gemm_tiled_workspace_bytes()
{
    return GEMM_SPLITK_WS + GEMM_SPLITK_TILES * sizeof(unsigned);
}

int gemm_tiled_splitk_max_m()
{
    static const int v = getenv("FTCF_GEMM_SPLITK_MAX_M") ? atoi(getenv("FTCF_GEMM_SPLITK_MAX_M"))
                                                            : std::max(GEMM_SPLITK_MAX_M_I8, GEMM_SPLITK_MAX_M_F16);
    return v;  // (the larger of the two weight forms' limits: who sizes a workspace by it covers both)
}

void launch_gemm_tiled(const f16* A, const void* W, const f16* scale, const f16* bias, int act, f16* C, int m, int n,
                       int k, bool int8, hipStream_t s, float* workspace)
{
    if (m == 0) {
        return;
    }
    FTCF_CHECK_ARG(k % GEMM_KSTEP == 0, "GEMM needs k % 64 == 0 (as the reference: fpA_intB_gemm_template.h:159-163)");
    FTCF_CHECK_ARG(n % 16 == 0, "GEMM needs n % 16 == 0");
    const int NT = n / 16;
    const char*      sk_env        = getenv("FTCF_GEMM_SPLITK");  // (read per call: the tests switch it inside one process)
    const int        splitk_target = sk_env ? atoi(sk_env) : 256;
    static const int splitk_max_env = getenv("FTCF_GEMM_SPLITK_MAX_M") ? atoi(getenv("FTCF_GEMM_SPLITK_MAX_M")) : -1;
    const int splitk_max_m = splitk_max_env >= 0 ? splitk_max_env : (int8 ? GEMM_SPLITK_MAX_M_I8 : GEMM_SPLITK_MAX_M_F16);
    if (workspace && m <= splitk_max_m && splitk_target > 0) {
        // 64-row tiles cut along K until ~1 workgroup per CU is in flight (FTCF_GEMM_SPLITK: the target number of workgroups)
        constexpr int deep = 4;  // ring depth of the split-K form (0 / 2: measured slower, profiles/r03_notes.md)
        // 64-row tiles; one row block (m <= 64): the lowest tile that covers it, 32 / 48 / 64 rows -- a 64-row tile on 17..32 rows
        // spends half its MFMAs and LDS reads on clamped duplicate rows (13B int8 prompt phase: 17 tokens 5.2 -> 4.4 ms, 33: 5.3 ->
        // 5.0).  With several row blocks lower tiles measured no better (a k-step's fixed costs -- barrier, A staging,
        // dequantisation -- outweigh its MFMAs at these heights).
        constexpr int rgsel = 1;
        constexpr int NG = 2, DQ = 4;
        // several row blocks: 128-row tiles above FTCF_GEMM_SK128_MIN_M rows (see GEMM_SK128_MIN_M_* for the measurements)
        const char*   sk128_env = getenv("FTCF_GEMM_SK128_MIN_M");  // rows above which the tiles are 128 rows high
        const int     sk128_min = sk128_env ? atoi(sk128_env) : (int8 ? GEMM_SK128_MIN_M_I8 : GEMM_SK128_MIN_M_F16);
        const int     RGs = (rgsel && m <= 64) ? std::max(2, (m + 15) / 16) : (m > sk128_min ? 8 : 4);
        const int BM = RGs * 16;
        const int gx = (NT + 8 * NG - 1) / (8 * NG), gy = (m + BM - 1) / BM, gx8 = 8 * ((gx + 7) / 8);
        const int ksteps = k / GEMM_KSTEP;
        int       KS     = std::max(1, std::min({8, splitk_target / (gx8 * gy), ksteps / 8}));
        KS               = std::max<long>(1, std::min<long>(KS, (long)(GEMM_SPLITK_WS / (size_t)(BM * 256 * 4)) / ((long)gx * gy)));
        if (gx * gy <= (int)GEMM_SPLITK_TILES) {
            // the deep form runs whole rounds of its ring: slices of a multiple of DQ k-steps (else the plain form)
            const bool dd  = deep > 0 && ksteps % DQ == 0;
            const int  rnd = dd ? DQ : 1;
            const int  per = ((ksteps + KS - 1) / KS + rnd - 1) / rnd * rnd;
            KS             = (ksteps + per - 1) / per;  // no empty slice
            f32x4*    ws   = reinterpret_cast<f32x4*>(workspace);
            unsigned* tk   = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + GEMM_SPLITK_WS);
            dim3      grid(gx8 * gy * KS);
#define FTCF_SK(I8, Dv, RGv)                                                                                                     \
    hipLaunchKernelGGL((k_gemm_tiled<I8, RGv, NG, 8, false, true, true, Dv, (Dv > 0)>), grid, dim3(512), 0, s, A, W, scale, bias, act, C, m, \
                       n, k, gx, gy, KS, per, ws, tk)
#define FTCF_SK_RG(I8, Dv)                                                                                                       \
    if (RGs == 2) {                                                                                                              \
        FTCF_SK(I8, Dv, 2);                                                                                                      \
    }                                                                                                                            \
    else if (RGs == 3) {                                                                                                         \
        FTCF_SK(I8, Dv, 3);                                                                                                      \
    }                                                                                                                            \
    else if (RGs == 8) {                                                                                                         \
        FTCF_SK(I8, Dv, 8);                                                                                                      \
    }                                                                                                                            \
    else {                                                                                                                       \
        FTCF_SK(I8, Dv, 4);                                                                                                      \
    }
            if (int8 && dd) {
                FTCF_SK_RG(true, DQ)
            }
            else if (int8) {
                FTCF_SK_RG(true, 0)
            }
            else if (dd) {
                FTCF_SK_RG(false, DQ)
            }
            else {
                FTCF_SK_RG(false, 0)
            }
#undef FTCF_SK_RG
#undef FTCF_SK
            FTCF_HIP_CHECK(hipGetLastError());
            return;
        }
    }
    if (m <= 32) {
        dim3 grid((NT + 3) / 4, (m + 31) / 32);
        if (int8) {
            hipLaunchKernelGGL((k_gemm_tiled<true, 2, 1, 4>), grid, dim3(256), 0, s, A, W, scale, bias, act, C, m, n, k);
        }
        else {
            hipLaunchKernelGGL((k_gemm_tiled<false, 2, 1, 4>), grid, dim3(256), 0, s, A, W, scale, bias, act, C, m, n, k);
        }
    }
    else {
        // Block tile 128 x 256 (RG, NG, WAVES) = (8, 2, 8).  Measured on the 13B layer at m = 1024 (tools/bench_gemm.py, us per
        // layer of four GEMMs): 128 x 512 tiles 1006, 64 x 256 tiles 857, this 760; with `nt` weight loads and the plain
        // 2-D grid 791 (int8) / 932 (fp16 weights).  hipBLASLt on the same fp16 shapes: 651-762 depending on the layout.
        // Two things matter more than the tile: (1) the weight panel of a column block is read by every row block, so it is
        // loaded with the default cache policy (the decode kernels' `nt` would drop it from the L2), and (2) the row blocks
        // of one panel are placed on ONE XCD, next to each other in time (1-D grid, workgroup b runs on XCD b % 8): the panel
        // crosses the fabric once instead of once per XCD.  Few tiles (m <= 512 with n = 5120: < 160 workgroups for 256
        // CUs) -> 64-row tiles.
        constexpr int NG = 2;
        const int     gx = (NT + 8 * NG - 1) / (8 * NG);
        constexpr int small_tiles = 160;
        // (above 1024 rows also when the 128-row tiles would leave a partial round of the 256 CUs: n = 5120 at 1025..1536 rows is
        // 180..240 tiles -- 13B prompt phase at 1536 tokens 51.2 -> 47.7 ms (int8), 54.0 -> 52.4 (fp16); at exactly 1024 rows, 160
        // tiles, the 64-row form is SLOWER: 31.1 -> 32.5 / 33.1 -> 38.8)
        const long    t128  = (long)gx * ((m + 127) / 128);
        const bool    small = t128 < small_tiles || (m > 1024 && t128 < 256);
        const int     gy = small ? (m + 63) / 64 : (m + 127) / 128;
        dim3          grid(8 * ((gx + 7) / 8) * gy);
        // The ring form, two k-steps deep (it needs an even number of k-steps; else the plain loop).  At most one workgroup per CU
        // (n = 5120 at m <= 1024: 160 tiles): with the A fragments prefetched, 163 VGPRs.  More tiles than that: without the
        // prefetch and held to 128 VGPRs, so that two workgroups share a CU as in the plain form (124 VGPRs).  13B int8 layer,
        // us, plain loop / ring at one workgroup per CU / ring at two: m = 1024: QKV 163 / 180 / 163, out-proj 91 / 80 / 93,
        // FFN1 218 / 235 / 216, FFN2 293 / 270 / 295; m = 2048: 348 / 375 / 340, 138 / 147 / 135, 392 / 418 / 380, 506 / 532 / 487.
        constexpr int big_deep = 2;
        const int        bd       = (k / GEMM_KSTEP) % 2 == 0 ? big_deep : 0;
        const bool       lone     = (long)gx * gy <= 256;
#define FTCF_BIG(I8, RGv, Dv, PFv, O2v)                                                                                          \
    hipLaunchKernelGGL((k_gemm_tiled<I8, RGv, NG, 8, false, true, false, Dv, PFv, O2v>), grid, dim3(512), 0, s, A, W, scale, bias, act, \
                       C, m, n, k, gx, gy)
#define FTCF_BIG_D(I8, RGv)                                                                                                      \
    if (bd == 2 && lone) {                                                                                                       \
        FTCF_BIG(I8, RGv, 2, true, false);                                                                                       \
    }                                                                                                                            \
    else if (bd == 2) {                                                                                                          \
        FTCF_BIG(I8, RGv, 2, !I8, I8);                                                                                           \
    }                                                                                                                            \
    else {                                                                                                                       \
        FTCF_BIG(I8, RGv, 0, false, false);                                                                                      \
    }
        // fp16 weights from 2048 rows: the same 128 x 256 tile on FOUR waves of four column groups each -- an A fragment read from LDS
        // feeds four MFMAs instead of two, two workgroups per CU (252 VGPRs, plain loop: the ring of two k-steps spills 58).  13B prompt
        // phase, fp16 weights: 2048 tokens 64.4 -> 63.2 ms, 4096: 133.2 -> 129.6; at 1024 tokens it is slower (33.1 -> 34.7), and with
        // int8 weights at every length (1024: 29.9 -> 33.4, 4096: 120.0 -> 121.0).
        if (!int8 && !small && m >= 2048) {
            hipLaunchKernelGGL((k_gemm_tiled<false, 8, 4, 4, false, true, false, 0, false, true>), grid, dim3(256), 0, s, A, W, scale, bias,
                               act, C, m, n, k, gx, gy);
            FTCF_HIP_CHECK(hipGetLastError());
            return;
        }
        if (small && int8) {
            FTCF_BIG_D(true, 4)
        }
        else if (small) {
            FTCF_BIG_D(false, 4)
        }
        else if (int8) {
            FTCF_BIG_D(true, 8)
        }
        else {
            FTCF_BIG_D(false, 8)
        }
#undef FTCF_BIG_D
#undef FTCF_BIG
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// Batched decode GEMM, 5 <= m <= 16 rows: HBM bound like the GEMV kernels, so it is built like them -- every wave streams
// the K extent of ONE 16-column group with two register batches of 8 tiles in flight -- while the (L2 resident)
// activations of the 4 waves' common K range pass through LDS in double-buffered chunks of 8 tiles (one barrier per chunk).
// A 16x16x32 MFMA tile takes 16 rows for the price of one, so the weights are read once for the whole batch (the tiled
// GEMM above is shaped for prefill: one 1 KiB weight tile in flight per wave, 11x off the HBM roofline at m = 16).
// Small n (out-proj, FFN2) are cut along K as well (grid.y): partial sums go to an fp32 workspace [ks][m][n] and
// k_smallm_reduce adds them in split order and applies the epilogue.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SM_U = 8;  // tiles per chunk / register batch

template<bool INT8>
__global__ __launch_bounds__(256) void k_gemm_smallm(const f16* __restrict__ A, const void* __restrict__ W,
                                                     const f16* __restrict__ scale, const f16* __restrict__ bias,
                                                     int act, f16* __restrict__ C, float* __restrict__ partial, int m,
                                                     int n, int k)
{
    constexpr int TK  = INT8 ? TILE_K_I8 : TILE_K_F16;
    constexpr int KC  = SM_U * TK;  // k per chunk
    constexpr int LDA = KC + 8;     // halves per LDS row: rows start in different banks
    __shared__ __attribute__((aligned(16))) f16 As[2][16 * LDA];

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int NT = n / 16, KT = k / TK;
    const int nt = blockIdx.x * 4 + wid;
    const bool active = nt < NT;
    // this block's K range in tiles
    const int ks = gridDim.y;
    const int t0 = (int)((long)KT * blockIdx.y / ks), t1 = (int)((long)KT * (blockIdx.y + 1) / ks);
    const int nch = (t1 - t0 + SM_U - 1) / SM_U;

    const u32x4* wp = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(W)
                                                     + (((size_t)(active ? nt : 0) * KT + t0) * 64 + lane) * 16);
    f16x2 scale2 = {(f16)1.f, (f16)1.f};
    if constexpr (INT8) {
        if (active) {
            const f16 sc = scale[nt * 16 + c];
            scale2       = f16x2{sc, sc};
        }
    }
    // x chunk: 16 rows x KC halves = 16 * KC / 8 16-byte pieces, (16 * KC / 8) / 256 per thread
    constexpr int XPT = 16 * KC / 8 / 256;  // 4 (int8) / 2 (fp16)
    u32x4         xreg[XPT];
    auto load_x = [&](int ch) {
        const int ch2 = ch < nch ? ch : nch - 1;
#pragma unroll
        for (int i = 0; i < XPT; i++) {
            const int pc  = threadIdx.x + i * 256;  // piece index: row = pc / (KC / 8), column piece = pc % (KC / 8)
            int       row = pc / (KC / 8);
            row           = row < m ? row : m - 1;
            int kk        = (t0 + ch2 * SM_U) * TK + (pc % (KC / 8)) * 8;
            kk            = kk < k ? kk : k - 8;  // the last chunk of a range may be short
            xreg[i]       = *reinterpret_cast<const u32x4*>(A + (size_t)row * k + kk);
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XPT; i++) {
            const int pc = threadIdx.x + i * 256;
            *reinterpret_cast<u32x4*>(&As[buf][(pc / (KC / 8)) * LDA + (pc % (KC / 8)) * 8]) = xreg[i];
        }
    };
    u32x4 RA[SM_U], RB[SM_U];
    auto load_w = [&](u32x4 (&r)[SM_U], int ch) {
        const int ch2 = ch < nch ? ch : nch - 1;
#pragma unroll
        for (int u = 0; u < SM_U; u++) {
            int t = ch2 * SM_U + u;
            t     = t < t1 - t0 ? t : t1 - t0 - 1;
            r[u]  = __builtin_nontemporal_load(wp + (size_t)t * 64);
        }
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto consume = [&](const u32x4 (&r)[SM_U], int buf, int ch) {
        const int  nt_ch = (t1 - t0 - ch * SM_U < SM_U) ? t1 - t0 - ch * SM_U : SM_U;
        const f16* xr    = &As[buf][c * LDA + g * (INT8 ? 16 : 8)];
        if (nt_ch == SM_U) {
#pragma unroll
            for (int u = 0; u < SM_U; u++) {
                consume_tile_raw<INT8>(r[u], xr + u * TK, scale2, acc);
            }
        }
        else {
#pragma unroll
            for (int u = 0; u < SM_U; u++) {
                if (u < nt_ch) {
                    consume_tile_raw<INT8>(r[u], xr + u * TK, scale2, acc);
                }
            }
        }
    };
    if (nch > 0) {
        load_x(0);
        load_w(RA, 0);
        store_x(0);
        for (int ch = 0; ch < nch; ch += 2) {
            __syncthreads();  // chunk ch staged, buffer 1 free
            load_x(ch + 1);
            load_w(RB, ch + 1);
            __builtin_amdgcn_sched_barrier(0);
            consume(RA, 0, ch);
            __builtin_amdgcn_sched_barrier(0);
            store_x(1);
            if (ch + 1 >= nch) {
                break;
            }
            __syncthreads();
            load_x(ch + 2);
            load_w(RA, ch + 2);
            __builtin_amdgcn_sched_barrier(0);
            consume(RB, 1, ch + 1);
            __builtin_amdgcn_sched_barrier(0);
            store_x(0);
        }
    }
    if (!active) {
        return;
    }
    const int col = nt * 16 + c;
    if (ks > 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = g * 4 + j;
            if (row < m) {
                partial[((size_t)blockIdx.y * m + row) * n + col] = acc[j];
            }
        }
        return;
    }
    const float bv = bias ? (float)bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = g * 4 + j;
        if (row < m) {
            float v = acc[j];
            f16   h;
            if constexpr (INT8) {  // fused fp32 epilogue (epilogue_helpers.h:52-62)
                v += bv;
                if (act == 1) {
                    v = gelu_f32(v);
                }
                h = (f16)v;
            }
            else {  // cuBLAS rounds to half; bias/gelu follow in half (activation_kernels.cu:401-426)
                h = (f16)v;
                if (act == 1) {
                    h = gelu_f16(bias ? (f16)(h + bias[col]) : h);
                }
                else if (bias) {
                    h = h + bias[col];
                }
            }
            C[(size_t)row * n + col] = h;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// "Burst" form of the batched decode GEMM (used whenever a split-K workspace is available).  The chunked kernel above is
// latency bound: a workgroup lives for three dependent memory round trips to move 20 KiB per wave.  Here K is cut into
// slices of <= SMB_T tiles and every wave requests its WHOLE slice (20 KiB, 80 VGPRs) in one go, together with the
// workgroup's x slice (16 rows x slice_k, through LDS once): one round trip per workgroup, the whole matrix is in flight
// at once and the launch lasts about bytes / HBM rate -- IF a workgroup's slot (registers, LDS) is handed to the next
// workgroup as soon as its weights have arrived.  Hence the shape of the split-K reduction: partial sums travel as 8-byte
// {tag, value} granules (attn_device.hip.h; one relaxed agent-scope store each, the data is its own flag), every workgroup
// but the one that owns the LAST slice of a column block leaves right after issuing its stores, and that last one --
// dispatched after all of its siblings -- polls their granules, adds them in slice order (deterministic) and applies the
// epilogue.  (The first version took a ticket per workgroup after waiting for its stores to complete: two more memory
// round trips per workgroup; 40.2 -> 38.7 us per launch of the 13B layer's GEMM pairs at 16 rows.  What bounds the launch
// is that a slot streams nothing while its workgroup computes, leaves and is replaced: ~60 % of the in-flight capacity.)
// tag = f(decode step, launch) is unique per launch within a request; the engine zeroes the granules when a request begins.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SMB_T     = 20;  // tiles per wave and slice
constexpr int SMB_WAVES = 4;        // waves (16-column groups) per workgroup: they share the x slice
constexpr int SMB_SPINS = 1 << 22;  // bounded: a protocol bug must not hang the GPU
typedef unsigned long long                          smb_u64;
typedef __attribute__((address_space(1))) smb_u64   smb_gu64;

template<bool INT8>
__device__ __forceinline__ f16 smallm_epilogue(float v, const f16* __restrict__ bias, const int col, const int act)
{
    f16 h;
    if constexpr (INT8) {  // fused fp32 epilogue (epilogue_helpers.h:52-62)
        v += bias ? (float)bias[col] : 0.f;
        if (act == 1) {
            v = gelu_f32(v);
        }
        h = (f16)v;
    }
    else {  // cuBLAS rounds to half; bias/gelu follow in half (activation_kernels.cu:401-426)
        h = (f16)v;
        if (act == 1) {
            h = gelu_f16(bias ? (f16)(h + bias[col]) : h);
        }
        else if (bias) {
            h = h + bias[col];
        }
    }
    return h;
}

// One launch runs up to two independent GEMMs with the same m ("group": QKV with FFN1, out-proj with FFN2): a dependent
// launch costs ~8 us of dispatch latency however little it computes, which is most of a layer at tensor-parallel shard
// sizes.  1-D grid: problem 0's workgroups first; within a problem workgroup i = column block i / ks, slice i % ks.
struct SmallmProblem {
    const f16*  A;
    const void* W;
    const f16*  scale;
    const f16*  bias;
    f16*        C;
    smb_u64*    partial;  // granules [ks - 1][m][n]
    int         act, n, k, ks, bx;
};
struct SmallmGroup {
    SmallmProblem p[2];
    int           np, m;
    const int*    d_step;  // device-resident decode step (a replayed hipGraph freezes every scalar argument), or NULL
    unsigned      seq;     // launch counter of the owning workspace
    int*          err;     // sticky: a reducer gave up waiting
};
// (pointers that come out of the argument struct are global: the explicit address space keeps the accesses from
// becoming FLAT, which would also count on lgkmcnt and serialise with the LDS waits)
#define SMB_G(T, ptr) ((const __attribute__((address_space(1))) T*)(ptr))

template<bool INT8, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gemm_smallm_burst(const SmallmGroup G)
{
    constexpr int TK   = INT8 ? TILE_K_I8 : TILE_K_F16;
    constexpr int KMAX = SMB_T * TK;  // k per slice
    constexpr int LDA  = KMAX + 8;    // halves per LDS row: rows start in different banks
    constexpr int NTHR = 64 * WAVES;
    constexpr int BC   = 16 * WAVES;              // columns per workgroup
    constexpr int XP   = (16 * KMAX / 8 + NTHR - 1) / NTHR;  // 16-byte pieces of x per thread
    __shared__ __attribute__((aligned(16))) f16 As[16 * LDA];
    static_assert(sizeof(As) >= 16 * BC * sizeof(float), "the reducer parks its own partial sums in the x tile");
    const unsigned tag = ((G.d_step ? (unsigned)*SMB_G(int, G.d_step) : 0u) << 12) + (G.seq & 0xfffu) + 1u;
    int idx = blockIdx.x, pi = 0;
    if (G.np > 1 && idx >= G.p[0].bx * G.p[0].ks) {
        pi = 1;
        idx -= G.p[0].bx * G.p[0].ks;
    }
    const SmallmProblem& P = G.p[pi];
    const int      ks = P.ks;
    // slice major: the workgroups in flight at any moment read 20 KiB pieces spread over the whole matrix (measured: a
    // column block's slices as dispatch neighbours -- one contiguous region in flight -- is 12 % slower, an XCD-contiguous
    // walk 4 %); the owner of the last slice is dispatched after all its siblings
    const int bxl = idx % P.bx, sl = idx / P.bx;
    const int      m = G.m, n = P.n, k = P.k, act = P.act;
    const f16*     A = P.A;
    const f16*     bias = P.bias;
    f16*           C = P.C;
    smb_u64*       partial = P.partial;

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int NT = n / 16, KT = k / TK;
    const int nt = bxl * WAVES + wid;
    const bool active = nt < NT;
    const int t0 = (int)((long)KT * sl / ks), t1 = (int)((long)KT * (sl + 1) / ks);
    const int nts = t1 - t0;  // <= SMB_T

    // x first (L2 hits: they are back long before the weights and go to LDS while those are in flight), then the weights;
    // the tiles are consumed in request order as they land
    u32x4     xr[XP];
    const int ppr = nts * TK / 8;  // pieces per row of this slice
#pragma unroll
    for (int i = 0; i < XP; i++) {
        const int pc  = threadIdx.x + i * NTHR;
        int       row = pc / (KMAX / 8), p8 = pc % (KMAX / 8);
        row           = row < m ? row : m - 1;  // (also covers pc past the tile when XP rounds up)
        p8            = p8 < ppr ? p8 : ppr - 1;
        xr[i]         = *SMB_G(u32x4, reinterpret_cast<const u32x4*>(A + (size_t)row * k + (size_t)t0 * TK + p8 * 8));
    }
    f16x2 scale2 = {(f16)1.f, (f16)1.f};
    if constexpr (INT8) {
        const f16 sc = *SMB_G(f16, P.scale + (active ? nt : 0) * 16 + c);
        scale2       = f16x2{sc, sc};
    }
    __builtin_amdgcn_sched_barrier(0);
    const u32x4* wp = reinterpret_cast<const u32x4*>(P.W) + ((size_t)(active ? nt : 0) * KT + t0) * 64 + lane;
    u32x4        wr[SMB_T];
#pragma unroll
    for (int u = 0; u < SMB_T; u++) {
        const int t = u < nts ? u : nts - 1;  // clamped, never conditional
        wr[u]       = __builtin_nontemporal_load(SMB_G(u32x4, wp + (size_t)t * 64));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < XP; i++) {
        const int pc = threadIdx.x + i * NTHR;
        if (pc < 16 * KMAX / 8) {
            *reinterpret_cast<u32x4*>(&As[(pc / (KMAX / 8)) * LDA + (pc % (KMAX / 8)) * 8]) = xr[i];
        }
    }
    __syncthreads();
    f32x4      acc = {0.f, 0.f, 0.f, 0.f};
    const f16* xs  = &As[c * LDA + g * (INT8 ? 16 : 8)];
#pragma unroll
    for (int u = 0; u < SMB_T; u++) {
        if (u < nts) {
            consume_tile_raw<INT8>(wr[u], xs + u * TK, scale2, acc);
        }
    }
    const int col = nt * 16 + c;
    if (ks == 1) {
        if (active) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = g * 4 + j;
                if (row < m) {
                    C[(size_t)row * n + col] = smallm_epilogue<INT8>(acc[j], bias, col, act);
                }
            }
        }
        return;
    }
    if (sl < ks - 1) {  // publish and leave: nothing waits for these stores
        if (active) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = g * 4 + j;
                if (row < m) {
                    __hip_atomic_store((smb_gu64*)&partial[((size_t)sl * m + row) * n + col],
                                       ((smb_u64)tag << 32) | (smb_u64)__float_as_uint(acc[j]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        return;
    }
    // owner of the last slice: its own sums through LDS (the x tile is dead), the siblings' from their granules
    __syncthreads();
    float* own = reinterpret_cast<float*>(As);  // [16 rows][BC columns]
#pragma unroll
    for (int j = 0; j < 4; j++) {
        own[(g * 4 + j) * BC + wid * 16 + c] = acc[j];
    }
    __syncthreads();
    // BC columns x m rows: 4 outputs per thread, RS slices of each requested together (an agent-scope load is a memory
    // round trip)
    constexpr int RS = 5;
    const int     cl = threadIdx.x % BC, r0 = threadIdx.x / BC;  // rows r0, r0 + 4, r0 + 8, r0 + 12
    const int     cc = bxl * BC + cl;
    if (cc >= n) {
        return;
    }
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int   spins = 0;
    for (int s0 = 0; s0 < ks - 1; s0 += RS) {
        smb_u64 pv[4][RS];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < RS; j++) {
                const int s2 = s0 + j < ks - 1 ? s0 + j : ks - 2;
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    int row  = r0 + o * 4;
                    row      = row < m ? row : m - 1;
                    pv[o][j] = __hip_atomic_load((const smb_gu64*)&partial[((size_t)s2 * m + row) * n + cc], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int j = 0; j < RS; j++) {
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    ok &= (unsigned)(pv[o][j] >> 32) == tag;
                }
            }
            if (ok) {
                break;
            }
            if (++spins > SMB_SPINS) {
                *G.err = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int j = 0; j < RS; j++) {
            if (s0 + j < ks - 1) {
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    v[o] += __uint_as_float((unsigned)pv[o][j]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int row = r0 + o * 4;
        if (row < m) {
            C[(size_t)row * n + cc] = smallm_epilogue<INT8>(v[o] + own[row * BC + cl], bias, cc, act);
        }
    }
}

static int smallm_burst_slices(int k, bool int8)
{
    const int KT = k / (int8 ? TILE_K_I8 : TILE_K_F16);
    return (KT + SMB_T - 1) / SMB_T;
}

// split-K partials [slices][m][n] of ONE GEMM; take the maximum over the GEMMs that share the workspace
size_t gemm_smallm_workspace_bytes(int m, int n, int k, bool int8)
{
    return (size_t)std::max(8, smallm_burst_slices(k, int8)) * m * n * sizeof(smb_u64);  // {tag, value} granules
}
size_t gemm_smallm_ticket_bytes()
{
    return 256;  // [0] sticky error flag of the in-launch reduction, [1] launch counter (host side mirror: SmallmState)
}

// burst form of one or two GEMMs in one launch; `workspace` = [partial_bytes of partial sums][ticket table]
void launch_gemm_smallm_group(const SmallmDesc* d, int np, float* workspace, size_t partial_bytes, int m, bool int8,
                              hipStream_t s, const int* d_step, unsigned* seq, size_t partial_offset)
{
    FTCF_CHECK_ARG(np >= 1 && np <= 2 && m >= 1 && m <= 16 && workspace != nullptr && seq != nullptr,
                   "small-m GEMM group: bad arguments");
    SmallmGroup G{};
    G.np     = np;
    G.m      = m;
    G.d_step = d_step;
    G.seq    = (*seq)++;
    G.err    = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + partial_bytes);
    size_t poff = partial_offset;  // launches that may run concurrently use disjoint parts of the workspace
    long   wgs  = 0;
    for (int i = 0; i < np; i++) {
        FTCF_CHECK_ARG(d[i].k % GEMM_KSTEP == 0 && d[i].n % 16 == 0, "GEMM needs k % 64 == 0 and n % 16 == 0");
        SmallmProblem& P = G.p[i];
        P.A = d[i].A;
        P.W = d[i].W;
        P.scale = d[i].scale;
        P.bias = d[i].bias;
        P.C = d[i].C;
        P.act = d[i].act;
        P.n = d[i].n;
        P.k = d[i].k;
        P.ks = smallm_burst_slices(d[i].k, int8);
        P.bx = (d[i].n / 16 + SMB_WAVES - 1) / SMB_WAVES;
        P.partial = reinterpret_cast<smb_u64*>(reinterpret_cast<char*>(workspace) + poff);
        poff += gemm_smallm_workspace_bytes(m, d[i].n, d[i].k, int8);
        wgs += (long)P.bx * P.ks;
    }
    FTCF_CHECK_ARG(poff <= partial_bytes, "small-m GEMM: split-K workspace too small");
    dim3 grid((unsigned)wgs);
    if (int8) {
        hipLaunchKernelGGL((k_gemm_smallm_burst<true, SMB_WAVES>), grid, dim3(64 * SMB_WAVES), 0, s, G);
    }
    else {
        hipLaunchKernelGGL((k_gemm_smallm_burst<false, SMB_WAVES>), grid, dim3(64 * SMB_WAVES), 0, s, G);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

void launch_gemm_smallm(const f16* A, const void* W, const f16* scale, const f16* bias, int act, f16* C, float* workspace,
                        size_t partial_bytes, int m, int n, int k, bool int8, int num_cu, hipStream_t s, const int* d_step,
                        unsigned* seq)
{
    FTCF_CHECK_ARG(m >= 1 && m <= 16, "small-m GEMM handles 1..16 rows");
    FTCF_CHECK_ARG(k % GEMM_KSTEP == 0 && n % 16 == 0, "GEMM needs k % 64 == 0 and n % 16 == 0");
    const int NT = n / 16, bx = (NT + 3) / 4;
    (void)num_cu;
    if (workspace != nullptr) {
        const SmallmDesc d{A, W, scale, bias, act, C, n, k};
        launch_gemm_smallm_group(&d, 1, workspace, partial_bytes, m, int8, s, d_step, seq);
        return;
    }
    // no workspace (kernel-level entry points): the chunked form over the whole K extent
    dim3 grid(bx, 1);
    if (int8) {
        hipLaunchKernelGGL((k_gemm_smallm<true>), grid, dim3(256), 0, s, A, W, scale, bias, act, C, workspace, m, n, k);
    }
    else {
        hipLaunchKernelGGL((k_gemm_smallm<false>), grid, dim3(256), 0, s, A, W, scale, bias, act, C, workspace, m, n, k);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// logits_f32[m, n] = A[m,k] x W[n,k]^T for m > 4 (batched decode LM head).  W rows are k-contiguous, which is the
// B-operand order of the MFMA directly: lane (col = lane&15, kgroup = lane>>4) reads 16 B of row n0+col.
// Each wave takes NGR groups of 16 vocabulary rows: an A fragment (from L2) feeds NGR weight fragments (from HBM).
template<int NGR, int WPB, int UK>
__global__ __launch_bounds__(64 * WPB) void k_gemm_nk_f32out(const f16* __restrict__ A, const f16* __restrict__ W,
                                                        float* __restrict__ C, int m, int n, int k, int ldc)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n0 = (blockIdx.x * WPB + wid) * (16 * NGR);
    const int m0 = blockIdx.y * 16;
    if (n0 >= n) {
        return;
    }
    const int  arow = (m0 + c < m) ? m0 + c : m - 1;
    const f16* ap   = A + (size_t)arow * k + g * 8;
    const f16* wp[NGR];
    f32x4      acc[NGR];
#pragma unroll
    for (int j = 0; j < NGR; j++) {
        const int r = n0 + j * 16 + c;
        wp[j]       = W + (size_t)(r < n ? r : n - 1) * k + g * 8;
        acc[j]      = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // 16 weight fragments (16 KiB per wave) in flight per trip: the loop is HBM-latency bound.  The x fragments come from
    // the L2 through the same per-CU load path as the weights: one of them feeds NGR weight fragments (NGR = 2 made the
    // path carry 1 byte of x per 2 of weights: 2.9 TB/s on the 100864 x 5120 head)
    int           k0 = 0;
    for (; k0 + UK * 32 <= k; k0 += UK * 32) {
        f16x8 bw[NGR][UK], a[UK];
#pragma unroll
        for (int u = 0; u < UK; u++) {
#pragma unroll
            for (int j = 0; j < NGR; j++) {
                bw[j][u] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(wp[j] + k0 + u * 32));
            }
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            a[u] = *reinterpret_cast<const f16x8*>(ap + k0 + u * 32);
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
#pragma unroll
            for (int j = 0; j < NGR; j++) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], bw[j][u], acc[j], 0, 0, 0);
            }
        }
    }
    for (; k0 < k; k0 += 32) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(ap + k0);
#pragma unroll
        for (int j = 0; j < NGR; j++) {
            const f16x8 bw = *reinterpret_cast<const f16x8*>(wp[j] + k0);
            acc[j]         = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bw, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = m0 + g * 4 + q;
        if (row < m) {
#pragma unroll
            for (int j = 0; j < NGR; j++) {
                if (n0 + j * 16 + c < n) {
                    C[(size_t)row * ldc + n0 + j * 16 + c] = acc[j][q];
                }
            }
        }
    }
}

// The same product with the weight rows read as WHOLE 512-byte pieces (round 4).  The form above asks the memory for 64-byte
// pieces of sixteen rows per wave-load -- the B-fragment order taken straight from the row-major [V, H] image -- and reaches
// 3.6 TB/s on the 100864 x 5120 head; here a wave-load is two rows' 256 halves (lanes along k, as the GEMV LM head reads), the
// 16 x 256 tile goes through a wave-private LDS tile and comes back in fragment order, and the 16 x 256 tile of x is staged once
// per workgroup and chunk (double buffered, one barrier per chunk) instead of once per wave from the L2.  Every wave of a
// workgroup makes the same number of trips (16 vocabulary rows each); the next chunk's rows are requested before the current
// chunk's MFMAs.
constexpr int NKT_KC = 256;
template<int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gemm_nk_f32out_tr(const f16* __restrict__ A, const f16* __restrict__ W,
                                                                 float* __restrict__ C, int m, int n, int k, int ldc, int trips)
{
    constexpr int KC = NKT_KC, LDT = KC + 8, NTHR = 64 * WAVES;
    constexpr int XP = 16 * KC / 8 / NTHR;  // 16-byte pieces of the x tile per thread
    static_assert(16 * KC / 8 % NTHR == 0, "x tile pieces per thread");
    __shared__ __attribute__((aligned(16))) f16 xt[2][16 * LDT];
    __shared__ __attribute__((aligned(16))) f16 wt[WAVES][16 * LDT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16;
    const int NRG = (n + 15) / 16, nkc = k / KC;
    const int total = trips * nkc;
    f16*      wtw   = wt[wid];
    u32x4     wr[8], xr[XP];
    auto fetch = [&](const int it) {
        const int t = it / nkc, kc = it % nkc;
        int       rg = (t * (int)gridDim.x + (int)blockIdx.x) * WAVES + wid;
        rg           = rg < NRG ? rg : NRG - 1;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int row = rg * 16 + 2 * u + (lane >> 5);
            row     = row < n ? row : n - 1;
            wr[u]   = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + (size_t)row * k + (size_t)kc * KC + (lane & 31) * 8));
        }
#pragma unroll
        for (int u = 0; u < XP; u++) {
            const int idx = threadIdx.x + u * NTHR;
            int       row = m0 + idx / (KC / 8);
            row           = row < m ? row : m - 1;
            xr[u]         = *reinterpret_cast<const u32x4*>(A + (size_t)row * k + (size_t)kc * KC + (idx % (KC / 8)) * 8);
        }
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int it = 0; it < total; it++) {
        const int buf = it & 1;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            *reinterpret_cast<u32x4*>(&wtw[(2 * u + (lane >> 5)) * LDT + (lane & 31) * 8]) = wr[u];
        }
#pragma unroll
        for (int u = 0; u < XP; u++) {
            const int idx = threadIdx.x + u * NTHR;
            *reinterpret_cast<u32x4*>(&xt[buf][(idx / (KC / 8)) * LDT + (idx % (KC / 8)) * 8]) = xr[u];
        }
        __syncthreads();  // the x tile of this chunk (the readers of this buffer two chunks ago passed the previous barrier)
        if (it + 1 < total) {
            fetch(it + 1);
        }
#pragma unroll
        for (int ks = 0; ks < KC / 32; ks++) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(&xt[buf][c * LDT + ks * 32 + g * 8]);
            const f16x8 b = *reinterpret_cast<const f16x8*>(&wtw[c * LDT + ks * 32 + g * 8]);
            acc           = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
        if (it % nkc == nkc - 1) {  // the trip's last chunk: 16 rows of x by 16 vocabulary rows
            const int rg = ((it / nkc) * (int)gridDim.x + (int)blockIdx.x) * WAVES + wid;
            if (rg < NRG) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = m0 + g * 4 + q;
                    if (row < m && rg * 16 + c < n) {
                        C[(size_t)row * ldc + rg * 16 + c] = acc[q];
                    }
                }
            }
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

void launch_gemm_nk_f32out(const f16* A, const f16* W_nk, float* C, int m, int n, int k, int ldc, hipStream_t s)
{
    FTCF_CHECK_ARG(k % 32 == 0, "k must be a multiple of 32");
    {
        if (k % NKT_KC == 0 && n >= 16) {  // (else the fragment-order form below)
            constexpr int WAVES = 4;
            const int     NRG = (n + 15) / 16;
            // 100864 x 5120 at 16 rows, us per launch by trips per wave: 1 (1576 workgroups): 171.0, 2: 190.9, 3 (526 workgroups = 2.05 per
            // CU): 213.2, 4: 172.2, 6: 221.7 -- the fragment-order form: 283.7
            constexpr int trips = 1;
            const int     gx    = (NRG + WAVES * trips - 1) / (WAVES * trips);
            dim3          grid(gx, (m + 15) / 16);
            hipLaunchKernelGGL((k_gemm_nk_f32out_tr<WAVES>), grid, dim3(64 * WAVES), 0, s, A, W_nk, C, m, n, k, ldc, trips);
            FTCF_HIP_CHECK(hipGetLastError());
            return;
        }
    }
    // (fallback for k % 256 != 0; NGR, waves per workgroup, 32-k steps per trip) = (8, 2, 2): 289 us on the 100864 x 5120 head at 16 rows (3.6 TB/s);
    // (2, 4, 8) 353, (4, 4, 4) 328, (8, 2, 4) 320, (16, 2, 1) 307.  More fragments in flight do not help: the row-major
    // [V, H] image (the caller's buffer, not re-tiled) is read in 64-byte pieces of 16 rows per wave-load.
    constexpr int NGR = 8, WPB = 2;
    dim3          grid((n + 16 * NGR * WPB - 1) / (16 * NGR * WPB), (m + 15) / 16);
    hipLaunchKernelGGL((k_gemm_nk_f32out<NGR, WPB, 2>), grid, dim3(64 * WPB), 0, s, A, W_nk, C, m, n, k, ldc);
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
