// Host-side weight-only quantiser (see host_quant.h).  Pure CPU code: runs inside `libth_common`'s
// symmetric_quantize_last_axis_of_batched_matrix_int8 exactly like the reference's (WeightOnlyQuantOps.cc:140-233).
#include "host_quant.h"

#include <cmath>
#include <cstring>
#include <vector>

#include "ftcf_common.h"

namespace ftcf {

// Tile layout (ftcf_common.h): byte ((nt*KT + kt)*64 + lane)*16 + j  <-  u8(q[kt*64 + (lane>>4)*16 + j][nt*16 + (lane&15)] + 128)
// The +128 bias mirrors add_bias_and_interleave_int8s_inplace (cutlass_preprocessors.cc:350-370): the kernels
// convert u8 -> f16 with the 0x6400 magic number and subtract 1152.
// Both directions walk the row-major matrix a 64 x 64 block at a time (four column tiles of one k tile): its rows are read / written
// as whole 64-byte lines (walking one column tile over all K touched every line of the matrix sixteen bytes at a time, from four
// threads: 13.6 GB of a 13B model took minutes)
void host_int8_rowmajor_to_tiled(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    FTCF_CHECK_ARG(K % TILE_K_I8 == 0 && N % TILE_N == 0, "int8 tiling needs K % 64 == 0 and N % 16 == 0");
    const size_t KT = K / TILE_K_I8, NT = N / TILE_N, NB4 = (NT + 3) / 4;
    uint8_t*     o  = reinterpret_cast<uint8_t*>(out);
#pragma omp parallel for collapse(2) schedule(static)
    for (size_t kt = 0; kt < KT; kt++) {
        for (size_t nb = 0; nb < NB4; nb++) {
            uint8_t      buf[64][64];
            const size_t n0 = nb * 64, w = std::min<size_t>(64, N - n0);
            for (int kk = 0; kk < 64; kk++) {
                std::memcpy(buf[kk], q + (kt * 64 + kk) * N + n0, w);
            }
            for (size_t t = 0; t < 4 && nb * 4 + t < NT; t++) {
                uint8_t* tile = o + ((nb * 4 + t) * KT + kt) * TILE_BYTES;
                for (int lane = 0; lane < 64; lane++) {
                    const int c = (int)t * 16 + (lane & 15), k0 = (lane >> 4) * 16;
                    for (int j = 0; j < 16; j++) {
                        tile[lane * 16 + j] = (uint8_t)((int)(int8_t)buf[k0 + j][c] + 128);
                    }
                }
            }
        }
    }
}

void host_int8_tiled_to_rowmajor(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    FTCF_CHECK_ARG(K % TILE_K_I8 == 0 && N % TILE_N == 0, "int8 tiling needs K % 64 == 0 and N % 16 == 0");
    const size_t   KT = K / TILE_K_I8, NT = N / TILE_N, NB4 = (NT + 3) / 4;
    const uint8_t* in = reinterpret_cast<const uint8_t*>(q);
#pragma omp parallel for collapse(2) schedule(static)
    for (size_t kt = 0; kt < KT; kt++) {
        for (size_t nb = 0; nb < NB4; nb++) {
            int8_t       buf[64][64];
            const size_t n0 = nb * 64, w = std::min<size_t>(64, N - n0);
            for (size_t t = 0; t < 4 && nb * 4 + t < NT; t++) {
                const uint8_t* tile = in + ((nb * 4 + t) * KT + kt) * TILE_BYTES;
                for (int lane = 0; lane < 64; lane++) {
                    const int c = (int)t * 16 + (lane & 15), k0 = (lane >> 4) * 16;
                    for (int j = 0; j < 16; j++) {
                        buf[k0 + j][c] = (int8_t)((int)tile[lane * 16 + j] - 128);
                    }
                }
            }
            for (int kk = 0; kk < 64; kk++) {
                std::memcpy(out + (kt * 64 + kk) * N + n0, buf[kk], w);
            }
        }
    }
}

// bfloat16 as the reference's __nv_bfloat16 instantiation sees it (WeightOnlyQuantOps.cc:205,
// symmetric_quantize<__nv_bfloat16, __nv_bfloat16>): float(x) is exact, T(float) rounds to nearest even
struct bf16_t {
    uint16_t bits;
    bf16_t() = default;
    explicit bf16_t(float f)
    {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) {
            bits = (uint16_t)((u >> 16) | 0x40u);  // quiet NaN
        }
        else {
            u += 0x7fffu + ((u >> 16) & 1u);
            bits = (uint16_t)(u >> 16);
        }
    }
    explicit operator float() const
    {
        const uint32_t u = (uint32_t)bits << 16;
        float          f;
        memcpy(&f, &u, 4);
        return f;
    }
};

template<typename T>
static void quantize_one(const T* w, size_t K, size_t N, int8_t* q_rowmajor, T* scale)
{
    std::vector<float> col_max(N, 0.f);
    for (size_t i = 0; i < K; i++) {
        const T* row = w + i * N;
        for (size_t j = 0; j < N; j++) {
            const float a = std::fabs((float)row[j]);
            if (a > col_max[j]) {
                col_max[j] = a;
            }
        }
    }
    for (size_t j = 0; j < N; j++) {
        col_max[j] *= (1.f / 128.f);       // quant_range_scale = 1 / 2^(bits-1)
        scale[j] = (T)col_max[j];          // stored in the weight dtype (:618-621)
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < K; i++) {
        const T* row = w + i * N;
        for (size_t j = 0; j < N; j++) {
            const float s = std::round((float)row[j] / col_max[j]);  // divides by the UNROUNDED fp32 scale (:626-632)
            float       c = (s < 127.f) ? s : 127.f;                 // std::min(127.f, s): NaN -> 127.f
            if (!(c > -128.f)) {
                c = (c != c) ? 127.f : -128.f;
            }
            q_rowmajor[i * N + j] = (int8_t)c;
        }
    }
}

void host_symmetric_quantize_int8(const void* weight, int dtype, size_t E, size_t K, size_t N, int8_t* out_q,
                                  void* out_scale)
{
    FTCF_CHECK_ARG(dtype >= 0 && dtype <= 2, "weight dtype must be fp32 (0), fp16 (1) or bf16 (2)");
    FTCF_CHECK_ARG(weight && out_q && out_scale, "NULL tensor");
    FTCF_CHECK_ARG(E >= 1 && K >= 1 && N >= 1, "empty weight");
    FTCF_CHECK_ARG(K % TILE_K_I8 == 0 && N % TILE_N == 0,
                   "weight-only int8 needs K % 64 == 0 (as the reference, fpA_intB_gemm_template.h:159-163) and N % 16 == 0");
    std::vector<int8_t> tmp(K * N);
    for (size_t e = 0; e < E; e++) {
        if (dtype == 1) {
            quantize_one<f16>(reinterpret_cast<const f16*>(weight) + e * K * N, K, N, tmp.data(),
                              reinterpret_cast<f16*>(out_scale) + e * N);
        }
        else if (dtype == 2) {
            quantize_one<bf16_t>(reinterpret_cast<const bf16_t*>(weight) + e * K * N, K, N, tmp.data(),
                                 reinterpret_cast<bf16_t*>(out_scale) + e * N);
        }
        else {
            quantize_one<float>(reinterpret_cast<const float*>(weight) + e * K * N, K, N, tmp.data(),
                                reinterpret_cast<float*>(out_scale) + e * N);
        }
        host_int8_rowmajor_to_tiled(tmp.data(), K, N, out_q + e * K * N);
    }
}

// ---- importer / exporter of the CUDA build's SM75..SM89 int8 layout ------------------------------------------------
// Closed form of the four steps of preprocess_weights_for_mixed_gemm (cutlass_preprocessors.cc:500-539) for int8:
// element (k, n) of the row-major matrix ends up at byte
//   word  = (n / 2) * (K / 4) * 2  +  2 * 16 * (kp / 64)  +  16 * (n % 2)  +  (kp % 64) / 4
//   byte  = {0, 2, 1, 3}[kp % 4]
// with kp the row AFTER the 16-row permutation (row kp of the permuted matrix holds original row
// 16*(kp/16) + MAP[kp%16], MAP = 0 1 8 9 2 3 10 11 4 5 12 13 6 7 14 15), and the value stored is q + 128.
static const int kSm80RowMap[16] = {0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15};
static const int kSm80ByteSwz[4] = {0, 2, 1, 3};

static inline size_t sm80_byte_index(size_t kp, size_t n, size_t K)
{
    const size_t word = (n / 2) * (K / 4) * 2 + 32 * (kp / 64) + 16 * (n % 2) + (kp % 64) / 4;
    return word * 4 + (size_t)kSm80ByteSwz[kp % 4];
}

void host_int8_cuda_sm80_to_rowmajor(const int8_t* q_cuda, size_t K, size_t N, int8_t* out)
{
#pragma omp parallel for schedule(static)
    for (long long kpl = 0; kpl < (long long)K; kpl++) {
        const size_t kp = (size_t)kpl;
        const size_t k  = 16 * (kp / 16) + (size_t)kSm80RowMap[kp % 16];
        for (size_t n = 0; n < N; n++) {
            out[k * N + n] = (int8_t)((int)(uint8_t)q_cuda[sm80_byte_index(kp, n, K)] - 128);
        }
    }
}

void host_int8_rowmajor_to_cuda_sm80(const int8_t* q, size_t K, size_t N, int8_t* out_cuda)
{
#pragma omp parallel for schedule(static)
    for (long long kpl = 0; kpl < (long long)K; kpl++) {
        const size_t kp = (size_t)kpl;
        const size_t k  = 16 * (kp / 16) + (size_t)kSm80RowMap[kp % 16];
        for (size_t n = 0; n < N; n++) {
            out_cuda[sm80_byte_index(kp, n, K)] = (int8_t)(uint8_t)((int)q[k * N + n] + 128);
        }
    }
}

}  // namespace ftcf
