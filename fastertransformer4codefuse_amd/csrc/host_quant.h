// Host-side weight-only quantiser and tile-layout transforms (libth_common counterpart).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ftcf {
// cutlass_preprocessors.cc:576-673 (symmetric_quantize) followed by the gfx950 tiling that plays the role of
// preprocess_weights_for_mixed_gemm (:500-539).  weight: [E,K,N] fp32 (dtype 0), fp16 (1) or bf16 (2, raw bits);
// out_scale in the weight dtype.
void host_symmetric_quantize_int8(const void* weight, int dtype, size_t E, size_t K, size_t N, int8_t* out_q,
                                  void* out_scale);
void host_int8_rowmajor_to_tiled(const int8_t* q, size_t K, size_t N, int8_t* out);
void host_int8_tiled_to_rowmajor(const int8_t* q, size_t K, size_t N, int8_t* out);
// int8 weights as the CUDA build stores them for SM75..SM89 (preprocess_weights_for_mixed_gemm,
// cutlass_preprocessors.cc:500-539: row permutation in groups of 16, column major, 64-row x 2-column interleave, +128 and
// the [0,2,1,3] byte order of each register) <-> row major [K,N].  K % 64 == 0, N % 2 == 0.
void host_int8_cuda_sm80_to_rowmajor(const int8_t* q_cuda, size_t K, size_t N, int8_t* out);
void host_int8_rowmajor_to_cuda_sm80(const int8_t* q, size_t K, size_t N, int8_t* out_cuda);
}  // namespace ftcf
