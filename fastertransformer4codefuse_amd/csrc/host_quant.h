// Host-side weight-only quantiser and tile-layout transforms (libth_common counterpart).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ftcf {
// cutlass_preprocessors.cc:576-673 (symmetric_quantize) followed by the gfx950 tiling that plays the role of
// preprocess_weights_for_mixed_gemm (:500-539).  weight: [E,K,N] fp32 or fp16; out_scale in the weight dtype.
void host_symmetric_quantize_int8(const void* weight, bool is_half, size_t E, size_t K, size_t N, int8_t* out_q,
                                  void* out_scale);
void host_int8_rowmajor_to_tiled(const int8_t* q, size_t K, size_t N, int8_t* out);
void host_int8_tiled_to_rowmajor(const int8_t* q, size_t K, size_t N, int8_t* out);
}  // namespace ftcf
