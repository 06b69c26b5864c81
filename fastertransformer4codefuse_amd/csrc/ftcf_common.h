// Common device/host helpers for the gfx950 engine.  HIP only -- no CUDA compatibility layer.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <stdio.h>

#include <stdexcept>
#include <string>

namespace ftcf {

typedef _Float16 f16;
typedef f16      f16x2 __attribute__((ext_vector_type(2)));
typedef f16      f16x4 __attribute__((ext_vector_type(4)));
typedef f16      f16x8 __attribute__((ext_vector_type(8)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// ---- engine-private weight tile layouts (DESIGN.md "Data layout in HBM") ------------------------------------------
// int8 : tile = 16 columns x 64 k = 1024 B ; byte address ((nt*KT + kt)*64 + lane)*16 + j holds
//        u8(q[k = kt*64 + (lane>>4)*16 + j][n = nt*16 + (lane&15)] + 128)
// fp16 : tile = 16 columns x 32 k = 1024 B ; half index ((nt*KT + kt)*64 + lane)*8 + j holds
//        w[k = kt*32 + (lane>>4)*8 + j][n = nt*16 + (lane&15)]
// A wave's load instruction therefore reads 1 KiB of contiguous memory, a column group's K extent is contiguous,
// and the same image feeds the VALU GEMV (m <= 4) and the MFMA GEMM (lane = B-operand lane of mfma_f32_16x16x32_f16).
constexpr int TILE_N        = 16;
constexpr int TILE_K_I8     = 64;
constexpr int TILE_K_F16    = 32;
constexpr int TILE_BYTES    = 1024;

struct Error : public std::runtime_error {
    int code;
    Error(int c, const std::string& s): std::runtime_error(s), code(c) {}
};

#define FTCF_HIP_CHECK(expr)                                                                                           \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess) {                                                                                        \
            throw ::ftcf::Error(-2, std::string("HIP error ") + hipGetErrorString(_e) + " at " + __FILE__ + ":"       \
                                        + std::to_string(__LINE__) + " (" #expr ")");                                  \
        }                                                                                                              \
    } while (0)

#define FTCF_CHECK_ARG(cond, msg)                                                                                      \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            throw ::ftcf::Error(-1, std::string("invalid argument: ") + (msg) + " [" #cond "]");                       \
        }                                                                                                              \
    } while (0)

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        v += __shfl_xor(v, o, 64);
    }
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        v = fmaxf(v, __shfl_xor(v, o, 64));
    }
    return v;
}

// block-wide sum of up to 2 floats; `red` needs 2*(blockDim/64) floats of LDS. All threads get the result.
template<int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        v[i] = wave_sum(v[i]);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            red[wid * NV + i] = v[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
        float s = 0.f;
        for (int w = 0; w < nw; w++) {
            s += red[w * NV + i];
        }
        v[i] = s;
    }
    __syncthreads();
}

// u8 (q + 128) -> f16 exactly, two at a time: the 0x6400 magic-number trick the reference's converter uses
// (cutlass_extensions/include/cutlass_extensions/interleaved_numeric_conversion.h:50-83): 0x64xx is the half
// 1024 + xx, subtracting 1152 = 1024 + 128 leaves q exactly.
__device__ __forceinline__ void dequant4(uint32_t w, f16x2 scale2, f16x2& lo, f16x2& hi)
{
    const uint32_t magic = 0x64646464u;
    uint32_t       a     = __builtin_amdgcn_perm(magic, w, 0x04010400u);  // {b0, 0x64, b1, 0x64}
    uint32_t       b     = __builtin_amdgcn_perm(magic, w, 0x04030402u);  // {b2, 0x64, b3, 0x64}
    const f16x2    bias  = {(f16)1152.0f, (f16)1152.0f};
    f16x2          fa    = __builtin_bit_cast(f16x2, a) - bias;
    f16x2          fb    = __builtin_bit_cast(f16x2, b) - bias;
    lo                   = fa * scale2;  // B_f16 = half(q) * scale, rounded to half (mma_tensorop_dequantizer.h)
    hi                   = fb * scale2;
}

__device__ __forceinline__ float dot2(f16x2 a, f16x2 b, float c)
{
    return __builtin_amdgcn_fdot2(a, b, c, false);
}

// tanh-GELU in fp32: cutlass_extensions/.../ft_fused_activations.h:72-90 (GELU_taylor<float>)
__device__ __forceinline__ float gelu_f32(float z)
{
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    return 0.5f * z * (1.0f + tanhf(k0 * z * (1.0f + k1 * z * z)));
}
// kernels/activation_kernels.cu:54-85 GeluActivation<half2> (fp16 engine, non fused path)
__device__ __forceinline__ f16 gelu_f16(f16 v)
{
    f16   p3  = v * (v * v);
    float cdf = 0.5f * (1.0f + tanhf(0.7978845608028654f * ((float)v + 0.044715f * (float)p3)));
    return v * (f16)cdf;
}
#endif  // __HIPCC__

}  // namespace ftcf
