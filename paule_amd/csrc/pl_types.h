// Shared device-side types and helpers (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pl {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int WAVE = 64;

// every feature dimension is padded to a multiple of 32 elements (zero filled) so that
// K loops run in whole 16-byte chunks for both f32 (4 el) and bf16 (8 el) and MFMA k-steps
__host__ __device__ constexpr int pad32(int x) { return (x + 31) / 32 * 32; }
__host__ __device__ constexpr int pad16(int x) { return (x + 15) / 16 * 16; }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return static_cast<float>(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return static_cast<bf16_t>(v); }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// bf16 path: hardware exp2 / rcp (v_exp_f32, v_rcp_f32; ~1 ulp each), far inside the bf16 output rounding
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {   // 1 - 2 / (1 + e^{2x}); saturates cleanly at +-1
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)), 1.0f);
}
// c_t = f c_{t-1} + i g with the rounding spelled out: left to the compiler, WHICH product is fused into the FMA differs from
// kernel to kernel, and the per-layer and the fused sweeps (lstm_persist.hip, lstm_fused.hip) are compared bit for bit
__device__ __forceinline__ float cell_c(float f, float c_prev, float i, float g) { return __builtin_fmaf(f, c_prev, i * g); }


// activation functions by arithmetic type: exact libm forms on the f32 path (parity bar 1e-5), hardware forms on bf16
template <typename AT> __device__ __forceinline__ float act_sigmoid(float x);
template <> __device__ __forceinline__ float act_sigmoid<float>(float x) { return sigmoid_f(x); }
template <> __device__ __forceinline__ float act_sigmoid<bf16_t>(float x) { return sigmoid_fast(x); }
template <typename AT> __device__ __forceinline__ float act_tanh(float x);
template <> __device__ __forceinline__ float act_tanh<float>(float x) { return tanhf(x); }
template <> __device__ __forceinline__ float act_tanh<bf16_t>(float x) { return tanh_fast(x); }

}  // namespace pl
