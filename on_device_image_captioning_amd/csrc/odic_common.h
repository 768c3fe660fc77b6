// Shared device/host helpers for libodic_hip.so (gfx950 only — no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/odic_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // one 16x16 MFMA accumulator
typedef unsigned short bf16_raw;

#define ODIC_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_raw h) {
  return __uint_as_float(((unsigned)h) << 16);
}
// round-to-nearest-even; NaN stays NaN (the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_raw, b);
}

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_raw>(const bf16_raw* p) { return bf16_to_f32(*p); }

template <typename T> __device__ __forceinline__ void store_from_f32(T* p, float v);
template <> __device__ __forceinline__ void store_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_from_f32<bf16_raw>(bf16_raw* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (fp32 path) and a 1.5e-7-abs-error erf (Abramowitz–Stegun 7.1.26) for the bf16
// path, where the result is rounded to 8 bits of mantissa anyway.
__device__ __forceinline__ float gelu_exact(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float e = 1.0f - poly * t * __expf(-z * z);
  const float erfv = x < 0.f ? -e : e;
  return 0.5f * x * (1.0f + erfv);
}

template <bool FAST> __device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ODIC_ACT_GELU: return FAST ? gelu_fast(v) : gelu_exact(v);
    case ODIC_ACT_RELU: return fmaxf(v, 0.f);
    case ODIC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

static inline int odic_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
