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

// Kernels of the encode pass (bf16 GEMM, window attention, LayerNorm) raise their wave priority: on a CU they
// share with a decoder-step block their instructions win the issue arbitration, which shortens the tail every
// GEMM launch waits for (the step kernels are latency-bound and barely notice).  -DODIC_NO_ENCODE_PRIO disables.
#ifdef ODIC_NO_ENCODE_PRIO
#define ODIC_ENCODE_PRIO() ((void)0)
#else
#define ODIC_ENCODE_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

__device__ __forceinline__ float bf16_to_f32(bf16_raw h) {
  return __uint_as_float(((unsigned)h) << 16);
}
// round-to-nearest-even; NaN stays NaN (the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_raw, b);
}

// four floats → four OCP e4m3 bytes, saturating at ±448 (the converter alone returns NaN past the range)
__device__ __forceinline__ unsigned f32x4_to_fp8(float a, float b, float c, float d) {
  const float lim = 448.0f;
  a = fminf(fmaxf(a, -lim), lim); b = fminf(fmaxf(b, -lim), lim);
  c = fminf(fmaxf(c, -lim), lim); d = fminf(fmaxf(d, -lim), lim);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
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

// GELU for four accumulator values at once, transcendental-free so it compiles to packed FP32 FMAs
// (v_pk_fma_f32: 2 lanes-worth per issue):  gelu(x) = relu(x) - h(|x|),  h(u) = u·Φ(-u) is even and
// decays to 0, so h(u) = u·(0.5 - u·P(u²)) with u clamped at 4.25 (h(4.25) = 4.5e-5).  P is a degree-8
// weighted-minimax fit; max |error| over all x is 3.8e-5 in fp32 arithmetic — below half a bf16 ulp
// for every |gelu| > 0.02 (tests/test_hip_ops.py::test_gelu_poly_accuracy).  bf16 path only.
__device__ __forceinline__ f32x4_t gelu_poly4(f32x4_t x) {
  const f32x4_t u = __builtin_elementwise_min(__builtin_elementwise_abs(x), (f32x4_t)(4.25f));
  const f32x4_t t = u * u;
  f32x4_t p = (f32x4_t)(4.547085625e-11f);
  p = p * t + (f32x4_t)(-4.515313901e-09f);
  p = p * t + (f32x4_t)(1.986162346e-07f);
  p = p * t + (f32x4_t)(-5.147519914e-06f);
  p = p * t + (f32x4_t)(8.848919970e-05f);
  p = p * t + (f32x4_t)(-1.079078298e-03f);
  p = p * t + (f32x4_t)(9.718779474e-03f);
  p = p * t + (f32x4_t)(-6.619028002e-02f);
  p = p * t + (f32x4_t)(3.988192081e-01f);
  const f32x4_t w = (f32x4_t)(0.5f) - u * p;
  return __builtin_elementwise_max(x, (f32x4_t)(0.f)) - u * w;
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
