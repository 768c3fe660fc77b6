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

// ---------------------------------------------------------------------------------------------
// Split-fp16 ("h2", dtype code ODIC_H2) — the operand format of the near-exact fast mode (`precision='x3'`).
// A value is carried as hi + lo with hi = fp16(x), lo = fp16(x − hi): 22 significand bits, and a product
// a·w ≈ ah·wh + ah·wl + al·wh runs as THREE fp16 MFMAs (fp32 accumulate) instead of one 16x-slower fp32 MFMA.
// Memory layout: 4 bytes per element like fp32 — leading dimensions, strides and buffer sizes are those of an fp32
// tensor — but in groups of 8 consecutive K-elements: [8 x hi fp16 | 8 x lo fp16] (32 bytes), so that a lane's
// 16-byte MFMA fragment (8 consecutive k of one plane) is ONE aligned 16-byte chunk and a producer that owns 8
// adjacent columns stores 32 contiguous bytes.  All-zero bytes are the value 0.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
struct h2_t { unsigned v; };                    // tag type: one h2 element slot (4 bytes)

__device__ __forceinline__ void h2_split(float x, f16_t& hi, f16_t& lo) {
  x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);       // saturate instead of producing inf (inf − inf = NaN)
  hi = (f16_t)x;
  lo = (f16_t)(x - (float)hi);
}
// eight consecutive elements (col % 8 == 0) → the group's 32 bytes at element address `dst`
__device__ __forceinline__ void h2_store8(h2_t* dst, const float* v) {
  f16x8_t hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) { f16_t h, l; h2_split(v[e], h, l); hi[e] = h; lo[e] = l; }
  ((f16x8_t*)dst)[0] = hi;
  ((f16x8_t*)dst)[1] = lo;
}
// four consecutive elements starting at column c (c % 4 == 0) of the row that starts at `row`: 8 bytes into each plane
__device__ __forceinline__ void h2_store4(h2_t* row, int c, float a, float b, float cc, float d) {
  f16x4_t hi, lo;
  f16_t h, l;
  h2_split(a, h, l); hi[0] = h; lo[0] = l;
  h2_split(b, h, l); hi[1] = h; lo[1] = l;
  h2_split(cc, h, l); hi[2] = h; lo[2] = l;
  h2_split(d, h, l); hi[3] = h; lo[3] = l;
  char* g = (char*)row + (long)(c >> 3) * 32 + (c & 4) * 2;
  *(f16x4_t*)g = hi;
  *(f16x4_t*)(g + 16) = lo;
}
__device__ __forceinline__ void h2_store1(h2_t* row, int c, float x) {
  f16_t h, l;
  h2_split(x, h, l);
  char* g = (char*)row + (long)(c >> 3) * 32 + (c & 7) * 2;
  *(f16_t*)g = h;
  *(f16_t*)(g + 16) = l;
}
__device__ __forceinline__ float h2_load1(const h2_t* row, int c) {
  const char* g = (const char*)row + (long)(c >> 3) * 32 + (c & 7) * 2;
  return (float)*(const f16_t*)g + (float)*(const f16_t*)(g + 16);
}

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_raw>(const bf16_raw* p) { return bf16_to_f32(*p); }

template <typename T> __device__ __forceinline__ void store_from_f32(T* p, float v);
template <> __device__ __forceinline__ void store_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_from_f32<bf16_raw>(bf16_raw* p, float v) { *p = f32_to_bf16(v); }
// one h2 element by its slot address (rows start on 32-byte boundaries: the group and the position inside it follow
// from the address alone)
template <> __device__ __forceinline__ void store_from_f32<h2_t>(h2_t* p, float v) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  char* g = reinterpret_cast<char*>(a & ~(uintptr_t)31) + ((a >> 2) & 7) * 2;
  f16_t h, l;
  h2_split(v, h, l);
  *(f16_t*)g = h;
  *(f16_t*)(g + 16) = l;
}

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

// XCD partition of a GEMM's tile grid: pm x pn = 8 rectangles, one per XCD (= per 4 MiB L2; blocks b, b + 8, ... share an
// XCD).  Every row panel of A is fetched by the pn XCDs of its rectangle row and W by the pm XCDs of its rectangle column,
// so the split that moves the fewest bytes across the fabric minimises  pn·|A| + pm·|W|;  a W sub-panel that does not fit
// an L2 share beside the streaming A panels is fetched again by every round of its rectangle.  (Rounds 1-2 only asked for
// the W sub-panel to fit: with A >> W — fc2: 56 MB of hidden activations against 4.7 MB of weights — that read A two
// to four times: PMC FETCH_SIZE 3.3x the algorithmic bytes on the split-fp16 fc2.)
static inline void odic_xcd_partition(int tiles_m, int tiles_n, double a_bytes, double w_bytes, int slots_per_xcd,
                                      int* pm_out, int* pn_out) {
  double best = 1e300;
  int bpm = 0, bpn = 0;
  for (int pn = 1; pn <= 8; pn *= 2) {
    const int pm = 8 / pn;
    if (pn > tiles_n || pm > tiles_m) continue;
    const long rect = (long)((tiles_m + pm - 1) / pm) * ((tiles_n + pn - 1) / pn);
    const long rounds = (rect + slots_per_xcd - 1) / slots_per_xcd;
    const double wf = (w_bytes / pn > 2.5 * 1024 * 1024) ? (double)rounds : 1.0;
    const double cost = pn * a_bytes + pm * w_bytes * wf;
    if (cost < best) { best = cost; bpm = pm; bpn = pn; }
  }
  if (!bpm) {                                             // fewer tiles than XCDs along both axes: as many row parts as there are
    bpm = 8;
    while (bpm > tiles_m && bpm > 1) bpm /= 2;
    bpn = 8 / bpm;
  }
  *pm_out = bpm; *pn_out = bpn;
}

static inline int odic_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
