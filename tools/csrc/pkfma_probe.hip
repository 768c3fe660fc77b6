// Minimal hardware probe for the round-2 cross-attention incident (tools/xattn_ab.py located it on
// `v_pk_fma_f32 ... op_sel:[0,1,0]` with vector-memory loads in flight).  Every lane computes the same sum of products
// four ways — two packed forms whose source selection differs only in HOW the same operand dword is named, and a scalar
// form — while global loads that nothing waits for keep returning into other registers.  Any lane where the results
// differ is a hardware / code-generation fault, independent of this library's kernels.
//
//   form A   v_pk_fma_f32 acc, a, b,  acc op_sel:[0,1,0]        low = a.lo·b.HI + acc.lo   high = a.hi·b.HI + acc.hi
//   form B   v_pk_fma_f32 acc, a, b', acc op_sel_hi:[1,0,1]     b' = (b.hi, junk):  low = a.lo·b'.lo, high = a.hi·b'.lo
//   scalar   v_fma_f32 twice
// `mode` bit 0: keep 12 global loads in flight across the FMA block; bit 1: write the UNUSED dword b.lo with a VALU
// instruction right after form A (the instruction order of the failing kernel); bit 2: s_nop 7 between A and that write.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void pkfma_probe_kernel(const float* __restrict__ buf, unsigned mask, int iters,
                                                          unsigned* __restrict__ mism, float* __restrict__ sink) {
  // (buf holds mask + 1 floats, a power of two)
  const unsigned tid = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float x = buf[tid & mask] + 1.5f, y = buf[(tid * 7u + 3u) & mask] - 0.25f;
  f32x2 a = {x, x};
  f32x2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
  float accS = 0.f, s = 0.f;
  unsigned idx = (tid * 97u) & mask;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    float l[12];
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 12; ++i) l[i] = buf[(idx + (unsigned)i * 1000003u) & mask];      // far apart: HBM / L2 misses
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x2 b = {1.0e30f, y};                 // form A must never read b.lo
      f32x2 bsw = {y, -1.0e30f};              // form B must never read bsw.hi
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(accA) : "v"(a), "v"(b));
      if (MODE & 4) asm volatile("s_nop 7");
      if (MODE & 2) asm volatile("v_add_u32_e32 %0, 12345, %0" : "+v"(b.x));           // VALU write of the unused dword
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accB) : "v"(a), "v"(bsw));
      accS = __builtin_fmaf(x, y, accS);
      asm volatile("" : "+v"(b.x), "+v"(accS));
      y = y * 1.0009765625f;                  // exact: keeps the operands changing without rounding
    }
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s += l[i];
    }
    idx = (idx + 7919u * 64u) & mask;
    const bool ok = __float_as_uint(accA.x) == __float_as_uint(accS) && __float_as_uint(accA.y) == __float_as_uint(accS) &&
                    __float_as_uint(accB.x) == __float_as_uint(accS) && __float_as_uint(accB.y) == __float_as_uint(accS);
    if (!ok) {
      bad |= (__float_as_uint(accA.x) != __float_as_uint(accS) ? 1u : 0u) | (__float_as_uint(accA.y) != __float_as_uint(accS) ? 2u : 0u) |
             (__float_as_uint(accB.x) != __float_as_uint(accS) ? 4u : 0u) | (__float_as_uint(accB.y) != __float_as_uint(accS) ? 8u : 0u);
      accA = f32x2{accS, accS}; accB = accA;  // resynchronise so that one event is counted once
      atomicAdd(&mism[0], 1u);
      atomicAdd(&mism[1 + (lane >> 4)], 1u);  // which 16-lane quarter
    }
  }
  if (bad) atomicOr(&mism[8], bad);
  if (s == 123.456f) sink[0] = s;             // keeps the loads alive
}

// A neighbour that does NOTHING but hold registers: NV VGPRs (the clobber makes the kernel descriptor allocate them),
// optional LDS bytes, `iters` rounds of s_sleep.  If the probe misbehaves beside this, the cause is where the hardware
// places the probe's registers, not what the neighbour computes.
template <int NV>
__global__ __launch_bounds__(256) void hold_regs_kernel(int iters, int* out) {
  extern __shared__ char dyn[];
  int x = threadIdx.x;
  if constexpr (NV == 32) asm volatile("v_mov_b32 v31, %0" :: "v"(x) : "v31");
  if constexpr (NV == 48) asm volatile("v_mov_b32 v47, %0" :: "v"(x) : "v47");
  if constexpr (NV == 64) asm volatile("v_mov_b32 v63, %0" :: "v"(x) : "v63");
  if constexpr (NV == 96) asm volatile("v_mov_b32 v95, %0" :: "v"(x) : "v95");
  if constexpr (NV == 128) asm volatile("v_mov_b32 v127, %0" :: "v"(x) : "v127");
  if constexpr (NV == 160) asm volatile("v_mov_b32 v159, %0" :: "v"(x) : "v159");
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(100);
  if (x == 12345678) { out[0] = x; dyn[0] = 1; }
}

extern "C" int hold_regs(int nv, int iters, int blocks, int lds_bytes, int* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define GO(M) case M: hipLaunchKernelGGL(hold_regs_kernel<M>, dim3(blocks), dim3(256), lds_bytes, st, iters, out); break;
  switch (nv) { GO(32) GO(48) GO(64) GO(96) GO(128) GO(160) default: return -1; }
#undef GO
  return (int)hipGetLastError();
}

extern "C" int pkfma_probe(int mode, const float* buf, int64_t n, int iters, int blocks, unsigned* mism, float* sink,
                           void* stream) {
  if (n <= 0 || (n & (n - 1))) return -2;      // power of two
  const unsigned mask = (unsigned)(n - 1);
  hipStream_t st = (hipStream_t)stream;
#define GO(M) case M: hipLaunchKernelGGL(pkfma_probe_kernel<M>, dim3(blocks), dim3(256), 0, st, buf, mask, iters, mism, sink); break;
  switch (mode) { GO(0) GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) default: return -1; }
#undef GO
  return (int)hipGetLastError();
}
