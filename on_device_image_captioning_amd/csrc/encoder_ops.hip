// Static-expansion (encoder) glue kernels for gfx950 — layers.py:45-102.
//
// The five contractions of the block run through odic_gemm (all as NT products, see
// engine.py::ExpansionEncoder); what is left are the relu / mask / L1-normalise steps on the
// score matrix z = Q·Kᵀ/sqrt(d)  [B, nq, S] and the selector mix.  All of it is small, fp32 and
// bandwidth/latency-bound, so these are plain coalesced VALU kernels:
//
//   stcexp_fw_kernel      one wave per (b, q) row: relu(±z) · key-valid, divide by (row sum + eps)
//   stcexp_colsum_kernel  per (b, group, s): Σ_q∈group relu(±z[b,q,s])   (lanes walk s → coalesced)
//   stcexp_bw_kernel      32x32 LDS-tile transpose: out[b,s,q] = relu(±z[b,q,s]) / (colsum + eps) / G
//   selector_mix_kernel   out = x + σ(sel)·a + (1-σ(sel))·b
#include "odic_common.h"

namespace {

// fw: one wave per (b, q) row.  Output rows have leading dimension ld (>= S); columns S..ld-1 are
// written as zeros so the buffers can feed a K-padded GEMM directly.
template <typename OutT>
__global__ __launch_bounds__(256) void stcexp_fw_kernel(const float* __restrict__ z,
                                                        const int* __restrict__ enc_len,
                                                        OutT* __restrict__ pos_fw, OutT* __restrict__ neg_fw,
                                                        int B, int nq, int S, int ld, float eps, float osc) {
  ODIC_ENCODE_PRIO();
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * nq) return;
  const int b = row / nq;
  const int len = enc_len[b];
  const float* zr = z + row * S;
  float sp = 0.f, sn = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float v = s < len ? zr[s] : 0.f;
    sp += fmaxf(v, 0.f);
    sn += fmaxf(-v, 0.f);
  }
  sp = wave_sum(sp); sn = wave_sum(sn);
  const float ip = 1.0f / (sp + eps), in_ = 1.0f / (sn + eps);
  for (int s = lane; s < ld; s += 64) {
    const float v = (s < S && s < len) ? zr[s] : 0.f;
    store_from_f32<OutT>(pos_fw + row * ld + s, fmaxf(v, 0.f) * ip * osc);     // (osc = 1: exactly the reference value)
    store_from_f32<OutT>(neg_fw + row * ld + s, fmaxf(-v, 0.f) * in_ * osc);
  }
}

// grid (ceil(S/64), ngroups, B), block (64, 16): 16 slices of the group's query range (the 512-query group is 32 rows
// per thread, requested eight at a time: the loop is a chain of L2 round trips, not bandwidth), LDS reduce
constexpr int CS_SL = 16;
__global__ __launch_bounds__(64 * CS_SL) void stcexp_colsum_kernel(const float* __restrict__ z,
                                                            const int* __restrict__ group_start,
                                                            float* __restrict__ colsum,   // [B, G, 2, S]
                                                            int nq, int S) {
  ODIC_ENCODE_PRIO();
  __shared__ float rp[CS_SL][64];
  __shared__ float rn[CS_SL][64];
  const int s = blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y, b = blockIdx.z, G = gridDim.y;
  const int q0 = group_start[g], q1 = group_start[g + 1];
  const float* zb = z + (long)b * nq * S;
  float sp = 0.f, sn = 0.f;
  if (s < S) {
    int q = q0 + threadIdx.y;
    for (; q + 7 * CS_SL < q1; q += 8 * CS_SL) {            // (fixed order of additions: deterministic)
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = zb[(long)(q + i * CS_SL) * S + s];
#pragma unroll
      for (int i = 0; i < 8; ++i) { sp += fmaxf(v[i], 0.f); sn += fmaxf(-v[i], 0.f); }
    }
    for (; q < q1; q += CS_SL) {
      const float v = zb[(long)q * S + s];
      sp += fmaxf(v, 0.f);
      sn += fmaxf(-v, 0.f);
    }
  }
  rp[threadIdx.y][threadIdx.x] = sp;
  rn[threadIdx.y][threadIdx.x] = sn;
  __syncthreads();
  if (threadIdx.y == 0 && s < S) {
    float tp = 0.f, tn = 0.f;
#pragma unroll
    for (int i = 0; i < CS_SL; ++i) { tp += rp[i][threadIdx.x]; tn += rn[i][threadIdx.x]; }
    colsum[(((long)b * G + g) * 2 + 0) * S + s] = tp;
    colsum[(((long)b * G + g) * 2 + 1) * S + s] = tn;
  }
}

// grid (ceil(S/32), ceil(ld/32), B), block (32, 8); output [B, S, ld] with zeroed columns nq..ld-1
template <typename OutT>
__global__ __launch_bounds__(256) void stcexp_bw_kernel(const float* __restrict__ z,
                                                        const float* __restrict__ colsum,
                                                        const int* __restrict__ group_of_q,
                                                        OutT* __restrict__ pos_bw, OutT* __restrict__ neg_bw,
                                                        int nq, int S, int ld, int G, float eps, float inv_g) {
  ODIC_ENCODE_PRIO();
  __shared__ float tp[32][33];
  __shared__ float tn[32][33];
  const int b = blockIdx.z;
  const int s0 = blockIdx.x * 32, q0 = blockIdx.y * 32;
  const float* zb = z + (long)b * nq * S;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int q = q0 + i, s = s0 + threadIdx.x;
    float vp = 0.f, vn = 0.f;
    if (q < nq && s < S) {
      const float v = zb[(long)q * S + s];
      const int g = group_of_q[q];
      const float cp = colsum[(((long)b * G + g) * 2 + 0) * S + s];
      const float cn = colsum[(((long)b * G + g) * 2 + 1) * S + s];
      vp = fmaxf(v, 0.f) / (cp + eps) * inv_g;
      vn = fmaxf(-v, 0.f) / (cn + eps) * inv_g;
    }
    tp[i][threadIdx.x] = vp;
    tn[i][threadIdx.x] = vn;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int s = s0 + i, q = q0 + threadIdx.x;
    if (s < S && q < ld) {
      store_from_f32<OutT>(pos_bw + ((long)b * S + s) * ld + q, tp[threadIdx.x][i]);
      store_from_f32<OutT>(neg_bw + ((long)b * S + s) * ld + q, tn[threadIdx.x][i]);
    }
  }
}

__global__ __launch_bounds__(256) void selector_mix_kernel(const float* __restrict__ x, long ldx,
                                                           const float* __restrict__ sel, long lds_,
                                                           const float* __restrict__ a, long lda,
                                                           const float* __restrict__ b, long ldb,
                                                           float* __restrict__ out, long ldo, int M, int d) {
  ODIC_ENCODE_PRIO();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * d) return;
  const int r = i / d, c = i - (long)r * d;
  const float sg = 1.0f / (1.0f + expf(-sel[r * lds_ + c]));
  out[r * ldo + c] = x[r * ldx + c] + sg * a[r * lda + c] + (1.0f - sg) * b[r * ldb + c];
}

}  // namespace

template <typename OutT>
static int stcexp_launch(const float* z, const int32_t* enc_len, const int32_t* group_meta, int32_t ngroups,
                         void* pos_fw, void* neg_fw, int64_t ld_fw, void* pos_bw, void* neg_bw, int64_t ld_bw,
                         float* colsum_ws, int32_t B, int32_t nq, int32_t S, float eps, float scale_fw, float scale_bw,
                         hipStream_t s) {
  const long rows = (long)B * nq;
  hipLaunchKernelGGL(stcexp_fw_kernel<OutT>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, z, enc_len,
                     (OutT*)pos_fw, (OutT*)neg_fw, B, nq, S, (int)ld_fw, eps, scale_fw);
  hipLaunchKernelGGL(stcexp_colsum_kernel, dim3((S + 63) / 64, ngroups, B), dim3(64, CS_SL), 0, s, z, group_meta,
                     colsum_ws, nq, S);
  hipLaunchKernelGGL(stcexp_bw_kernel<OutT>, dim3((S + 31) / 32, (unsigned)((ld_bw + 31) / 32), B), dim3(32, 8), 0, s,
                     z, colsum_ws, group_meta + ngroups + 1, (OutT*)pos_bw, (OutT*)neg_bw, nq, S, (int)ld_bw, ngroups,
                     eps, scale_bw / (float)ngroups);
  return odic_launch_status();
}

extern "C" int odic_stcexp_normalize(const float* z, const int32_t* enc_len, const int32_t* group_meta,
                                     int32_t ngroups, void* pos_fw, void* neg_fw, int64_t ld_fw, void* pos_bw,
                                     void* neg_bw, int64_t ld_bw, float* colsum_ws, int32_t B, int32_t nq, int32_t S,
                                     float eps, float scale_fw, float scale_bw, int32_t out_dtype, void* stream) {
  if (!z || !enc_len || !group_meta || !pos_fw || !neg_fw || !pos_bw || !neg_bw || !colsum_ws) return ODIC_ENULL;
  if (B <= 0 || nq <= 0 || S <= 0 || ngroups <= 0 || B > 65535 || ld_fw < S || ld_bw < nq) return ODIC_EINVAL;
  if (!(scale_fw > 0.f) || !(scale_bw > 0.f)) return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == ODIC_F32)
    return stcexp_launch<float>(z, enc_len, group_meta, ngroups, pos_fw, neg_fw, ld_fw, pos_bw, neg_bw, ld_bw,
                                colsum_ws, B, nq, S, eps, scale_fw, scale_bw, s);
  if (out_dtype == ODIC_BF16)
    return stcexp_launch<bf16_raw>(z, enc_len, group_meta, ngroups, pos_fw, neg_fw, ld_fw, pos_bw, neg_bw, ld_bw,
                                   colsum_ws, B, nq, S, eps, scale_fw, scale_bw, s);
  if (out_dtype == ODIC_H2) {             // split-fp16 operands of the x3 GEMM: whole [8 hi | 8 lo] groups per row
    if ((ld_fw & 7) || (ld_bw & 7) || ((uintptr_t)pos_fw & 31) || ((uintptr_t)neg_fw & 31) || ((uintptr_t)pos_bw & 31) ||
        ((uintptr_t)neg_bw & 31))
      return ODIC_EINVAL;
    return stcexp_launch<h2_t>(z, enc_len, group_meta, ngroups, pos_fw, neg_fw, ld_fw, pos_bw, neg_bw, ld_bw,
                               colsum_ws, B, nq, S, eps, scale_fw, scale_bw, s);
  }
  return ODIC_EINVAL;
}

extern "C" int odic_selector_mix(const float* x, int64_t ldx, const float* sel_pre, int64_t lds_, const float* a,
                                 int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int32_t M,
                                 int32_t d, void* stream) {
  if (!x || !sel_pre || !a || !b || !out) return ODIC_ENULL;
  if (M <= 0 || d <= 0) return ODIC_EINVAL;
  const long n = (long)M * d;
  hipLaunchKernelGGL(selector_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, sel_pre, lds_, a, lda, b, ldb, out, ldo, M, d);
  return odic_launch_status();
}
