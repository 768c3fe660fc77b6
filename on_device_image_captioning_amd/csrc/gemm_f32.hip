// fp32 NT GEMM with fused epilogue for gfx950:  out = act(alpha·A·Wᵀ + bias) + residual
//
// Exact-fp32 path (parity mode, expansion encoder, decoder): v_mfma_f32_16x16x4_f32 is bit-for-bit
// a K-ordered fp32 fma chain (no TF32-like truncation exists on gfx950), so results differ from a
// CPU sgemm only by summation order.
//
//   * 64x64x16 block tile, 256 threads = 4 waves (2x2), each wave 32x32 = 2x2 MFMA tiles.
//   * register-staged double buffering: the loads of K-tile t+1 are issued before the MFMAs of
//     tile t and written to the other LDS buffer after them (cdna_hip_programming.md T14).
//   * LDS rows padded to 17 floats: the fragment read A[row=lane&15][k=lane>>4] is conflict-free.
//   * any M, N, K: out-of-range rows/cols/k are zero-filled; 16-byte loads when lda, ldw, K are
//     multiples of 4 and the bases are aligned, scalar loads otherwise.
#include "odic_common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDP = BK + 1;
#ifndef ODIC_SKINNY_LDS_KIB
#define ODIC_SKINNY_LDS_KIB 12      // stay below the ~16 KiB a CU has left beside the encode stream's GEMM blocks (sweep: 6→1420, 9→1455, 15→1448, 48→1430 captions/s)
#endif

struct Params {
  const float* A; const float* W; const float* bias; const float* residual; void* out;
  int M, N, K;
  long lda, ldw, ldr, ldc;
  long strideA, strideW, strideBias, strideR, strideC;
  float alpha; int act; int bias_axis; int vec_ok;
  const float* ln_gamma; const float* ln_beta; float ln_eps;
};

__device__ __forceinline__ float4 load4(const float* base, long ld, int row, int nrows, int k, int K,
                                        bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < nrows) {
    const float* p = base + (long)row * ld + k;
    if (vec && k + 3 < K) {
      v = *(const float4*)p;
    } else {
      if (k + 0 < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
      if (k + 3 < K) v.w = p[3];
    }
  }
  return v;
}

template <typename OutT>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(Params p) {
  __shared__ float As[2][BM][LDP];
  __shared__ float Ws[2][BN][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const long bz = blockIdx.z;
  const float* A = p.A + bz * p.strideA;
  const float* W = p.W + bz * p.strideW;
  const bool vec = p.vec_ok != 0;

  const int lrow = tid >> 2, lk = (tid & 3) * 4;      // this thread stages row lrow, k lk..lk+3
  const int nk = (p.K + BK - 1) / BK;

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  float4 ra = load4(A, p.lda, m0 + lrow, p.M, lk, p.K, vec);
  float4 rw = load4(W, p.ldw, n0 + lrow, p.N, lk, p.K, vec);
  As[0][lrow][lk + 0] = ra.x; As[0][lrow][lk + 1] = ra.y; As[0][lrow][lk + 2] = ra.z; As[0][lrow][lk + 3] = ra.w;
  Ws[0][lrow][lk + 0] = rw.x; Ws[0][lrow][lk + 1] = rw.y; Ws[0][lrow][lk + 2] = rw.z; Ws[0][lrow][lk + 3] = rw.w;
  __syncthreads();

  const int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      ra = load4(A, p.lda, m0 + lrow, p.M, (kt + 1) * BK + lk, p.K, vec);
      rw = load4(W, p.ldw, n0 + lrow, p.N, (kt + 1) * BK + lk, p.K, vec);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[2], w[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = As[cur][wm * 32 + mi * 16 + frow][ks * 4 + fq];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) w[ni] = Ws[cur][wn * 32 + ni * 16 + frow][ks * 4 + fq];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], w[ni], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      const int nx = cur ^ 1;
      As[nx][lrow][lk + 0] = ra.x; As[nx][lrow][lk + 1] = ra.y; As[nx][lrow][lk + 2] = ra.z; As[nx][lrow][lk + 3] = ra.w;
      Ws[nx][lrow][lk + 0] = rw.x; Ws[nx][lrow][lk + 1] = rw.y; Ws[nx][lrow][lk + 2] = rw.z; Ws[nx][lrow][lk + 3] = rw.w;
    }
    __syncthreads();
  }

  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  OutT* out = (OutT*)p.out + bz * p.strideC;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = m0 + wm * 32 + mi * 16 + fq * 4 + j;
      if (row >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 32 + ni * 16 + frow;
        if (col >= p.N) continue;
        float v = acc[mi][ni][j] * p.alpha;
        if (bias) v += p.bias_axis ? bias[row] : bias[col];
        v = apply_act<false>(v, p.act);
        if (resid) v += resid[(long)row * p.ldr + col];
        store_from_f32<OutT>(out + (long)row * p.ldc + col, v);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Skinny-M variant (decoder steps: M = images x beams = 48..192 rows against 0.25-20 MB weights).
// These products are weight-streaming and latency-bound, so the grid is cut for parallelism:
//   * one block per 16 output columns, every block covers ALL M rows (MT = ceil(M/16) MFMA tiles);
//   * the block's waves split K between them (in-block split-K), each wave streams its K-slice of
//     the 16 W rows straight from global memory into MFMA operands (no LDS round trip: W is read
//     once, A is L2-resident) as float4 loads = 4 MFMA K-steps per load;
//   * partial accumulators are summed through LDS, then the usual epilogue.
// K-order inside an MFMA is permuted (slot (kq, s) ↔ k = k0 + 4·kq + s) identically on both
// operands, which only changes the summation order.
// ---------------------------------------------------------------------------------------------
template <int MT, bool LN, typename OutT>
__global__ __launch_bounds__(1024) void gemm_f32_skinny_kernel(Params p, int kslice) {
  extern __shared__ float red[];                   // [nwaves][MT][4][64]  (+ [2][MT*16] row moments if LN)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int n0 = blockIdx.x * 16;
  const long bz = blockIdx.z;
  const float* A = p.A + bz * p.strideA;
  const float* W = p.W + bz * p.strideW;
  const int fr = lane & 15, fq = lane >> 4;
  const int k_begin = wave * kslice, k_end = min(p.K, k_begin + kslice);

  f32x4_t acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fused LayerNorm of A: every block recomputes the (tiny, L2-resident) row moments — wave w owns rows
  // w, w+nwaves, ...; two-pass mean / centred variance with the row held in registers.
  float* ln_mean = red + nwaves * MT * 256;
  float* ln_rstd = ln_mean + MT * 16;
  if constexpr (LN) {
    // one 16-lane group per row (blockDim/16 rows per pass), the row slice held in registers
    const int grp = tid >> 4, gl = tid & 15, ngrp = blockDim.x >> 4;
    for (int r0 = 0; r0 < p.M; r0 += ngrp) {
      const int r = r0 + grp;
      const float* xr = A + (long)min(r, p.M - 1) * p.lda;
      float4 v[16];                                  // K <= 1024: 16 lanes x 16 float4
      float sm = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int c = (gl + 16 * t) * 4;
        v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < p.K) { v[t] = *(const float4*)(xr + c); sm += (v[t].x + v[t].y) + (v[t].z + v[t].w); }
      }
      sm += __shfl_xor(sm, 8, 64); sm += __shfl_xor(sm, 4, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 1, 64);
      const float mean = sm / (float)p.K;
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int c = (gl + 16 * t) * 4;
        if (c < p.K) {
          const float a0 = v[t].x - mean, a1 = v[t].y - mean, a2 = v[t].z - mean, a3 = v[t].w - mean;
          q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
      }
      q += __shfl_xor(q, 8, 64); q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 1, 64);
      if (gl == 0 && r < p.M) { ln_mean[r] = mean; ln_rstd[r] = rsqrtf(q / (float)p.K + p.ln_eps); }
    }
    __syncthreads();
  }
  float mu[MT], rs[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    mu[i] = 0.f; rs[i] = 1.f;
    if constexpr (LN) { const int m = min(i * 16 + fr, p.M - 1); mu[i] = ln_mean[m]; rs[i] = ln_rstd[m]; }
  }

  const bool wn_ok = (n0 + fr) < p.N;
  const float* wrow = W + (long)min(n0 + fr, p.N - 1) * p.ldw + 4 * fq;
  const float* arow[MT];
  bool am_ok[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = i * 16 + fr;
    am_ok[i] = m < p.M;
    arow[i] = A + (long)min(m, p.M - 1) * p.lda + 4 * fq;
  }
  // All loads of a 64-deep K run are issued before the first MFMA (these kernels are latency-bound:
  // the more requests in flight per wave the better); guards are applied on the loaded values.
  constexpr int UN = (MT <= 4) ? 4 : 2;
  for (int k = k_begin; k < k_end; k += 16 * UN) {
    float4 w4[UN], a4[UN][MT];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kk = min(k + 16 * u, p.K - 16);          // clamped address, masked below
      w4[u] = *(const float4*)(wrow + kk);
#pragma unroll
      for (int i = 0; i < MT; ++i) a4[u][i] = *(const float4*)(arow[i] + kk);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (k + 16 * u >= k_end || !wn_ok) w4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 g4, b4;
      if constexpr (LN) {
        const int kk = min(k + 16 * u, p.K - 16) + 4 * fq;
        g4 = *(const float4*)(p.ln_gamma + kk); b4 = *(const float4*)(p.ln_beta + kk);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if constexpr (LN) {
          a4[u][i].x = (a4[u][i].x - mu[i]) * rs[i] * g4.x + b4.x; a4[u][i].y = (a4[u][i].y - mu[i]) * rs[i] * g4.y + b4.y;
          a4[u][i].z = (a4[u][i].z - mu[i]) * rs[i] * g4.z + b4.z; a4[u][i].w = (a4[u][i].w - mu[i]) * rs[i] * g4.w + b4.w;
        }
        if (!am_ok[i]) a4[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][i].x, w4[u].x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][i].y, w4[u].y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][i].z, w4[u].z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][i].w, w4[u].w, acc[i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[((wave * MT + i) * 4 + j) * 64 + lane] = acc[i][j];
  __syncthreads();

  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  OutT* out = (OutT*)p.out + bz * p.strideC;
  for (int e = tid; e < MT * 256; e += blockDim.x) {
    const int i = e >> 8, j = (e >> 6) & 3, l = e & 63;
    float v = 0.f;
    for (int w = 0; w < nwaves; ++w) v += red[((w * MT + i) * 4 + j) * 64 + l];
    const int row = i * 16 + (l >> 4) * 4 + j, col = n0 + (l & 15);
    if (row < p.M && col < p.N) {
      v *= p.alpha;
      if (bias) v += p.bias_axis ? bias[row] : bias[col];
      v = apply_act<false>(v, p.act);
      if (resid) v += resid[(long)row * p.ldr + col];
      store_from_f32<OutT>(out + (long)row * p.ldc + col, v);
    }
  }
}

template <int MT>
static void launch_skinny(const Params& p, int out_dtype, int batch, hipStream_t stream) {
  // waves: aim at K-slices of ~64, at most 16 waves, LDS for the reduction <= 48 KiB
  int nw = (p.K + 63) / 64;
  if (nw > 16) nw = 16;
  while (nw > 1 && nw * MT > ODIC_SKINNY_LDS_KIB) --nw;      // per-wave split-K slabs: nw·MT KiB of LDS
  if (nw < 1) nw = 1;
  int kslice = ((p.K + nw - 1) / nw + 15) / 16 * 16;
  dim3 grid((p.N + 15) / 16, 1, batch), block(64 * nw);
  const size_t shmem = (size_t)(nw * MT * 256 + 2 * MT * 16) * sizeof(float);
  const bool ln = p.ln_gamma != nullptr;
  if (out_dtype == ODIC_BF16) {
    if (ln) hipLaunchKernelGGL((gemm_f32_skinny_kernel<MT, true, bf16_raw>), grid, block, shmem, stream, p, kslice);
    else hipLaunchKernelGGL((gemm_f32_skinny_kernel<MT, false, bf16_raw>), grid, block, shmem, stream, p, kslice);
  } else {
    if (ln) hipLaunchKernelGGL((gemm_f32_skinny_kernel<MT, true, float>), grid, block, shmem, stream, p, kslice);
    else hipLaunchKernelGGL((gemm_f32_skinny_kernel<MT, false, float>), grid, block, shmem, stream, p, kslice);
  }
}

}  // namespace

int odic_gemm_f32_launch(const odic_gemm_args* a, hipStream_t stream) {
  Params p;
  p.A = (const float*)a->A; p.W = (const float*)a->W; p.bias = a->bias; p.residual = a->residual;
  p.out = a->out; p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldr = a->ldr; p.ldc = a->ldc;
  p.strideA = a->strideA; p.strideW = a->strideW; p.strideBias = a->strideBias;
  p.strideR = a->strideR; p.strideC = a->strideC;
  p.alpha = a->alpha; p.act = a->act; p.bias_axis = a->bias_axis;
  p.ln_gamma = a->ln_gamma; p.ln_beta = a->ln_beta; p.ln_eps = a->ln_eps;
  p.vec_ok = (a->lda % 4 == 0) && (a->ldw % 4 == 0) && (a->strideA % 4 == 0) && (a->strideW % 4 == 0) &&
             (((uintptr_t)a->A & 15) == 0) && (((uintptr_t)a->W & 15) == 0);
  const bool want_ln = a->ln_gamma != nullptr;
  if (want_ln && (!a->ln_beta || !(p.vec_ok && a->M <= 192 && a->K % 16 == 0 && a->K <= 1024 && a->batch == 1)))
    return ODIC_EUNSUPPORTED;
  if (p.vec_ok && a->M <= 192 && a->K % 16 == 0 && (a->N >= 64 || want_ln)) {
    const int mt = (a->M + 15) / 16;
    if (mt <= 1) launch_skinny<1>(p, a->out_dtype, a->batch, stream);
    else if (mt <= 2) launch_skinny<2>(p, a->out_dtype, a->batch, stream);
    else if (mt <= 3) launch_skinny<3>(p, a->out_dtype, a->batch, stream);
    else if (mt <= 4) launch_skinny<4>(p, a->out_dtype, a->batch, stream);
    else if (mt <= 6) launch_skinny<6>(p, a->out_dtype, a->batch, stream);
    else if (mt <= 9) launch_skinny<9>(p, a->out_dtype, a->batch, stream);
    else launch_skinny<12>(p, a->out_dtype, a->batch, stream);
    return odic_launch_status();
  }
  dim3 grid((a->N + BN - 1) / BN, (a->M + BM - 1) / BM, a->batch);
  if (grid.y > 65535) return ODIC_EINVAL;
  if (a->out_dtype == ODIC_BF16)
    hipLaunchKernelGGL(gemm_f32_nt_kernel<bf16_raw>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(gemm_f32_nt_kernel<float>, grid, dim3(256), 0, stream, p);
  return odic_launch_status();
}
