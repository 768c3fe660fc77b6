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
#ifndef ODIC_SKINNY_MAXW
#define ODIC_SKINNY_MAXW 16         // K-slices (= waves) per block of the one-row-tile-per-block decomposition
#endif
#ifndef ODIC_SKINNY_LDS_KIB
#define ODIC_SKINNY_LDS_KIB 12      // stay below the ~16 KiB a CU has left beside the encode stream's GEMM blocks (sweep: 6→1420, 9→1455, 15→1448, 48→1430 captions/s)
#endif

struct Params {
  const float* A; const float* W; const float* bias; const float* residual; void* out;
  int M, N, K;
  long lda, ldw, ldr, ldc;
  long strideA, strideW, strideBias, strideR, strideC;
  float alpha; int act; int bias_axis; int vec_ok;
  const float* ln_colsum; float ln_eps;
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
// These products are weight-streaming and latency-bound, so the grid is cut for parallelism and
// every wave is kept SHORT (the step chain runs beside the encoder's GEMM blocks: a long-lived wave
// delays a whole GEMM tile):
//   * one block per 16 output columns, covering ALL M rows (MT = ceil(M/16) MFMA row tiles);
//   * waves = KS x MS: wave (ks, ms) owns the K-slice ks and MTW of the row tiles; it streams its
//     slice of the 16 W rows and of its A rows straight from global memory into MFMA operands (no
//     LDS round trip: W is read once per block, A is L2-resident) as float4 loads = 4 MFMA K-steps
//     per load, a whole round (UN x 16 of K) issued at once and the next round prefetched before
//     the MFMAs of the current one;
//   * partial accumulators are summed through LDS (KS x MT KiB), then the usual epilogue.
// K-order inside an MFMA is permuted (slot (kq, s) <-> k = k0 + 4·kq + s) identically on both
// operands, which only changes the summation order.
//
// Folded LayerNorm (FOLD): with W' = W·diag(gamma), colsum[n] = Σ_k W'[n][k] and
// bias' = bias + W·beta prepared by the caller,
//     LayerNorm(a)·Wᵀ + bias = rstd·(a·W'ᵀ − mean·colsum) + bias'
// so the product runs on the RAW rows and only the epilogue needs the row moments — which every
// wave accumulates for free from the A fragments it loads anyway (Σa, Σa² over its K-slice).
// ---------------------------------------------------------------------------------------------
//
// Decomposition (launch_skinny): grid = (N/16 column tiles) x (row blocks); a block owns MT row tiles starting at
// tile blockIdx.y·MT.  Three shapes of it are built (odic_gemm_args.tile_cfg 0 / 1 / 2, -1 = by block count):
//   0  all rows in one block, waves = K-slices x row tiles (<= 16 waves, K-slices of >= 64)           [round 1]
//   1  one row tile per block (grid.y = MT): waves = min(16, K/64) K-slices, each wave ONE round trip of loads at
//      K = 512 and two at K = 2048 — the fp32 MFMA work of a launch (157 TFLOP/s chip-wide = 0.6 per CU) spreads
//      over 3x the CUs and the per-wave chain of dependent load rounds shrinks 3x (48x512x2048: 24 -> 11 us);
//   2  three row tiles per WAVE (W fragments loaded once per block, 8 K-slices, 512-thread blocks that fit four to a
//      CU): for the wide products (vocabulary, 625 column tiles) where shape 1 would be thousands of blocks.
// ONE: the K-slice of a wave is a single round of loads (K/KS <= 16·UN) — no second register buffer, which is
// what lets two or three of the 512-thread blocks of decomposition 2 share a CU.
template <int MTW, int UN, bool FOLD, typename OutT, int MAXT, bool ONE>
__global__ __launch_bounds__(MAXT) void gemm_f32_skinny_kernel(Params p, int kslice, int KS, int MT) {
  extern __shared__ float red[];                   // [KS][MT][4][64] partial tiles (+ [KS][MT*16][2] row sums if FOLD)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave % KS, ms = wave / KS;
  const int n0 = blockIdx.x * 16;
  const long bz = blockIdx.z;
  const float* A = p.A + bz * p.strideA;
  const float* W = p.W + bz * p.strideW;
  const int fr = lane & 15, fq = lane >> 4;
  const int k_begin = ks * kslice, k_end = min(p.K, k_begin + kslice);
  const int t0 = ms * MTW;                          // first row tile of this wave (block-local)
  const int tb = blockIdx.y * MT;                   // first row tile of this block

  f32x4_t acc[MTW];
  float sx[MTW], sxx[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) { acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; sx[i] = 0.f; sxx[i] = 0.f; }

  const bool wn_ok = (n0 + fr) < p.N;
  const float* wrow = W + (long)min(n0 + fr, p.N - 1) * p.ldw + 4 * fq;
  const float* arow[MTW];
  bool am_ok[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int m = (tb + t0 + i) * 16 + fr;
    am_ok[i] = m < p.M;
    arow[i] = A + (long)min(m, p.M - 1) * p.lda + 4 * fq;
  }
  constexpr int NBUF = ONE ? 1 : 2;
  float4 w4[NBUF][UN], a4[NBUF][UN][MTW];
  auto load_round = [&](int buf, int k) {            // addresses clamped, values masked in compute()
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kk = min(k + 16 * u, p.K - 16);
      w4[buf][u] = *(const float4*)(wrow + kk);
#pragma unroll
      for (int i = 0; i < MTW; ++i) a4[buf][u][i] = *(const float4*)(arow[i] + kk);
    }
  };
  auto compute = [&](int buf, int k) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool live = k + 16 * u < k_end;
      float4 w = w4[buf][u];
      if (!live || !wn_ok) w = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        float4 a = a4[buf][u][i];
        if (!live || !am_ok[i]) a = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FOLD) {
          sx[i] += (a.x + a.y) + (a.z + a.w);
          sxx[i] += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[i], 0, 0, 0);
      }
    }
  };
  constexpr int STEP = 16 * UN;
  if constexpr (ONE) {
    if (t0 < MT && k_begin < k_end) { load_round(0, k_begin); compute(0, k_begin); }
  } else if (t0 < MT && k_begin < k_end) {
    load_round(0, k_begin);
    for (int k = k_begin; k < k_end; k += 2 * STEP) {
      if (k + STEP < k_end) load_round(NBUF - 1, k + STEP);
      compute(0, k);
      if (k + STEP < k_end) {
        if (k + 2 * STEP < k_end) load_round(0, k + 2 * STEP);
        compute(NBUF - 1, k + STEP);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    if (t0 + i < MT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((ks * MT + t0 + i) * 4 + j) * 64 + lane] = acc[i][j];
    }
  }
  float* rsum = red + KS * MT * 256;               // [KS][MT*16][2]
  if constexpr (FOLD) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      float a = sx[i], b = sxx[i];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (fq == 0 && t0 + i < MT) {
        rsum[((ks * MT + t0 + i) * 16 + fr) * 2 + 0] = a;
        rsum[((ks * MT + t0 + i) * 16 + fr) * 2 + 1] = b;
      }
    }
  }
  __syncthreads();

  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  OutT* out = (OutT*)p.out + bz * p.strideC;
  if constexpr (FOLD) {                              // row moments: combine the K-slices once per row, not per element
    const float inv_k = 1.0f / (float)p.K;
    float mean = 0.f, rstd = 0.f;
    if (tid < MT * 16) {
      float s1 = 0.f, s2 = 0.f;
      for (int w = 0; w < KS; ++w) { s1 += rsum[(w * MT * 16 + tid) * 2]; s2 += rsum[(w * MT * 16 + tid) * 2 + 1]; }
      mean = s1 * inv_k;
      rstd = rsqrtf(fmaxf(s2 * inv_k - mean * mean, 0.f) + p.ln_eps);
    }
    __syncthreads();
    if (tid < MT * 16) { rsum[tid * 2] = mean; rsum[tid * 2 + 1] = rstd; }
    __syncthreads();
  }
  for (int e = tid; e < MT * 256; e += blockDim.x) {
    const int i = e >> 8, j = (e >> 6) & 3, l = e & 63;
    float v = 0.f;
    for (int w = 0; w < KS; ++w) v += red[((w * MT + i) * 4 + j) * 64 + l];
    const int r16 = (l >> 4) * 4 + j;
    const int row = (tb + i) * 16 + r16, col = n0 + (l & 15);
    if (row < p.M && col < p.N) {
      v *= p.alpha;
      if constexpr (FOLD) {
        const float mean = rsum[(i * 16 + r16) * 2], rstd = rsum[(i * 16 + r16) * 2 + 1];
        v = (v - mean * p.ln_colsum[col]) * rstd;
      }
      if (bias) v += p.bias_axis ? bias[row] : bias[col];
      v = apply_act<false>(v, p.act);
      if (resid) v += resid[(long)row * p.ldr + col];
      store_from_f32<OutT>(out + (long)row * p.ldc + col, v);
    }
  }
}

template <int MTW, int UN, int MAXW>
static void launch_skinny(const Params& p, int out_dtype, int batch, hipStream_t stream, int row_blocks, int lds_kib) {
  const int MT_all = (p.M + 15) / 16;
  const int MT = (MT_all + row_blocks - 1) / row_blocks;             // row tiles per block
  const int MS = (MT + MTW - 1) / MTW;
  // K-slices of >= 64, KS·MS <= MAXW waves, KS·MT KiB of LDS for the partial tiles <= lds_kib
  int KS = (p.K + 63) / 64;
  if (KS > MAXW / MS) KS = MAXW / MS;
  while (KS > 1 && KS * MT > lds_kib) --KS;
  if (KS < 1) KS = 1;
  const int kslice = ((p.K + KS - 1) / KS + 15) / 16 * 16;
  dim3 grid((p.N + 15) / 16, (MT_all + MT - 1) / MT, batch), block(64 * KS * MS);
  const size_t shmem = (size_t)(KS * MT * 256 + KS * MT * 32) * sizeof(float);
  const bool fold = p.ln_colsum != nullptr;
  constexpr int MAXT = 64 * MAXW;
  const bool one = kslice <= 16 * UN;
#define ODIC_SKINNY_GO(FOLD_, OUT_, ONE_)                                                                          \
  hipLaunchKernelGGL((gemm_f32_skinny_kernel<MTW, UN, FOLD_, OUT_, MAXT, ONE_>), grid, block, shmem, stream, p, kslice, KS, MT)
  if (out_dtype == ODIC_BF16) {
    if (fold) { if (one) ODIC_SKINNY_GO(true, bf16_raw, true); else ODIC_SKINNY_GO(true, bf16_raw, false); }
    else { if (one) ODIC_SKINNY_GO(false, bf16_raw, true); else ODIC_SKINNY_GO(false, bf16_raw, false); }
  } else {
    if (fold) { if (one) ODIC_SKINNY_GO(true, float, true); else ODIC_SKINNY_GO(true, float, false); }
    else { if (one) ODIC_SKINNY_GO(false, float, true); else ODIC_SKINNY_GO(false, float, false); }
  }
#undef ODIC_SKINNY_GO
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
  p.ln_colsum = a->ln_colsum; p.ln_eps = a->ln_eps;
  p.vec_ok = (a->lda % 4 == 0) && (a->ldw % 4 == 0) && (a->strideA % 4 == 0) && (a->strideW % 4 == 0) &&
             (((uintptr_t)a->A & 15) == 0) && (((uintptr_t)a->W & 15) == 0);
  const bool want_ln = a->ln_colsum != nullptr;
  const bool skinny_ok = p.vec_ok && a->M <= 192 && a->K % 16 == 0;
  if (want_ln && !(skinny_ok && a->bias_axis == 0)) return ODIC_EUNSUPPORTED;
  if (skinny_ok && (a->N >= 64 || want_ln) && !(a->tile_cfg == 3 && !want_ln)) {     // (3: the 64x64 tile kernel)
    const int mt = (a->M + 15) / 16, ct = (a->N + 15) / 16;
    // The decomposition fixes the K-summation order (up to 16 K-slices in shape 1, up to 8 in shape 2), so it is
    // chosen from the product's (N, K) alone — never from M: the same logical product (a decoder weight) must sum in
    // the same order whether 48 rows (one batch) or 96 rows (a decode group of two) go through it, or a grouped
    // search would differ from the per-batch search at near-ties.  Wide products (the vocabulary: 625 column tiles)
    // take the three-row-tiles-per-wave shape, everything else one row tile per block.
    int shape = a->tile_cfg;
    if (shape < 0 || shape > 2) shape = ct <= 256 ? 1 : 2;
    if (shape == 1) {
      launch_skinny<1, 4, ODIC_SKINNY_MAXW>(p, a->out_dtype, a->batch, stream, mt, 16);
    } else if (shape == 2) {
      launch_skinny<3, 4, 8>(p, a->out_dtype, a->batch, stream, (mt + 2) / 3, 24);
    } else {
      // (MTW, UN): two rounds of UN·(1 + MTW) float4 per lane are in flight — 64 / 48 / 64 registers
      if (mt <= 4) launch_skinny<1, 4, 16>(p, a->out_dtype, a->batch, stream, 1, ODIC_SKINNY_LDS_KIB);   // one row tile per wave
      else if (mt <= 8) launch_skinny<2, 2, 16>(p, a->out_dtype, a->batch, stream, 1, ODIC_SKINNY_LDS_KIB);
      else launch_skinny<3, 2, 16>(p, a->out_dtype, a->batch, stream, 1, ODIC_SKINNY_LDS_KIB);
    }
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
