// bf16 NT GEMM with fused epilogue for gfx950:  out = act(alpha·A·Wᵀ + bias) + residual
//
//   A [M,K], W [N,K] bf16 row-major (K contiguous — the nn.Linear layout, so both MFMA operands are
//   read as 16-byte K-runs and nothing is ever transposed), fp32 accumulation.
//
// Structure (cdna_hip_programming.md §5, "minimum 2-phase" form):
//   * 128x128x64 block tile, 256 threads = 4 waves in a 2x2 grid, each wave a 64x64 patch made of
//     4x4 v_mfma_f32_16x16x32_bf16 tiles (16 accumulators x 4 VGPRs).
//   * global → LDS by global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction, no VGPR round
//     trip).  The LDS image is lane-linear, so the bank-conflict XOR swizzle (16-byte chunk index
//     ^ (row & 7) inside each 128-byte row) is applied to the per-lane SOURCE address and again on
//     the ds_read_b128 fragment reads (rule 21: both sides or neither).
//   * two LDS buffers: the DMA of K-tile t+1 is in flight while tile t feeds the MFMAs; one
//     s_waitcnt vmcnt(0) + barrier per K-tile.
//   * rows/cols beyond M/N are clamped on load (valid memory, discarded on store).
#include "odic_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand per buffer

struct Params {
  const bf16_raw* A; const bf16_raw* W; const float* bias; const float* residual; void* out;
  int M, N, K;
  long lda, ldw, ldr, ldc;
  long strideA, strideW, strideBias, strideR, strideC;
  float alpha; int act; int bias_axis;
  int tiles_m, tiles_n;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename OutT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];   // A0 | A1 | B0 | B1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
  // contiguous run of tiles; inside a run tiles walk N fastest so neighbours share the A panel.
  const int ntiles = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const long bz = blockIdx.z;
  const bf16_raw* A = p.A + bz * p.strideA;
  const bf16_raw* W = p.W + bz * p.strideW;

  // ---- LDS-DMA source addresses: instruction i of this wave fills rows (i*4+wave)*8 .. +7
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;         // logical 16-byte chunk this lane must fetch
  const bf16_raw* a_src[4];
  const bf16_raw* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + srow;
    const int ra = min(m0 + row, p.M - 1);
    const int rw = min(n0 + row, p.N - 1);
    a_src[i] = A + (long)ra * p.lda + schunk * 8;
    w_src[i] = W + (long)rw * p.ldw + schunk * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * TILE_BYTES;
    char* lw = lds + (2 + buf) * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (i * 4 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + (long)kt * BK), (lptr_t)(la + off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + (long)kt * BK), (lptr_t)(lw + off), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  const int frow = lane & 15, fq = lane >> 4;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);

    const char* la = lds + cur * TILE_BYTES;
    const char* lw = lds + (2 + cur) * TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[4], wf[4];
      const int chunk = ((kk * 4 + fq) ^ (frow & 7)) << 4;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        af[mi] = *(const bf16x8_t*)(la + (wm * 64 + mi * 16 + frow) * 128 + chunk);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        wf[ni] = *(const bf16x8_t*)(lw + (wn * 64 + ni * 16 + frow) * 128 + chunk);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], wf[ni], acc[mi][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 16x16 MFMA is col = lane&15, row = (lane>>4)*4 + reg
  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  OutT* out = (OutT*)p.out + bz * p.strideC;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = m0 + wm * 64 + mi * 16 + fq * 4 + j;
      if (row >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = n0 + wn * 64 + ni * 16 + frow;
        if (col >= p.N) continue;
        float v = acc[mi][ni][j] * p.alpha;
        if (bias) v += p.bias_axis ? bias[row] : bias[col];
        v = apply_act<true>(v, p.act);
        if (resid) v += resid[(long)row * p.ldr + col];
        store_from_f32<OutT>(out + (long)row * p.ldc + col, v);
      }
    }
  }
}

}  // namespace

int odic_gemm_bf16_launch(const odic_gemm_args* a, hipStream_t stream) {
  if (a->K % BK != 0 || a->lda % 8 != 0 || a->ldw % 8 != 0) return ODIC_EINVAL;
  if (((uintptr_t)a->A & 15) || ((uintptr_t)a->W & 15)) return ODIC_EINVAL;
  if ((a->strideA % 8) || (a->strideW % 8)) return ODIC_EINVAL;
  Params p;
  p.A = (const bf16_raw*)a->A; p.W = (const bf16_raw*)a->W; p.bias = a->bias; p.residual = a->residual;
  p.out = a->out; p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldr = a->ldr; p.ldc = a->ldc;
  p.strideA = a->strideA; p.strideW = a->strideW; p.strideBias = a->strideBias;
  p.strideR = a->strideR; p.strideC = a->strideC;
  p.alpha = a->alpha; p.act = a->act; p.bias_axis = a->bias_axis;
  p.tiles_m = (a->M + BM - 1) / BM; p.tiles_n = (a->N + BN - 1) / BN;
  dim3 grid(p.tiles_m * p.tiles_n, 1, a->batch);
  if (a->out_dtype == ODIC_BF16)
    hipLaunchKernelGGL(gemm_bf16_nt_kernel<bf16_raw>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(gemm_bf16_nt_kernel<float>, grid, dim3(256), 0, stream, p);
  return odic_launch_status();
}
