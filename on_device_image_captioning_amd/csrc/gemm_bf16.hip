// bf16 NT GEMM with fused epilogue for gfx950:  out = act(alpha·A·Wᵀ + bias) + residual
//
//   A [M,K], W [N,K] bf16 row-major (K contiguous — the nn.Linear layout, so both MFMA operands are
//   read as 16-byte K-runs and nothing is ever transposed), fp32 accumulation.
//
// Structure (cdna_hip_programming.md §5):
//   * block tile BM x BN x 64 with NWM x NWN waves, each wave a (16·MI) x (16·NI) patch of
//     v_mfma_f32_16x16x32_bf16 tiles.  Three instantiations:
//        256x256 (2x4 waves of 128x64)  arithmetic intensity 128 FLOP per LDS-filled byte — the
//                                       large stage-2/3 products (73 % of all Swin FLOPs)
//        128x128 (2x2 waves of 64x64)   small N / tail-friendly
//        128x64  (2x2 waves of 64x32)   N = 192 style panels
//   * global → LDS by global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction, no VGPR round
//     trip).  The LDS image is lane-linear, so the bank-conflict XOR swizzle (16-byte chunk index
//     ^ (row & 7) inside each 128-byte row) is applied to the per-lane SOURCE address and again on
//     the ds_read_b128 fragment reads (rule 21: both sides or neither).
//   * two LDS stages: the DMA of K-tile t+1 is in flight while tile t feeds the MFMAs; one
//     s_waitcnt vmcnt(0) + barrier per K-tile.
//   * epilogue through LDS: each wave parks 16 x TN fp32 accumulator rows in its own LDS slab and
//     reads them back row-contiguous, so bias / activation / fp32 residual / output all move as
//     16-byte (fp32) or 8-byte (bf16) per-lane vectors on full rows instead of 2-4-byte scatters.
//   * XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); each XCD gets a
//     contiguous run of tiles, N fastest, so concurrently resident tiles share A and W panels.
//   * rows/cols beyond M/N are clamped on load (valid memory, discarded on store).
#include "odic_common.h"

namespace {

constexpr int BK = 64;

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

template <int NWM, int NWN, int MI, int NI, typename OutT>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_bf16_nt_kernel(Params p) {
  constexpr int NW = NWM * NWN;
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16, TN = NI * 16;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, W_INSTR = BN / 8 / NW;      // 1-KiB DMA instructions per wave
  constexpr int EP_LD = TN + 4;                                      // fp32 words per staged row
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
  static_assert(NW * 16 * EP_LD * 4 <= 2 * STAGE, "epilogue slabs must fit in the pipeline LDS");
  extern __shared__ __attribute__((aligned(16))) char lds[];          // stage0 {A,W} | stage1 {A,W}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;

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

  // ---- LDS-DMA source addresses: instruction i of this wave fills rows (i*NW+wave)*8 .. +7
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;         // logical 16-byte chunk this lane must fetch
  const bf16_raw* a_src[A_INSTR];
  const bf16_raw* w_src[W_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + srow;
    a_src[i] = A + (long)min(m0 + row, p.M - 1) * p.lda + schunk * 8;
  }
#pragma unroll
  for (int i = 0; i < W_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + srow;
    w_src[i] = W + (long)min(n0 + row, p.N - 1) * p.ldw + schunk * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * STAGE;
    char* lw = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + (long)kt * BK), (lptr_t)(la + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + (long)kt * BK), (lptr_t)(lw + (i * NW + wave) * 1024), 16, 0, 0);
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  const int frow = lane & 15, fq = lane >> 4;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);

    const char* la = lds + cur * STAGE + (wm * MI * 16 + frow) * 128;
    const char* lw = lds + cur * STAGE + A_BYTES + (wn * NI * 16 + frow) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[MI], wf[NI];
      const int chunk = ((kk * 4 + fq) ^ (frow & 7)) << 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *(const bf16x8_t*)(la + mi * 16 * 128 + chunk);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8_t*)(lw + ni * 16 * 128 + chunk);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], wf[ni], acc[mi][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.
  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  OutT* out = (OutT*)p.out + bz * p.strideC;
  float* slab = (float*)lds + wave * 16 * EP_LD;          // private to this wave
  constexpr int VPR = TN / 4;                              // float4 vectors per staged row
  constexpr int RPP = 64 / VPR;                            // rows per read-back pass
  const int rcol = (lane % VPR) * 4, rrow = lane / VPR;
  const int col = n0 + wn * TN + rcol;
  const bool vec_ok = (col + 3 < p.N) && ((p.ldc & 3) == 0) && (!resid || (p.ldr & 3) == 0);
  float4 bcol = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias && !p.bias_axis) {
    bcol.x = col + 0 < p.N ? bias[col + 0] : 0.f; bcol.y = col + 1 < p.N ? bias[col + 1] : 0.f;
    bcol.z = col + 2 < p.N ? bias[col + 2] : 0.f; bcol.w = col + 3 < p.N ? bias[col + 3] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) slab[(fq * 4 + j) * EP_LD + ni * 16 + frow] = acc[mi][ni][j];
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 16 / RPP; ++pass) {
      const int r = pass * RPP + rrow;
      const int row = m0 + (wm * MI + mi) * 16 + r;
      if (row < p.M) {
        float4 v = *(const float4*)(slab + r * EP_LD + rcol);
        v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
        if (bias) {
          if (p.bias_axis) { const float b = bias[row]; v.x += b; v.y += b; v.z += b; v.w += b; }
          else { v.x += bcol.x; v.y += bcol.y; v.z += bcol.z; v.w += bcol.w; }
        }
        v.x = apply_act<true>(v.x, p.act); v.y = apply_act<true>(v.y, p.act);
        v.z = apply_act<true>(v.z, p.act); v.w = apply_act<true>(v.w, p.act);
        if (vec_ok) {
          if (resid) {
            const float4 rr = *(const float4*)(resid + (long)row * p.ldr + col);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          OutT* dst = out + (long)row * p.ldc + col;
          if constexpr (sizeof(OutT) == 4) {
            *(float4*)dst = v;
          } else {
            ushort4 pk;
            pk.x = f32_to_bf16(v.x); pk.y = f32_to_bf16(v.y); pk.z = f32_to_bf16(v.z); pk.w = f32_to_bf16(v.w);
            *(ushort4*)dst = pk;
          }
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col + e < p.N) {
              float x = vv[e];
              if (resid) x += resid[(long)row * p.ldr + col + e];
              store_from_f32<OutT>(out + (long)row * p.ldc + col + e, x);
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int NWM, int NWN, int MI, int NI>
int launch_cfg(Params& p, int out_dtype, int batch, hipStream_t stream) {
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16;
  constexpr int SHMEM = 2 * (BM + BN) * BK * 2;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch), block(64 * NWM * NWN);
  auto kb = gemm_bf16_nt_kernel<NWM, NWN, MI, NI, bf16_raw>;
  auto kf = gemm_bf16_nt_kernel<NWM, NWN, MI, NI, float>;
  if (SHMEM > 64 * 1024) {
    static bool done = false;       // idempotent; racing first calls set the same value
    if (!done) {
      (void)hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
      (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
      done = true;
    }
  }
  if (out_dtype == ODIC_BF16) hipLaunchKernelGGL(kb, grid, block, SHMEM, stream, p);
  else hipLaunchKernelGGL(kf, grid, block, SHMEM, stream, p);
  return odic_launch_status();
}

int g_force_cfg = -1;      // test / tuning hook: odic_gemm_bf16_force_config()

}  // namespace

extern "C" void odic_gemm_bf16_force_config(int cfg) { g_force_cfg = cfg; }

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
  // Tile choice = fewest "rounds x per-tile cost": a launch runs in ceil(tiles / resident slots)
  // rounds (256 CUs x 3 / 2 / 1 blocks for the 128x64 / 128x128 / 256x256 tiles, set by their LDS
  // footprints); relative per-tile costs 1 : 1.38 : 2.6 were measured on MI355X over the Swin-L
  // shapes (tools/gemm_tune.py; profiles/r01_gemm_tile_sweep.txt).
  int cfg = g_force_cfg;
  if (cfg < 0) {
    auto rounds = [&](int bm, int bn, int slots) {
      const long t = (long)((a->M + bm - 1) / bm) * ((a->N + bn - 1) / bn) * a->batch;
      return (double)((t + slots - 1) / slots);
    };
    const double c0 = rounds(128, 64, 768) * 1.0, c1 = rounds(128, 128, 512) * 1.38;
    const double c2 = (a->N % 256 == 0) ? rounds(256, 256, 256) * 2.6 : 1e30;
    cfg = (c0 <= c1 && c0 <= c2) ? 0 : (c1 <= c2 ? 1 : 2);
  }
  switch (cfg) {
    case 0: return launch_cfg<2, 2, 4, 2>(p, a->out_dtype, a->batch, stream);     // 128 x 64
    case 1: return launch_cfg<2, 2, 4, 4>(p, a->out_dtype, a->batch, stream);     // 128 x 128
    case 2: return launch_cfg<2, 4, 8, 4>(p, a->out_dtype, a->batch, stream);     // 256 x 256
    default: return ODIC_EINVAL;
  }
}
