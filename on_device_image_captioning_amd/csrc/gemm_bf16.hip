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
//   * operands are SWAPPED in the MFMA (A-slot = W fragment, B-slot = activation fragment): the
//     accumulator tile is then Cᵀ, i.e. each lane holds 4 CONSECUTIVE OUTPUT COLUMNS of one output
//     row, so bias / activation / fp32 residual / output move as 16-byte (fp32) or 8-byte (bf16)
//     per-lane vectors straight from the accumulator registers — no LDS staging, no shuffles.
//   * XCD-aware 2-D tile partition: blocks b, b+8, ... share an XCD (and its 4 MiB L2); each XCD
//     owns a PM x PN rectangle of the tile grid chosen so that its W sub-panel stays L2-resident
//     while its A panels stream through once (PMC: L2 hit rate 70 % → see profiles/).
//   * rows/cols beyond M/N are clamped on load (valid memory, discarded on store).
#include "odic_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {


struct Params {
  const bf16_raw* A; const bf16_raw* W; const float* bias; const float* residual; void* out;
  int M, N, K;
  long lda, ldw, ldr, ldc;
  long strideA, strideW, strideBias, strideR, strideC;
  float alpha; int act; int bias_axis;
  int tiles_m, tiles_n;
  int pm, pn;          // XCD partition of the tile grid, pm * pn == 8
  int* ws;             // persistent kernels: 8 per-partition tile counters + 1 exit counter (all zero at launch)
  // LayerNorm folded across two products (one-block-per-tile kernel only):
  //   producer (fp32 out): also writes the bf16 copy of its output rows and, per row and 32-column group, the
  //                        mean and centred sum of squares of those bf16 values;
  //   consumer: A is that bf16 copy; row moments are combined from the K/32 groups and the epilogue computes
  //             rstd·(A·W'ᵀ − mean·colsum) + bias' = LayerNorm(A)·Wᵀ + bias (W', colsum, bias' packed by the caller).
  bf16_raw* out16; long ld16; float* stats_out;
  const float* ln_stats; const float* ln_colsum; float ln_eps;
  int skew_from, skew_to, skew_sleeps;   // blocks [skew_from, skew_to) start skew_sleeps x 64·127 clocks late
  const float* a_ln; long ld_aln;        // A-resident kernels: fp32 rows whose LayerNorm (no affine) is the A operand
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 16-byte-chunk swizzle inside an LDS row.  BK = 64 (128-byte rows, 8 chunks): chunk ^ (row & 7).
// BK = 32 (64-byte rows, 4 chunks; 4 rows share a 256-byte bank row): chunk ^ f((row >> 2) & 3) with
// f = {0, 2, 3, 1}, which makes every 16-lane service group of ds_read_b128 hit 16 distinct slots.
template <int BK> __device__ __forceinline__ int swz(int chunk, int row) {
  if constexpr (BK == 64) return chunk ^ (row & 7);
  else return chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);
}

// LDS row r of the W tile holds tile-local output column wperm(r): inside each 32-row group the two
// 16-row MFMA tiles interleave in runs of 4, so that MFMA slot 4·fq + j of tile h is column
// 8·fq + 4·h + j and a lane's registers across the tile pair are 8 adjacent output columns.
__device__ __forceinline__ int wperm(int r) {
  return (r & ~31) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3);
}

// LNF: 0 plain, 1 producer of a folded LayerNorm (fp32 out + bf16 copy + row-group moments), 2 consumer
// (separate instantiations: the extra registers of the fold must not cost the plain products their occupancy)
// (second launch bound: the 4-wave blocks with 128 accumulator registers per lane must stay within 256 registers
//  so that two of them share a CU — left alone hipcc takes 158 + 128)
// KS = 2: TWO groups of NWM x NWN waves share the tile and its LDS stages; group g takes the 32-deep half g of every 64-deep
// K-tile (half the MFMAs each), group 1 hands its accumulators to group 0 through LDS after the loop (a fixed order of
// additions: deterministic, but not the summation order of the KS = 1 form).  For the tiles whose wave count does not fill
// four SIMDs evenly — 144 x 192 is six waves, the exact-round tile of the 9216 x 768 products — this gives every SIMD three
// waves.  Only group 0 stages (its vmcnt waits precede the barrier both groups meet at) and only group 0 stores.
template <int NWM, int NWN, int MI, int NI, int NSTAGE, int BK, typename OutT, int LNF = 0, int KS = 1>
__global__ __launch_bounds__(64 * NWM * NWN * KS, (NWM * NWN * KS == 4 && MI * NI == 32) ? 2 : 1) void gemm_bf16_nt_kernel(Params p) {
  static_assert(KS == 1 || (KS == 2 && BK == 64 && LNF == 0), "the K-split form is two groups on 64-deep K-tiles");
  constexpr int NW = NWM * NWN;
  constexpr int ROWB = BK * 2;                 // bytes per LDS row
  constexpr int RPI = 1024 / ROWB;             // rows per 1-KiB DMA instruction
  constexpr int CPR = ROWB / 16;               // 16-byte chunks per row
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / RPI / NW, W_INSTR = BN / RPI / NW;  // 1-KiB DMA instructions per wave
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile rows must split evenly over the waves");
  constexpr int G = A_INSTR + W_INSTR;                                 // LDS-DMA instructions per wave per K-tile
  static_assert(NI % 2 == 0, "the epilogue pairs MFMA column tiles");
  constexpr int D = NSTAGE - 1;                                        // prefetch distance in K-tiles
  extern __shared__ __attribute__((aligned(16))) char lds[];          // stage0 {A,W} | stage1 {A,W}

  const int tid = threadIdx.x, lane = tid & 63;
  const int kg = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((tid >> 6) / NW);      // K group of this wave
  const int wave = KS == 1 ? (tid >> 6) : __builtin_amdgcn_readfirstlane((tid >> 6) % NW);
  const int wm = wave / NWN, wn = wave % NWN;
  ODIC_ENCODE_PRIO();
  if ((int)blockIdx.x >= p.skew_from && (int)blockIdx.x < p.skew_to)
    for (int i = 0; i < p.skew_sleeps; ++i) __builtin_amdgcn_s_sleep(127);


  // XCD x = blockIdx % 8 owns tile rows [r0,r1) x cols [c0,c1); inside the rectangle tiles run N-fastest
  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int xm = xcd / p.pn, xn = xcd - xm * p.pn;
    const int r0 = xm * p.tiles_m / p.pm, r1 = (xm + 1) * p.tiles_m / p.pm;
    const int c0 = xn * p.tiles_n / p.pn, c1 = (xn + 1) * p.tiles_n / p.pn;
    const int w = c1 - c0;
    if (idx >= (r1 - r0) * w) return;             // rectangles differ by at most one row/col of tiles
    const int lr = idx / w;
    tm = r0 + lr; tn = c0 + (idx - lr * w);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const long bz = blockIdx.z;
  const bf16_raw* A = p.A + bz * p.strideA;
  const bf16_raw* W = p.W + bz * p.strideW;

  // ---- LDS-DMA source addresses: instruction i of this wave fills rows (i*NW+wave)*RPI .. +RPI-1
  const int srow = lane / CPR;
  const int schunk = swz<BK>(lane % CPR, srow);  // logical 16-byte chunk this lane must fetch
  const bf16_raw* a_src[A_INSTR];
  const bf16_raw* w_src[W_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * RPI + srow;
    a_src[i] = A + (long)min(m0 + row, p.M - 1) * p.lda + schunk * 8;
  }
#pragma unroll
  for (int i = 0; i < W_INSTR; ++i) {
    const int row = (i * NW + wave) * RPI + srow;
    w_src[i] = W + (long)min(n0 + wperm(row), p.N - 1) * p.ldw + schunk * 8;
  }

  auto stage = [&](int buf, int kt) {
    if (KS == 2 && kg != 0) return;
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

  // Pipeline: tiles kt+1 .. kt+D-1 stay in flight across the barrier (counted vmcnt: the wave's own
  // DMA groups retire in order, G instructions per tile; the raw s_barrier carries no implicit
  // vmcnt(0) drain).  After the barrier every wave's share of tile kt has landed AND every wave has
  // finished reading tile kt-1, whose buffer the prefetch of tile kt+D then overwrites.
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nk) stage(t, t);

  // folded LayerNorm, consumer side: (mean, rstd) of this tile's BM rows of A from the per-32-column-group
  // moments its producer left (Chan's combination of equal-sized groups) → LDS tail, read in the epilogue.
  // Four lanes per row, each with a quarter of the K/32 groups in registers (all loads issued before the first
  // use: a dependent load per group would cost more than the tile's whole K-loop).
  float2* s_stat = (float2*)(lds + NSTAGE * STAGE);
  if constexpr (LNF == 2) {
    const int ng = p.K >> 5;                                      // groups per row (<= 48); lane q takes q, q+4, ...
    for (int t = tid; t < 4 * BM; t += 64 * NW) {
      const int r = t >> 2, q = t & 3;
      const int row = min(m0 + r, p.M - 1);
      const float2* st = (const float2*)p.ln_stats + (long)row * ng + q;
      float2 gv[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) gv[i] = q + 4 * i < ng ? st[4 * i] : make_float2(0.f, 0.f);
      float msum = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) msum += gv[i].x;
      msum += __shfl_xor(msum, 1, 64);
      msum += __shfl_xor(msum, 2, 64);
      const float mean = msum / (float)ng;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float d = gv[i].x - mean;
        m2 += q + 4 * i < ng ? gv[i].y + 32.0f * d * d : 0.f;
      }
      m2 += __shfl_xor(m2, 1, 64);
      m2 += __shfl_xor(m2, 2, 64);
      if (q == 0) s_stat[r] = make_float2(mean, rsqrtf(m2 / (float)p.K + p.ln_eps));
    }
  }

  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(D - 1, nk - 1 - kt);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + D < nk) stage((kt + D) % NSTAGE, kt + D);

    const int cur = kt % NSTAGE;
    const char* la = lds + cur * STAGE + (wm * MI * 16 + frow) * ROWB;
    const char* lw = lds + cur * STAGE + A_BYTES + (wn * NI * 16 + frow) * ROWB;
#pragma unroll
    for (int kq = 0; kq < BK / 32 / KS; ++kq) {
      const int kk = KS == 1 ? kq : kg;
      bf16x8_t af[MI], wf[NI];
      const int chunk = swz<BK>(kk * 4 + fq, frow) << 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *(const bf16x8_t*)(la + mi * 16 * ROWB + chunk);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8_t*)(lw + ni * 16 * ROWB + chunk);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
  }
  if constexpr (KS == 2) {
    // group 1 → LDS → group 0 (the stages are free: every wave is past its last fragment read at the barrier)
    f32x4_t* red = (f32x4_t*)lds + (long)wave * MI * NI * 64 + lane;
    __builtin_amdgcn_s_barrier();
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) red[(i * NI + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] += red[(i * NI + j) * 64];
  }
  // ---- epilogue.  With the operands swapped the 16x16 accumulator is Cᵀ: lane (frow, fq) register j
  //      holds output row frow, W-slot 4·fq + j.  The W rows were staged permuted (wperm above), so
  //      the slot pair (2q, 2q+1) of a lane covers 8 CONSECUTIVE output columns 32q + 8·fq .. +7:
  //      one 16-byte bf16 store (or two adjacent float4) per lane, 64/128 contiguous bytes per row.
  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  OutT* out = (OutT*)p.out + bz * p.strideC;
  const bool ld_ok = ((p.ldc & 7) == 0) && (!resid || (p.ldr & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if constexpr (sizeof(OutT) == 2 && LNF == 0 && NI % 4 == 0) {
    // bf16 output in WHOLE 128-byte lines.  A lane's 8 columns are 16 bytes, the four fq lanes of a row 64 contiguous
    // bytes: half a line per row and store instruction, and half-line stores run at 4 TB/s where whole lines run at
    // 6.8 (tools/smallk_probe.py).  The line's other half is the same lanes' NEXT column group: the two groups are
    // exchanged between lanes frow and frow ^ 8 (one DPP row rotation per dword), so that the low eight lanes of each
    // 16 hold both rows' first halves and the high eight both second halves — 8 rows x 128 bytes per store.
    if (ld_ok && (p.N & 63) == 0) {
      typedef __attribute__((ext_vector_type(4))) int i32x4_t;
      const bool lo = frow < 8;
#pragma unroll
      for (int q2 = 0; q2 < NI / 4; ++q2) {
        const int cbase = n0 + wn * NI * 16 + q2 * 64;            // wave-uniform
        if (cbase >= p.N) continue;
        f32x4_t bc[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              bc[g][h][e] = (bias && !p.bias_axis) ? bias[cbase + g * 32 + fq * 8 + 4 * h + e] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int r16 = m0 + (wm * MI + mi) * 16;
          const float brow = (bias && p.bias_axis) ? bias[min(r16 + frow, p.M - 1)] : 0.f;
          i32x4_t own[2];
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            bf16x8_t pk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x4_t pre = acc[mi][4 * q2 + 2 * g + h] * p.alpha + bc[g][h] + brow;
              if (p.act == ODIC_ACT_GELU) {
                pre = gelu_poly4(pre);
              } else if (p.act != ODIC_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pre[e] = apply_act<true>(pre[e], p.act);
              }
              if (resid) {
                const f32x4_t rr = *(const f32x4_t*)(resid + (long)min(r16 + frow, p.M - 1) * p.ldr + cbase + g * 32 + fq * 8 + 4 * h);
                pre += rr;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) pk[4 * h + e] = (short)f32_to_bf16(pre[e]);
            }
            own[g] = __builtin_bit_cast(i32x4_t, pk);
          }
          const i32x4_t send = lo ? own[1] : own[0];
          i32x4_t recv;
#pragma unroll
          for (int e = 0; e < 4; ++e) recv[e] = __builtin_amdgcn_update_dpp(0, send[e], 0x128, 0xf, 0xf, false);   // row_ror:8
          const int ra = r16 + (frow & 7);
          bf16_raw* dst = (bf16_raw*)out + (long)ra * p.ldc + cbase + (lo ? 0 : 32) + fq * 8;
          if (ra < p.M) *(i32x4_t*)dst = lo ? own[0] : recv;
          if (ra + 8 < p.M) *(i32x4_t*)(dst + 8 * p.ldc) = lo ? recv : own[1];
        }
      }
      return;
    }
  }
  if constexpr (LNF == 0 && !(sizeof(OutT) == 2 && NI % 4 == 0)) {   // (bf16 with NI % 4 == 0: whole-line path above, or the plain loop below)
    // The general vector path.  vmcnt retires in order and counts stores, so ANY load between two groups of stores
    // waits for every store before it: a bias or residual load per column group turned the epilogue into a chain of
    // memory round trips (9 per 144 x 192 tile; the "+16-20 us for the fp32 residual form" of DESIGN.md §4.1).  Here
    // every bias value is requested before the first store, and the residual rows of column group g + 1 are requested
    // BEFORE the stores of group g (they are then older than those stores, and waiting for them waits for nothing else).
    if (ld_ok && (p.N & 7) == 0) {
      constexpr int NG = NI / 2;
      // residual prefetch depth: 2 = pipelined as above (4- and 6-wave blocks: the registers are there), 1 = a group's
      // rows requested together (12-wave blocks), 0 = row by row (8-wave blocks: 32 more registers would cost them
      // their second resident block)
      constexpr int RD = sizeof(OutT) == 4 ? (NW <= 6 ? 2 : (NW >= 12 ? 1 : 0)) : 0;
      const int cw = n0 + wn * NI * 16 + fq * 8;
      f32x4_t bc[NG][2];
      float brow[MI];
#pragma unroll
      for (int nq = 0; nq < NG; ++nq) bc[nq][0] = bc[nq][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) brow[mi] = 0.f;
      if (bias && !p.bias_axis) {                       // (uniform branches around whole groups of loads)
        if ((reinterpret_cast<uintptr_t>(bias) & 15) == 0) {
#pragma unroll
          for (int nq = 0; nq < NG; ++nq) {
            const f32x4_t* bp = (const f32x4_t*)(bias + min(cw + nq * 32, p.N - 8));
            bc[nq][0] = bp[0]; bc[nq][1] = bp[1];
          }
        } else {
#pragma unroll
          for (int nq = 0; nq < NG; ++nq) {
            const float* bp = bias + min(cw + nq * 32, p.N - 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bc[nq][0][e] = bp[e]; bc[nq][1][e] = bp[4 + e]; }
          }
        }
      } else if (bias) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) brow[mi] = bias[min(m0 + (wm * MI + mi) * 16 + frow, p.M - 1)];
      }
      // (two copies of the store loop, with and without a residual, so that the residual loads are unconditional:
      //  a load under a lane-dependent branch makes hipcc's wait insertion fall back to vmcnt(0))
      auto store_groups = [&](auto has_res) {
        constexpr bool HR = decltype(has_res)::value;
        f32x4_t rv[RD == 2 ? 2 : 1][(HR && RD) ? MI : 1][2];
        auto loadg = [&](int nq, int slot) {
          if constexpr (HR && RD > 0) {
            const int colc = min(cw + nq * 32, p.N - 8);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const f32x4_t* rp = (const f32x4_t*)(resid + (long)min(m0 + (wm * MI + mi) * 16 + frow, p.M - 1) * p.ldr + colc);
              rv[slot][mi][0] = rp[0]; rv[slot][mi][1] = rp[1];
            }
          }
        };
        if constexpr (RD == 2) loadg(0, 0);
#pragma unroll
        for (int nq = 0; nq < NG; ++nq) {
          if constexpr (RD == 2) { if (nq + 1 < NG) loadg(nq + 1, (nq + 1) & 1); }
          if constexpr (RD == 1) loadg(nq, 0);
          const int col = cw + nq * 32;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int row = m0 + (wm * MI + mi) * 16 + frow;
            f32x4_t v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x4_t pre = acc[mi][2 * nq + h] * p.alpha + bc[nq][h] + brow[mi];
              if (p.act == ODIC_ACT_GELU) {
                pre = gelu_poly4(pre);
              } else if (p.act != ODIC_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pre[e] = apply_act<true>(pre[e], p.act);
              }
              v[h] = pre;
            }
            if constexpr (HR) {
              if constexpr (RD == 0) {
                const f32x4_t* rp = (const f32x4_t*)(resid + (long)min(row, p.M - 1) * p.ldr + min(col, p.N - 8));
                v[0] += rp[0]; v[1] += rp[1];
              } else {
                v[0] += rv[RD == 2 ? (nq & 1) : 0][mi][0]; v[1] += rv[RD == 2 ? (nq & 1) : 0][mi][1];
              }
            }
            if (row < p.M && col < p.N) {
              OutT* dst = out + (long)row * p.ldc + col;
              if constexpr (sizeof(OutT) == 4) {
                ((f32x4_t*)dst)[0] = v[0]; ((f32x4_t*)dst)[1] = v[1];
              } else {
                bf16x8_t pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pk[e] = (short)f32_to_bf16(v[0][e]); pk[4 + e] = (short)f32_to_bf16(v[1][e]); }
                *(bf16x8_t*)dst = pk;
              }
            }
          }
        }
      };
      if (resid) store_groups(std::true_type{});
      else store_groups(std::false_type{});
      return;
    }
  }
#pragma unroll
  for (int nq = 0; nq < NI / 2; ++nq) {
    const int col = n0 + wn * NI * 16 + nq * 32 + fq * 8;
    if (col >= p.N) continue;
    float bc[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bc[e] = (bias && !p.bias_axis && col + e < p.N) ? bias[col + e] : 0.f;
      cs[e] = (LNF == 2 && col + e < p.N) ? p.ln_colsum[col + e] : 0.f;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int rloc = (wm * MI + mi) * 16 + frow;
      const int row = m0 + rloc;
      if (row >= p.M) continue;
      const float brow = (bias && p.bias_axis) ? bias[row] : 0.f;
      float2 mr = make_float2(0.f, 1.f);
      if constexpr (LNF == 2) mr = s_stat[rloc];
      f32x4_t v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4_t lin = acc[mi][2 * nq + h] * p.alpha;
        if constexpr (LNF == 2)
          lin = (lin - f32x4_t{cs[4 * h], cs[4 * h + 1], cs[4 * h + 2], cs[4 * h + 3]} * mr.x) * mr.y;
        f32x4_t pre = lin + f32x4_t{bc[4 * h], bc[4 * h + 1], bc[4 * h + 2], bc[4 * h + 3]} + brow;
        if (p.act == ODIC_ACT_GELU) {
          pre = gelu_poly4(pre);
        } else if (p.act != ODIC_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) pre[e] = apply_act<true>(pre[e], p.act);
        }
        v[h] = pre;
      }
      if (ld_ok && col + 7 < p.N) {
        if (resid) {
          const f32x4_t* rp = (const f32x4_t*)(resid + (long)row * p.ldr + col);
          v[0] += rp[0]; v[1] += rp[1];
        }
        OutT* dst = out + (long)row * p.ldc + col;
        if constexpr (sizeof(OutT) == 4) {
          ((f32x4_t*)dst)[0] = v[0]; ((f32x4_t*)dst)[1] = v[1];
          if constexpr (LNF == 1) {
            // folded LayerNorm, producer side: the bf16 copy the next product reads as its A operand, and the
            // moments of THOSE bf16 values over this row's 32-column group (the 4 fq lanes of a row hold it)
            bf16x8_t pk;
            float a8[8], sum = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const bf16_raw hb = f32_to_bf16(v[e >> 2][e & 3]);
              pk[e] = (short)hb;
              a8[e] = bf16_to_f32(hb);
              sum += a8[e];
            }
            *(bf16x8_t*)(p.out16 + (long)row * p.ld16 + col) = pk;
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float gmean = sum * (1.0f / 32.0f);
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = a8[e] - gmean; m2 += d * d; }
            m2 += __shfl_xor(m2, 16, 64);
            m2 += __shfl_xor(m2, 32, 64);
            if (fq == 0) ((float2*)p.stats_out)[(long)row * (p.N >> 5) + (col >> 5)] = make_float2(gmean, m2);
          }
        } else {
          bf16x8_t pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) { pk[e] = (short)f32_to_bf16(v[0][e]); pk[4 + e] = (short)f32_to_bf16(v[1][e]); }
          *(bf16x8_t*)dst = pk;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (col + e < p.N) {
            float x = v[e >> 2][e & 3];
            if (resid) x += resid[(long)row * p.ldr + col + e];
            store_from_f32<OutT>(out + (long)row * p.ldc + col + e, x);
          }
        }
      }
    }
  }
}



#ifdef ODIC_EXPERIMENTAL_GEMM   // persistent and 256x256 phase-pipelined kernels: measured, never selected
// One K-tile of LDS-DMA for the persistent kernel: `buffer_load ... lds` with the tile origins in scalar resource
// descriptors, per-lane 32-bit offsets and the K advance as the scalar offset.  (A free function: a local of the
// buffer-resource type inside a lambda of a __global__ template suppresses the kernel's host stub on ROCm 7.2.)
template <int A_INSTR, int W_INSTR, int NW>
__device__ __forceinline__ void persist_stage(char* la, int a_bytes, const bf16_raw* a_base, const bf16_raw* w_base,
                                              const int* voff_a, const int* voff_w, int wave, int koff) {
  char* lw = la + a_bytes;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(la + (i * NW + wave) * 1024), 16, voff_a[i], koff, 0, 0);
#pragma unroll
  for (int i = 0; i < W_INSTR; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(lw + (i * NW + wave) * 1024), 16, voff_w[i], koff, 0, 0);
}

// =================================================================================================
// Persistent, dynamically scheduled form of the generic kernel (tile_cfg 16 + c runs tile config c this way).
//
// Why: a Swin-L product is 100..1700 tiles on 256 CUs x (1..3 resident blocks), i.e. 1-3 "rounds" with a ragged
// last one, and beside the decoder-step kernels of the other HIP streams some CUs are slower than others — with
// one block per tile the launch ends with its slowest CU (the straggler effect).  Here a launch is
// only as many blocks as fit the chip; every block pulls tiles from per-XCD-partition atomic counters until the
// product is done, so a slowed CU simply takes fewer tiles, and a partition that runs dry steals from the next.
//   * tile order inside a partition is unchanged (N-fastest inside the XCD's rectangle → its W sub-panel stays in
//     the XCD's L2 and an A panel is consumed by all its column tiles while resident);
//   * the NEXT tile's ticket is drawn (one returning atomic by one lane) right after the current tile's first
//     K-tiles have been requested and is only looked at after the K-loop, so its latency costs nothing;
//     (vmcnt retires in order and counts stores, so a tile's first K-tile cannot be consumed before the previous
//     epilogue's stores are acknowledged: requesting it ahead of the epilogue buys nothing inside one wave —
//     prologue and store tail overlap across the 2-3 resident blocks of a CU instead);
//   * LDS-DMA as `buffer_load ... lds`: the tile's base sits in a scalar resource descriptor, the per-lane part
//     is a 32-bit offset computed once per tile and the K advance is the scalar offset — no 64-bit address
//     arithmetic per issue.
// Workspace protocol: ws[0..7] tile counters, ws[8] exit counter; the caller hands zeros, the last block to
// leave zeroes them again (kernel boundary = release), so one workspace serves all launches of a stream.
// Blocks never wait for each other: no residency requirement, nothing can hang.
// =================================================================================================
template <int NWM, int NWN, int MI, int NI, int NSTAGE, int BK, typename OutT>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_bf16_nt_persist_kernel(Params p) {
  constexpr int NW = NWM * NWN;
  constexpr int ROWB = BK * 2;
  constexpr int RPI = 1024 / ROWB;
  constexpr int CPR = ROWB / 16;
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / RPI / NW, W_INSTR = BN / RPI / NW;
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile rows must split evenly over the waves");
  constexpr int G = A_INSTR + W_INSTR;
  static_assert(NI % 2 == 0, "the epilogue pairs MFMA column tiles");
  constexpr int D = NSTAGE - 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];          // stages | int slot[2] (next-tile mailbox)
  typedef __attribute__((address_space(3))) int lds_int;
  lds_int* slot = (lds_int*)(lds + NSTAGE * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  ODIC_ENCODE_PRIO();

  // ---- tile source.  Partition `part` of the tile grid (the XCD rectangles of the one-block-per-tile kernel)
  //      has its own counter; a block starts on partition blockIdx % 8 and moves on when that one runs dry.
  const int xcd = blockIdx.x & 7;
  int dry = 0;                                                        // partitions this block has seen exhausted
  int p_r0 = 0, p_c0 = 0, p_w = 1, p_size = 0;                        // rectangle of the current partition
  auto load_part = [&]() {
    const int part = (xcd + dry) & 7;
    const int xm = part / p.pn, xn = part - xm * p.pn;
    const int r1 = (xm + 1) * p.tiles_m / p.pm, c1 = (xn + 1) * p.tiles_n / p.pn;
    p_r0 = xm * p.tiles_m / p.pm; p_c0 = xn * p.tiles_n / p.pn;
    p_w = c1 - p_c0; p_size = (r1 - p_r0) * p_w;
  };
  auto decode = [&](int idx) -> int {
    const int lr = idx / p_w;
    return ((p_r0 + lr) << 16) | (p_c0 + idx - lr * p_w);
  };
  auto draw = [&]() -> int {                                          // returning atomic, result not waited for here
    return __hip_atomic_fetch_add(p.ws + ((xcd + dry) & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // ticket → tile; run by ALL lanes of wave 0 with the ticket in lane 0.  When the partition is dry, lanes 0..7 read
  // the eight counters in ONE wave instruction (past the L1) and the partition with the most tiles left is taken;
  // nothing left anywhere → -1.  (Eight dependent loads by one lane cost more than the tile itself.)
  auto resolve = [&](int ticket_lane0) -> int {
    int idx = __builtin_amdgcn_readfirstlane(ticket_lane0);
    while (idx >= p_size) {
      const int i = lane & 7;
      const int xm = i / p.pn, xn = i - xm * p.pn;
      const int sz = ((xm + 1) * p.tiles_m / p.pm - xm * p.tiles_m / p.pm) *
                     ((xn + 1) * p.tiles_n / p.pn - xn * p.tiles_n / p.pn);
      int key = ((sz - __hip_atomic_load(p.ws + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) << 3) | i;
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o, 64));      // lanes 0..7 hold the 8 partitions
      key = __builtin_amdgcn_readfirstlane(key);
      if ((key >> 3) <= 0) return -1;
      dry = ((key & 7) - xcd) & 7;
      load_part();
      int t = 0;
      if (lane == 0) t = draw();
      idx = __builtin_amdgcn_readfirstlane(t);
    }
    return decode(idx);
  };
  load_part();

  const int srow = lane / CPR;
  const int schunk = swz<BK>(lane % CPR, srow);
  const int frow = lane & 15, fq = lane >> 4;
  const int nk = p.K / BK;
  const bf16_raw* A = p.A;
  const bf16_raw* W = p.W;

  int m0 = 0, n0 = 0;
  const bf16_raw* a_base = A;                                          // tile origins (wave-uniform)
  const bf16_raw* w_base = W;
  int voff_a[A_INSTR], voff_w[W_INSTR];
  auto setup = [&](int tile) {                                         // tile is wave-uniform (SGPR)
    m0 = (tile >> 16) * BM; n0 = (tile & 0xffff) * BN;
    a_base = A + (long)m0 * p.lda;
    w_base = W + (long)n0 * p.ldw;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const int row = (i * NW + wave) * RPI + srow;
      voff_a[i] = (min(m0 + row, p.M - 1) - m0) * (int)p.lda * 2 + schunk * 16;
    }
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
      const int row = (i * NW + wave) * RPI + srow;
      voff_w[i] = (min(n0 + wperm(row), p.N - 1) - n0) * (int)p.ldw * 2 + schunk * 16;
    }
  };
  auto stage = [&](int buf, int kt) {
    persist_stage<A_INSTR, W_INSTR, NW>(lds + buf * STAGE, A_BYTES, a_base, w_base, voff_a, voff_w, wave, kt * ROWB);
  };

  // ---- first tile
  if (wave == 0) {
    int t = 0;
    if (lane == 0) t = draw();
    const int first = resolve(t);
    if (lane == 0) slot[0] = first;
  }
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(slot[0]);
  const float* bias = p.bias;
  const float* resid = p.residual;
  OutT* out = (OutT*)p.out;
  const bool ld_ok = ((p.ldc & 7) == 0) && (!resid || (p.ldr & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0);

  for (int it = 0; tile >= 0; ++it) {
    setup(tile);
#pragma unroll
    for (int t = 0; t < D; ++t)
      if (t < nk) stage(t, t);
    // ticket for the NEXT tile: issued now, looked at after the K-loop (its latency hides behind the whole tile;
    // it is younger than this tile's first DMA groups, so the counted waits below at most over-wait by one)
    int ticket = 0;
    if (tid == 0) ticket = draw();

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      // (the previous tile's epilogue stores are OLDER than this tile's DMA: a counted wait covers them too)
      const int ahead = min(D - 1, nk - 1 - kt);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + D < nk) stage((kt + D) % NSTAGE, kt + D);

      const int cur = kt % NSTAGE;
      const char* la = lds + cur * STAGE + (wm * MI * 16 + frow) * ROWB;
      const char* lw = lds + cur * STAGE + A_BYTES + (wn * NI * 16 + frow) * ROWB;
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        bf16x8_t af[MI], wf[NI];
        const int chunk = swz<BK>(kk * 4 + fq, frow) << 4;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = *(const bf16x8_t*)(la + mi * 16 * ROWB + chunk);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8_t*)(lw + ni * 16 * ROWB + chunk);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
      }
    }
    // next tile: resolve the ticket (steals only at the very end of a launch), hand it to the other waves.
    // slot[] alternates, so a fast wave 0 cannot overwrite a value a slow wave has not read yet.
    if (wave == 0) {
      const int nx = resolve(ticket);
      if (lane == 0) slot[(it + 1) & 1] = nx;
    }

    // ---- epilogue (lane → 8 adjacent output columns, as in the generic kernel)
#pragma unroll
    for (int nq = 0; nq < NI / 2; ++nq) {
      const int col = n0 + wn * NI * 16 + nq * 32 + fq * 8;
      if (col >= p.N) continue;
      float bc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bc[e] = (bias && !p.bias_axis && col + e < p.N) ? bias[col + e] : 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = m0 + (wm * MI + mi) * 16 + frow;
        if (row >= p.M) continue;
        const float brow = (bias && p.bias_axis) ? bias[row] : 0.f;
        f32x4_t v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4_t pre = acc[mi][2 * nq + h] * p.alpha + f32x4_t{bc[4 * h], bc[4 * h + 1], bc[4 * h + 2], bc[4 * h + 3]} + brow;
          if (p.act == ODIC_ACT_GELU) {
            pre = gelu_poly4(pre);
          } else if (p.act != ODIC_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[e] = apply_act<true>(pre[e], p.act);
          }
          v[h] = pre;
        }
        if (ld_ok && col + 7 < p.N) {
          if (resid) {
            const f32x4_t* rp = (const f32x4_t*)(resid + (long)row * p.ldr + col);
            v[0] += rp[0]; v[1] += rp[1];
          }
          OutT* dst = out + (long)row * p.ldc + col;
          if constexpr (sizeof(OutT) == 4) {
            ((f32x4_t*)dst)[0] = v[0]; ((f32x4_t*)dst)[1] = v[1];
          } else {
            bf16x8_t pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pk[e] = (short)f32_to_bf16(v[0][e]); pk[4 + e] = (short)f32_to_bf16(v[1][e]); }
            *(bf16x8_t*)dst = pk;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (col + e < p.N) {
              float x = v[e >> 2][e & 3];
              if (resid) x += resid[(long)row * p.ldr + col + e];
              store_from_f32<OutT>(out + (long)row * p.ldc + col + e, x);
            }
          }
        }
      }
    }
    // every wave has finished reading this tile's LDS stages (its last ds_reads fed the MFMAs above) and wave
    // 0's mailbox write has landed: one barrier, then the stages may be refilled.  (A raw barrier: __syncthreads()
    // would also wait for the acknowledgement of the epilogue's stores.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    tile = __builtin_amdgcn_readfirstlane(slot[(it + 1) & 1]);
  }
  // ---- leave: the last block re-arms the workspace for the next launch on this stream
  if (tid == 0) {
    const int prev = __hip_atomic_fetch_add(p.ws + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) __hip_atomic_store(p.ws + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// =================================================================================================
// 256 x 256 x 64 tile, one block of 8 waves per CU, four phases per K-tile (config 12).
//
// The generic kernel above fills LDS at ~half of what a CU can take from L2 and keeps the matrix pipe
// ~30 % busy: per K-tile it is one wait + one barrier + a compiler-scheduled blob.  This one follows
// the phase structure of cdna_hip_programming.md §5 ("256² 8-phase"): the 256 x 256 accumulator is
// cut into four 128 x 128 QUADRANTS (A half i x W half j); a phase = every wave's 64 x 32 patch of
// one quadrant over the whole K-tile (16 MFMAs) and the LDS-DMA of ONE 16-KiB half-tile (2 per wave):
//
//   phase  quadrant   LDS fragment reads           stages (dest. = LDS parity of that K-tile)
//     1    (A0, W0)   W0: 4, A0: 8 ds_read_b128    A1 of K-tile c+1
//     2    (A0, W1)   W1: 4 (A0 stays in regs)     W0 of K-tile c+1
//     3    (A1, W1)   A1: 8 (W1 stays)             A0 of K-tile c+2   (A0 of c: last read in phase 1)
//     4    (A1, W0)   none (W0 kept from phase 1)  W1 of K-tile c+2   (W1 of c: last read in phase 2)
//
// so every half-tile buffer is restaged two phases after its last read (WAR) and is read no earlier
// than the phase after the counted wait + barrier that retires it (RAW): the
// only vmcnt is in phase 4, vmcnt(4) = the two half-tiles of K-tile c+2 just issued stay in flight,
// everything of K-tile c+1 has landed.  128 KiB of LDS (2 parities x 4 halves), 128 accumulator
// registers per lane, operands swapped / W rows permuted exactly as in the generic kernel.
// Requires K % 128 == 0 (K-tiles are processed in pairs so that LDS parities are compile-time).
// =================================================================================================
template <typename OutT>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256sq_kernel(Params p) {
  constexpr int HALF = 128 * 128;               // bytes of one half-tile: 128 rows x 64 bf16
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [parity 2][A0, A1, W0, W1][128][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  ODIC_ENCODE_PRIO();

  int tm, tn;
  {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int xm = xcd / p.pn, xn = xcd - xm * p.pn;
    const int r0 = xm * p.tiles_m / p.pm, r1 = (xm + 1) * p.tiles_m / p.pm;
    const int c0 = xn * p.tiles_n / p.pn, c1 = (xn + 1) * p.tiles_n / p.pn;
    const int w = c1 - c0;
    if (idx >= (r1 - r0) * w) return;
    const int lr = idx / w;
    tm = r0 + lr; tn = c0 + (idx - lr * w);
  }
  const int m0 = tm * 256, n0 = tn * 256;
  const long bz = blockIdx.z;
  const bf16_raw* A = p.A + bz * p.strideA;
  const bf16_raw* W = p.W + bz * p.strideW;

  // ---- LDS-DMA sources: a half-tile is 16 pieces of 1 KiB (8 rows x 128 B); wave w moves pieces 2w, 2w+1.
  //      buffer_load ... lds with the tile's base in a scalar resource descriptor, a 32-bit per-lane byte
  //      offset and the K-tile advance as the scalar offset: no per-issue 64-bit address arithmetic, and the
  //      LDS-DMA issues cheaper than the flat-address form (it is the long pole of a phase's load segment).
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;          // swizzle: chunk ^ (row & 7)
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int voff[4][2];                                                  // [A0, A1, W0, W1][piece] byte offsets
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + srow;                     // row inside the half-tile
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      voff[h][i] = (min(m0 + h * 128 + row, p.M - 1) - m0) * (int)p.lda * 2 + schunk * 16;
      voff[2 + h][i] = (min(n0 + h * 128 + wperm(row), p.N - 1) - n0) * (int)p.ldw * 2 + schunk * 16;
    }
  }
  auto stage = [&](int parity, int h, int kt) {
    char* dst = lds + (parity * 4 + h) * HALF + wave * 2048;
    const __amdgpu_buffer_rsrc_t r = h < 2 ? rsrc_a : rsrc_w;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)dst, 16, voff[h][0], kt * 128, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(dst + 1024), 16, voff[h][1], kt * 128, 0, 0);
  };

  f32x4_t acc[2][2][4][2];                                         // [A half][W half][mi][ni]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[i][j][mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / 64;
  const int frow = lane & 15, fq = lane >> 4;
  const int a_off = (wr * 64 + frow) * 128, w_off = (wc * 32 + frow) * 128;
  const int c0 = ((0 + fq) ^ (frow & 7)) << 4, c1 = ((4 + fq) ^ (frow & 7)) << 4;   // kk = 0 / 1 chunks

  bf16x8_t af[4][2], wf[2][2][2];                                  // [mi][kk], [W half][ni][kk]: both W halves stay live
  auto read_a = [&](int parity, int h) {
    const char* b = lds + (parity * 4 + h) * HALF + a_off;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      af[mi][0] = *(const bf16x8_t*)(b + mi * 16 * 128 + c0);
      af[mi][1] = *(const bf16x8_t*)(b + mi * 16 * 128 + c1);
    }
  };
  auto read_w = [&](int parity, int h) {
    const char* b = lds + (parity * 4 + 2 + h) * HALF + w_off;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      wf[h][ni][0] = *(const bf16x8_t*)(b + ni * 16 * 128 + c0);
      wf[h][ni][1] = *(const bf16x8_t*)(b + ni * 16 * 128 + c1);
    }
  };
  auto mma = [&](int i, int j) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[i][j][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ni][kk], af[mi][kk], acc[i][j][mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // prologue: all of K-tile 0, plus A0 / W1 of K-tile 1 (its A1 / W0 follow in phases 1 and 2)
  stage(0, 0, 0); stage(0, 2, 0); stage(0, 3, 0); stage(0, 1, 0);
  if (nk > 1) {
    stage(1, 0, 1); stage(1, 3, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  // The two wave rows run half a phase apart (one extra barrier for row 1 here, one for row 0 after the
  // loop): a SIMD holds one wave of each row, so while one of them issues its 16 MFMAs the other is in
  // its load segment (fragment reads, LDS-DMA issue, counted wait) — the matrix pipe never waits for LDS.
  // Every segment ends in a block-wide barrier.  Hazards with the half-phase skew: a buffer is restaged
  // two phases (four segments) after its last read, the later row finishes that read two segments after
  // the earlier one started it; a retired K-tile is first read two segments after the earlier row's wait,
  // i.e. one segment after the later row's wait + barrier.
  auto ktile = [&](int c, int parity) {           // parity is a literal at both call sites
    // ---- phase 1: quadrant (A0, W0)
    read_w(parity, 0); read_a(parity, 0);
    if (c + 1 < nk) stage(parity ^ 1, 1, c + 1);
    __builtin_amdgcn_s_barrier();
    mma(0, 0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: quadrant (A0, W1)
    read_w(parity, 1);
    if (c + 1 < nk) stage(parity ^ 1, 2, c + 1);
    __builtin_amdgcn_s_barrier();
    mma(0, 1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: quadrant (A1, W1)
    read_a(parity, 1);
    if (c + 2 < nk) stage(parity, 0, c + 2);
    __builtin_amdgcn_s_barrier();
    mma(1, 1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: quadrant (A1, W0) — W0 fragments are still in registers from phase 1; the counted wait
    //      retires every half-tile of K-tile c+1
    if (c + 2 < nk) {
      stage(parity, 3, c + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    mma(1, 0);
    __builtin_amdgcn_s_barrier();
  };
  if (wr == 1) __builtin_amdgcn_s_barrier();
  for (int c = 0; c < nk; c += 2) {
    ktile(c, 0);
    ktile(c + 1, 1);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue (same lane → 8 adjacent output columns mapping as the generic kernel)
  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  OutT* out = (OutT*)p.out + bz * p.strideC;
  const bool ld_ok = ((p.ldc & 7) == 0) && (!resid || (p.ldr & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + j * 128 + wc * 32 + fq * 8;
    if (col >= p.N) continue;
    float bc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bc[e] = (bias && !p.bias_axis && col + e < p.N) ? bias[col + e] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row = m0 + i * 128 + wr * 64 + mi * 16 + frow;
        if (row >= p.M) continue;
        const float brow = (bias && p.bias_axis) ? bias[row] : 0.f;
        f32x4_t v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4_t pre = acc[i][j][mi][h] * p.alpha + f32x4_t{bc[4 * h], bc[4 * h + 1], bc[4 * h + 2], bc[4 * h + 3]} + brow;
          if (p.act == ODIC_ACT_GELU) {
            pre = gelu_poly4(pre);
          } else if (p.act != ODIC_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[e] = apply_act<true>(pre[e], p.act);
          }
          v[h] = pre;
        }
        if (ld_ok && col + 7 < p.N) {
          if (resid) {
            const f32x4_t* rp = (const f32x4_t*)(resid + (long)row * p.ldr + col);
            v[0] += rp[0]; v[1] += rp[1];
          }
          OutT* dst = out + (long)row * p.ldc + col;
          if constexpr (sizeof(OutT) == 4) {
            ((f32x4_t*)dst)[0] = v[0]; ((f32x4_t*)dst)[1] = v[1];
          } else {
            bf16x8_t pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pk[e] = (short)f32_to_bf16(v[0][e]); pk[4 + e] = (short)f32_to_bf16(v[1][e]); }
            *(bf16x8_t*)dst = pk;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (col + e < p.N) {
              float x = v[e >> 2][e & 3];
              if (resid) x += resid[(long)row * p.ldr + col + e];
              store_from_f32<OutT>(out + (long)row * p.ldc + col + e, x);
            }
          }
        }
      }
    }
  }
}

#endif  // ODIC_EXPERIMENTAL_GEMM

// =================================================================================================
// A-resident streaming form for the K = 192 / 384 products of Swin stages 0-1 (tile_cfg 50-53).
//
// Those products (147456 x {576,192,768} x 192, 36864 x {1152,384,1536} x 384 at B = 16) are output-bandwidth-bound —
// 227-283 MB of traffic for 33-44 GFLOP — and the tiled kernel above spends 1.5-3x the time of a copy of the same
// bytes on them (tools/smallk_probe.py: fc1 of stage 0 145 us against 84 us for a device copy of twice its bytes):
// a K-loop of 3-6 steps is all pipeline fill, every tile re-stages its A rows once per column tile, and the stores of a
// round of tiles leave together.  Here nothing is re-staged and the stores never stop:
//   * a wave keeps its 16·MI rows of A over the WHOLE K extent in registers (MI · K/32 fragments, read once, straight
//     from global memory in MFMA operand layout — A never touches LDS);
//   * the block's four waves walk the output columns in chunks of 16·NI; a chunk of W (16·NI rows x K, 24 KiB) comes in
//     by LDS-DMA one chunk ahead into the other of two buffers, in the generic kernel's swizzled sub-tile images;
//   * one barrier per chunk; the counted wait in front of it lets the previous chunk's stores stay in flight
//     (vmcnt retires in order: the chunk's DMA was issued BEFORE those stores, so `vmcnt(stores per chunk)` proves it
//     landed) — the store stream of a wave overlaps its own next MFMA block, and the resident blocks of a CU
//     overlap each other's epilogue arithmetic.  (Tried and dropped: a fifth wave that only issues the DMA, so that
//     the compute waves never wait on vmcnt at all — 10 % slower; the store stream is not what the counted wait holds
//     back, the same kernel with MFMA and DMA switched off stores no faster.);
//   * whole tiles only (M % 64·MI == 0, N % 16·NI == 0, vector-store alignment): every wave issues exactly the counted
//     number of stores; anything else is refused and the caller's tuner falls back to the tiled kernel.
// Grid: row panels x `nsplit` column ranges; the ranges of one panel run on one XCD (its A rows come from that L2).
// =================================================================================================
template <int MI, int NI, int KT, int KH, typename OutT, bool RES, bool LNA = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_apanel_kernel(Params p, int nsplit) {
  constexpr int NW = 4;
  constexpr int BM = NW * MI * 16, BNC = NI * 16;
  constexpr int SUB = BNC * 128;                         // one 64-deep W sub-tile: BNC rows x 128 B
  // a chunk of W arrives in KH pieces along K (KH = 2 for K = 384: a 24-KiB buffer then holds half the K extent of 64
  // columns, so that the wave's column groups still pair into whole 128-byte output lines); one pipeline step per piece
  static_assert(KH == 1 || KH == 2, "one or two K pieces per chunk");
  constexpr int KTS = KT / KH;                           // 64-deep sub-tiles per step
  constexpr int CHUNK = KTS * SUB;                       // bytes per buffer
  static_assert(CHUNK % (1024 * NW) == 0, "a chunk must split into whole DMA instructions per wave");
  constexpr int INSTR = CHUNK / 1024 / NW;               // 1-KiB LDS-DMA instructions per wave per chunk
  constexpr int RG = BNC / 8;                            // 8-row DMA groups per sub-tile
  constexpr int NST = MI * (NI / 2) * (sizeof(OutT) == 4 ? 2 : 1);     // stores per wave per chunk
  static_assert(NI % 2 == 0 && NST <= 48, "store count must fit the vmcnt field with the DMA group on top");
  extern __shared__ __attribute__((aligned(16))) char lds[];          // W chunk buffers 0 | 1 | the block's bias values

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ODIC_ENCODE_PRIO();
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int split = idx % nsplit, panel = (idx / nsplit) * 8 + xcd;
  if (panel >= p.M / BM) return;
  const int nchunks = p.N / BNC;
  const int c0 = split * nchunks / nsplit, c1 = (split + 1) * nchunks / nsplit;
  if (c0 >= c1) return;
  const int m0 = panel * BM;
  const int frow = lane & 15, fq = lane >> 4;

  // ---- this wave's rows of A, whole K, as MFMA B-operand fragments: lane (frow, fq) holds A[row frow][32k + 8fq ..]
  bf16x8_t af[2 * KT][MI];
  if constexpr (LNA) {
    // LayerNorm while reading: the row's K fp32 values are spread over the four fq lanes (8 per 32-deep step each);
    // mean and centred variance in two passes over the registers, reduced across lanes frow, frow + 16, + 32, + 48,
    // then (x - mean)·rstd is rounded to bf16 straight into the fragments (gamma / beta are folded into W / bias).
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const float* xr = p.a_ln + (long)(m0 + (wave * MI + mi) * 16 + frow) * p.ld_aln + fq * 8;
      if constexpr (KT > 3) {
        // K = 384: 96 fp32 values per lane and row would cost the kernel a resident block — moments in ONE pass over the
        // loads (shifted by the row's first element, which keeps E[(x-c)²] − E[x-c]² well conditioned), then the row is
        // read again (L1 / L2) and normalised into the fragments
        f32x4_t a0 = *(const f32x4_t*)xr;
        const float c = __shfl(a0[0], frow, 64);                         // x[row][0] (lane fq = 0 holds it)
        f32x4_t s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2 * KT; ++k) {
          const f32x4_t u0 = *(const f32x4_t*)(xr + k * 32) - c, u1 = *(const f32x4_t*)(xr + k * 32 + 4) - c;
          s1 += u0 + u1;
          s2 += u0 * u0 + u1 * u1;
          if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);             // (at most 8 row loads in flight: registers)
        }
        float t1 = (s1[0] + s1[1]) + (s1[2] + s1[3]), t2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
        t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
        t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
        const float m1 = t1 * (1.0f / (64 * KT));
        const float mean = c + m1;
        const float rstd = rsqrtf(fmaxf(t2 * (1.0f / (64 * KT)) - m1 * m1, 0.f) + p.ln_eps);
#pragma unroll
        for (int k = 0; k < 2 * KT; ++k) {
          const f32x4_t u0 = (*(const f32x4_t*)(xr + k * 32) - mean) * rstd, u1 = (*(const f32x4_t*)(xr + k * 32 + 4) - mean) * rstd;
          bf16x8_t f;
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[e] = (short)f32_to_bf16(u0[e]); f[4 + e] = (short)f32_to_bf16(u1[e]); }
          af[k][mi] = f;
          if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        continue;
      }
      f32x4_t xv[2 * KT][2];
#pragma unroll
      for (int k = 0; k < 2 * KT; ++k) { xv[k][0] = *(const f32x4_t*)(xr + k * 32); xv[k][1] = *(const f32x4_t*)(xr + k * 32 + 4); }
      f32x4_t s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 2 * KT; ++k) s4 += xv[k][0] + xv[k][1];
      float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum / (float)(64 * KT);       // (a true division: a constant row then normalises to exactly 0)
      f32x4_t q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 2 * KT; ++k) {
        xv[k][0] -= mean; xv[k][1] -= mean;
        q4 += xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1];
      }
      float ssq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
      ssq += __shfl_xor(ssq, 16, 64);
      ssq += __shfl_xor(ssq, 32, 64);
      const float rstd = rsqrtf(ssq / (float)(64 * KT) + p.ln_eps);
#pragma unroll
      for (int k = 0; k < 2 * KT; ++k) {
        bf16x8_t f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = (short)f32_to_bf16(xv[k][0][e] * rstd);
          f[4 + e] = (short)f32_to_bf16(xv[k][1][e] * rstd);
        }
        af[k][mi] = f;
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const bf16_raw* ar = p.A + (long)(m0 + (wave * MI + mi) * 16 + frow) * p.lda + fq * 8;
#pragma unroll
      for (int k = 0; k < 2 * KT; ++k) af[k][mi] = *(const bf16x8_t*)(ar + k * 32);
    }
  }

  // ---- W chunk DMA: instruction i of this wave fills 1 KiB = rows 8·rg .. +7 of sub-tile kt, j = i·NW + wave
  int w_off[INSTR];                                       // element offsets inside a chunk (< 2^31: host-checked)
#pragma unroll
  for (int i = 0; i < INSTR; ++i) {
    const int j = i * NW + wave, kt = j / RG, rg = j - kt * RG;
    const int srow = lane >> 3, r = rg * 8 + srow;
    w_off[i] = wperm(r) * (int)p.ldw + kt * 64 + swz<64>(lane & 7, srow) * 8;
  }
  auto issue = [&](int c, int h, int buf) {
    const bf16_raw* wb = p.W + (long)c * BNC * p.ldw + h * KTS * 64;
    char* lb = lds + buf * CHUNK;
#pragma unroll
    for (int i = 0; i < INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wb + w_off[i]), (lptr_t)(lb + (i * NW + wave) * 1024), 16, 0, 0);
  };

  const float* resid = p.residual;
  OutT* out = (OutT*)p.out;

  issue(c0, 0, 0);
  // the block's bias values go through LDS: a global load inside the chunk loop would be younger than the next
  // piece's DMA, and waiting for it (vmcnt retires in order) would wait for that DMA and every store before it
  float* sbias = (float*)(lds + 2 * CHUNK);
  for (int t = tid; t < (c1 - c0) * BNC; t += 256) sbias[t] = p.bias ? p.bias[c0 * BNC + t] : 0.f;
  // A fragments, first piece and bias staging complete.  (The builtin, not inline asm: hipcc's own wait insertion
  // must SEE that the A fragments have arrived, or it waits for vmcnt(0) at their first use in every iteration.)
  __builtin_amdgcn_s_waitcnt(0x0070);                                  // vmcnt(0) lgkmcnt(0)
  for (int c = c0; c < c1; ++c) {
    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      // this piece has landed: its DMA is older than the previous chunk's NST stores, which may stay in flight when
      // they are the only younger operations (h == 0); a second piece has nothing younger than itself ...
      if (h == 0) { if (c != c0) __builtin_amdgcn_s_waitcnt(0x0F70 | (NST & 15) | ((NST >> 4) << 14)); }   // vmcnt(NST)
      else __builtin_amdgcn_s_waitcnt(0x0F70);                                                             // vmcnt(0)
      // ... for every wave, and every wave is done reading the other buffer (the previous piece)
      __builtin_amdgcn_s_barrier();
      const int buf = KH == 2 ? h : ((c - c0) & 1);
      if (h + 1 < KH) issue(c, h + 1, buf ^ 1);
      else if (c + 1 < c1) issue(c + 1, 0, buf ^ 1);
      const char* lw = lds + buf * CHUNK + frow * 128;
#pragma unroll
      for (int kt = 0; kt < KTS; ++kt) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8_t wf[NI];
          const int chunk = swz<64>(kk * 4 + fq, frow) << 4;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8_t*)(lw + kt * SUB + ni * 16 * 128 + chunk);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[2 * (h * KTS + kt) + kk][mi], acc[mi][ni], 0, 0, 0);
        }
      }
    }

    // ---- epilogue of the chunk (lane → 8 adjacent columns as in the generic kernel; no bounds: whole tiles only)
    // the eight finished values of accumulator pair (mi, nq): alpha, bias, activation, residual
    auto finish = [&](int mi, int nq, f32x4_t* v) {
      const int col = c * BNC + nq * 32 + fq * 8;
      const f32x4_t* sb = (const f32x4_t*)(sbias + (c - c0) * BNC + nq * 32 + fq * 8);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4_t pre = acc[mi][2 * nq + h] * p.alpha + sb[h];
        if (p.act == ODIC_ACT_GELU) {
          pre = gelu_poly4(pre);
        } else if (p.act != ODIC_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) pre[e] = apply_act<true>(pre[e], p.act);
        }
        v[h] = pre;
      }
      if constexpr (RES) {
        const f32x4_t* rp = (const f32x4_t*)(resid + (long)(m0 + (wave * MI + mi) * 16 + frow) * p.ldr + col);
        v[0] += rp[0]; v[1] += rp[1];
      }
    };
    if constexpr (sizeof(OutT) == 2 && NI % 4 == 0) {
      // bf16 output, WHOLE 128-byte lines per store instruction.  A lane's 8 columns are 16 bytes and the four fq lanes
      // of a row make 64 contiguous bytes — half a cache line per row, and half-line stores run at 4 TB/s where whole
      // lines run at 6.8 (tools/smallk_probe.py).  The other half of the line is the same lanes' NEXT column group, so
      // the two groups are exchanged between lanes frow and frow ^ 8 (one DPP row rotation by 8 per dword): the low
      // eight lanes of each 16 then hold both rows' first halves, the high eight both rows' second halves, and each
      // store instruction writes 8 rows x 128 bytes.
      typedef __attribute__((ext_vector_type(4))) int i32x4_t;
      const bool lo = frow < 8;
#pragma unroll
      for (int q2 = 0; q2 < NI / 4; ++q2) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4_t v0[2], v1[2];
          finish(mi, 2 * q2, v0);
          finish(mi, 2 * q2 + 1, v1);
          bf16x8_t p0, p1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p0[e] = (short)f32_to_bf16(v0[0][e]); p0[4 + e] = (short)f32_to_bf16(v0[1][e]);
            p1[e] = (short)f32_to_bf16(v1[0][e]); p1[4 + e] = (short)f32_to_bf16(v1[1][e]);
          }
          const i32x4_t own0 = __builtin_bit_cast(i32x4_t, p0), own1 = __builtin_bit_cast(i32x4_t, p1);
          const i32x4_t send = lo ? own1 : own0;
          i32x4_t recv;
#pragma unroll
          for (int e = 0; e < 4; ++e) recv[e] = __builtin_amdgcn_update_dpp(0, send[e], 0x128, 0xf, 0xf, false);   // row_ror:8
          // first store: rows 0-7 of the 16 (low lanes: own first half; high lanes: row frow - 8's second half)
          const long rbase = (long)(m0 + (wave * MI + mi) * 16 + (frow & 7)) * p.ldc + c * BNC + q2 * 64 + (lo ? 0 : 32) + fq * 8;
          *(i32x4_t*)(out + rbase) = lo ? own0 : recv;
          *(i32x4_t*)(out + rbase + 8 * p.ldc) = lo ? recv : own1;
        }
      }
    } else {
#pragma unroll
      for (int nq = 0; nq < NI / 2; ++nq) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4_t v[2];
          finish(mi, nq, v);
          OutT* dst = out + (long)(m0 + (wave * MI + mi) * 16 + frow) * p.ldc + c * BNC + nq * 32 + fq * 8;
          if constexpr (sizeof(OutT) == 4) {
            ((f32x4_t*)dst)[0] = v[0]; ((f32x4_t*)dst)[1] = v[1];
          } else {
            bf16x8_t pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pk[e] = (short)f32_to_bf16(v[0][e]); pk[4 + e] = (short)f32_to_bf16(v[1][e]); }
            *(bf16x8_t*)dst = pk;
          }
        }
      }
    }
  }
}

template <int MI, int NI, int KT, int KH = 1>
int launch_apanel(Params& p, int out_dtype, int batch, hipStream_t stream) {
  constexpr int BM = 4 * MI * 16, BNC = NI * 16;
  if (batch != 1 || p.K != KT * 64 || p.M % BM != 0 || p.N % BNC != 0 || p.bias_axis != 0 || p.out16 || p.ln_stats)
    return ODIC_EUNSUPPORTED;
  if ((long)BNC * p.ldw >= (1L << 30) || (p.residual && MI > 2))       // (the residual form of the 64-row waves would spill)
    return ODIC_EUNSUPPORTED;
  if ((p.ldc & 7) || ((uintptr_t)p.out & 15) || (p.residual && ((p.ldr & 3) || ((uintptr_t)p.residual & 15))))
    return ODIC_EUNSUPPORTED;
  const int panels = p.M / BM, nchunks = p.N / BNC;
  // column ranges per panel: the smallest divisor of the chunk count that gives the chip >= `want` blocks (3 rounds of 2
  // per CU) — a block should keep its A rows for as many chunks as the grid size allows
  static const int want = getenv("ODIC_APANEL_BLOCKS") ? atoi(getenv("ODIC_APANEL_BLOCKS")) : 1536;
  int nsplit = nchunks;
  for (int d = 1; d <= nchunks; ++d)
    if (nchunks % d == 0 && (long)panels * d >= want) { nsplit = d; break; }
  dim3 grid(8 * ((panels + 7) / 8) * nsplit), block(256);
  const int SHMEM = 2 * (KT / KH) * BNC * 128 + (nchunks + nsplit - 1) / nsplit * BNC * 4;     // W buffers + the block's bias values
  if (SHMEM > 64 * 1024) return ODIC_EUNSUPPORTED;
  if (p.a_ln) {                       // LayerNorm-while-reading form: bf16 / fp32 output, no residual
    if (p.residual || (p.ld_aln & 3) || ((uintptr_t)p.a_ln & 15)) return ODIC_EUNSUPPORTED;
    if (out_dtype == ODIC_BF16) hipLaunchKernelGGL((gemm_bf16_apanel_kernel<MI, NI, KT, KH, bf16_raw, false, true>), grid, block, SHMEM, stream, p, nsplit);
    else hipLaunchKernelGGL((gemm_bf16_apanel_kernel<MI, NI, KT, KH, float, false, true>), grid, block, SHMEM, stream, p, nsplit);
    return odic_launch_status();
  }
  if constexpr (MI <= 2) {
    if (p.residual) {
      if (out_dtype == ODIC_BF16) hipLaunchKernelGGL((gemm_bf16_apanel_kernel<MI, NI, KT, KH, bf16_raw, true>), grid, block, SHMEM, stream, p, nsplit);
      else hipLaunchKernelGGL((gemm_bf16_apanel_kernel<MI, NI, KT, KH, float, true>), grid, block, SHMEM, stream, p, nsplit);
      return odic_launch_status();
    }
  }
  if (out_dtype == ODIC_BF16) hipLaunchKernelGGL((gemm_bf16_apanel_kernel<MI, NI, KT, KH, bf16_raw, false>), grid, block, SHMEM, stream, p, nsplit);
  else hipLaunchKernelGGL((gemm_bf16_apanel_kernel<MI, NI, KT, KH, float, false>), grid, block, SHMEM, stream, p, nsplit);
  return odic_launch_status();
}

template <int NWM, int NWN, int MI, int NI, int NSTAGE, int BK = 64, bool FOLD = false, int KS = 1>
int launch_cfg(Params& p, int out_dtype, int batch, hipStream_t stream) {
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16;
  constexpr int SH_STAGES = NSTAGE * (BM + BN) * BK * 2 + BM * 8;      // + (mean, rstd) per row of the tile
  constexpr int SH_RED = KS == 2 ? NWM * NWN * MI * NI * 1024 : 0;     // K-split: one group's accumulators
  constexpr int SHMEM = SH_STAGES > SH_RED ? SH_STAGES : SH_RED;
  if (p.K % BK != 0) return ODIC_EINVAL;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  int pm, pn;                                 // XCD partition: the split with the least fabric traffic (odic_common.h)
  odic_xcd_partition(p.tiles_m, p.tiles_n, (double)p.M * p.K * 2.0, (double)p.N * p.K * 2.0,
                     32 * (NWM * NWN * KS <= 4 ? 2 : 1), &pm, &pn);
  p.pm = pm; p.pn = pn;
  int max_rect = 0;
  for (int xm = 0; xm < pm; ++xm)
    for (int xn = 0; xn < pn; ++xn) {
      const int r = ((xm + 1) * p.tiles_m / pm - xm * p.tiles_m / pm) * ((xn + 1) * p.tiles_n / pn - xn * p.tiles_n / pn);
      if (r > max_rect) max_rect = r;
    }
  dim3 grid(8 * max_rect, 1, batch), block(64 * NWM * NWN * KS);
  auto kb = gemm_bf16_nt_kernel<NWM, NWN, MI, NI, NSTAGE, BK, bf16_raw, 0, KS>;
  auto kf = gemm_bf16_nt_kernel<NWM, NWN, MI, NI, NSTAGE, BK, float, 0, KS>;
  if (SHMEM > 64 * 1024) {
    static bool done = false;       // idempotent; racing first calls set the same value
    if (!done) {
      (void)hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
      (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
      done = true;
    }
  }
  if constexpr (FOLD) {             // folded-LayerNorm forms exist for the tile configurations the tuner picks from
    auto kprod = gemm_bf16_nt_kernel<NWM, NWN, MI, NI, NSTAGE, BK, float, 1>;
    auto kcons = gemm_bf16_nt_kernel<NWM, NWN, MI, NI, NSTAGE, BK, bf16_raw, 2>;
    if (SHMEM > 64 * 1024) {
      static bool done2 = false;
      if (!done2) {
        (void)hipFuncSetAttribute((const void*)kprod, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
        (void)hipFuncSetAttribute((const void*)kcons, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
        done2 = true;
      }
    }
    if (p.out16) { hipLaunchKernelGGL(kprod, grid, block, SHMEM, stream, p); return odic_launch_status(); }
    if (p.ln_stats) {
      if (out_dtype != ODIC_BF16 || (p.K >> 5) > 48) return ODIC_EUNSUPPORTED;
      hipLaunchKernelGGL(kcons, grid, block, SHMEM, stream, p);
      return odic_launch_status();
    }
  } else {
    if (p.out16 || p.ln_stats) return ODIC_EUNSUPPORTED;
  }
  if (out_dtype == ODIC_BF16) hipLaunchKernelGGL(kb, grid, block, SHMEM, stream, p);
  else hipLaunchKernelGGL(kf, grid, block, SHMEM, stream, p);
  return odic_launch_status();
}


#ifdef ODIC_EXPERIMENTAL_GEMM
template <int NWM, int NWN, int MI, int NI, int NSTAGE, int BK = 64>
int launch_persist(Params& p, int out_dtype, int batch, hipStream_t stream) {
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16;
  constexpr int SHMEM = NSTAGE * (BM + BN) * BK * 2 + 16;
  if (p.K % BK != 0 || batch != 1 || !p.ws) return ODIC_EINVAL;
  if ((long)(BM - 1) * p.lda * 2 + 2L * p.K >= 0x7fffffffL || (long)(BN - 1) * p.ldw * 2 + 2L * p.K >= 0x7fffffffL)
    return ODIC_EINVAL;                        // 32-bit byte offsets inside a tile's buffer resource
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  if (p.tiles_m > 65535 || p.tiles_n > 65535) return ODIC_EINVAL;
  int pn = 1;
  while (pn < 8 && pn * 2 <= p.tiles_n && (double)p.N / pn * p.K * 2.0 > 2.5 * 1024 * 1024) pn *= 2;
  int pm = 8 / pn;
  while (pm > p.tiles_m && pm > 1) { pm /= 2; pn *= 2; }
  if (pn > p.tiles_n) { pn = 1; pm = 8; while (pm > p.tiles_m && pm > 1) pm /= 2; pn = 8 / pm; }
  p.pm = pm; p.pn = pn;
  auto kb = gemm_bf16_nt_persist_kernel<NWM, NWN, MI, NI, NSTAGE, BK, bf16_raw>;
  auto kf = gemm_bf16_nt_persist_kernel<NWM, NWN, MI, NI, NSTAGE, BK, float>;
  static int per_cu = 0;                       // resident blocks per CU of this instantiation (code-object property)
  if (!per_cu) {
    if (SHMEM > 64 * 1024) {
      (void)hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
      (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    }
    // resident blocks per CU from the code object's own numbers (LDS, registers, wave slots); blocks never wait
    // for each other, so an estimate that is one too high or too low only costs a little speed
    int nb = (160 * 1024) / SHMEM;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void*)kb) == hipSuccess && fa.numRegs > 0) {
      const int alloc = (fa.numRegs + 7) / 8 * 8;
      const int waves_per_simd = 512 / alloc < 8 ? 512 / alloc : 8;
      const int by_regs = waves_per_simd * 4 / (NWM * NWN);
      if (by_regs < nb) nb = by_regs;
    }
    if (32 / (NWM * NWN) < nb) nb = 32 / (NWM * NWN);
    per_cu = nb < 1 ? 1 : nb;
    if (getenv("ODIC_GEMM_DEBUG"))
      fprintf(stderr, "[odic_gemm] persistent %dx%dx%d stages %d: %d B LDS, %d regs -> %d blocks/CU\n", BM, BN, BK, NSTAGE,
              SHMEM, fa.numRegs, per_cu);
  }
  const long tiles = (long)p.tiles_m * p.tiles_n;
  const long slots = 256L * per_cu;
  dim3 grid((unsigned)(tiles < slots ? tiles : slots)), block(64 * NWM * NWN);
  if (out_dtype == ODIC_BF16) hipLaunchKernelGGL(kb, grid, block, SHMEM, stream, p);
  else hipLaunchKernelGGL(kf, grid, block, SHMEM, stream, p);
  return odic_launch_status();
}

int launch_256sq(Params& p, int out_dtype, int batch, hipStream_t stream) {
  constexpr int SHMEM = 128 * 1024;
  if (p.K % 128 != 0) return ODIC_EINVAL;
  if (256L * p.lda * 2 + 2L * p.K >= 0x7fffffffL || 256L * p.ldw * 2 + 2L * p.K >= 0x7fffffffL)
    return ODIC_EINVAL;                        // 32-bit byte offsets inside a tile's buffer resource
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  int pn = 1;
  while (pn < 8 && pn * 2 <= p.tiles_n && (double)p.N / pn * p.K * 2.0 > 2.5 * 1024 * 1024) pn *= 2;
  int pm = 8 / pn;
  while (pm > p.tiles_m && pm > 1) { pm /= 2; pn *= 2; }
  if (pn > p.tiles_n) { pn = 1; pm = 8; while (pm > p.tiles_m && pm > 1) pm /= 2; pn = 8 / pm; }
  p.pm = pm; p.pn = pn;
  int max_rect = 0;
  for (int xm = 0; xm < pm; ++xm)
    for (int xn = 0; xn < pn; ++xn) {
      const int r = ((xm + 1) * p.tiles_m / pm - xm * p.tiles_m / pm) * ((xn + 1) * p.tiles_n / pn - xn * p.tiles_n / pn);
      if (r > max_rect) max_rect = r;
    }
  dim3 grid(8 * max_rect, 1, batch), block(512);
  auto kb = gemm_bf16_256sq_kernel<bf16_raw>;
  auto kf = gemm_bf16_256sq_kernel<float>;
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    done = true;
  }
  if (out_dtype == ODIC_BF16) hipLaunchKernelGGL(kb, grid, block, SHMEM, stream, p);
  else hipLaunchKernelGGL(kf, grid, block, SHMEM, stream, p);
  return odic_launch_status();
}

#endif  // ODIC_EXPERIMENTAL_GEMM

// The default build carries the five tile configurations the host's tuner chooses from (0, 1, 7, 10) or the built-in
// model falls back to (2).  Everything else that was built and measured on the way — more stages, 16-wave and 4-wave
// 256-wide tiles, out-of-phase residents, the persistent and the 256x256 phase-pipelined kernels, the LayerNorm fold
// across two products — compiles only with -DODIC_EXPERIMENTAL_GEMM (make EXTRA=-DODIC_EXPERIMENTAL_GEMM): none of it
// is ever selected, and the folded-LayerNorm consumer is the one place where hipcc emits v_pk_fma_f32 with op_sel
// source selection, the instruction form behind the round-2 wrong-row incident.
#ifdef ODIC_EXPERIMENTAL_GEMM
constexpr bool kFold = true;
#else
constexpr bool kFold = false;
#endif

}  // namespace

int odic_gemm_bf16_launch(const odic_gemm_args* a, hipStream_t stream) {
  if (a->ln_colsum && !a->ln_stats) return ODIC_EUNSUPPORTED;          // (the in-kernel moments form is fp32 skinny only)
  if (a->K % 64 != 0 || (a->A && a->lda % 8 != 0) || a->ldw % 8 != 0) return ODIC_EINVAL;
  if (((uintptr_t)a->A & 15) || ((uintptr_t)a->W & 15)) return ODIC_EINVAL;
  if (a->a_ln && (a->tile_cfg < 50 || a->tile_cfg > 53)) return ODIC_EUNSUPPORTED;   // (A-resident kernels only)
  if (!a->a_ln && !a->A) return ODIC_EINVAL;
  if ((a->strideA % 8) || (a->strideW % 8)) return ODIC_EINVAL;
  Params p;
  p.skew_from = 0; p.skew_to = 0; p.skew_sleeps = 0;
  p.A = (const bf16_raw*)a->A; p.W = (const bf16_raw*)a->W; p.bias = a->bias; p.residual = a->residual;
  p.out = a->out; p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldr = a->ldr; p.ldc = a->ldc;
  p.strideA = a->strideA; p.strideW = a->strideW; p.strideBias = a->strideBias;
  p.strideR = a->strideR; p.strideC = a->strideC;
  p.alpha = a->alpha; p.act = a->act; p.bias_axis = a->bias_axis;
  p.ws = a->workspace;
  p.out16 = (bf16_raw*)a->out16; p.ld16 = a->ld16; p.stats_out = a->stats_out;
  p.ln_stats = a->ln_stats; p.ln_colsum = a->ln_colsum; p.ln_eps = a->ln_eps;
  p.a_ln = a->a_ln; p.ld_aln = a->ld_aln;
  if (p.out16) {          // producer of a folded LayerNorm: whole 32-column groups, vector stores, fp32 output
    if (!p.stats_out || a->out_dtype != ODIC_F32 || a->batch != 1 || (a->N & 31) || (a->ldc & 7) || (a->ld16 & 7) ||
        ((uintptr_t)a->out16 & 15) || ((uintptr_t)a->out & 15) || (a->residual && (a->ldr & 3)))
      return ODIC_EINVAL;
  }
  if (p.ln_stats) {       // consumer: A's row moments come in K/32 groups (a multiple of 4, at most 48)
    if (!p.ln_colsum || a->batch != 1 || (a->K & 31) || a->K > 1536 || a->bias_axis != 0) return ODIC_EINVAL;
  }
  // Tile choice = fewest "rounds x per-tile cost": a launch runs in ceil(tiles / resident slots)
  // rounds (256 CUs x 3 / 2 / 1 blocks for the 128x64 / 128x128 / 256x256 tiles, set by their LDS
  // footprints); relative per-tile costs 1 : 1.38 : 2.6 were measured on MI355X over the Swin-L
  // shapes (tools/gemm_tune.py; profiles/r01_gemm_tile_sweep.txt).
  int cfg = a->tile_cfg;
  if (cfg < 0) {
    auto rounds = [&](int bm, int bn, int slots) {
      const long t = (long)((a->M + bm - 1) / bm) * ((a->N + bn - 1) / bn) * a->batch;
      return (double)((t + slots - 1) / slots);
    };
    const double c0 = rounds(128, 64, 768) * 1.0, c1 = rounds(128, 128, 512) * 1.38;
    const double c2 = (a->N % 256 == 0 && !p.out16 && !p.ln_stats) ? rounds(256, 256, 256) * 2.6 : 1e30;   // (no folded-LN form)
    cfg = (c0 <= c1 && c0 <= c2) ? 0 : (c1 <= c2 ? 1 : 2);
  }
  switch (cfg) {
    case 0: return launch_cfg<2, 2, 4, 2, 2, 64, kFold>(p, a->out_dtype, a->batch, stream);     // 128 x 64, 2 stages
    case 1: return launch_cfg<2, 2, 4, 4, 2, 64, kFold>(p, a->out_dtype, a->batch, stream);     // 128 x 128
    case 2: return launch_cfg<2, 4, 8, 4, 2>(p, a->out_dtype, a->batch, stream);     // 256 x 256
    case 7: return launch_cfg<4, 2, 4, 4, 2, 32, kFold>(p, a->out_dtype, a->batch, stream); // 256 x 128 x 32 (48 KiB)
    case 10: return launch_cfg<4, 2, 4, 4, 3, 32, kFold>(p, a->out_dtype, a->batch, stream); // 256 x 128 x 32, 3 stages (72 KiB)
    // 144- / 288-row tiles of 48 x 96 wave patches (MI = 3, NI = 6): the Swin-L token counts carry factors of 9
    // (9216 = 32·288, 2304 = 16·144), so these tile grids are exact multiples of the 256 compute units where the
    // power-of-two tiles leave a ragged last round (9216 x 3072: 512 tiles of 288 x 192 = two full rounds of one
    // 12-wave block per CU, against 1.69 rounds of 256 x 128).  One block per CU; the fill traffic per MFMA is a
    // quarter below the 256 x 128 tile's.
    case 40: return launch_cfg<6, 2, 3, 6, 2, 64>(p, a->out_dtype, a->batch, stream);  // 288 x 192, 12 waves (120 KiB)
    case 41: return launch_cfg<3, 3, 3, 6, 2, 64>(p, a->out_dtype, a->batch, stream);  // 144 x 288,  9 waves (108 KiB)
    case 42: return launch_cfg<3, 2, 3, 6, 2, 64>(p, a->out_dtype, a->batch, stream);  // 144 x 192,  6 waves, 2 stages (84 KiB)
    case 43: return launch_cfg<3, 2, 3, 6, 3, 64>(p, a->out_dtype, a->batch, stream);  // 144 x 192,  6 waves, 3 stages (126 KiB)
    case 44: return launch_cfg<3, 1, 3, 6, 2, 64>(p, a->out_dtype, a->batch, stream);  // 144 x 96,   3 waves, 2 stages (60 KiB)
    case 45: return launch_cfg<3, 3, 3, 6, 3, 32>(p, a->out_dtype, a->batch, stream);  // 144 x 288 x 32, 9 waves, 3 stages (81 KiB)
    case 46: return launch_cfg<3, 1, 3, 6, 3, 64>(p, a->out_dtype, a->batch, stream);  // 144 x 96,   3 waves, 3 stages (90 KiB)
    case 47: return launch_cfg<3, 2, 3, 6, 2, 64, false, 2>(p, a->out_dtype, a->batch, stream);  // 144 x 192, TWO K groups of 6 waves (108 KiB)
    // 64 x 64 tiles (4 waves of 32 x 32): the expansion encoder's products are 2304 rows (or 16 batches of 144) by 512 columns —
    // 72 tiles of 128 x 128 leave 184 of the 256 CUs idle while each tile walks a K of 512 ... 2048 alone
    case 48: return launch_cfg<2, 2, 2, 2, 3, 64>(p, a->out_dtype, a->batch, stream);  // 64 x 64, 3 stages (48 KiB)
    case 49: return launch_cfg<2, 2, 2, 2, 2, 64>(p, a->out_dtype, a->batch, stream);  // 64 x 64, 2 stages (32 KiB)
    // A-resident streaming kernels for K = 192 / 384 (whole tiles only; see gemm_bf16_apanel_kernel)
    case 50: return launch_apanel<4, 4, 3>(p, a->out_dtype, a->batch, stream);  // K = 192: 256-row panels, 64-column chunks
    case 51: return launch_apanel<2, 2, 6>(p, a->out_dtype, a->batch, stream);  // K = 384: 128-row panels, 32-column chunks
    case 52: return launch_apanel<2, 4, 3>(p, a->out_dtype, a->batch, stream);  // K = 192: 128-row panels, 64-column chunks
    case 53: return launch_apanel<2, 4, 6, 2>(p, a->out_dtype, a->batch, stream);  // K = 384: 128-row panels, 64-column chunks in two K pieces
#ifdef ODIC_EXPERIMENTAL_GEMM
    case 3: return launch_cfg<2, 2, 4, 2, 3>(p, a->out_dtype, a->batch, stream);     // 128 x 64, 3 stages
    case 4: return launch_cfg<2, 2, 4, 4, 3>(p, a->out_dtype, a->batch, stream);     // 128 x 128, 3 stages
    case 5: return launch_cfg<4, 2, 4, 4, 3>(p, a->out_dtype, a->batch, stream);     // 256 x 128, 3 stages (144 KiB)
    case 6: return launch_cfg<2, 2, 4, 2, 4>(p, a->out_dtype, a->batch, stream);     // 128 x 64, 4 stages
    case 8: return launch_cfg<2, 2, 4, 4, 2, 32>(p, a->out_dtype, a->batch, stream); // 128 x 128 x 32 (32 KiB)
    case 9: return launch_cfg<2, 4, 8, 4, 2, 32>(p, a->out_dtype, a->batch, stream); // 256 x 256 x 32 (64 KiB)
    case 11: return launch_cfg<2, 4, 8, 4, 3, 32>(p, a->out_dtype, a->batch, stream); // 256 x 256 x 32, 3 stages (96 KiB)
    case 13: return launch_cfg<4, 4, 4, 4, 3, 32>(p, a->out_dtype, a->batch, stream); // 256 x 256 x 32, 16 waves of 64 x 64, 3 stages (96 KiB)
    case 14: return launch_cfg<4, 4, 4, 4, 2, 64>(p, a->out_dtype, a->batch, stream); // 256 x 256 x 64, 16 waves of 64 x 64, 2 stages (128 KiB)
    case 15: return launch_cfg<4, 4, 4, 4, 4, 32>(p, a->out_dtype, a->batch, stream); // 256 x 256 x 32, 16 waves, 4 stages (128 KiB)
    // 4 waves of 128 x 64 (64 x 128): 12 KiB of LDS reads per 32 MFMAs instead of 8 KiB per 16, two blocks per CU
    case 28: return launch_cfg<2, 2, 8, 4, 3, 32>(p, a->out_dtype, a->batch, stream); // 256 x 128 x 32, 3 stages (72 KiB)
    case 29: return launch_cfg<2, 2, 4, 8, 3, 32>(p, a->out_dtype, a->batch, stream); // 128 x 256 x 32, 3 stages (72 KiB)
    case 30: return launch_cfg<2, 2, 8, 4, 2, 32>(p, a->out_dtype, a->batch, stream); // 256 x 128 x 32, 2 stages (48 KiB)
    case 31: return launch_cfg<2, 2, 8, 4, 2, 64>(p, a->out_dtype, a->batch, stream); // 256 x 128 x 64, 2 stages (96 KiB)
    // 32 / 33: config 10 with the blocks that become the SECOND resident block of a CU (ids 256..511) started 8 / 12 us
    // late, so that the two blocks of a CU are out of phase — one's pipeline fill and store tail under the other's
    // K-loop — instead of running prologue, loop and epilogue in lockstep (fc1 / fc2 of stage 2: -5 %)
    case 32: p.skew_from = 256; p.skew_to = 512; p.skew_sleeps = 2;
             return launch_cfg<4, 2, 4, 4, 3, 32, kFold>(p, a->out_dtype, a->batch, stream);
    case 33: p.skew_from = 256; p.skew_to = 512; p.skew_sleeps = 3;
             return launch_cfg<4, 2, 4, 4, 3, 32, kFold>(p, a->out_dtype, a->batch, stream);
    case 12: if (p.out16 || p.ln_stats) return ODIC_EUNSUPPORTED;
             return launch_256sq(p, a->out_dtype, a->batch, stream);                 // 256 x 256 x 64, 4 phases per K-tile (128 KiB)
    // 16 + c: tile config c as a persistent, dynamically scheduled launch (needs args->workspace, batch == 1)
    case 16: case 17: case 18: case 19: case 20: case 21: case 23: case 24: case 25: case 26: case 27:
      if (p.out16 || p.ln_stats) return ODIC_EUNSUPPORTED;
      break;
    default: break;
  }
  switch (cfg) {
    case 16: return launch_persist<2, 2, 4, 2, 2>(p, a->out_dtype, a->batch, stream);
    case 17: return launch_persist<2, 2, 4, 4, 2>(p, a->out_dtype, a->batch, stream);
    case 18: return launch_persist<2, 4, 8, 4, 2>(p, a->out_dtype, a->batch, stream);
    case 19: return launch_persist<2, 2, 4, 2, 3>(p, a->out_dtype, a->batch, stream);
    case 20: return launch_persist<2, 2, 4, 4, 3>(p, a->out_dtype, a->batch, stream);
    case 21: return launch_persist<4, 2, 4, 4, 3>(p, a->out_dtype, a->batch, stream);
    case 23: return launch_persist<4, 2, 4, 4, 2, 32>(p, a->out_dtype, a->batch, stream);
    case 24: return launch_persist<2, 2, 4, 4, 2, 32>(p, a->out_dtype, a->batch, stream);
    case 25: return launch_persist<2, 4, 8, 4, 2, 32>(p, a->out_dtype, a->batch, stream);
    case 26: return launch_persist<4, 2, 4, 4, 3, 32>(p, a->out_dtype, a->batch, stream);
    case 27: return launch_persist<2, 4, 8, 4, 3, 32>(p, a->out_dtype, a->batch, stream);
    default: return ODIC_EINVAL;
  }
#else
    default: return ODIC_EINVAL;             // (tile configurations 3-6, 8, 9, 11-33: -DODIC_EXPERIMENTAL_GEMM builds only)
  }
#endif
}
