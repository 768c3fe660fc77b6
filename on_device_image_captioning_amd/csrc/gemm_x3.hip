// Split-fp16 ("h2") NT GEMM with fused epilogue for gfx950 — the contraction of the near-exact fast mode
// (`precision='x3'`):
//
//     out = act(alpha · A·Wᵀ + bias) + residual,     A·Wᵀ ≈ Ah·Whᵀ + Ah·Wlᵀ + Al·Whᵀ
//
// Both operands are carried as hi + lo fp16 pairs (odic_common.h: 22 significand bits, 4 bytes per element in
// groups of [8 hi | 8 lo]); every 16x16x32 step issues THREE v_mfma_f32_16x16x32_f16 into one fp32 accumulator
// (the dropped Al·Wl term is 2^-22 of the product).  Against the exact fp32 MFMA (1/16 of the fp16 rate) that
// is a 5x higher matrix-pipe ceiling at fp32-class accuracy; against the bf16 kernel the LDS image carries twice
// the bytes per K but feeds 1.5x the MFMAs per staged byte, which is what these L2→LDS-fill-bound tiles want.
//
// Same structure as gemm_bf16.hip's one-block-per-tile kernel (LDS-DMA staging, XOR chunk swizzle on the source
// address and on the fragment read, counted vmcnt + raw barrier, operands swapped so a lane owns 8 adjacent
// output columns of a row, W rows staged permuted, XCD-aware tile partition).  What differs:
//   * a K-tile is 32 elements = one 128-byte LDS row per matrix row, staged PLANAR: LDS chunks 0..3 hold the hi
//     fragments of k-groups 0..3, chunks 4..7 their lo fragments (the per-lane DMA source address does the
//     de-interleave for free), so the hi / lo fragment of lane (row, fq) is chunk fq / 4 + fq — the conflict-free
//     ds_read_b128 pattern of the bf16 kernel, twice;
//   * out_dtype ODIC_H2: the lane's 8 adjacent columns are exactly one h2 group → 32 contiguous bytes (hi | lo).
//   * GELU is the exact erf form: this mode exists to reproduce the fp32 reference's captions.
#include "odic_common.h"
#include <type_traits>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct Params {
  const char* A; const char* W; const float* bias; const float* residual; void* out;
  int M, N, K;                // K in elements
  long lda, ldw, ldr, ldc;    // elements (4 bytes each for h2 operands)
  long strideA, strideW, strideBias, strideR, strideC;
  float alpha; int act; int bias_axis;
  int tiles_m, tiles_n, pm, pn;
};

constexpr int ROWB = 128;      // bytes per LDS row = 32 h2 elements

__device__ __forceinline__ int swz(int chunk, int row) { return chunk ^ (row & 7); }
__device__ __forceinline__ int wperm(int r) {
  return (r & ~31) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3);
}

template <int NWM, int NWN, int MI, int NI, int NSTAGE, typename OutT>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_x3_nt_kernel(Params p) {
  constexpr int NW = NWM * NWN;
  constexpr int BK = 32;
  constexpr int RPI = 1024 / ROWB;             // 8 rows per 1-KiB DMA instruction
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / RPI / NW, W_INSTR = BN / RPI / NW;
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile rows must split evenly over the waves");
  constexpr int G = A_INSTR + W_INSTR;
  static_assert(NI % 2 == 0, "the epilogue pairs MFMA column tiles");
  constexpr int D = NSTAGE - 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
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
  const int m0 = tm * BM, n0 = tn * BN;
  const long bz = blockIdx.z;
  const char* A = p.A + bz * p.strideA * 4;
  const char* W = p.W + bz * p.strideW * 4;

  // LDS-DMA source: LDS slot `pos` (0..7) of a row holds logical image chunk c = swz(pos): c < 4 → hi fragment of
  // k-group c (memory chunk 2c of the 128-byte K-tile), c >= 4 → lo fragment of k-group c-4 (memory chunk 2(c-4)+1)
  const int srow = lane >> 3;
  const int simg = swz(lane & 7, srow);
  const int smem_chunk = simg < 4 ? 2 * simg : 2 * (simg - 4) + 1;
  const char* a_src[A_INSTR];
  const char* w_src[W_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * RPI + srow;
    a_src[i] = A + (long)min(m0 + row, p.M - 1) * p.lda * 4 + smem_chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < W_INSTR; ++i) {
    const int row = (i * NW + wave) * RPI + srow;
    w_src[i] = W + (long)min(n0 + wperm(row), p.N - 1) * p.ldw * 4 + smem_chunk * 16;
  }
  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * STAGE;
    char* lw = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + (long)kt * ROWB), (lptr_t)(la + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + (long)kt * ROWB), (lptr_t)(lw + (i * NW + wave) * 1024), 16, 0, 0);
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nk) stage(t, t);

  const int ch_hi = swz(fq, frow) << 4, ch_lo = swz(4 + fq, frow) << 4;
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
    f16x8_t ah[MI], al[MI], wh[NI], wl[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[mi] = *(const f16x8_t*)(la + mi * 16 * ROWB + ch_hi);
      al[mi] = *(const f16x8_t*)(la + mi * 16 * ROWB + ch_lo);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      wh[ni] = *(const f16x8_t*)(lw + ni * 16 * ROWB + ch_hi);
      wl[ni] = *(const f16x8_t*)(lw + ni * 16 * ROWB + ch_lo);
    }
    // the two small cross terms first, the dominant hi·hi product last
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], ah[mi], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], al[mi], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], ah[mi], acc[mi][ni], 0, 0, 0);
      }
  }

  // ---- epilogue: lane (frow, fq) owns output row frow, 8 adjacent columns 32q + 8·fq .. +7 of each column pair
  const float* bias = p.bias ? p.bias + bz * p.strideBias : nullptr;
  const float* resid = p.residual ? p.residual + bz * p.strideR : nullptr;
  constexpr bool OUT_H2 = __is_same(OutT, h2_t);
  char* out = (char*)p.out + bz * p.strideC * 4;
  const bool ld_ok = ((p.ldc & 7) == 0) && (!resid || (p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 31) == 0);
  // The vector path (as gemm_bf16.hip's): vmcnt retires in order and counts stores, so a bias or residual load between two
  // groups of stores waits for every store before it — every bias value is requested before the first store, the residual
  // rows of column group g + 1 before the stores of group g (4- and 6-wave blocks; wider blocks request a group's rows together).
  if (ld_ok && (p.N & 7) == 0 && !p.bias_axis && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) {
    constexpr int NG = NI / 2;
    const int cw = n0 + wn * NI * 16 + fq * 8;
    f32x4_t bc[NG][2];
#pragma unroll
    for (int nq = 0; nq < NG; ++nq) bc[nq][0] = bc[nq][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int nq = 0; nq < NG; ++nq) {
        const f32x4_t* bp = (const f32x4_t*)(bias + min(cw + nq * 32, p.N - 8));
        bc[nq][0] = bp[0]; bc[nq][1] = bp[1];
      }
    }
    auto store_groups = [&](auto has_res) {
      constexpr bool HR = decltype(has_res)::value;
      constexpr int RD = HR ? (NW <= 6 ? 2 : 1) : 0;
      f32x4_t rv[RD == 2 ? 2 : 1][RD ? MI : 1][2];
      auto loadg = [&](int nq, int slot) {
        if constexpr (RD > 0) {
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
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = acc[mi][2 * nq + (e >> 2)][e & 3] * p.alpha + bc[nq][e >> 2][e & 3];
            v[e] = apply_act<false>(x, p.act);
          }
          if constexpr (HR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[RD == 2 ? (nq & 1) : 0][mi][e >> 2][e & 3];
          }
          if (row < p.M && col < p.N) {
            if constexpr (OUT_H2) {
              h2_store8((h2_t*)out + (long)row * p.ldc + col, v);
            } else {
              float* dst = (float*)out + (long)row * p.ldc + col;
              ((f32x4_t*)dst)[0] = f32x4_t{v[0], v[1], v[2], v[3]};
              ((f32x4_t*)dst)[1] = f32x4_t{v[4], v[5], v[6], v[7]};
            }
          }
        }
      }
    };
    if (resid) store_groups(std::true_type{});
    else store_groups(std::false_type{});
    return;
  }
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
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = acc[mi][2 * nq + (e >> 2)][e & 3] * p.alpha + bc[e] + brow;
        v[e] = apply_act<false>(x, p.act);
      }
      const bool full = ld_ok && col + 7 < p.N;
      if (resid) {
        if (full) {
          const f32x4_t* rp = (const f32x4_t*)(resid + (long)row * p.ldr + col);
          const f32x4_t r0 = rp[0], r1 = rp[1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < p.N) v[e] += resid[(long)row * p.ldr + col + e];
        }
      }
      if constexpr (OUT_H2) {
        h2_t* drow = (h2_t*)out + (long)row * p.ldc;
        if (full) {
          h2_store8(drow + col, v);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < p.N) h2_store1(drow, col + e, v[e]);
        }
      } else {
        float* dst = (float*)out + (long)row * p.ldc + col;
        if (full) {
          ((f32x4_t*)dst)[0] = f32x4_t{v[0], v[1], v[2], v[3]};
          ((f32x4_t*)dst)[1] = f32x4_t{v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < p.N) dst[e] = v[e];
        }
      }
    }
  }
}

// =================================================================================================
// A-resident streaming form for the K = 192 products of Swin stage 0 (tile configs 20 / 21) — gemm_bf16.hip's
// gemm_bf16_apanel_kernel on split-fp16 operands.  Those products are output-bandwidth-bound (fc1: 453 MB of h2 output for
// 44 GFLOP) and the tiled kernel spends 2-3x a device copy's time on them (profiles/r03_gemm_x3_tile_sweep.txt: 260 us).
//   * a wave keeps its 16·MI rows of A over the whole K as hi / lo MFMA fragments in registers (2 x 4 registers per row
//     tile and 32-deep step) — read once from the h2 tensor, or, LNA, from the fp32 rows of the residual stream, normalised
//     in registers and split into hi + lo there (LayerNorm while reading: gamma / beta folded into W / bias by the caller);
//   * the four waves walk the output columns in chunks of 16·NI; a W chunk (16·NI rows x K in the planar swizzled image of
//     the tiled kernel, 24 KiB) arrives by LDS-DMA one chunk ahead; one barrier per chunk; the counted wait in front of it
//     is vmcnt(stores of a chunk): the chunk's DMA was issued before the previous chunk's stores;
//   * bias through LDS; whole tiles only (refused otherwise: the caller's tuner falls back to the tiled kernel).
// Same MFMAs in the same order as the tiled kernel: bit-identical (tests/test_x3_gpu.py).
// =================================================================================================
struct PanelParams {
  const char* A; const float* a_ln; const char* W; const float* bias; void* out;
  int M, N, K;
  long lda, ld_aln, ldw, ldc;
  float alpha, ln_eps; int act;
};

template <int MI, int NI, int KT, typename OutT, bool LNA>
__global__ __launch_bounds__(256, 2) void gemm_x3_apanel_kernel(PanelParams p, int nsplit) {
  constexpr int NW = 4;
  constexpr int BM = NW * MI * 16, BNC = NI * 16;
  constexpr int SUB = BNC * ROWB;                        // one 32-deep K-tile of a W chunk
  constexpr int CHUNK = KT * SUB;
  static_assert(CHUNK % (1024 * NW) == 0 && NI % 2 == 0, "whole DMA instructions per wave; the epilogue pairs column tiles");
  constexpr int INSTR = CHUNK / 1024 / NW;
  constexpr int RG = BNC / 8;
  constexpr bool OUT_H2 = __is_same(OutT, h2_t);
  constexpr int NST = MI * (NI / 2) * 2;                 // 16-byte stores per wave per chunk (32 bytes per lane and pair)
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

  f16x8_t ah[KT][MI], al[KT][MI];
  if constexpr (LNA) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const float* xr = p.a_ln + (long)(m0 + (wave * MI + mi) * 16 + frow) * p.ld_aln + fq * 8;
      f32x4_t xv[KT][2];
#pragma unroll
      for (int k = 0; k < KT; ++k) { xv[k][0] = *(const f32x4_t*)(xr + k * 32); xv[k][1] = *(const f32x4_t*)(xr + k * 32 + 4); }
      f32x4_t s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KT; ++k) s4 += xv[k][0] + xv[k][1];
      float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      asm volatile("" : "+v"(sum));       // (keeps hipcc's SLP pass from pairing the row tiles' scalar moments into packed
                                          //  fp32 ops with `op_sel` source selection — tests/test_isa_lint.py)
      const float mean = sum / (float)(32 * KT);       // (a true division: a constant row then normalises to exactly 0)
      f32x4_t q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        xv[k][0] -= mean; xv[k][1] -= mean;
        q4 += xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1];
      }
      float ssq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
      ssq += __shfl_xor(ssq, 16, 64);
      ssq += __shfl_xor(ssq, 32, 64);
      asm volatile("" : "+v"(ssq));
      const float rstd = rsqrtf(ssq / (float)(32 * KT) + p.ln_eps);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        f16x8_t h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) { f16_t hh, ll; h2_split(xv[k][e >> 2][e & 3] * rstd, hh, ll); h[e] = hh; l[e] = ll; }
        ah[k][mi] = h; al[k][mi] = l;
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const char* ar = p.A + ((long)(m0 + (wave * MI + mi) * 16 + frow) * p.lda) * 4 + fq * 32;
#pragma unroll
      for (int k = 0; k < KT; ++k) {             // k-group 4k + fq of the row: [8 hi | 8 lo]
        ah[k][mi] = *(const f16x8_t*)(ar + k * 128);
        al[k][mi] = *(const f16x8_t*)(ar + k * 128 + 16);
      }
    }
  }

  // ---- W chunk DMA: instruction i of this wave fills 1 KiB = rows 8·rg .. +7 of K-tile kt, j = i·NW + wave
  int w_off[INSTR];                                     // byte offsets inside a chunk
#pragma unroll
  for (int i = 0; i < INSTR; ++i) {
    const int j = i * NW + wave, kt = j / RG, rg = j - kt * RG;
    const int srow = lane >> 3, r = rg * 8 + srow;
    const int simg = swz(lane & 7, srow);
    const int smem_chunk = simg < 4 ? 2 * simg : 2 * (simg - 4) + 1;
    w_off[i] = wperm(r) * (int)p.ldw * 4 + kt * ROWB + smem_chunk * 16;
  }
  auto issue = [&](int c, int buf) {
    const char* wb = p.W + (long)c * BNC * p.ldw * 4;
    char* lb = lds + buf * CHUNK;
#pragma unroll
    for (int i = 0; i < INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wb + w_off[i]), (lptr_t)(lb + (i * NW + wave) * 1024), 16, 0, 0);
  };
  char* out = (char*)p.out;
  issue(c0, 0);
  float* sbias = (float*)(lds + 2 * CHUNK);
  for (int t = tid; t < (c1 - c0) * BNC; t += 256) sbias[t] = p.bias ? p.bias[c0 * BNC + t] : 0.f;
  __builtin_amdgcn_s_waitcnt(0x0070);                                  // vmcnt(0) lgkmcnt(0) (the builtin: hipcc must see it)
  const int ch_hi = swz(fq, frow) << 4, ch_lo = swz(4 + fq, frow) << 4;
  for (int c = c0; c < c1; ++c) {
    if (c != c0) __builtin_amdgcn_s_waitcnt(0x0F70 | (NST & 15) | ((NST >> 4) << 14));      // vmcnt(NST)
    __builtin_amdgcn_s_barrier();
    const int buf = (c - c0) & 1;
    if (c + 1 < c1) issue(c + 1, buf ^ 1);

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* lw = lds + buf * CHUNK + frow * ROWB;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      f16x8_t wh[NI], wl[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        wh[ni] = *(const f16x8_t*)(lw + kt * SUB + ni * 16 * ROWB + ch_hi);
        wl[ni] = *(const f16x8_t*)(lw + kt * SUB + ni * 16 * ROWB + ch_lo);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], ah[kt][mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], al[kt][mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], ah[kt][mi], acc[mi][ni], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nq = 0; nq < NI / 2; ++nq) {
      const int col = c * BNC + nq * 32 + fq * 8;
      const f32x4_t* sb = (const f32x4_t*)(sbias + (c - c0) * BNC + nq * 32 + fq * 8);
      const f32x4_t b0 = sb[0], b1 = sb[1];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = m0 + (wave * MI + mi) * 16 + frow;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = acc[mi][2 * nq + (e >> 2)][e & 3] * p.alpha + (e < 4 ? b0[e & 3] : b1[e & 3]);
          v[e] = apply_act<false>(x, p.act);
        }
        if constexpr (OUT_H2) {
          h2_store8((h2_t*)out + (long)row * p.ldc + col, v);
        } else {
          float* dst = (float*)out + (long)row * p.ldc + col;
          ((f32x4_t*)dst)[0] = f32x4_t{v[0], v[1], v[2], v[3]};
          ((f32x4_t*)dst)[1] = f32x4_t{v[4], v[5], v[6], v[7]};
        }
      }
    }
  }
}

template <int MI, int NI, int KT>
int launch_panel(const odic_gemm_args* a, hipStream_t stream) {
  constexpr int BM = 4 * MI * 16, BNC = NI * 16;
  if (a->batch != 1 || a->K != KT * 32 || a->M % BM != 0 || a->N % BNC != 0 || a->bias_axis != 0 || a->residual)
    return ODIC_EUNSUPPORTED;
  if ((a->ldc & 7) || ((uintptr_t)a->out & 31) || (long)BNC * a->ldw * 4 >= (1L << 30)) return ODIC_EUNSUPPORTED;
  if (a->a_ln && ((a->ld_aln & 3) || ((uintptr_t)a->a_ln & 15))) return ODIC_EUNSUPPORTED;
  PanelParams p;
  p.A = (const char*)a->A; p.a_ln = a->a_ln; p.W = (const char*)a->W; p.bias = a->bias; p.out = a->out;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ld_aln = a->ld_aln; p.ldw = a->ldw; p.ldc = a->ldc;
  p.alpha = a->alpha; p.ln_eps = a->ln_eps; p.act = a->act;
  const int panels = a->M / BM, nchunks = a->N / BNC;
  int nsplit = nchunks;
  for (int d = 1; d <= nchunks; ++d)
    if (nchunks % d == 0 && (long)panels * d >= 1536) { nsplit = d; break; }
  dim3 grid(8 * ((panels + 7) / 8) * nsplit), block(256);
  const int SHMEM = 2 * KT * BNC * ROWB + (nchunks + nsplit - 1) / nsplit * BNC * 4;
  if (SHMEM > 64 * 1024) return ODIC_EUNSUPPORTED;
  const bool h2 = a->out_dtype == ODIC_H2;
  if (a->a_ln) {
    if (h2) hipLaunchKernelGGL((gemm_x3_apanel_kernel<MI, NI, KT, h2_t, true>), grid, block, SHMEM, stream, p, nsplit);
    else hipLaunchKernelGGL((gemm_x3_apanel_kernel<MI, NI, KT, float, true>), grid, block, SHMEM, stream, p, nsplit);
  } else {
    if (h2) hipLaunchKernelGGL((gemm_x3_apanel_kernel<MI, NI, KT, h2_t, false>), grid, block, SHMEM, stream, p, nsplit);
    else hipLaunchKernelGGL((gemm_x3_apanel_kernel<MI, NI, KT, float, false>), grid, block, SHMEM, stream, p, nsplit);
  }
  return odic_launch_status();
}

template <int NWM, int NWN, int MI, int NI, int NSTAGE>
int launch(Params& p, int out_dtype, int batch, hipStream_t stream) {
  constexpr int BM = NWM * MI * 16, BN = NWN * NI * 16;
  constexpr int SHMEM = NSTAGE * (BM + BN) * ROWB;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  int pm, pn;                                 // XCD partition: the split with the least fabric traffic (odic_common.h)
  odic_xcd_partition(p.tiles_m, p.tiles_n, (double)p.M * p.K * 4.0, (double)p.N * p.K * 4.0, 32 * (NWM * NWN <= 4 ? 2 : 1), &pm, &pn);
  p.pm = pm; p.pn = pn;
  int max_rect = 0;
  for (int xm = 0; xm < pm; ++xm)
    for (int xn = 0; xn < pn; ++xn) {
      const int r = ((xm + 1) * p.tiles_m / pm - xm * p.tiles_m / pm) * ((xn + 1) * p.tiles_n / pn - xn * p.tiles_n / pn);
      if (r > max_rect) max_rect = r;
    }
  dim3 grid(8 * max_rect, 1, batch), block(64 * NWM * NWN);
  auto k32 = gemm_x3_nt_kernel<NWM, NWN, MI, NI, NSTAGE, float>;
  auto kh2 = gemm_x3_nt_kernel<NWM, NWN, MI, NI, NSTAGE, h2_t>;
  if (SHMEM > 64 * 1024) {
    static bool done = false;       // code-object attribute; idempotent
    if (!done) {
      (void)hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
      (void)hipFuncSetAttribute((const void*)kh2, hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
      done = true;
    }
  }
  if (out_dtype == ODIC_F32) hipLaunchKernelGGL(k32, grid, block, SHMEM, stream, p);
  else hipLaunchKernelGGL(kh2, grid, block, SHMEM, stream, p);
  return odic_launch_status();
}

}  // namespace

int odic_gemm_x3_launch(const odic_gemm_args* a, hipStream_t stream) {
  if (a->ln_colsum || a->col_scale || a->out16 || a->ln_stats) return ODIC_EUNSUPPORTED;
  if (a->out_dtype != ODIC_F32 && a->out_dtype != ODIC_H2) return ODIC_EINVAL;
  // h2 rows are whole [8 hi | 8 lo] groups and K-tiles are four of them
  if (a->K % 32 != 0 || (a->A && a->lda % 8 != 0) || a->ldw % 8 != 0 || (a->strideA % 8) || (a->strideW % 8)) return ODIC_EINVAL;
  if (((uintptr_t)a->A & 31) || ((uintptr_t)a->W & 31)) return ODIC_EINVAL;
  if (a->a_ln && a->tile_cfg != 20 && a->tile_cfg != 21) return ODIC_EUNSUPPORTED;   // (A-resident kernels only)
  if (!a->a_ln && !a->A) return ODIC_EINVAL;
  if (a->tile_cfg == 20) return launch_panel<2, 2, 6>(a, stream);      // K = 192: 128-row panels, 32-column chunks
  if (a->tile_cfg == 21) return launch_panel<1, 2, 6>(a, stream);      // K = 192:  64-row panels
  if (a->out_dtype == ODIC_H2 && ((a->ldc % 8) || (a->strideC % 8) || ((uintptr_t)a->out & 31))) return ODIC_EINVAL;
  Params p;
  p.A = (const char*)a->A; p.W = (const char*)a->W; p.bias = a->bias; p.residual = a->residual; p.out = a->out;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldr = a->ldr; p.ldc = a->ldc;
  p.strideA = a->strideA; p.strideW = a->strideW; p.strideBias = a->strideBias; p.strideR = a->strideR; p.strideC = a->strideC;
  p.alpha = a->alpha; p.act = a->act; p.bias_axis = a->bias_axis;
  int cfg = a->tile_cfg;
  if (cfg < 0) {
    auto rounds = [&](int bm, int bn, int slots) {
      const long t = (long)((a->M + bm - 1) / bm) * ((a->N + bn - 1) / bn) * a->batch;
      return (double)((t + slots - 1) / slots);
    };
    const double c0 = rounds(128, 64, 768) * 1.0, c1 = rounds(128, 128, 512) * 1.5, c2 = rounds(256, 128, 256) * 2.6;
    cfg = (c0 <= c1 && c0 <= c2) ? 0 : (c1 <= c2 ? 1 : 2);
  }
  switch (cfg) {
    case 0: return launch<2, 2, 4, 2, 2>(p, a->out_dtype, a->batch, stream);      // 128 x 64,  2 stages (48 KiB)
    case 1: return launch<2, 2, 4, 4, 2>(p, a->out_dtype, a->batch, stream);      // 128 x 128, 2 stages (64 KiB)
    case 2: return launch<4, 2, 4, 4, 2>(p, a->out_dtype, a->batch, stream);      // 256 x 128, 2 stages (96 KiB)
    case 3: return launch<2, 2, 4, 4, 3>(p, a->out_dtype, a->batch, stream);      // 128 x 128, 3 stages (96 KiB)
    case 4: return launch<4, 2, 4, 4, 3>(p, a->out_dtype, a->batch, stream);      // 256 x 128, 3 stages (144 KiB)
    case 5: return launch<2, 2, 4, 2, 3>(p, a->out_dtype, a->batch, stream);      // 128 x 64,  3 stages (72 KiB)
    // 48 x 96 wave patches in 144- / 288-row tiles (gemm_bf16.hip tile configs 40-42): exact multiples of 256 tiles on
    // the 9216- / 2304-row products
    case 6: return launch<6, 2, 3, 6, 2>(p, a->out_dtype, a->batch, stream);      // 288 x 192, 12 waves (120 KiB)
    case 7: return launch<3, 3, 3, 6, 2>(p, a->out_dtype, a->batch, stream);      // 144 x 288,  9 waves (108 KiB)
    case 8: return launch<3, 2, 3, 6, 2>(p, a->out_dtype, a->batch, stream);      // 144 x 192,  6 waves (84 KiB)
    case 9: return launch<4, 2, 4, 4, 3>(p, a->out_dtype, a->batch, stream);      // 256 x 128, 3 stages (144 KiB) [as 4]
    default: return ODIC_EINVAL;
  }
}
