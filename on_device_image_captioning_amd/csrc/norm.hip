// LayerNorm family for gfx950 — HBM-bound row kernels, one 64-lane wave per row.
//
//   layernorm            x fp32 [M,C] → fp32 | bf16           (nn.LayerNorm; every norm on the path)
//   patch_merge_layernorm gathers the 2x2 neighbourhood (PatchMerging, swin_transformer_mod.py
//                        :386-395) while loading, so the (B,L/4,4C) concat is never materialised
//   patch_embed          Conv2d(k=s=patch) as a per-token dot product + LayerNorm (:511-519)
//
// Each lane keeps its slice of the row in registers (float4 loads, 16 B/lane coalesced), the mean
// and the centred variance are reduced with wave shuffles (two-pass, like ATen's RowwiseMoments up
// to summation order), and the normalised row is written once.
#include "odic_common.h"

namespace {

// VPL = float4 vectors per lane (row length <= 256*VPL)
template <int VPL, bool MERGE, typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, OutT* __restrict__ out,
                                                        int M, int C, float eps, int res, int Cin) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (M >= 1024) ODIC_ENCODE_PRIO();          // backbone-sized launches only (the decoder's 48-row norms stay at 0)
  const int C4 = C >> 2;

  // source pointer of float4 index v of this row
  long base0 = 0, base1 = 0, base2 = 0, base3 = 0;
  int Cin4 = 0;
  if (MERGE) {
    const int half = res >> 1;
    const int per_img = half * half;
    const int b = row / per_img, r = row - b * per_img;
    const int i = r / half, j = r - i * half;
    const long img = (long)b * res * res;
    base0 = (img + (long)(2 * i) * res + 2 * j) * Cin;          // (0,0)
    base1 = (img + (long)(2 * i + 1) * res + 2 * j) * Cin;      // (1,0)
    base2 = (img + (long)(2 * i) * res + 2 * j + 1) * Cin;      // (0,1)
    base3 = (img + (long)(2 * i + 1) * res + 2 * j + 1) * Cin;  // (1,1)
    Cin4 = Cin >> 2;
  } else {
    base0 = (long)row * ldx;
  }

  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < VPL; ++t) {
    const int idx = lane + 64 * t;
    v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < C4) {
      const float* src;
      if (MERGE) {
        const int seg = idx / Cin4, off = idx - seg * Cin4;
        const long b_ = seg == 0 ? base0 : (seg == 1 ? base1 : (seg == 2 ? base2 : base3));
        src = x + b_ + 4 * off;
      } else {
        src = x + base0 + 4 * idx;
      }
      v[t] = *(const float4*)src;
      s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < VPL; ++t) {
    const int idx = lane + 64 * t;
    if (idx < C4) {
      const float a = v[t].x - mean, b = v[t].y - mean, c = v[t].z - mean, d = v[t].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int t = 0; t < VPL; ++t) {
    const int idx = lane + 64 * t;
    if (idx < C4) {
      const float4 g = *(const float4*)(gamma + 4 * idx);
      const float4 bb = *(const float4*)(beta + 4 * idx);
      const float o0 = (v[t].x - mean) * rstd * g.x + bb.x;
      const float o1 = (v[t].y - mean) * rstd * g.y + bb.y;
      const float o2 = (v[t].z - mean) * rstd * g.z + bb.z;
      const float o3 = (v[t].w - mean) * rstd * g.w + bb.w;
      OutT* dst = out + (long)row * C + 4 * idx;
      if constexpr (__is_same(OutT, h2_t)) {     // split-fp16 operand of the x3 GEMM: 8 bytes into each plane
        h2_store4(out + (long)row * C, 4 * idx, o0, o1, o2, o3);
      } else if constexpr (sizeof(OutT) == 4) {
        *(float4*)dst = make_float4(o0, o1, o2, o3);
      } else if constexpr (sizeof(OutT) == 2) {
        ushort4 pk;
        pk.x = f32_to_bf16(o0); pk.y = f32_to_bf16(o1); pk.z = f32_to_bf16(o2); pk.w = f32_to_bf16(o3);
        *(ushort4*)dst = pk;
      } else {                                   // fp8 e4m3 (the consumer's scale is folded into gamma / beta)
        *(unsigned*)dst = f32x4_to_fp8(o0, o1, o2, o3);
      }
    }
  }
}

template <bool MERGE, typename OutT>
int launch_ln(const float* x, long ldx, const float* g, const float* b, void* out, int M, int C, float eps,
              int res, int Cin, hipStream_t stream) {
  dim3 grid((M + 3) / 4), block(256);
  const int C4 = C / 4;
#define ODIC_LN_CASE(V)                                                                                   \
  hipLaunchKernelGGL((layernorm_kernel<V, MERGE, OutT>), grid, block, 0, stream, x, ldx, g, b, (OutT*)out, \
                     M, C, eps, res, Cin)
  if (C4 <= 64) ODIC_LN_CASE(1);
  else if (C4 <= 128) ODIC_LN_CASE(2);
  else if (C4 <= 256) ODIC_LN_CASE(4);
  else if (C4 <= 512) ODIC_LN_CASE(8);
  else if (C4 <= 1024) ODIC_LN_CASE(16);
  else ODIC_LN_CASE(32);
#undef ODIC_LN_CASE
  return odic_launch_status();
}

// ---------------------------------------------------------------------------------------------
// patch embed: a block of 256 threads owns 64 consecutive tokens.  Inputs (64 x K, float4 loads along
// the patch rows, coalesced over consecutive tokens) and the transposed weights [K][C] sit in LDS;
// thread (token = tid & 63, group = tid >> 6) accumulates C/4 output channels with broadcast weight
// reads, LayerNorm moments are combined across the 4 channel groups through LDS, and the normalised
// 64 x C tile leaves through an LDS slab as one linear, fully coalesced copy.
// ---------------------------------------------------------------------------------------------
template <int CPT>     // channels per thread; C == 4 * CPT exactly
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ img,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out,
                                                          int B, int in_chans, int H, int W, int patch, int C,
                                                          float eps) {
  ODIC_ENCODE_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = in_chans * patch * patch;
  constexpr int cpt = CPT;
  float* wt = sm;                          // [K][C]
  float* xin = wt + K * C;                 // [K][64]
  float* red = xin + K * 64;               // [4][64]
  float* slab = sm;                        // [64][C+1]  — reuses the weight/input area after the FMAs
  const int tid = threadIdx.x, token = tid & 63, grp = tid >> 6;
  const int gh = H / patch, gw = W / patch;
  const int per_img = gh * gw;
  const long ntok = (long)B * per_img;
  const long tok0 = (long)blockIdx.x * 64;
  const int b0 = (int)(tok0 / per_img);                 // one 64-bit divide per block (wave-uniform)
  const int r0 = (int)(tok0 - (long)b0 * per_img);
  const int nvalid = (int)min((long)64, ntok - tok0);

  // weights, transposed on the fly: thread c reads row c of w (K contiguous floats, 16-byte loads when K % 4 == 0) and
  // writes column c of wt — consecutive lanes, consecutive LDS words; no index division
  for (int c = tid; c < C; c += 256) {
    const float* wr = w + (long)c * K;
    if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
      for (int k = 0; k < K; k += 4) {
        const float4 v = *(const float4*)(wr + k);
        wt[(k + 0) * C + c] = v.x; wt[(k + 1) * C + c] = v.y; wt[(k + 2) * C + c] = v.z; wt[(k + 3) * C + c] = v.w;
      }
    } else {
      for (int k = 0; k < K; ++k) wt[k * C + c] = wr[k];
    }
  }
  // input rows: segment s = (channel, patch row) holds 4 contiguous floats per token
  const int nseg = in_chans * patch;
  for (int i = tid; i < nseg * 64; i += 256) {
    const int t = i & 63, sgm = i >> 6;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < nvalid) {
      int r = r0 + t, b = b0;
      if (r >= per_img) { r -= per_img; ++b; }           // a 64-token block spans at most two images
      const int ph = r / gw, pw = r - ph * gw;
      const int c = sgm / patch, kh = sgm - c * patch;
      v = *(const float4*)(img + (((long)b * in_chans + c) * H + ph * patch + kh) * W + pw * patch);
    }
    const int k0 = sgm * patch;
    xin[(k0 + 0) * 64 + t] = v.x; xin[(k0 + 1) * 64 + t] = v.y;
    xin[(k0 + 2) * 64 + t] = v.z; xin[(k0 + 3) * 64 + t] = v.w;
  }
  __syncthreads();

  float acc[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) acc[c] = bias[grp * cpt + c];
  for (int k = 0; k < K; ++k) {
    const float xv = xin[k * 64 + token];
    const float4* wr = (const float4*)(wt + k * C + grp * cpt);     // wave-uniform address: LDS broadcast
    float4 wv[CPT / 4];
#pragma unroll
    for (int c = 0; c < CPT / 4; ++c) wv[c] = wr[c];                  // all reads in flight before the FMAs
#pragma unroll
    for (int c = 0; c < CPT / 4; ++c) {
      acc[4 * c + 0] = fmaf(xv, wv[c].x, acc[4 * c + 0]); acc[4 * c + 1] = fmaf(xv, wv[c].y, acc[4 * c + 1]);
      acc[4 * c + 2] = fmaf(xv, wv[c].z, acc[4 * c + 2]); acc[4 * c + 3] = fmaf(xv, wv[c].w, acc[4 * c + 3]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CPT; ++c) s += acc[c];
  red[grp * 64 + token] = s;
  __syncthreads();
  const float mean = (red[token] + red[64 + token] + red[128 + token] + red[192 + token]) / (float)C;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CPT; ++c) { const float dlt = acc[c] - mean; q += dlt * dlt; }
  red[grp * 64 + token] = q;
  __syncthreads();
  const float rstd = rsqrtf((red[token] + red[64 + token] + red[128 + token] + red[192 + token]) / (float)C + eps);
  __syncthreads();                         // everyone is done with wt / xin / red: the slab may overwrite them
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int ch = grp * cpt + c;
    slab[token * (C + 1) + ch] = (acc[c] - mean) * rstd * gamma[ch] + beta[ch];
  }
  __syncthreads();
  // 64 x C tile → global: row t = grp + 4r, lane = float4 column; rows are contiguous in `out`
  float* dst = out + tok0 * C;
  const int C4 = C >> 2;
  for (int t = grp; t < nvalid; t += 4) {
    for (int c4 = token; c4 < C4; c4 += 64) {
      const float* sp = slab + t * (C + 1) + 4 * c4;
      *(float4*)(dst + (long)t * C + 4 * c4) = make_float4(sp[0], sp[1], sp[2], sp[3]);
    }
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, long ldx,
                                                   OutT* __restrict__ out, long ldo, int M, int C4) {
  ODIC_ENCODE_PRIO();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * C4) return;
  const int r = i / C4, c = (i - (long)r * C4) * 4;
  const float4 v = *(const float4*)(x + r * ldx + c);
  if constexpr (__is_same(OutT, h2_t)) {
    h2_store4(out + r * ldo, c, v.x, v.y, v.z, v.w);
  } else {
    ushort4 pk;
    pk.x = f32_to_bf16(v.x); pk.y = f32_to_bf16(v.y); pk.z = f32_to_bf16(v.z); pk.w = f32_to_bf16(v.w);
    *(ushort4*)(out + r * ldo + c) = pk;
  }
}

// plain copy of 16-byte units, grid-stride (the K/V hand-off of the pipeline: the consumer kernels then read what a
// kernel of this library wrote, under the same launch-boundary ordering as every other hand-off)
__global__ __launch_bounds__(256) void copy16_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

extern "C" int odic_copy(const void* src, void* dst, int64_t nbytes, void* stream) {
  if (!src || !dst) return ODIC_ENULL;
  if (nbytes <= 0 || (nbytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return ODIC_EINVAL;
  const long n16 = nbytes / 16;
  const long want = (n16 + 255) / 256;
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)src, (float4*)dst, n16);
  return odic_launch_status();
}

extern "C" int odic_cast_f32_to_bf16(const float* x, int64_t ldx, void* out, int64_t ldo, int32_t M, int32_t C,
                                     void* stream) {
  if (!x || !out) return ODIC_ENULL;
  if (M <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldo & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 7))
    return ODIC_EINVAL;
  const long n = (long)M * (C / 4);
  hipLaunchKernelGGL(cast_kernel<bf16_raw>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)ldx, (bf16_raw*)out, (long)ldo, M, C / 4);
  return odic_launch_status();
}

extern "C" int odic_cast_f32_to_h2(const float* x, int64_t ldx, void* out, int64_t ldo, int32_t M, int32_t C,
                                   void* stream) {
  if (!x || !out) return ODIC_ENULL;
  if (M <= 0 || C <= 0 || (C & 7) || (ldx & 3) || (ldo & 7) || ((uintptr_t)x & 15) || ((uintptr_t)out & 31))
    return ODIC_EINVAL;
  const long n = (long)M * (C / 4);
  hipLaunchKernelGGL(cast_kernel<h2_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)ldx, (h2_t*)out, (long)ldo, M, C / 4);
  return odic_launch_status();
}

extern "C" int odic_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, void* out,
                              int32_t M, int32_t C, float eps, int32_t out_dtype, void* stream) {
  if (!x || !gamma || !beta || !out) return ODIC_ENULL;
  if (M <= 0 || C <= 0 || (C & 3) || C > 8192 || (ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 7))
    return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == ODIC_F32) return launch_ln<false, float>(x, ldx, gamma, beta, out, M, C, eps, 0, 0, s);
  if (out_dtype == ODIC_BF16) return launch_ln<false, bf16_raw>(x, ldx, gamma, beta, out, M, C, eps, 0, 0, s);
  if (out_dtype == ODIC_FP8) return launch_ln<false, unsigned char>(x, ldx, gamma, beta, out, M, C, eps, 0, 0, s);
  if (out_dtype == ODIC_H2) {
    if ((C & 7) || ((uintptr_t)out & 31)) return ODIC_EINVAL;
    return launch_ln<false, h2_t>(x, ldx, gamma, beta, out, M, C, eps, 0, 0, s);
  }
  return ODIC_EINVAL;
}

extern "C" int odic_patch_merge_layernorm(const float* x, const float* gamma, const float* beta, void* out,
                                          int32_t B, int32_t res, int32_t C, float eps, int32_t out_dtype,
                                          void* stream) {
  if (!x || !gamma || !beta || !out) return ODIC_ENULL;
  if (B <= 0 || res <= 0 || (res & 1) || C <= 0 || (C & 3) || 4 * C > 8192) return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int M = B * (res / 2) * (res / 2);
  if (out_dtype == ODIC_F32) return launch_ln<true, float>(x, 0, gamma, beta, out, M, 4 * C, eps, res, C, s);
  if (out_dtype == ODIC_BF16) return launch_ln<true, bf16_raw>(x, 0, gamma, beta, out, M, 4 * C, eps, res, C, s);
  if (out_dtype == ODIC_H2) {
    if ((C & 1) || ((uintptr_t)out & 31)) return ODIC_EINVAL;
    return launch_ln<true, h2_t>(x, 0, gamma, beta, out, M, 4 * C, eps, res, C, s);
  }
  return ODIC_EINVAL;
}

extern "C" int odic_patch_embed(const float* img, const float* w, const float* b, const float* gamma,
                                const float* beta, float* out, int32_t B, int32_t in_chans, int32_t H,
                                int32_t W, int32_t patch, int32_t C, float eps, void* stream) {
  if (!img || !w || !b || !gamma || !beta || !out) return ODIC_ENULL;
  const int K = in_chans * patch * patch;
  if (B <= 0 || K > 64 || C > 256 || H % patch || W % patch) return ODIC_EINVAL;
  const long ntok = (long)B * (H / patch) * (W / patch);
  if (patch != 4 || (C & 15) || (W & 3) || ((uintptr_t)img & 15)) return ODIC_EUNSUPPORTED;
  dim3 grid((unsigned)((ntok + 63) / 64)), block(256);
  const size_t main_words = (size_t)C * K + K * 64 + 256, slab_words = (size_t)64 * (C + 1);
  const size_t shmem = (main_words > slab_words ? main_words : slab_words) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](auto kern) {
    static bool raised = false;          // one flag per kernel instantiation; first call is never inside a capture
    if (!raised) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      raised = true;
    }
    hipLaunchKernelGGL(kern, grid, block, shmem, s, img, w, b, gamma, beta, out, B, in_chans, H, W, patch, C, eps);
  };
  switch (C) {
    case 64: launch(patch_embed_kernel<16>); break;
    case 96: launch(patch_embed_kernel<24>); break;
    case 128: launch(patch_embed_kernel<32>); break;
    case 192: launch(patch_embed_kernel<48>); break;
    case 256: launch(patch_embed_kernel<64>); break;
    default: return ODIC_EUNSUPPORTED;       // Swin embed dims: 96 (T/S), 128 (B), 192 (L)
  }
  return odic_launch_status();
}
