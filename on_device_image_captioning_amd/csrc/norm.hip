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
      if constexpr (sizeof(OutT) == 4) {
        *(float4*)dst = make_float4(o0, o1, o2, o3);
      } else {
        ushort4 pk;
        pk.x = f32_to_bf16(o0); pk.y = f32_to_bf16(o1); pk.z = f32_to_bf16(o2); pk.w = f32_to_bf16(o3);
        *(ushort4*)dst = pk;
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
// patch embed: one wave per token; the K = in_chans*patch² inputs of the token are broadcast by
// shuffles, each lane owns output channels lane, lane+64, ... (<= 8 per lane → C <= 512).
// Weights are read through L1/L2 (C*K*4 B = 36 KiB for Swin-L, shared by every wave).
// ---------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ img,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out,
                                                          int B, int in_chans, int H, int W, int patch, int C,
                                                          float eps) {
  extern __shared__ float wt[];            // [K][C] transposed weights: lanes read consecutive c
  const int K = in_chans * patch * patch;
  for (int i = threadIdx.x; i < C * K; i += blockDim.x) {
    const int c = i / K, k = i - c * K;
    wt[k * C + c] = w[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int gh = H / patch, gw = W / patch;
  const long ntok = (long)B * gh * gw;
  const long tok0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;    // 8 tokens per wave
  for (int it = 0; it < 8; ++it) {
    const long tok = tok0 + it;
    if (tok >= ntok) return;
    const int b = tok / (gh * gw);
    const int r = tok - (long)b * gh * gw;
    const int ph = r / gw, pw = r - ph * gw;
    // lane k (< K) loads input element k = (c, kh, kw)
    float xin = 0.f;
    if (lane < K) {
      const int c = lane / (patch * patch), rem = lane - c * patch * patch;
      const int kh = rem / patch, kw = rem - kh * patch;
      xin = img[(((long)b * in_chans + c) * H + ph * patch + kh) * W + pw * patch + kw];
    }
    float acc[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
      const int c = lane + 64 * t;
      acc[t] = c < C ? bias[c] : 0.f;
    }
    for (int k = 0; k < K; ++k) {
      const float xv = __shfl(xin, k, 64);
#pragma unroll
      for (int t = 0; t < CPL; ++t) {
        const int c = lane + 64 * t;
        if (c < C) acc[t] = fmaf(xv, wt[k * C + c], acc[t]);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < CPL; ++t) if (lane + 64 * t < C) s += acc[t];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < CPL; ++t) if (lane + 64 * t < C) { const float d = acc[t] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
      const int c = lane + 64 * t;
      if (c < C) out[tok * C + c] = (acc[t] - mean) * rstd * gamma[c] + beta[c];
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, long ldx,
                                                        bf16_raw* __restrict__ out, long ldo, int M, int C4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * C4) return;
  const int r = i / C4, c = (i - (long)r * C4) * 4;
  const float4 v = *(const float4*)(x + r * ldx + c);
  ushort4 pk;
  pk.x = f32_to_bf16(v.x); pk.y = f32_to_bf16(v.y); pk.z = f32_to_bf16(v.z); pk.w = f32_to_bf16(v.w);
  *(ushort4*)(out + r * ldo + c) = pk;
}

}  // namespace

extern "C" int odic_cast_f32_to_bf16(const float* x, int64_t ldx, void* out, int64_t ldo, int32_t M, int32_t C,
                                     void* stream) {
  if (!x || !out) return ODIC_ENULL;
  if (M <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldo & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 7))
    return ODIC_EINVAL;
  const long n = (long)M * (C / 4);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)ldx, (bf16_raw*)out, (long)ldo, M, C / 4);
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
  return ODIC_EINVAL;
}

extern "C" int odic_patch_embed(const float* img, const float* w, const float* b, const float* gamma,
                                const float* beta, float* out, int32_t B, int32_t in_chans, int32_t H,
                                int32_t W, int32_t patch, int32_t C, float eps, void* stream) {
  if (!img || !w || !b || !gamma || !beta || !out) return ODIC_ENULL;
  const int K = in_chans * patch * patch;
  if (B <= 0 || K > 64 || C > 512 || H % patch || W % patch) return ODIC_EINVAL;
  const long ntok = (long)B * (H / patch) * (W / patch);
  dim3 grid((unsigned)((ntok + 31) / 32)), block(256);
  const size_t shmem = (size_t)C * K * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (C <= 128) hipLaunchKernelGGL(patch_embed_kernel<2>, grid, block, shmem, s, img, w, b, gamma, beta, out, B, in_chans, H, W, patch, C, eps);
  else if (C <= 256) hipLaunchKernelGGL(patch_embed_kernel<4>, grid, block, shmem, s, img, w, b, gamma, beta, out, B, in_chans, H, W, patch, C, eps);
  else hipLaunchKernelGGL(patch_embed_kernel<8>, grid, block, shmem, s, img, w, b, gamma, beta, out, B, in_chans, H, W, patch, C, eps);
  return odic_launch_status();
}
