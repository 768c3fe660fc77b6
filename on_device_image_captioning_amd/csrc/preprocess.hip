// Device-side image preprocessing (reference utils/image_utils.py:5-23):
//   Resize((S,S)) with PIL's BILINEAR resampler (antialiased: the triangle filter widens with the
//   down-scale factor), ToTensor (/255, HWC → CHW) and Normalize(mean, std) — bit-exact with
//   PIL.Image.resize + torch float32 arithmetic.
//
// PIL resamples 8-bit images in two integer passes (libImaging/Resample.c): per output coordinate a
// window [xmin, xmin+n) of taps with coefficients round(w·2^22) (computed on the host exactly as
// precompute_coeffs / normalize_coeffs_8bpc do, ops.pil_bilinear_coeffs), accumulator = 2^21 + Σ pix·k,
// result = clip8(acc >> 22); horizontal pass first into an 8-bit intermediate, then the vertical one.
// Both passes here are one thread per output pixel (3 channels), coalesced along x.
#include "odic_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[y][xx][c] = clip8(2^21 + Σ_i src[y][xmin+i][c]·kx[xx][i])
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ src, long sstride, int H,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ksize, unsigned char* __restrict__ tmp, int out) {
  const int xx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (xx >= out) return;
  const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = kk + (long)xx * ksize;
  const unsigned char* p = src + (long)y * sstride + 3L * x0;
  int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
    a0 += p[3 * i] * w; a1 += p[3 * i + 1] * w; a2 += p[3 * i + 2] * w;
  }
  unsigned char* q = tmp + ((long)y * out + xx) * 3;
  q[0] = clip8(a0); q[1] = clip8(a1); q[2] = clip8(a2);
}

// dst[c][yy][xx] = (clip8(2^21 + Σ_i tmp[ymin+i][xx][c]·ky[yy][i]) / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const unsigned char* __restrict__ tmp,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk,
                                                            int ksize, float* __restrict__ dst, int out, float m0,
                                                            float m1, float m2, float s0, float s1, float s2) {
  const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
  if (xx >= out) return;
  const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = kk + (long)yy * ksize;
  const unsigned char* p = tmp + ((long)y0 * out + xx) * 3;
  int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
    const unsigned char* r = p + (long)i * out * 3;
    a0 += r[0] * w; a1 += r[1] * w; a2 += r[2] * w;
  }
  const long plane = (long)out * out, o = (long)yy * out + xx;
  // ToTensor: uint8 → float32 / 255 ; Normalize: (x - mean) / std — each step rounded to fp32 as torch does
  dst[o] = (__fdiv_rn((float)clip8(a0), 255.0f) - m0) / s0;
  dst[plane + o] = (__fdiv_rn((float)clip8(a1), 255.0f) - m1) / s1;
  dst[2 * plane + o] = (__fdiv_rn((float)clip8(a2), 255.0f) - m2) / s2;
}

}  // namespace

extern "C" int odic_resize_bilinear_normalize(const uint8_t* src_rgb, int32_t H, int32_t W, int64_t src_stride_bytes,
                                              const int32_t* bounds_x, const int32_t* coef_x, int32_t ksize_x,
                                              const int32_t* bounds_y, const int32_t* coef_y, int32_t ksize_y,
                                              uint8_t* tmp, float* dst, int32_t out_size, const float* mean3,
                                              const float* std3, void* stream) {
  if (!src_rgb || !bounds_x || !coef_x || !bounds_y || !coef_y || !tmp || !dst || !mean3 || !std3) return ODIC_ENULL;
  if (H <= 0 || W <= 0 || out_size <= 0 || ksize_x <= 0 || ksize_y <= 0 || src_stride_bytes < 3L * W || H > 65535 ||
      out_size > 65535)
    return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 block(256);
  hipLaunchKernelGGL(resize_h_kernel, dim3((out_size + 255) / 256, H), block, 0, s, src_rgb, (long)src_stride_bytes, H,
                     bounds_x, coef_x, ksize_x, tmp, out_size);
  hipLaunchKernelGGL(resize_v_norm_kernel, dim3((out_size + 255) / 256, out_size), block, 0, s, tmp, bounds_y, coef_y,
                     ksize_y, dst, out_size, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return odic_launch_status();
}
