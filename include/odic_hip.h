/*
 * odic_hip.h — C ABI of libodic_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * ExpansionNet v2 inference path (Swin-L/384 backbone → expansion encoder → beam-search decoder).
 *
 * The reference (nighting0le01/On_Device_Image_Captioning) has NO native code and NO FFI: its
 * boundary is a Python class contract.  This header is therefore the boundary
 * that the build's own Python host code (the on_device_image_captioning_amd package, ctypes) binds; each
 * entry point names the reference computation it replaces (file:line under /root/reference).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted.
 *   - no allocation, no ownership transfer, no host synchronisation inside any entry point: the
 *     caller allocates outputs/workspaces and passes the HIP stream (hipStream_t as void*).
 *     All entry points are therefore legal inside a stream capture (hipGraph).
 *   - return 0 on success, a negative ODIC_E* code on a rejected argument, or the positive
 *     hipError_t of a failed launch.  Nothing is printed.
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 *   - dtype codes: ODIC_F32 = 0 (float), ODIC_BF16 = 1 (bfloat16, raw uint16 storage), ODIC_FP8 = 2 (OCP e4m3,
 *     raw uint8), ODIC_F16 = 3 (IEEE half), ODIC_H2 = 4 (split fp16, see below).
 *   - ODIC_H2 ("split fp16": the operand format of the near-exact fast mode): a value x is carried as hi + lo with
 *     hi = fp16(x), lo = fp16(x - hi) — 22 significand bits — and a contraction runs as three fp16 MFMAs
 *     (hi·hi + hi·lo + lo·hi, fp32 accumulate).  Storage is 4 bytes per element, so shapes, leading dimensions and
 *     strides (in ELEMENTS) are those of the fp32 tensor it replaces; inside a row, every group of 8 consecutive
 *     elements is 32 bytes: [8 x hi | 8 x lo].  Rows start on 32-byte boundaries (base 32-byte aligned, ld % 8 == 0).
 *     All-zero bytes are the value 0, so zero-filled K padding is valid.  |x| saturates at 65504.
 */
#ifndef ODIC_HIP_H
#define ODIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODIC_F32 0
#define ODIC_BF16 1
#define ODIC_FP8 2     /* OCP e4m3 (e4m3fn), one byte per element: the low-precision backbone mode's GEMM operands */
#define ODIC_F16 3     /* IEEE half: that mode's qkv / attention activations */
#define ODIC_H2 4      /* split fp16 (hi + lo pairs, 4 bytes per element): the near-exact fast mode's operands */

#define ODIC_ACT_NONE 0
#define ODIC_ACT_GELU 1    /* exact erf GELU (nn.GELU, swin_transformer_mod.py:87) */
#define ODIC_ACT_RELU 2
#define ODIC_ACT_SIGMOID 3

#define ODIC_EINVAL (-1)   /* bad shape / alignment / enum */
#define ODIC_ENULL (-2)    /* required pointer is NULL */
#define ODIC_EUNSUPPORTED (-3)

/* ABI version of this header; bumped on any signature change. */
#define ODIC_ABI_VERSION 15
int odic_abi_version(void);

/* Human-readable build string ("gfx950 hipcc ..."), static storage. */
const char* odic_build_info(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   out = act(alpha * A·Wᵀ + bias) + residual
 *   A [M,K] (lda), W [N,K] (ldw) — the nn.Linear weight layout, so no transposes anywhere.
 *   bias: fp32, NULL or length N (bias_axis 0, per column) / length M (bias_axis 1, per row)
 *   residual: fp32 [M,N] (ldr) or NULL; out: `out_dtype` [M,N] (ldc)
 *   batch > 1: operands advance by the given element strides (0 = shared operand).
 * Replaces every nn.Linear on the path: swin_transformer_mod.py:190,212 (qkv/proj), :94-97
 * (Mlp fc1/fc2), :396 (PatchMerging.reduction), layers.py:49-51,99,154-161,274-276,293,306-307,
 * End_ExpansionNet_v2.py:82,97,134,137.
 * in_dtype ODIC_BF16: MFMA 16x16x32 bf16, fp32 accumulate; needs K % 64 == 0, lda/ldw % 8 == 0,
 *   16-byte aligned A/W.   in_dtype ODIC_F32: MFMA 16x16x4 f32 (exact fp32 FMA chain), any M,N,K.
 * in_dtype ODIC_FP8 / ODIC_F16 (low-precision backbone mode, BASELINE.json configs[4]): MFMA 16x16x32 fp8 / f16,
 *   fp32 accumulate, out = cast(act(alpha·col_scale[n]·(A·Wᵀ) + bias)·out_scale) + residual with out_dtype in
 *   {ODIC_F32, ODIC_F16, ODIC_FP8}; K a multiple of 64 (fp8) / 32 (f16); batch == 1.  The caller quantises: W per
 *   output channel at pack time, A per tensor with a static scale — col_scale[n] is their product.
 * in_dtype ODIC_H2 (near-exact fast mode): A and W are split-fp16 tensors, three MFMA 16x16x32 f16 per step into one
 *   fp32 accumulator; out_dtype ODIC_F32 or ODIC_H2; exact-erf GELU; K % 32 == 0, lda / ldw / strides % 8 == 0,
 *   32-byte aligned operands; any batch.  Weights may be pre-scaled by a power of two at pack time (undone in alpha).
 * ------------------------------------------------------------------------------------------- */
typedef struct odic_gemm_args {
  const void* A; const void* W; const float* bias; const float* residual; void* out;
  int32_t M, N, K;
  int64_t lda, ldw, ldr, ldc;
  int32_t batch;
  int64_t strideA, strideW, strideBias, strideR, strideC;
  float alpha;
  int32_t act;        /* ODIC_ACT_* */
  int32_t bias_axis;  /* 0: bias[n]   1: bias[m] */
  int32_t in_dtype;   /* dtype of A and W */
  int32_t out_dtype;  /* dtype of out */
  int32_t tile_cfg;   /* tile configuration, -1 = built-in choice.  bf16 (csrc/gemm_bf16.hip): 0, 1, 2, 7, 10 power-of-two tiles,
                       * 40..47 tiles of 48 x 96 wave patches (144 / 288 rows), 48 / 49 64 x 64, 50..53 the A-resident streaming
                       * kernels for K = 192 / 384 (whole tiles only; the only ones that take `a_ln`); 3..33 further variants in
                       * -DODIC_EXPERIMENTAL_GEMM builds (16 + c = config c as a persistent launch: needs `workspace`, batch == 1).
                       * fp8 / fp16 (gemm_lowp.hip): 0..4, 5..9 = the same on the block-scaled fp8 MFMA.  split fp16 (gemm_x3.hip):
                       * 0..9, 20 / 21 the A-resident kernels for K = 192.  An unsupported (shape, configuration) pair is refused. */
  /* Optional LayerNorm of the A operand, folded (fp32 skinny-M path only: M <= 192, K % 16 == 0,
   * bias_axis 0).  The caller prepares  W' = W·diag(gamma),  ln_colsum[n] = Σ_k W'[n][k]  and
   * bias' = bias + W·beta, passes W' / bias' as W / bias, and the kernel computes
   *     out = act(rstd[m]·(alpha·A·W'ᵀ − mean[m]·ln_colsum) + bias') + residual
   *         = act(LayerNorm(A; gamma, beta, ln_eps)·Wᵀ + bias) + residual
   * with mean/rstd the moments of row m of the RAW fp32 A, accumulated from the operand fragments.
   * Replaces the separate norm_1/2/3 + dec_reduce_norm launches of the decoder step
   * (layers.py:225,228,232; End_ExpansionNet_v2.py:135).  NULL = plain GEMM. */
  const float* ln_colsum; float ln_eps;
  /* bf16 persistent tile configurations only: 16 int32 of device memory, all zero when the launch starts;
   * the kernel leaves them zero again, so ONE buffer serves every launch of a stream (launches of different
   * streams that may overlap need their own).  NULL otherwise. */
  int32_t* workspace;
  /* fp8 / fp16 inputs only: per-output-column dequantisation factor (fp32 [N], NULL = 1) and the factor applied
   * before the output cast (0 = 1; 1/scale of the consumer's fp8 operand). */
  const float* col_scale; float out_scale;
  /* LayerNorm folded across two bf16 products (one-block-per-tile tile configurations 0..11, batch == 1) — removes
   * the separate norm1 / norm2 launches of a Swin block (swin_transformer_mod.py:309,338) and their fp32 re-read
   * of the residual stream:
   *   producer (out_dtype ODIC_F32, e.g. the proj / fc2 product with the residual added): out16 (bf16 [M,N], ld16)
   *     receives the rounded copy of `out`; stats_out (fp32 [M, N/32, 2]) receives, per row and 32-column group, the
   *     mean and the centred sum of squares of those bf16 values.  N % 32 == 0.
   *   consumer (A = that bf16 copy): ln_stats = the producer's stats_out (K/32 groups per row), ln_colsum and the
   *     packed W / bias as for the fp32 form above; out = act(rstd·(alpha·A·W'ᵀ − mean·ln_colsum) + bias') + residual. */
  void* out16; int64_t ld16; float* stats_out;
  const float* ln_stats;
  /* LayerNorm of the A operand computed while A is read (A-resident tile configurations only: bf16 50-53, split fp16 20 / 21 — the
   * K = 192 / 384 products of Swin stages 0-1; whole tiles, batch == 1): A = NULL and the operand is
   *     (x − mean(x)) / sqrt(var(x) + ln_eps)   of each fp32 row of a_ln [M,K] (ld_aln elements),
   * rounded to bf16 in registers; W and bias are folded by the caller as above (W' = W·diag(gamma), bias' = bias + W·beta), so
   *     out = act(alpha·LayerNorm(x; gamma, beta)·Wᵀ + bias) + residual.
   * Replaces norm1 → qkv and norm2 → fc1 (swin_transformer_mod.py:309-310, 338) by one launch each and removes the bf16
   * copy of the residual stream between them.  NULL = A is the operand. */
  const float* a_ln; int64_t ld_aln;
} odic_gemm_args;
int odic_gemm(const odic_gemm_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps inside sqrt, biased variance — torch.nn.LayerNorm).
 *   x fp32 [M,C] (ldx) → out `out_dtype` [M,C] contiguous (ODIC_F32 / ODIC_BF16 / ODIC_FP8 / ODIC_H2; an fp8 consumer's
 *   quantisation scale is folded into gamma / beta by the caller).   C % 4 == 0 (ODIC_H2: % 8), C <= 8192.
 * Replaces swin_transformer_mod.py:309,338 (norm1/norm2), :639 (final norm), layers.py:119,121,
 * 225,228,232 and the reduce norms End_ExpansionNet_v2.py:99,135.
 * ------------------------------------------------------------------------------------------- */
int odic_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, void* out,
                   int32_t M, int32_t C, float eps, int32_t out_dtype, void* stream);

/* Device-to-device copy of nbytes (a multiple of 16; both pointers 16-byte aligned) by a kernel of this library:
 * the pipeline's K/V hand-off from the encode stream's staging buffer to a decode lane. */
int odic_copy(const void* src, void* dst, int64_t nbytes, void* stream);

/* Row-strided fp32 → bf16 conversion (feeds fp32 residual streams / caller tensors to the bf16 MFMA
 * GEMM).  x fp32 [M,C] (ldx) → out bf16 [M,C] (ldo); C, ldx, ldo multiples of 4. */
int odic_cast_f32_to_bf16(const float* x, int64_t ldx, void* out, int64_t ldo, int32_t M, int32_t C,
                          void* stream);
/* The same into split fp16 (ODIC_H2): C % 8 == 0, ldo % 8 == 0 (elements of 4 bytes), out 32-byte aligned. */
int odic_cast_f32_to_h2(const float* x, int64_t ldx, void* out, int64_t ldo, int32_t M, int32_t C, void* stream);

/* PatchMerging gather + LayerNorm(4C)  (swin_transformer_mod.py:386-395):
 *   x fp32 [B, res*res, C] → out `out_dtype` (ODIC_F32 / ODIC_BF16 / ODIC_H2) [B, (res/2)², 4C]; channel blocks in
 *   the order (0,0),(1,0),(0,1),(1,1) of the 2x2 neighbourhood (row offset, col offset). */
int odic_patch_merge_layernorm(const float* x, const float* gamma, const float* beta, void* out,
                               int32_t B, int32_t res, int32_t C, float eps, int32_t out_dtype,
                               void* stream);

/* PatchEmbed: Conv2d(in_chans→C, k=s=patch) + flatten + LayerNorm(C)
 * (swin_transformer_mod.py:511-519).  img fp32 [B,in_chans,H,W]; w fp32 [C,in_chans*patch*patch];
 * out fp32 [B,(H/patch)*(W/patch),C].  patch == 4, C in {64,96,128,192,256}, in_chans*16 <= 64. */
int odic_patch_embed(const float* img, const float* w, const float* b, const float* gamma,
                     const float* beta, float* out, int32_t B, int32_t in_chans, int32_t H,
                     int32_t W, int32_t patch, int32_t C, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image preprocessing on the device (utils/image_utils.py:5-23: Resize((S,S)) → ToTensor → Normalize),
 * bit-exact with PIL.Image.resize(..., BILINEAR) followed by the fp32 /255, -mean, /std of torch.
 *   src_rgb  host-decoded RGB8 image in device memory, H rows of W pixels, row pitch src_stride_bytes
 *   bounds_* int32 [S, 2] (first tap, tap count), coef_* int32 [S, ksize_*] fixed-point (2^22) weights of
 *            PIL's antialiased triangle filter for that axis — computed on the host exactly as
 *            libImaging/Resample.c does (on_device_image_captioning_amd.image_utils.pil_bilinear_coeffs)
 *   tmp      uint8 workspace [H, S, 3] (the horizontally resampled image)
 *   dst      fp32 [3, S, S];  mean3 / std3 are HOST pointers to 3 floats.
 * ------------------------------------------------------------------------------------------- */
int odic_resize_bilinear_normalize(const uint8_t* src_rgb, int32_t H, int32_t W, int64_t src_stride_bytes,
                                   const int32_t* bounds_x, const int32_t* coef_x, int32_t ksize_x,
                                   const int32_t* bounds_y, const int32_t* coef_y, int32_t ksize_y,
                                   uint8_t* tmp, float* dst, int32_t out_size, const float* mean3,
                                   const float* std3, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Swin (shifted-)window attention core  (WindowAttention.forward swin_transformer_mod.py:193-211
 * plus the roll / window_partition / window_reverse / roll index maps of :312-334, folded into the
 * kernel's loads and stores so no permuted copy is ever materialised):
 *   qkv  `dtype` [B*res*res, 3C]   token-major output of the qkv Linear, columns (3, heads, 32)
 *   bias_table fp32 [(2ws-1)², heads]   relative_position_bias_table; the index buffer (:163-173)
 *        and the SW-MSA mask (:281-297, values 0/-100) are recomputed from coordinates.
 *   bias_shifted_prescaled (optional, bf16 path, ws = 12) fp32 [heads, 4, 576]: per head four copies of the
 *        x-reversed bias table with rows padded to 24, R[r·24 + c'] = bias_table[r·23 + (22 - c')] / scale,
 *        copy s holding R shifted by s floats (copy_s[i] = R[i + s]) — the four keys of an accumulator quad
 *        are then ONE aligned 16-byte LDS read (the gather of :196-198 without a dense [ws², ws²] tensor).
 *        The fast kernel keeps the 9 KiB in LDS, loads the bias as the accumulator init of the q·kᵀ MFMA and
 *        applies scale·log2(e) afterwards (base-2 softmax).  NULL → the table is used (slower kernel).
 *   out  `dtype` [B*res*res, C]    softmax(q·kᵀ·scale + bias + mask)·v, heads concatenated,
 *        written back at the un-shifted token positions (ready for the proj Linear).
 * head_dim is 32 (every Swin-L stage), ws*ws <= 144, res % ws == 0, 0 <= shift < ws.
 * dtype ODIC_H2 (near-exact fast mode; needs bias_shifted_prescaled, ws = 12): qkv and out are split-fp16 tensors,
 *        q·kᵀ and P·v run as three fp16 MFMAs each, the softmax between them in fp32.
 * ------------------------------------------------------------------------------------------- */
int odic_window_attention(const void* qkv, const float* bias_table, const float* bias_shifted_prescaled,
                          void* out, int32_t B, int32_t res, int32_t C, int32_t heads, int32_t ws,
                          int32_t shift, float scale, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * norm1 → qkv Linear → window attention core of one Swin block in ONE launch (swin_transformer_mod.py:309-334 with
 * WindowAttention.forward :222-263 up to, not including, the proj Linear), for the stage of width C = 192 (ws = 12):
 *   x        fp32 [B*res*res, C] (ldx)   the residual stream (token-major, un-shifted, un-partitioned)
 *   w_qkv_folded bf16 [3C, C], b_qkv_folded fp32 [3C]:  W·diag(gamma) and bias + W·beta of norm1 → qkv (the caller folds the
 *            LayerNorm's affine part at pack time; the kernel computes (x − mean)/sqrt(var + ln_eps) in registers)
 *   bias_shifted_prescaled fp32 [heads, 4, 576]   as for odic_window_attention
 *   out      bf16 [B*res*res, C]   attention output at the un-shifted token positions (ready for the proj Linear)
 * One block per window keeps its 144 normalised rows as MFMA fragments in registers; q / k / v of a head never leave the
 * chip.  Results are bit-identical to odic_gemm(a_ln = x, …) followed by odic_window_attention (bf16).
 * ------------------------------------------------------------------------------------------- */
int odic_swin_qkv_attention(const float* x, int64_t ldx, const void* w_qkv_folded, const float* b_qkv_folded,
                            const float* bias_shifted_prescaled, void* out, int32_t B, int32_t res, int32_t C,
                            int32_t heads, int32_t ws, int32_t shift, float scale, float ln_eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Static expansion (encoder) helpers — layers.py:45-102.  The contractions run through
 * odic_gemm; these kernels do the relu/mask/L1-normalise steps in between.
 *   z fp32 [B, nq, S]: Q·Kᵀ/sqrt(d).
 *   fw:  pos = relu(z)·valid, neg = relu(-z)·valid, each row divided by (rowsum + eps) over S
 *        (layers.py:56-61) → pos_fw, neg_fw `out_dtype` [B, nq, ld_fw].  enc_len[b] = #valid keys.
 *   bw:  relu(±zᵀ), each of the `ngroups` column groups L1-normalised separately (layers.py:67-79)
 *        and pre-divided by ngroups (:84-85) → pos_bw, neg_bw `out_dtype` [B, S, ld_bw].
 *   ld_fw >= S and ld_bw >= nq: the padding columns are written as zeros, so the outputs can be the
 *        K-padded operands of odic_gemm (bf16 needs K % 64 == 0, ODIC_H2 K % 32 == 0); out_dtype ODIC_F32 / ODIC_BF16 /
 *        ODIC_H2.
 *   group_meta: device int32 [ngroups+1+nq] = exclusive prefix sums of the group sizes (last = nq)
 *        followed by the group index of every query row.
 *   colsum_ws: fp32 scratch [B*ngroups*2*S].
 *   scale_fw / scale_bw: factors applied to the fw / bw tables on output (1 = as the reference; the split-fp16 mode
 *        writes them times a power of two so that the lo halves of these small weights stay fp16 normals, and
 *        undoes it in the consuming product's alpha).
 * ------------------------------------------------------------------------------------------- */
int odic_stcexp_normalize(const float* z, const int32_t* enc_len, const int32_t* group_meta,
                          int32_t ngroups, void* pos_fw, void* neg_fw, int64_t ld_fw, void* pos_bw,
                          void* neg_bw, int64_t ld_bw, float* colsum_ws, int32_t B, int32_t nq,
                          int32_t S, float eps, float scale_fw, float scale_bw, int32_t out_dtype, void* stream);

/* out = x + sigmoid(sel_pre)·a + (1-sigmoid(sel_pre))·b     (layers.py:99-100 + the residual add of
 * EncoderLayer :120); all fp32 [M, d] with row strides. */
int odic_selector_mix(const float* x, int64_t ldx, const float* sel_pre, int64_t lds,
                      const float* a, int64_t lda, const float* b, int64_t ldb, float* out,
                      int64_t ldo, int32_t M, int32_t d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Incremental decoder step (exact because the decoder is causal).
 * N = number of live sequences (images × beams), laid out image-major: seq = img*beams + beam.
 * `pos` (device int32 scalar) is the position being processed; kernels read it from memory so one
 * captured graph can be replayed for every step.
 * ------------------------------------------------------------------------------------------- */

/* y[n,:] = embed[tok[n]]·sqrt(d) + pos_table[pos]   (layers.py:16-17, End_ExpansionNet_v2.py:118-121).
 * pos_rows = rows of pos_table (pos_encoder = nn.Embedding(max_seq_len, d), End_ExpansionNet_v2.py:105): *pos is device
 * memory, so the bound is checked ON the device — a launch with *pos outside [0, pos_rows) writes nothing. */
int odic_dec_embed(const int64_t* tokens, const float* embed, const float* pos_table,
                   const int32_t* pos, float* y, int64_t ldy, int32_t N, int32_t d, int32_t pos_rows,
                   float scale, void* stream);

/* Dynamic expansion for the newest position (layers.py:152-204), with per-position caches.
 *   lin fp32 [N, >=5d] (ldlin): cond | key | class_a | class_b | selector-pre-activation of
 *        LN1(y) at `pos`
 *   qexp, bexp fp32 [E, d]: query_exp_vectors / bias_exp_vectors
 *   caches (fp32), indexed [pos][seq_slot]:  cond_c, key_c, va_c, vb_c  [T, N, d]
 *                                            wfa_c, wfb_c  [T, N, T, E] — the normalised FORWARD weights of that
 *                                               position's E expansion queries over its keys 0..pos (:165-176);
 *                                               the (t·E) x d class matrices of :177-180,199-200 are never formed:
 *                                               the backward sum of :183-200 is re-associated onto va / vb / cond
 *                                            qk_c [T, N, E]   (query_exp[e]·key of that position)
 *   anc int32 [N, T]: for sequence n and position j < pos, the slot (sequence index) whose cache
 *        entry at j belongs to n's history (beam re-ordering without copying caches); position
 *        `pos` itself is always slot n.
 *   row_valid int32 [N]: 0 → padded row (finished beam): the block contributes 0 (masked rows of
 *        utils/masking.py:37-47), caches are still written.
 *   y_in fp32 [N,d] (ldy_in) → y fp32 [N,d] (ldy):  y = y_in + sel·A' + (1-sel)·B'  (may alias).
 *   T <= 128, E in {4, 8, 16, 32}, d a multiple of 64; one launch (one 1024-thread block per sequence).
 *   A launch with *pos outside [0, T) changes nothing (the caches hold T positions; checked on the device).
 */
int odic_dynexp_step(const float* lin, int64_t ldlin, const float* qexp, const float* bexp,
                     float* cond_c, float* key_c, float* va_c, float* vb_c, float* wfa_c,
                     float* wfb_c, float* qk_c, const int32_t* anc, const int32_t* row_valid,
                     const int32_t* pos, const float* y_in, int64_t ldy_in, float* y, int64_t ldy,
                     int32_t N, int32_t T, int32_t d, int32_t E, float eps, void* stream);

/* Cross attention of one query row per sequence against per-IMAGE cached K/V (layers.py:266-295;
 * the reference re-projects K/V of the 144 encoder tokens every step for every beam copy).
 *   q fp32 [N, d] (ldq; already Wq-projected, bias included);
 *   kv fp32 [n_img, S, ldkv]: projected keys at column koff, values at column voff;
 *   enc_len int32 [n_img]; beams = N / n_img; row_valid as above (0 → all scores masked to -1e4,
 *   i.e. a uniform average over all S positions, exactly what masked_fill + softmax gives).
 *   out fp32 [N, d] (ldo) = softmax(q·kᵀ/sqrt(d/heads))·v, heads concatenated.  d/heads in {16,32,64}. */
int odic_cross_attn_step(const float* q, int64_t ldq, const float* kv, int64_t ldkv, int32_t koff,
                         int32_t voff, const int32_t* enc_len, const int32_t* row_valid, float* out,
                         int64_t ldo, int32_t N, int32_t n_img, int32_t S, int32_t d, int32_t heads,
                         void* stream);

/* log_softmax over V + top-k (captioning_model.py:126-127,162-170).  logits fp32 [N, V] (ldl);
 * writes logp_out fp32 [N, V] (ldp) if non-NULL, top_val fp32 [N,k] / top_idx int32 [N,k] sorted
 * descending (ties: lower index first).  k <= 16. */
int odic_logsoftmax_topk(const float* logits, int64_t ldl, float* logp_out, int64_t ldp,
                         float* top_val, int32_t* top_idx, int32_t N, int32_t V, int32_t k,
                         void* stream);

/* The `sample` variants of the search (captioning_model.py:128-131,166-168: exp(log_probs).multinomial(k,
 * replacement=False); :59-109 ancestral sampling with k = 1): k words drawn WITHOUT replacement from
 * softmax(logits[n]) on the device (Gumbel-top-k), top_idx int32 [N,k] in draw order, top_val fp32 [N,k] =
 * their log-probabilities; logp_out as in odic_logsoftmax_topk.  Noise = Philox4x32-10(seed; row, word/4,
 * *pos): `pos` (device int32 scalar, may be NULL = 0) separates the steps of a captured graph. */
int odic_logsoftmax_sample(const float* logits, int64_t ldl, float* logp_out, int64_t ldp, float* top_val,
                           int32_t* top_idx, int32_t N, int32_t V, int32_t k, uint64_t seed,
                           const int32_t* pos, void* stream);

/* Ensemble step distribution (ensemble_captioning_model.py:66-83): `logits` is a HOST array of M (<= 8)
 * device pointers to fp32 [N, V] logits (row pitch ldl); out[n][v] = log(mean_m softmax(logits_m[n])[v]). */
int odic_ensemble_logprobs(const float* const* logits, int32_t M, int64_t ldl, float* out, int64_t ldo,
                           int32_t N, int32_t V, void* stream);
/* k largest entries of every row (ties → lower index), values taken as they are (rows already hold
 * log-probabilities, e.g. the output of odic_ensemble_logprobs): top_val fp32 [N,k], top_idx int32 [N,k]. */
int odic_topk_rows(const float* logp, int64_t ldl, float* top_val, int32_t* top_idx, int32_t N, int32_t V,
                   int32_t k, void* stream);

/* Beam bookkeeping of one search step on device (captioning_model.py:172-223; the call with
 * *pos == 0 is the seeding of :126-140).  All arrays are device resident.
 *   cand_val/cand_idx [n_img*beams, beams]: per-sequence top-k log-probs / words
 *   tokens int64 [n_img, beams, T] prefixes (tokens[:, :, 0] = SOS before the first call),
 *   logprobs fp32 [n_img, beams, T] per-token log-probs (slot 0 = 0), anc int32 [N, T],
 *   cumul fp32 [N], n_elem int32 [N] (length incl. SOS/EOS), has_eos int32 [N],
 *   row_valid int32 [N] (output: 1 while the beam was still growing), next_tok int64 [N] (output:
 *   token to feed at the next step), pos int32 scalar (incremented at the end),
 *   done int32 scalar (set to 1 when every beam has stopped growing, :222).
 */
typedef struct odic_beam_state {
  int64_t* tokens; float* logprobs; int32_t* anc;
  float* cumul; int32_t* n_elem; int32_t* has_eos; int32_t* row_valid; int64_t* next_tok;
  int32_t* pos; int32_t* done;
  int32_t* ctr;      /* int32 scalar, zero before the first call: inter-block arrival counter */
} odic_beam_state;
/* Optional tail of the launch that chooses the next words: the decoder input of the next position,
 *   y[n] = embed[word_n]·scale + pos_table[pos + 1]   (EmbeddingLayer, layers.py:118-121 — what odic_dec_embed does in
 * a launch of its own), written while pos + 1 <= T - 2 (the last prefix position is never fed back);
 * embed fp32 [V, d], pos_table fp32 [pos_rows, d], y fp32 [n_img·beams, d] (ldy); nothing is written for a position
 * pos + 1 >= pos_rows (device-side bound, as odic_dec_embed). */
typedef struct odic_embed_args {
  const float* embed; const float* pos_table; float* y; int64_t ldy; int32_t d; float scale; int32_t pos_rows;
} odic_embed_args;
/*   Limits: beams <= 16, 2 <= T <= 128 (per-token log-probs are staged in LDS), n_img <= 32767 (the
 *   arrival counter packs {arrivals, still-growing images} into one int32): ODIC_EINVAL otherwise.
 *   A call with *pos == T - 1 (the prefix is full) changes nothing.  emb: NULL or the embedding tail above. */
int odic_beam_step(const float* cand_val, const int32_t* cand_idx, const odic_beam_state* st,
                   const odic_embed_args* emb, int32_t n_img, int32_t beams, int32_t T, int64_t eos_idx,
                   void* stream);

/* The tail of a single-model search step in ONE launch (captioning_model.py:150-223): log_softmax of the step's
 * logits rows (fp32 [n_img·beams, V], ldl), their `beams` best words (ties → lower index), odic_beam_step on those
 * candidates (which never leave LDS) and, with emb, the next position's input.  Same limits as odic_beam_step. */
int odic_beam_search_step(const float* logits, int64_t ldl, int32_t V, const odic_beam_state* st,
                          const odic_embed_args* emb, int32_t n_img, int32_t beams, int32_t T, int64_t eos_idx,
                          void* stream);

/* Initial state of a search (captioning_model.py:117-125): tokens[:, :, 0] = sos, logprobs[:, :, 0] = 0,
 * next_tok = sos, row_valid = 1, *pos = *done = *ctr = 0; with emb, also the input of position 0 (the embedded
 * start token).  One launch instead of six fills on the latency-bound decode stream. */
int odic_beam_reset(const odic_beam_state* st, const odic_embed_args* emb, int32_t n_img, int32_t beams, int32_t T,
                    int64_t sos_idx, void* stream);

/* Final selection (captioning_model.py:225-241): score = cumul / n_elem, descending order per
 * image → order int32 [n_img, beams], score fp32 [n_img, beams]. */
int odic_beam_finalize(const odic_beam_state* st, int32_t* order, float* score, int32_t n_img,
                       int32_t beams, void* stream);

/* odic_beam_finalize + the best caption of every image in the fixed-shape form the multi-GPU gather
 * ships (captioning_model.py:225-241 with how_many_outputs = 1; test.py:216-224 takes output_words[i][0]):
 * out_tok int32 [n_img, T] = tokens of the best beam, positions >= its length filled with pad_idx;
 * out_len int32 [n_img] = its length (SOS and EOS included). */
int odic_beam_finalize_best(const odic_beam_state* st, int32_t* order, float* score, int32_t* out_tok,
                            int32_t* out_len, int32_t n_img, int32_t beams, int32_t T, int32_t pad_idx,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ODIC_HIP_H */
