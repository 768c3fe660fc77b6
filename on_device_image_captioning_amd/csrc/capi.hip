// C-ABI glue: argument validation + dtype dispatch for the entry points whose kernels live in
// several translation units.  No state, no allocation, no synchronisation.
#include "odic_common.h"

int odic_gemm_bf16_launch(const odic_gemm_args* a, hipStream_t stream);
int odic_gemm_f32_launch(const odic_gemm_args* a, hipStream_t stream);
int odic_gemm_lowp_launch(const odic_gemm_args* a, hipStream_t stream);
int odic_gemm_x3_launch(const odic_gemm_args* a, hipStream_t stream);

extern "C" int odic_abi_version(void) { return ODIC_ABI_VERSION; }

extern "C" const char* odic_build_info(void) {
#ifdef ODIC_EXPERIMENTAL_GEMM
  return "libodic_hip gfx950 (CDNA4) experimental-gemm " __DATE__ " " __TIME__ " hipcc " __clang_version__;
#else
  return "libodic_hip gfx950 (CDNA4) " __DATE__ " " __TIME__ " hipcc " __clang_version__;
#endif
}

extern "C" int odic_gemm(const odic_gemm_args* a, void* stream) {
  if (!a || !a->W || !a->out) return ODIC_ENULL;
  if (a->a_ln) {                           // LayerNorm-while-reading form: bf16 W, no A, whole fp32 rows of K elements
    if (a->A || (a->in_dtype != ODIC_BF16 && a->in_dtype != ODIC_H2) || a->ld_aln < a->K || a->batch != 1) return ODIC_EINVAL;
  } else if (!a->A) {
    return ODIC_ENULL;
  }
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) return ODIC_EINVAL;
  if ((a->A && a->lda < a->K) || a->ldw < a->K || a->ldc < a->N) return ODIC_EINVAL;
  if (a->residual && a->ldr < a->N) return ODIC_EINVAL;
  if (a->act < ODIC_ACT_NONE || a->act > ODIC_ACT_SIGMOID) return ODIC_EINVAL;
  if (a->bias_axis != 0 && a->bias_axis != 1) return ODIC_EINVAL;
  if (a->batch > 65535) return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->in_dtype == ODIC_FP8 || a->in_dtype == ODIC_F16) return odic_gemm_lowp_launch(a, s);
  if (a->in_dtype == ODIC_H2) return odic_gemm_x3_launch(a, s);
  if (a->out_dtype != ODIC_F32 && a->out_dtype != ODIC_BF16) return ODIC_EINVAL;
  if (a->in_dtype == ODIC_BF16) return odic_gemm_bf16_launch(a, s);
  if (a->in_dtype == ODIC_F32) return odic_gemm_f32_launch(a, s);
  return ODIC_EINVAL;
}
