"""Thin torch-tensor wrappers over the C ABI (include/odic_hip.h).

PyTorch is plumbing here: tensors own device memory and the current stream; every function hands
`data_ptr()`s and sizes to libodic_hip.so.  Nothing in this module computes on the CPU and nothing
falls back — a non-CUDA tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _hip
from ._hip import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, F16, F32, FP8, H2  # noqa: F401  (re-exported)

FP8_DTYPE = torch.float8_e4m3fn          # OCP e4m3: the fp8 format of gfx950's MFMA
FP8_MAX = 448.0
# Split-fp16 tensors (ODIC_H2: hi + lo fp16 pairs, 4 bytes per element in [8 hi | 8 lo] groups — include/odic_hip.h)
# travel as torch.int32 tensors of the LOGICAL shape: same sizes, strides and zero bytes as the fp32 tensor they
# replace, and a dtype no other operand of this library uses.
H2_DTYPE = torch.int32
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16, FP8_DTYPE: FP8, H2_DTYPE: H2}


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline pass): HIP events recorded on the launch stream
# around every C-ABI call, tagged with the kernel family and its algorithmic FLOPs / bytes.
# ----------------------------------------------------------------------------------------------
_PROFILE: Optional[list] = None


class profile:
    """`with ops.profile() as recs:` → recs = [(family, flops, bytes, start_evt, end_evt, detail), ...]"""

    def __enter__(self):
        global _PROFILE
        _PROFILE = []
        return _PROFILE

    def __exit__(self, *exc):
        global _PROFILE
        _PROFILE = None
        return False


class _timed:
    __slots__ = ("family", "flops", "bytes", "start", "detail")

    def __init__(self, family: str, flops: float, nbytes: float, detail: str = ""):
        self.family, self.flops, self.bytes, self.start, self.detail = family, flops, nbytes, None, detail

    def __enter__(self):
        if _PROFILE is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            _PROFILE.append((self.family, self.flops, self.bytes, self.start, end, self.detail))
        return False


#: position index of the decoder step being launched, set by the host loop while a profile is active
#: (the kernels read *pos from device memory; the host needs it only to price the step's cache reads)
_STEP_T: Optional[int] = None


def set_step_hint(t: Optional[int]) -> None:
    global _STEP_T
    _STEP_T = t


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("odic ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {dt}") from None


# ----------------------------------------------------------------------------------------------
# bf16 GEMM tile autotuning: "measure, don't guess".  While `autotune()` is active (the eager warm-up
# pass of a pipeline, never inside a stream capture) the first call of every distinct problem shape
# times the candidate tile configurations with HIP events on scratch outputs and remembers the
# fastest; later calls (including the captured ones) reuse the choice.
_TILE_CHOICE: dict = {}
_TUNING = False
# ODIC_TILE_CACHE=<file>: tile choices are loaded from / appended to a JSON file, so a later process (a profiler
# pass that must not contain tuning launches, a server restart) starts from the measured choices
_TILE_CACHE_FILE = os.environ.get("ODIC_TILE_CACHE")


def _cache_key(key) -> str:
    return "|".join(str(k) for k in key)


def _load_tile_cache() -> dict:
    if _TILE_CACHE_FILE and os.path.exists(_TILE_CACHE_FILE):
        import json
        with open(_TILE_CACHE_FILE) as f:
            return json.load(f)
    return {}


_TILE_CACHE: dict = _load_tile_cache()
_TILE_CANDIDATES = tuple(int(c) for c in os.environ.get("ODIC_TILE_CANDIDATES", "0,1,7,10,40,41,42,50,51,52,53").split(","))

# persistent tile configurations (16 + c) draw tiles from atomic counters in a 16-int workspace that is zero at
# launch and left zero by the kernel: launches of one stream are ordered, so one buffer per stream suffices
_GEMM_WS: dict = {}


def _gemm_workspace(device: torch.device) -> torch.Tensor:
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _GEMM_WS.get(key)
    if ws is None:
        ws = _GEMM_WS[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return ws


class autotune:
    def __enter__(self):
        global _TUNING
        self._prev, _TUNING = _TUNING, True
        return _TILE_CHOICE

    def __exit__(self, *exc):
        global _TUNING
        _TUNING = self._prev
        return False


_LOWP_CANDIDATES = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9)   # gemm_lowp.hip tile configurations (fp8 / fp16 operands; 5-9: the
                                                    # block-scaled fp8 MFMA, rejected for fp16 / K % 128 != 0 and then skipped)
_X3_CANDIDATES = tuple(int(c) for c in os.environ.get("ODIC_X3_TILE_CANDIDATES", "0,1,2,5,6,7,8").split(","))   # gemm_x3.hip


def _tune_gemm(args: "_hip.GemmArgs", key, out: torch.Tensor, candidates=None) -> int:
    lib = _hip.load()
    scratch = torch.empty_like(out)
    real_out = args.out
    args.out = scratch.data_ptr()
    best, best_ms = -1, float("inf")
    try:
        for cfg in (candidates or _TILE_CANDIDATES):
            args.tile_cfg = cfg
            if lib.odic_gemm(C.byref(args), _stream()) != 0:
                continue
            lib.odic_gemm(C.byref(args), _stream())
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(4):
                lib.odic_gemm(C.byref(args), _stream())
            en.record()
            en.synchronize()
            ms = st.elapsed_time(en)
            if ms < best_ms:
                best, best_ms = cfg, ms
    finally:
        args.out = real_out
    _TILE_CHOICE[key] = best
    if _TILE_CACHE_FILE:
        import json
        _TILE_CACHE[_cache_key(key)] = best
        with open(_TILE_CACHE_FILE, "w") as f:
            json.dump(_TILE_CACHE, f, indent=0)
    return best


def gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, *, act: int = ACT_NONE,
         alpha: float = 1.0, bias_axis: int = 0, out_dtype: Optional[torch.dtype] = None,
         M: Optional[int] = None, N: Optional[int] = None, K: Optional[int] = None,
         lda: Optional[int] = None, ldw: Optional[int] = None, ldr: Optional[int] = None,
         ldc: Optional[int] = None, batch: int = 1, strideA: int = 0, strideW: int = 0, strideBias: int = 0,
         strideR: int = 0, strideC: int = 0, ln_fold: Optional[tuple] = None, tile_cfg: int = -1,
         col_scale: Optional[torch.Tensor] = None, out_scale: float = 1.0,
         out16: Optional[torch.Tensor] = None, stats_out: Optional[torch.Tensor] = None,
         ln_stats: Optional[torch.Tensor] = None, a_ln: Optional[torch.Tensor] = None, ln_eps: float = 1e-5
         ) -> torch.Tensor:
    """out = act(alpha·A·Wᵀ + bias) + residual.  With no explicit dims, A is [..., K] (flattened to
    [M,K]) and W is [N,K], both contiguous.  Explicit dims / leading dimensions / batch strides allow
    strided sub-matrices (elements).  ln_fold = (colsum, eps): W and bias come from fold_layernorm() and
    the rows of A are LayerNorm-ed inside the product (fp32 skinny-M path only).  fp8 / fp16 operands (the
    low-precision backbone mode): `col_scale` fp32 [N] multiplies column n of A·Wᵀ (activation scale x weight channel
    scale) and `out_scale` the result before an fp8 / fp16 output cast.
    LayerNorm folded across two bf16 products: the PRODUCER (fp32 output) is given `out16` (bf16 [M,N]) and
    `stats_out` (fp32 [M, N/32, 2]); the CONSUMER reads that copy as A with `ln_stats=stats_out` and
    `ln_fold=(colsum, eps)` from fold_layernorm_bf16().
    LayerNorm while reading (A = None, `a_ln` = the fp32 rows [M,K], W / bias from fold_layernorm_bf16()): the bf16
    A-resident kernels normalise each row in registers — one launch for norm → linear (K = 192 / 384, whole tiles)."""
    if a_ln is not None:
        return _gemm_a_ln(a_ln, W, bias, out, act=act, alpha=alpha, out_dtype=out_dtype, ln_eps=ln_eps, tile_cfg=tile_cfg)
    _need_cuda(A, W, bias, residual, out, col_scale, out16, stats_out, ln_stats)
    if A.dtype != W.dtype:
        raise RuntimeError("A and W must share a dtype")
    if M is None:
        K = A.shape[-1]
        M = A.numel() // K
        N = W.shape[0]
        lda, ldw = K, W.shape[-1]
        if not (A.is_contiguous() and W.is_contiguous()):
            raise RuntimeError("gemm: implicit-shape operands must be contiguous")
    odt = out_dtype or (out.dtype if out is not None else A.dtype)
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (*A.shape[:-1], N), dtype=odt, device=A.device)
        if ldc is None:
            ldc = N
        if batch > 1 and strideC == 0:
            strideC = M * N
    if ldc is None:
        ldc = N
    if residual is not None and ldr is None:
        ldr = N
    if bias is not None and bias.dtype != torch.float32:
        raise RuntimeError("bias must be fp32")
    if residual is not None and residual.dtype != torch.float32:
        raise RuntimeError("residual must be fp32")
    a = _hip.GemmArgs(_p(A), _p(W), _p(bias), _p(residual), _p(out), M, N, K, lda, ldw, ldr or 0, ldc, batch,
                      strideA, strideW, strideBias, strideR, strideC, alpha, act, bias_axis,
                      dtype_code(A.dtype), dtype_code(out.dtype), tile_cfg,
                      _p(ln_fold[0]) if ln_fold else None, float(ln_fold[1]) if ln_fold else 0.0, None,
                      _p(col_scale), float(out_scale), _p(out16), out16.stride(0) if out16 is not None else 0,
                      _p(stats_out), _p(ln_stats))
    if A.dtype in (torch.bfloat16, torch.float16, FP8_DTYPE, H2_DTYPE):
        if batch == 1 and A.dtype == torch.bfloat16:
            a.workspace = _gemm_workspace(A.device).data_ptr()
        key = (A.dtype, M, N, K, batch, out.dtype, act, residual is not None, out16 is not None, ln_stats is not None)
        cfg = _TILE_CHOICE.get(key) if tile_cfg < 0 else tile_cfg
        if cfg is None and tile_cfg < 0 and _TILE_CACHE:
            cfg = _TILE_CACHE.get(_cache_key(key))
            if cfg is not None:
                _TILE_CHOICE[key] = cfg
        if cfg is None and _TUNING and _PROFILE is None and ldc == N and batch == 1 \
                and not torch.cuda.is_current_stream_capturing():
            cfg = _tune_gemm(a, key, out, None if A.dtype == torch.bfloat16 else
                             (_X3_CANDIDATES if A.dtype == H2_DTYPE else _LOWP_CANDIDATES))
        if cfg is not None:
            a.tile_cfg = cfg
    isz, osz = A.element_size(), out.element_size()
    nbytes = batch * ((M * K if strideA or batch == 1 else M * K / batch) * isz +
                      (N * K if strideW or batch == 1 else N * K / batch) * isz + M * N * osz +
                      (M * N * 4 if residual is not None else 0))
    fam = {torch.bfloat16: "gemm_bf16", torch.float16: "gemm_f16", FP8_DTYPE: "gemm_fp8",
           H2_DTYPE: "gemm_x3"}.get(A.dtype, "gemm_f32")
    with _timed(fam, 2.0 * M * N * K * batch, nbytes,
                f"{M}x{N}x{K}" + (f"x{batch}" if batch > 1 else "")):
        _hip.check(_hip.load().odic_gemm(C.byref(a), _stream()), "odic_gemm")
    return out


_A_LN_CANDIDATES = (50, 52, 51, 53)          # the A-resident tile configurations (gemm_bf16_apanel_kernel)
_A_LN_X3_CANDIDATES = (20, 21)               # gemm_x3_apanel_kernel (K = 192)


def _gemm_a_ln(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor], *, act: int,
               alpha: float, out_dtype: Optional[torch.dtype], ln_eps: float, tile_cfg: int) -> torch.Tensor:
    """out = act(alpha·LN0(x)·Wᵀ + bias), LN0 = LayerNorm without affine, computed while the fp32 rows are read
    (odic_gemm_args.a_ln).  x fp32 [M,K] contiguous, W bf16 [N,K] contiguous."""
    _need_cuda(x, W, bias, out)
    if x.dtype != torch.float32 or W.dtype not in (torch.bfloat16, H2_DTYPE) or not (x.is_contiguous() and W.is_contiguous()):
        raise RuntimeError("gemm(a_ln=...): contiguous fp32 rows and a contiguous bf16 / split-fp16 weight")
    K = x.shape[-1]
    M, N = x.numel() // K, W.shape[0]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=out_dtype or W.dtype, device=x.device)
    a = _hip.GemmArgs(None, _p(W), _p(bias), None, _p(out), M, N, K, K, K, 0, N, 1, 0, 0, 0, 0, 0, alpha, act, 0,
                      dtype_code(W.dtype), dtype_code(out.dtype), tile_cfg, None, float(ln_eps), None, None, 1.0, None, 0, None, None,
                      _p(x), K)
    key = ("a_ln", W.dtype, M, N, K, out.dtype, act)
    cfg = _TILE_CHOICE.get(key) if tile_cfg < 0 else tile_cfg
    if cfg is None and _TILE_CACHE:
        cfg = _TILE_CACHE.get(_cache_key(key))
        if cfg is not None:
            _TILE_CHOICE[key] = cfg
    if cfg is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("gemm(a_ln=...): no tile choice for this shape yet — run it once outside graph capture")
        cfg = _tune_gemm(a, key, out, _A_LN_CANDIDATES if W.dtype == torch.bfloat16 else _A_LN_X3_CANDIDATES)
        if cfg < 0:
            raise RuntimeError(f"gemm(a_ln=...): no A-resident tile configuration takes {M}x{N}x{K}")
    a.tile_cfg = cfg
    with _timed("gemm_bf16" if W.dtype == torch.bfloat16 else "gemm_x3", 2.0 * M * N * K,
                M * K * 4 + N * K * W.element_size() + M * N * out.element_size(), f"ln+{M}x{N}x{K}"):
        _hip.check(_hip.load().odic_gemm(C.byref(a), _stream()), "odic_gemm")
    return out


def a_ln_supported(M: int, N: int, K: int, dtype=torch.bfloat16) -> bool:
    """Shapes the LayerNorm-while-reading form takes: K = 192 (128-row panels, 64-column chunks) or 384 (128 / 32); split
    fp16: K = 192 (64-row panels, 32-column chunks)."""
    if dtype == H2_DTYPE:
        return K == 192 and M % 64 == 0 and N % 32 == 0
    return (K == 192 and M % 128 == 0 and N % 64 == 0) or (K == 384 and M % 128 == 0 and N % 32 == 0)


def fold_layernorm(W: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """Weight-pack-time half of odic_gemm's folded LayerNorm: (W·diag(gamma), bias + W·beta, row sums of
    W·diag(gamma)), all fp32, computed in fp64.  LayerNorm(a)·Wᵀ + bias = rstd·(a·W'ᵀ − mean·colsum) + bias'."""
    W64 = W.double()
    Wg = W64 * gamma.double()[None, :]
    b2 = W64 @ beta.double() + (bias.double() if bias is not None else 0.0)
    return Wg.float().contiguous(), b2.float().contiguous(), Wg.sum(1).float().contiguous()


def fold_layernorm_bf16(W: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """Weight-pack-time half of the LayerNorm folded across two bf16 products: (bf16 W·diag(gamma), fp32 bias + W·beta,
    fp32 row sums of the ROUNDED bf16 weights — the kernel subtracts mean·colsum from a product of exactly those)."""
    Wg, b2, _ = fold_layernorm(W, bias, gamma, beta)
    W16 = Wg.to(torch.bfloat16).contiguous()
    return W16, b2, W16.double().sum(1).float().contiguous()


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, eps: float = 1e-5,
              out_dtype: torch.dtype = torch.float32, out: Optional[torch.Tensor] = None,
              M: Optional[int] = None, C_: Optional[int] = None, ldx: Optional[int] = None) -> torch.Tensor:
    _need_cuda(x, gamma, beta, out)
    if M is None:
        C_ = x.shape[-1]
        M = x.numel() // C_
        ldx = C_
        if not x.is_contiguous():
            raise RuntimeError("layernorm: implicit-shape input must be contiguous")
    if out is None:
        out = torch.empty((M, C_) if x.dim() < 2 or ldx != C_ else x.shape, dtype=out_dtype, device=x.device)
    with _timed("layernorm", 8.0 * M * C_, M * C_ * (4 + out.element_size())):
        _hip.check(_hip.load().odic_layernorm(_p(x), ldx, _p(gamma), _p(beta), _p(out), M, C_, eps,
                                              dtype_code(out.dtype), _stream()), "odic_layernorm")
    return out


def cast_bf16(x: torch.Tensor, *, M: Optional[int] = None, C_: Optional[int] = None, ldx: Optional[int] = None
              ) -> torch.Tensor:
    """fp32 [M,C] (row stride ldx) → contiguous bf16 [M,C]."""
    _need_cuda(x)
    if M is None:
        C_ = x.shape[-1]
        M = x.numel() // C_
        ldx = C_
    out = torch.empty((M, C_), dtype=torch.bfloat16, device=x.device)
    with _timed("cast_bf16", 0.0, M * C_ * 6.0):
        _hip.check(_hip.load().odic_cast_f32_to_bf16(_p(x), ldx, _p(out), C_, M, C_, _stream()), "odic_cast_f32_to_bf16")
    return out


def cast_h2(x: torch.Tensor, *, M: Optional[int] = None, C_: Optional[int] = None, ldx: Optional[int] = None
            ) -> torch.Tensor:
    """fp32 [M,C] (row stride ldx) → contiguous split-fp16 [M,C] (an int32 tensor, see H2_DTYPE)."""
    _need_cuda(x)
    if M is None:
        C_ = x.shape[-1]
        M = x.numel() // C_
        ldx = C_
    out = torch.empty((M, C_), dtype=H2_DTYPE, device=x.device)
    with _timed("cast_h2", 0.0, M * C_ * 8.0):
        _hip.check(_hip.load().odic_cast_f32_to_h2(_p(x), ldx, _p(out), C_, M, C_, _stream()), "odic_cast_f32_to_h2")
    return out


def h2_from_f32(x: torch.Tensor) -> torch.Tensor:
    """Weight-pack-time split of an fp32 tensor [..., K] (K % 8 == 0) into the h2 layout, on the tensor's device with
    plain torch ops (exact: two round-to-nearest-even fp16 conversions)."""
    if x.shape[-1] % 8:
        raise RuntimeError("h2 rows are whole groups of 8 elements")
    x = x.detach().float().clamp(-65504.0, 65504.0)
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    g = torch.stack([hi.reshape(*x.shape[:-1], -1, 8), lo.reshape(*x.shape[:-1], -1, 8)], dim=-2)   # [..., K/8, 2, 8]
    return g.contiguous().view(torch.int32).reshape(x.shape)


def h2_to_f32(t: torch.Tensor) -> torch.Tensor:
    """Inverse of h2_from_f32 (tests, taps): hi + lo in fp32."""
    g = t.contiguous().view(torch.float16).reshape(*t.shape[:-1], -1, 2, 8).float()
    return (g[..., 0, :] + g[..., 1, :]).reshape(t.shape)


def pow2_scale_for_h2(w: torch.Tensor) -> float:
    """Power of two s such that max|w|·s lies in [2^12, 2^13): the hi parts sit far from the fp16 overflow and the lo
    parts (|lo| <= |w|·s·2^-11) of all but negligible elements stay fp16 normals.  Exact to undo (alpha = 1/s)."""
    import math
    m = float(w.detach().abs().max())
    if not (m > 0.0) or not math.isfinite(m):
        return 1.0
    return float(2.0 ** (12 - math.floor(math.log2(m))))


def copy(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst ← src (contiguous, same shape and dtype, a multiple of 16 bytes) by a kernel of this library."""
    _need_cuda(src, dst)
    if src.shape != dst.shape or src.dtype != dst.dtype or not src.is_contiguous() or not dst.is_contiguous():
        raise RuntimeError("copy: contiguous tensors of one shape and dtype expected")
    nbytes = src.numel() * src.element_size()
    with _timed("copy", 0.0, 2.0 * nbytes):
        _hip.check(_hip.load().odic_copy(_p(src), _p(dst), nbytes, _stream()), "odic_copy")
    return dst


def patch_merge_layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, B: int, res: int, Cin: int,
                          *, eps: float = 1e-5, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    _need_cuda(x, gamma, beta)
    out = torch.empty((B, (res // 2) ** 2, 4 * Cin), dtype=out_dtype, device=x.device)
    with _timed("patch_merge_layernorm", 8.0 * out.numel(), out.numel() * (4 + out.element_size())):
        _hip.check(_hip.load().odic_patch_merge_layernorm(_p(x), _p(gamma), _p(beta), _p(out), B, res, Cin, eps,
                                                          dtype_code(out_dtype), _stream()),
                   "odic_patch_merge_layernorm")
    return out


def patch_embed(img: torch.Tensor, w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                patch: int, *, eps: float = 1e-5) -> torch.Tensor:
    _need_cuda(img, w, b, gamma, beta)
    B, Cin, H, Wd = img.shape
    Cout = w.shape[0]
    if not img.is_contiguous() or img.dtype != torch.float32:
        raise RuntimeError("patch_embed: image batch must be contiguous fp32 [B,C,H,W]")
    out = torch.empty((B, (H // patch) * (Wd // patch), Cout), dtype=torch.float32, device=img.device)
    with _timed("patch_embed", 2.0 * out.numel() * Cin * patch * patch, (img.numel() + out.numel()) * 4):
        _hip.check(_hip.load().odic_patch_embed(_p(img), _p(w), _p(b), _p(gamma), _p(beta), _p(out), B, Cin, H, Wd,
                                                patch, Cout, eps, _stream()), "odic_patch_embed")
    return out


def shifted_bias_prescaled(bias_table: torch.Tensor, ws: int, scale: float) -> torch.Tensor:
    """[heads, 4, 576] fp32 — weight-pack-time plumbing for odic_window_attention's fast path (ws = 12): the
    relative-position bias table (reference swin_transformer_mod.py:163-173,196-198) with its x axis reversed and
    rows padded to 24, divided by `scale`, in four copies shifted by 0..3 floats, so that the biases of the four
    consecutive keys of an MFMA accumulator quad are one aligned 16-byte LDS read (include/odic_hip.h)."""
    if ws != 12:
        raise RuntimeError("the packed bias layout is specialised for 12x12 windows")
    heads = bias_table.shape[1]
    t = (bias_table.double() / scale).float().view(23, 23, heads)           # [r, c, head]
    R = torch.zeros(heads, 23 * 24 + 32, dtype=torch.float32, device=bias_table.device)
    R[:, :23 * 24].view(heads, 23, 24)[:, :, :23] = t.flip(1).permute(2, 0, 1)
    out = torch.zeros(heads, 4, 576, dtype=torch.float32, device=bias_table.device)
    for s_ in range(4):
        out[:, s_, :] = R[:, s_:s_ + 576]
    return out.contiguous()


def swin_qkv_attention(x: torch.Tensor, w_folded: torch.Tensor, b_folded: torch.Tensor, bias_shifted_prescaled: torch.Tensor,
                       B: int, res: int, C_: int, heads: int, ws: int, shift: int, *, eps: float = 1e-5,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """norm1 → qkv → attention core in one launch (odic_swin_qkv_attention; width 192, ws 12).  x fp32 [B·res², C] contiguous,
    (w_folded, b_folded) from fold_layernorm_bf16(qkv.weight, qkv.bias, norm1.weight, norm1.bias) → bf16 [B·res², C]."""
    _need_cuda(x, w_folded, b_folded, bias_shifted_prescaled, out)
    if x.dtype != torch.float32 or w_folded.dtype != torch.bfloat16 or not (x.is_contiguous() and w_folded.is_contiguous()):
        raise RuntimeError("swin_qkv_attention: contiguous fp32 rows and a contiguous bf16 folded weight")
    if out is None:
        out = torch.empty(B * res * res, C_, dtype=torch.bfloat16, device=x.device)
    nwh = B * (res // ws) ** 2 * heads
    with _timed("swin_qkv_attention", 2.0 * B * res * res * C_ * 3 * C_ + nwh * 2654208.0,
                B * res * res * C_ * (4 + 2) + 3 * C_ * C_ * 2, f"res{res}h{heads}"):
        _hip.check(_hip.load().odic_swin_qkv_attention(_p(x), C_, _p(w_folded), _p(b_folded), _p(bias_shifted_prescaled),
                                                      _p(out), B, res, C_, heads, ws, shift, (C_ // heads) ** -0.5, eps,
                                                      _stream()), "odic_swin_qkv_attention")
    return out


def window_attention(qkv: torch.Tensor, bias_table: torch.Tensor, B: int, res: int, C_: int, heads: int, ws: int,
                     shift: int, *, scale: Optional[float] = None, out: Optional[torch.Tensor] = None,
                     bias_shifted_prescaled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`bias_shifted_prescaled` must have been packed with the same `scale` (default head_dim^-0.5)."""
    _need_cuda(qkv, bias_table, out, bias_shifted_prescaled)
    if scale is None:
        scale = (C_ // heads) ** -0.5
    if out is None:
        out = torch.empty((B * res * res, C_), dtype=qkv.dtype, device=qkv.device)
    # algorithmic work per (window, head): QKᵀ + PV = 4·N²·hd FLOP; q,k,v in + o out = 4·N·hd elements
    inst = B * (res // ws) ** 2 * heads
    n = ws * ws
    fam = ("window_attention_bf16" if qkv.dtype in (torch.bfloat16, torch.float16) else
           "window_attention_x3" if qkv.dtype == H2_DTYPE else "window_attention_f32")
    with _timed(fam, inst * 4.0 * n * n * 32, inst * 4.0 * n * 32 * qkv.element_size(), f"res{res}h{heads}"):
        _hip.check(_hip.load().odic_window_attention(_p(qkv), _p(bias_table), _p(bias_shifted_prescaled), _p(out), B, res, C_,
                                                     heads, ws, shift,
                                                     scale, dtype_code(qkv.dtype), _stream()),
                   "odic_window_attention")
    return out


# ----------------------------------------------------------------------------------------------
def stcexp_group_meta(groups: Sequence[int], device) -> torch.Tensor:
    starts = [0]
    gid = []
    for g, n in enumerate(groups):
        starts.append(starts[-1] + n)
        gid += [g] * n
    return torch.tensor(starts + gid, dtype=torch.int32, device=device)


def stcexp_normalize(z: torch.Tensor, enc_len: torch.Tensor, group_meta: torch.Tensor, ngroups: int,
                     pos_fw: torch.Tensor, neg_fw: torch.Tensor, pos_bw: torch.Tensor, neg_bw: torch.Tensor,
                     colsum_ws: torch.Tensor, *, eps: float = 1e-9, scale_fw: float = 1.0, scale_bw: float = 1.0) -> None:
    """Outputs may be fp32, bf16 or split fp16 and wider than (S / nq): padding columns are zero-filled.  scale_fw /
    scale_bw multiply the forward / backward tables (powers of two that keep a split-fp16 table's lo halves in the
    fp16 normal range; the consumer undoes them in its alpha)."""
    _need_cuda(z, enc_len, group_meta, pos_fw, neg_fw, pos_bw, neg_bw, colsum_ws)
    B, nq, S = z.shape
    # algorithmic bytes: z read once, the four normalised weight tables written once (incl. their zero K padding)
    osz = pos_fw.element_size()
    with _timed("stcexp_normalize", 6.0 * B * nq * S,
                B * nq * S * 4.0 + 2.0 * B * nq * pos_fw.shape[-1] * osz + 2.0 * B * S * pos_bw.shape[-1] * osz):
        _hip.check(_hip.load().odic_stcexp_normalize(_p(z), _p(enc_len), _p(group_meta), ngroups, _p(pos_fw),
                                                     _p(neg_fw), pos_fw.shape[-1], _p(pos_bw), _p(neg_bw),
                                                     pos_bw.shape[-1], _p(colsum_ws), B, nq, S, eps, scale_fw, scale_bw,
                                                     dtype_code(pos_fw.dtype), _stream()), "odic_stcexp_normalize")


def selector_mix(x, ldx, sel_pre, lds, a, lda, b, ldb, out, ldo, M, d) -> None:
    _need_cuda(x, sel_pre, a, b, out)
    with _timed("selector_mix", 6.0 * M * d, 5.0 * M * d * 4):     # x, selector, A', B' in; y out
        _hip.check(_hip.load().odic_selector_mix(_p(x), ldx, _p(sel_pre), lds, _p(a), lda, _p(b), ldb, _p(out), ldo,
                                                 M, d, _stream()), "odic_selector_mix")


# ----------------------------------------------------------------------------------------------
def dec_embed(tokens, embed, pos_table, pos, y, ldy, N, d, scale) -> None:
    _need_cuda(tokens, embed, pos_table, pos, y)
    with _timed("dec_embed", 2.0 * N * d, N * (8.0 + 2 * d * 4) + d * 4):     # token id, embedding row in, y row out, one pos row
        _hip.check(_hip.load().odic_dec_embed(_p(tokens), _p(embed), _p(pos_table), _p(pos), _p(y), ldy, N, d,
                                              pos_table.shape[0], scale, _stream()), "odic_dec_embed")


def dynexp_step(lin, ldlin, qexp, bexp, cond_c, key_c, va_c, vb_c, wfa_c, wfb_c, qk_c, anc, row_valid, pos,
                y_in, ldy_in, y, ldy, N, T, d, E, eps=1e-9) -> None:
    _need_cuda(lin, qexp, bexp, cond_c, key_c, va_c, vb_c, wfa_c, wfb_c, qk_c, anc, row_valid, pos, y_in, y)
    # algorithmic bytes of one incremental step at position t (layers.py:152-204 on the newest row only, in the
    # re-associated form of csrc/decoder_ops.hip): in  lin [N,5d], y_in; through the ancestor table, per earlier
    # position cond, key, va, vb [d each], its (j+1)·E forward weights (x2) and query·key [E];  out  the position's
    # cache rows and y.   t = the host's step hint, else the mean position of a T-long search.
    t = _STEP_T if _STEP_T is not None else (T - 1) / 2.0
    nbytes = N * ((5 * d + 2 * d) * 4.0 + (t + 1) * (4 * d + E) * 4.0 + (t + 1) * (t + 2) * E * 4.0
                  + (4 * d + 2 * (t + 1) * E + E) * 4.0 + (t + 1) * 4.0)
    flops = N * ((2 * t + E + 1) * 2.0 * d + 6.0 * (t + 1) * d + 2.0 * (t + 1) * (t + 2) * E)
    with _timed("dynexp_step", flops, nbytes):
        _hip.check(_hip.load().odic_dynexp_step(_p(lin), ldlin, _p(qexp), _p(bexp), _p(cond_c), _p(key_c), _p(va_c),
                                                _p(vb_c), _p(wfa_c), _p(wfb_c), _p(qk_c), _p(anc), _p(row_valid),
                                                _p(pos), _p(y_in), ldy_in, _p(y), ldy, N, T, d, E, eps,
                                                _stream()),
                   "odic_dynexp_step")


def cross_attn_step(q, ldq, kv, ldkv, koff, voff, enc_len, row_valid, out, ldo, N, n_img, S, d, heads) -> None:
    _need_cuda(q, kv, enc_len, row_valid, out)
    # q and out rows per sequence, K and V of the S encoder tokens ONCE per image (shared by its beams)
    with _timed("cross_attn_step", 4.0 * N * S * d, (2.0 * N * d + 2.0 * n_img * S * d) * 4):
        _hip.check(_hip.load().odic_cross_attn_step(_p(q), ldq, _p(kv), ldkv, koff, voff, _p(enc_len), _p(row_valid),
                                                    _p(out), ldo, N, n_img, S, d, heads, _stream()),
                   "odic_cross_attn_step")


def logsoftmax_topk(logits, ldl, logp_out, ldp, top_val, top_idx, N, V, k) -> None:
    _need_cuda(logits, logp_out, top_val, top_idx)
    with _timed("logsoftmax_topk", 4.0 * N * V, N * V * 4.0 * (2 if logp_out is not None else 1) + N * k * 8.0):
        _hip.check(_hip.load().odic_logsoftmax_topk(_p(logits), ldl, _p(logp_out), ldp, _p(top_val), _p(top_idx), N, V,
                                                    k, _stream()), "odic_logsoftmax_topk")


def logsoftmax_sample(logits, ldl, logp_out, ldp, top_val, top_idx, N, V, k, seed: int, pos=None) -> None:
    """k words per row drawn without replacement from softmax(logits) on the device (Gumbel-top-k, Philox noise
    keyed by `seed`, the row, the word and *pos) + their log-probs — the reference's multinomial draws."""
    _need_cuda(logits, logp_out, top_val, top_idx, pos)
    with _timed("logsoftmax_topk", 14.0 * N * V, N * V * 4.0 * (2 if logp_out is not None else 1) + N * k * 8.0):
        _hip.check(_hip.load().odic_logsoftmax_sample(_p(logits), ldl, _p(logp_out), ldp, _p(top_val), _p(top_idx), N, V,
                                                      k, seed & 0xFFFFFFFFFFFFFFFF, _p(pos), _stream()),
                   "odic_logsoftmax_sample")


def ensemble_logprobs(logits_list, out: torch.Tensor) -> None:
    """out[n] = log(mean_m softmax(logits_m[n])) — ensemble_captioning_model.py:66-83."""
    _need_cuda(out, *logits_list)
    M = len(logits_list)
    N, V = out.shape
    arr = (C.c_void_p * M)(*[t.data_ptr() for t in logits_list])
    with _timed("ensemble_logprobs", 0.0, 4.0 * (M + 1) * N * V):
        _hip.check(_hip.load().odic_ensemble_logprobs(arr, M, logits_list[0].stride(0), _p(out), out.stride(0), N, V,
                                                      _stream()), "odic_ensemble_logprobs")


def topk_rows(logp: torch.Tensor, top_val: torch.Tensor, top_idx: torch.Tensor, k: int) -> None:
    _need_cuda(logp, top_val, top_idx)
    N, V = logp.shape
    with _timed("logsoftmax_topk", 0.0, 4.0 * N * V):
        _hip.check(_hip.load().odic_topk_rows(_p(logp), logp.stride(0), _p(top_val), _p(top_idx), N, V, k, _stream()),
                   "odic_topk_rows")


def embed_args(embed, pos_table, y, ldy, d, scale) -> "_hip.EmbedArgs":
    """odic_embed_args for beam_step / beam_search_step / beam_reset (the caller keeps the tensors alive)."""
    _need_cuda(embed, pos_table, y)
    return _hip.EmbedArgs(_p(embed), _p(pos_table), _p(y), ldy, d, scale, pos_table.shape[0])


def _emb_ref(emb):
    return C.byref(emb) if emb is not None else None


def _beam_bytes(n_img, beams, T, emb):
    # the k prefixes (token int64, log-prob, ancestor) of t+1 positions read and re-written, per-beam flags, and
    # with the embedding tail one table row in and one input row out per beam
    t = _STEP_T if _STEP_T is not None else (T - 1) / 2.0
    return n_img * (2.0 * beams * (t + 1) * 16.0 + beams * 32.0 + (beams * emb.d * 8.0 if emb is not None else 0.0))


def beam_step(cand_val, cand_idx, state: "_hip.BeamState", n_img, beams, T, eos_idx, emb=None) -> None:
    with _timed("beam_step", 0.0, n_img * beams * beams * 8.0 + _beam_bytes(n_img, beams, T, emb)):
        _hip.check(_hip.load().odic_beam_step(_p(cand_val), _p(cand_idx), C.byref(state), _emb_ref(emb), n_img, beams,
                                              T, eos_idx, _stream()), "odic_beam_step")


def beam_search_step(logits, ldl, V, state: "_hip.BeamState", n_img, beams, T, eos_idx, emb=None) -> None:
    """log-softmax + top-k of the step's logits rows, the beam update and (emb) the next input, in one launch."""
    _need_cuda(logits)
    with _timed("beam_search_step", 4.0 * n_img * beams * V, n_img * beams * V * 4.0 + _beam_bytes(n_img, beams, T, emb)):
        _hip.check(_hip.load().odic_beam_search_step(_p(logits), ldl, V, C.byref(state), _emb_ref(emb), n_img, beams, T,
                                                     eos_idx, _stream()), "odic_beam_search_step")


def beam_finalize(state: "_hip.BeamState", order, score, n_img, beams) -> None:
    with _timed("beam_finalize", 0.0, n_img * beams * 16.0):
        _hip.check(_hip.load().odic_beam_finalize(C.byref(state), _p(order), _p(score), n_img, beams, _stream()),
                   "odic_beam_finalize")


def beam_finalize_best(state: "_hip.BeamState", order, score, out_tok, out_len, n_img, beams, T, pad_idx) -> None:
    """Final ranking + the best caption per image as an int32 [n_img, T] row padded with `pad_idx` and its length."""
    _need_cuda(order, score, out_tok, out_len)
    with _timed("beam_finalize", 0.0, n_img * (beams * 8.0 + T * 12.0)):
        _hip.check(_hip.load().odic_beam_finalize_best(C.byref(state), _p(order), _p(score), _p(out_tok), _p(out_len),
                                                       n_img, beams, T, pad_idx, _stream()), "odic_beam_finalize_best")


def beam_reset(state: "_hip.BeamState", n_img, beams, T, sos_idx, emb=None) -> None:
    with _timed("beam_reset", 0.0, n_img * beams * (28.0 + (emb.d * 4.0 if emb is not None else 0.0))):
        _hip.check(_hip.load().odic_beam_reset(C.byref(state), _emb_ref(emb), n_img, beams, T, sos_idx, _stream()),
                   "odic_beam_reset")


def quantize_fp8_per_channel(w: torch.Tensor):
    """Weight-pack-time plumbing of the fp8 mode: W fp32 [N, K] → (fp8 e4m3 [N, K], scale fp32 [N]) with
    W ≈ fp8·scale[:, None], scale = max|row| / 448 (the whole e4m3 range per output channel)."""
    w = w.detach().float()
    scale = (w.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX)
    q = (w / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).cpu().to(FP8_DTYPE)        # host cast: exact OCP round-to-nearest-even
    return q.to(w.device).contiguous(), scale.contiguous()
