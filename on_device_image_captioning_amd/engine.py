"""Host-side sequencing of the HIP kernels: Swin backbone, expansion encoder, incremental decoder.

Everything here is plumbing: weights are packed once per (state dict, device, precision) into the
layouts the kernels want, activations live in torch-owned device buffers, and every arithmetic
step is a call into libodic_hip.so on the current stream (so a whole forward can be captured into
a hipGraph by `torch.cuda.graph`).  There is no CPU path.

Data layout in HBM
  * residual stream            fp32 [B·L, C]  (both precisions; 24 blocks deep, kept exact)
  * LN outputs / qkv / attention out / MLP hidden   fp32 or bf16 [B·L, *] token-major, never
    window-permuted: the (shifted) window gather/scatter happens inside the attention kernel
  * encoder / decoder          fp32; per-layer residual streams are column slices of one
    [rows, N_layers·d] buffer so the reference's torch.cat (End_ExpansionNet_v2.py:97,133) is free
  * decoder caches             [T, N, ...] indexed by (position, slot) + an ancestor table
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _hip, ops
from .weights import Geometry

SD = Dict[str, torch.Tensor]
# precision → dtype of the backbone's bf16-class activations / weights.  "fp8" (BASELINE.json configs[4]) keeps that
# at bf16 for the patch-merge reductions and the backbone output, and runs the Swin blocks with fp8 GEMM operands
# (qkv, fc1, fc2: 11/12 of the block FLOPs) and fp16 qkv / attention activations (proj on the fp16 MFMA).
# "x3" (the near-exact fast mode): activations and weights travel as split-fp16 pairs (ops.H2_DTYPE: hi + lo, 22
# significand bits, 4 bytes per element) and every contraction of the backbone and of the expansion encoder is three
# fp16 MFMAs (csrc/gemm_x3.hip, window_attention_h2_kernel); residual streams, normalisations and the decoder are fp32.
_CDT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp8": torch.bfloat16, "x3": ops.H2_DTYPE}


def pack_operand(t: torch.Tensor, cdt) -> Tuple[torch.Tensor, float]:
    """fp32 weight [N, K] → (GEMM operand in the engine's dtype, alpha that undoes its pack-time scale).  Split-fp16
    operands are scaled by a power of two so that the lo halves are fp16 normals (exact to undo)."""
    if cdt == ops.H2_DTYPE:
        sc = ops.pow2_scale_for_h2(t)
        return ops.h2_from_f32(t * sc), 1.0 / sc
    return t.to(cdt).contiguous(), 1.0


def _dev(t: torch.Tensor, device, dtype=None) -> torch.Tensor:
    t = t.detach()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(device).contiguous()


# =================================================================================================
# Swin backbone  
# =================================================================================================
class SwinEngine:
    def __init__(self, sd: SD, g: Geometry, device, precision: str = "fp32", calibration_images=None):
        if precision not in _CDT:
            raise ValueError(f"precision must be one of {list(_CDT)}")
        self.g, self.device, self.precision = g, device, precision
        self.cdt = _CDT[precision]
        fp8 = precision == "fp8"
        fast_attn = precision in ("bf16", "fp8", "x3")
        P = "swin_transf"
        f32 = lambda k: _dev(sd[k], device, torch.float32)          # noqa: E731
        cw = lambda k: pack_operand(_dev(sd[k], device, torch.float32), self.cdt)      # noqa: E731  (operand, alpha)
        self.pe_w = _dev(sd[f"{P}.patch_embed.proj.weight"].reshape(g.swin_embed_dim, -1), device, torch.float32)
        self.pe_b = f32(f"{P}.patch_embed.proj.bias")
        self.pe_g, self.pe_beta = f32(f"{P}.patch_embed.norm.weight"), f32(f"{P}.patch_embed.norm.bias")
        self.stages = []
        for s, depth in enumerate(g.swin_depths):
            blocks = []
            for b in range(depth):
                p = f"{P}.layers.{s}.blocks.{b}"
                (qkv_w, qkv_a), (proj_w, proj_a) = cw(p + ".attn.qkv.weight"), cw(p + ".attn.proj.weight")
                (fc1_w, fc1_a), (fc2_w, fc2_a) = cw(p + ".mlp.fc1.weight"), cw(p + ".mlp.fc2.weight")
                blocks.append(dict(
                    n1w=f32(p + ".norm1.weight"), n1b=f32(p + ".norm1.bias"),
                    qkv_w=qkv_w, qkv_a=qkv_a, qkv_b=f32(p + ".attn.qkv.bias"),
                    table=f32(p + ".attn.relative_position_bias_table"),
                    dense=(ops.shifted_bias_prescaled(f32(p + ".attn.relative_position_bias_table"), g.stage_window(s),
                                                      (g.stage_dim(s) // g.swin_num_heads[s]) ** -0.5)
                           if fast_attn and g.stage_window(s) == 12 else None),
                    proj_w=proj_w, proj_a=proj_a, proj_b=f32(p + ".attn.proj.bias"),
                    n2w=f32(p + ".norm2.weight"), n2b=f32(p + ".norm2.bias"),
                    fc1_w=fc1_w, fc1_a=fc1_a, fc1_b=f32(p + ".mlp.fc1.bias"),
                    fc2_w=fc2_w, fc2_a=fc2_a, fc2_b=f32(p + ".mlp.fc2.bias"),
                    shift=g.stage_shift(s, b)))
            down = None
            if s < len(g.swin_depths) - 1:
                p = f"{P}.layers.{s}.downsample"
                red_w, red_a = cw(p + ".reduction.weight")
                down = dict(nw=f32(p + ".norm.weight"), nb=f32(p + ".norm.bias"), red_w=red_w, red_a=red_a)
            self.stages.append((blocks, down))
        self.fn_w, self.fn_b = f32(f"{P}.norm.weight"), f32(f"{P}.norm.bias")
        self.stage_chunks = [int(v) for v in os.environ.get("ODIC_SWIN_CHUNKS", "1").split(",")]
        # Optional (ODIC_FOLD_BACKBONE_LN=1, bf16 mode): norm2 and the norm1 of every block but a stage's first folded
        # across the products around them (ops.gemm producer / consumer form): the proj / fc2 product leaves a bf16
        # copy of the residual stream plus row-group moments, the fc1 / next qkv product normalises in its epilogue.
        # Equally accurate (tools/fold_diag.py) and 44 launches fewer, but measured 1.5 % SLOWER end to end: the
        # LayerNorm kernels already run at 5 TB/s, and the producer's extra stores + the consumer's moment combine
        # cost the four products of a block 20 µs against the 16 µs of the two launches they replace.
        # (the fold's GEMM forms are compiled only into -DODIC_EXPERIMENTAL_GEMM builds of the library)
        self.fold_ln = precision == "bf16" and os.environ.get("ODIC_FOLD_BACKBONE_LN", "0") == "1" and \
            all(g.stage_dim(s) % 64 == 0 for s in range(len(g.swin_depths)))
        if self.fold_ln and b"experimental-gemm" not in _hip.load().odic_build_info():
            raise RuntimeError("ODIC_FOLD_BACKBONE_LN=1 needs a library built with make EXTRA=-DODIC_EXPERIMENTAL_GEMM")
        if self.fold_ln:
            for s, (blocks, _) in enumerate(self.stages):
                for b, w in enumerate(blocks):
                    p = f"{P}.layers.{s}.blocks.{b}"
                    w["qkv_f"] = ops.fold_layernorm_bf16(f32(p + ".attn.qkv.weight"), w["qkv_b"], w["n1w"], w["n1b"])
                    w["fc1_f"] = ops.fold_layernorm_bf16(f32(p + ".mlp.fc1.weight"), w["fc1_b"], w["n2w"], w["n2b"])
        # bf16 mode, the stage of width 192 (Swin-L stage 0): norm1 → qkv and norm2 → fc1 are ONE launch each — the A-resident
        # GEMM kernel normalises the fp32 rows while it reads them (odic_gemm_args.a_ln; gamma / beta folded into the
        # weights): 96 → 85 and 130 → 105 µs per pair at B = 16.  Not at width 384: there the fused form needs the registers
        # of a resident block and re-reads the fp32 rows once per column range — 95 against 62 µs (tools/ln_read_probe.py).
        # ODIC_FUSE_BACKBONE_LN_READ=0 keeps the two launches.
        # (split-fp16 mode: the same form exists — gemm_x3.hip tile configs 20 / 21 with a_ln — and is 1 % faster, but folding
        #  gamma into the weights changes WHICH near-ties of the xavier checkpoint round the other way: 255 / 256 captions equal
        #  to fp32 instead of 256 / 256 — the mode exists for that equality, so it is opt-in there: ODIC_FUSE_BACKBONE_LN_READ=x3)
        lr = os.environ.get("ODIC_FUSE_BACKBONE_LN_READ", "1")
        self.ln_read = not self.fold_ln and ((precision == "bf16" and lr != "0") or (precision == "x3" and lr == "x3"))
        self.fuse_qkv_attn = precision == "bf16" and os.environ.get("ODIC_FUSE_QKV_ATTENTION", "1") == "1"
        if self.ln_read:
            for s, (blocks, _) in enumerate(self.stages):
                if g.stage_dim(s) != 192:
                    continue
                for b, w in enumerate(blocks):
                    p = f"{P}.layers.{s}.blocks.{b}"
                    if precision == "bf16":
                        w["qkv_lnr"] = ops.fold_layernorm_bf16(f32(p + ".attn.qkv.weight"), w["qkv_b"], w["n1w"], w["n1b"])[:2] + (1.0,)
                        w["fc1_lnr"] = ops.fold_layernorm_bf16(f32(p + ".mlp.fc1.weight"), w["fc1_b"], w["n2w"], w["n2b"])[:2] + (1.0,)
                    else:       # split fp16: the folded weight goes through the same power-of-two packing as every x3 weight
                        for name, wk, bk, nw, nb in (("qkv_lnr", ".attn.qkv.weight", "qkv_b", "n1w", "n1b"),
                                                     ("fc1_lnr", ".mlp.fc1.weight", "fc1_b", "n2w", "n2b")):
                            Wg, b2, _ = ops.fold_layernorm(f32(p + wk), w[bk], w[nw], w[nb])
                            op, alpha = pack_operand(Wg, self.cdt)
                            w[name] = (op, b2, alpha)
        self.fp8_ready = False
        if fp8:
            self._pack_fp8(sd, calibration_images)

    # ------------------------------------------------------------------------------------------ fp8 mode
    def _pack_fp8(self, sd: SD, calibration_images) -> None:
        """Static quantisation of the Swin blocks: per-output-channel fp8 weights for qkv / fc1 / fc2, fp16 weights
        for proj, and per-tensor activation scales (LayerNorm outputs, GELU hidden) calibrated on one bf16 pass over
        `calibration_images` (default: two synthetic images) — amax x 1.25 mapped to the e4m3 maximum.  The activation
        scale of a LayerNorm output is folded into its gamma / beta, the product (activation scale x weight channel
        scale) is the consuming GEMM's `col_scale`, the GELU hidden is written as fp8 through `out_scale`."""
        from .weights import synth_images
        g, dv = self.g, self.device
        if any(g.stage_window(s) != 12 or g.stage_dim(s) % 64 for s in range(len(g.swin_depths))):
            raise RuntimeError("fp8 mode needs 12x12 windows and stage widths that are multiples of 64 (Swin-L: 192·2^s)")
        img = calibration_images if calibration_images is not None else synth_images(2, g, seed=7)
        amax = {}
        self.forward(img.to(dv, torch.float32), _amax=amax)                 # bf16 pass recording the operand ranges
        P = "swin_transf"
        margin = 1.25
        self.fp8_range = {k: margin * v for k, v in amax.items()}           # what each static scale can represent
        for s, (blocks, _) in enumerate(self.stages):
            for b, w in enumerate(blocks):
                p = f"{P}.layers.{s}.blocks.{b}"
                s1 = margin * amax[(s, b, "ln1")] / ops.FP8_MAX
                s2 = margin * amax[(s, b, "ln2")] / ops.FP8_MAX
                sh = margin * amax[(s, b, "hid")] / ops.FP8_MAX
                f32 = lambda k: _dev(sd[k], dv, torch.float32)      # noqa: E731
                q_qkv, sw_qkv = ops.quantize_fp8_per_channel(f32(p + ".attn.qkv.weight"))
                q_fc1, sw_fc1 = ops.quantize_fp8_per_channel(f32(p + ".mlp.fc1.weight"))
                q_fc2, sw_fc2 = ops.quantize_fp8_per_channel(f32(p + ".mlp.fc2.weight"))
                w.update(dict(
                    n1w8=(w["n1w"] / s1).contiguous(), n1b8=(w["n1b"] / s1).contiguous(),
                    n2w8=(w["n2w"] / s2).contiguous(), n2b8=(w["n2b"] / s2).contiguous(),
                    qkv_w8=q_qkv, qkv_cs=(sw_qkv * s1).contiguous(),
                    fc1_w8=q_fc1, fc1_cs=(sw_fc1 * s2).contiguous(), hid_inv=1.0 / sh,
                    fc2_w8=q_fc2, fc2_cs=(sw_fc2 * sh).contiguous(),
                    proj_w16=f32(p + ".attn.proj.weight").to(torch.float16).contiguous()))
        self.fp8_ready = True

    def fp8_saturation(self, img: torch.Tensor) -> dict:
        """Observed operand ranges of `img` (one bf16 pass) against the calibrated fp8 ranges: the tensors whose
        amax exceeds what the static scale represents are cast with clipping at ±448."""
        if not self.fp8_ready:
            raise RuntimeError("not an fp8 engine")
        seen = {}
        self.forward(img, _amax=seen)
        ratios = {k: seen[k] / self.fp8_range[k] for k in seen}
        over = {f"s{k[0]}b{k[1]}.{k[2]}": round(v, 3) for k, v in ratios.items() if v > 1.0}
        return {"tensors": len(ratios), "clipping_tensors": len(over), "worst_ratio": round(max(ratios.values()), 3),
                "median_ratio": round(sorted(ratios.values())[len(ratios) // 2], 3), "clipping": over}

    def forward(self, img: torch.Tensor, taps: Optional[dict] = None, out_dtype=torch.float32,
                _amax: Optional[dict] = None) -> torch.Tensor:
        """img fp32 [B,3,H,W] on the engine's device → features [B, res_last², C_last] (`out_dtype`)."""
        g, cdt = self.g, self.cdt
        if img.dtype != torch.float32 or not img.is_cuda:
            raise RuntimeError("SwinEngine.forward wants an fp32 CUDA image batch")
        img = img.contiguous()
        B = img.shape[0]
        fp8 = self.fp8_ready and _amax is None
        x = ops.patch_embed(img, self.pe_w, self.pe_b, self.pe_g, self.pe_beta, g.swin_patch_size)
        if taps is not None:
            taps["patch_embed"] = x.clone()
        for s, (blocks, down) in enumerate(self.stages):
            res, C_, heads, ws = g.stage_res(s), g.stage_dim(s), g.swin_num_heads[s], g.stage_window(s)
            x = x.view(B * res * res, C_)
            # image chunks (stage_chunks): a stage's blocks run over B/n images at a time, so that what one launch
            # writes (qkv, the MLP hidden, the residual stream) is still in the 256 MB Infinity Cache when the
            # next launch reads it — the stage-0/1 launches are HBM-bound
            n_ch = self.stage_chunks[s] if s < len(self.stage_chunks) else 1
            if taps is not None or _amax is not None or self.fold_ln or n_ch < 1 or B % n_ch:
                n_ch = 1
            x_all, B_all = x, B
            B = B_all // n_ch
            for ci, bi, w in ((ci, bi, w) for ci in range(n_ch) for bi, w in enumerate(blocks)):
                x = x_all[ci * B * res * res:(ci + 1) * B * res * res]
                if fp8:
                    # LN → fp8 | qkv: fp8 MFMA → fp16 | attention fp16 | proj: fp16 MFMA + residual
                    xn = ops.layernorm(x, w["n1w8"], w["n1b8"], out_dtype=ops.FP8_DTYPE)
                    qkv = ops.gemm(xn, w["qkv_w8"], w["qkv_b"], col_scale=w["qkv_cs"], out_dtype=torch.float16)
                    att = ops.window_attention(qkv, w["table"], B, res, C_, heads, ws, w["shift"],
                                               bias_shifted_prescaled=w["dense"])
                    ops.gemm(att, w["proj_w16"], w["proj_b"], residual=x, out=x)
                    # LN → fp8 | fc1: fp8 MFMA + GELU → fp8 | fc2: fp8 MFMA + residual
                    xn = ops.layernorm(x, w["n2w8"], w["n2b8"], out_dtype=ops.FP8_DTYPE)
                    h = ops.gemm(xn, w["fc1_w8"], w["fc1_b"], act=ops.ACT_GELU, col_scale=w["fc1_cs"],
                                 out_scale=w["hid_inv"], out_dtype=ops.FP8_DTYPE)
                    ops.gemm(h, w["fc2_w8"], w["fc2_b"], residual=x, out=x, col_scale=w["fc2_cs"])
                elif self.fold_ln and _amax is None:
                    if bi == 0:                                  # x comes from patch embed / patch merge: plain norm1
                        x16 = torch.empty(x.shape, dtype=cdt, device=x.device)
                        stats = torch.empty(x.shape[0], C_ // 32, 2, dtype=torch.float32, device=x.device)
                        xn = ops.layernorm(x, w["n1w"], w["n1b"], out_dtype=cdt)
                        qkv = ops.gemm(xn, w["qkv_w"], w["qkv_b"])
                    else:                                        # norm1 folded: A = bf16 copy left by the previous fc2
                        Wf, bf, cs = w["qkv_f"]
                        qkv = ops.gemm(x16, Wf, bf, ln_fold=(cs, 1e-5), ln_stats=stats)
                    att = ops.window_attention(qkv, w["table"], B, res, C_, heads, ws, w["shift"],
                                               bias_shifted_prescaled=w["dense"])
                    ops.gemm(att, w["proj_w"], w["proj_b"], residual=x, out=x, out16=x16, stats_out=stats)
                    Wf, bf, cs = w["fc1_f"]
                    h = ops.gemm(x16, Wf, bf, act=ops.ACT_GELU, ln_fold=(cs, 1e-5), ln_stats=stats)
                    if bi + 1 < len(blocks):
                        ops.gemm(h, w["fc2_w"], w["fc2_b"], residual=x, out=x, out16=x16, stats_out=stats)
                    else:
                        ops.gemm(h, w["fc2_w"], w["fc2_b"], residual=x, out=x)
                elif "qkv_lnr" in w and _amax is None and ops.a_ln_supported(x.shape[0], 3 * C_, C_, cdt):
                    if self.fuse_qkv_attn and ws == 12 and w["dense"] is not None:
                        # norm1 → qkv → attention core: one launch, q / k / v never leave the chip
                        att = ops.swin_qkv_attention(x, w["qkv_lnr"][0], w["qkv_lnr"][1], w["dense"], B, res, C_, heads,
                                                     ws, w["shift"])
                    else:
                        qkv = ops.gemm(None, w["qkv_lnr"][0], w["qkv_lnr"][1], a_ln=x, out_dtype=cdt, alpha=w["qkv_lnr"][2])
                        att = ops.window_attention(qkv, w["table"], B, res, C_, heads, ws, w["shift"],
                                                   bias_shifted_prescaled=w["dense"])
                    ops.gemm(att, w["proj_w"], w["proj_b"], residual=x, out=x, alpha=w["proj_a"])
                    h = ops.gemm(None, w["fc1_lnr"][0], w["fc1_lnr"][1], a_ln=x, act=ops.ACT_GELU, out_dtype=cdt,
                                 alpha=w["fc1_lnr"][2])
                    ops.gemm(h, w["fc2_w"], w["fc2_b"], residual=x, out=x, alpha=w["fc2_a"])
                else:
                    xn = ops.layernorm(x, w["n1w"], w["n1b"], out_dtype=cdt)
                    if _amax is not None:
                        _amax[(s, bi, "ln1")] = float(xn.float().abs().max())
                    qkv = ops.gemm(xn, w["qkv_w"], w["qkv_b"], alpha=w["qkv_a"])
                    att = ops.window_attention(qkv, w["table"], B, res, C_, heads, ws, w["shift"],
                                               bias_shifted_prescaled=w["dense"])
                    ops.gemm(att, w["proj_w"], w["proj_b"], residual=x, out=x, alpha=w["proj_a"])
                    xn = ops.layernorm(x, w["n2w"], w["n2b"], out_dtype=cdt)
                    h = ops.gemm(xn, w["fc1_w"], w["fc1_b"], act=ops.ACT_GELU, alpha=w["fc1_a"])
                    if _amax is not None:
                        _amax[(s, bi, "ln2")] = float(xn.float().abs().max())
                        _amax[(s, bi, "hid")] = float(h.float().abs().max())
                    ops.gemm(h, w["fc2_w"], w["fc2_b"], residual=x, out=x, alpha=w["fc2_a"])
                if taps is not None:
                    taps[f"s{s}b{bi}"] = x.view(B, res * res, C_).clone()
            x, B = x_all, B_all
            if down is not None:
                xm = ops.patch_merge_layernorm(x, down["nw"], down["nb"], B, res, C_, out_dtype=cdt)
                x = ops.gemm(xm.view(-1, 4 * C_), down["red_w"], out_dtype=torch.float32, alpha=down["red_a"])
                if taps is not None:
                    taps[f"merge{s}"] = x.view(B, (res // 2) ** 2, 2 * C_).clone()
        res = g.stage_res(len(g.swin_depths) - 1)
        out = ops.layernorm(x, self.fn_w, self.fn_b, out_dtype=out_dtype)
        return out.view(B, res * res, -1)


# =================================================================================================
# Expansion encoder + decoder, fp32
# =================================================================================================
class DecodeState:
    """Device-resident state of one search / teacher-forced run (see include/odic_hip.h)."""

    def __init__(self, eng: "CaptionerEngine", n_img: int, beams: int, T: int, kv: torch.Tensor,
                 enc_len: torch.Tensor, S: int):
        g, dv = eng.g, eng.device
        N = n_img * beams
        d, E, L = g.d_model, g.num_exp_dec, g.N_dec
        self.n_img, self.beams, self.N, self.T, self.S = n_img, beams, N, T, S
        self.kv, self.enc_len = kv, enc_len
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dv)      # noqa: E731
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dv)      # noqa: E731
        self.ycat = f(N, L * d)
        self.caches = [dict(cond=f(T, N, d), key=f(T, N, d), va=f(T, N, d), vb=f(T, N, d),
                            wfa=f(T, N, T, E), wfb=f(T, N, T, E), qk=f(T, N, E)) for _ in range(L)]
        self.anc = i32(N, T)
        self.row_valid = torch.ones(N, dtype=torch.int32, device=dv)
        self.next_tok = torch.zeros(N, dtype=torch.int64, device=dv)
        self.pos, self.done, self.ctr = i32(1), i32(1), i32(1)
        self.tokens = torch.zeros(n_img, beams, T, dtype=torch.int64, device=dv)
        self.logprobs = f(n_img, beams, T)
        self.cumul, self.n_elem, self.has_eos = f(N), i32(N), i32(N)
        self.cand_val, self.cand_idx = f(N, beams), i32(N, beams)
        self.logits = f(N, g.vocab_size)
        self.beam_state = _hip.BeamState(*(t.data_ptr() for t in (
            self.tokens, self.logprobs, self.anc, self.cumul, self.n_elem, self.has_eos, self.row_valid,
            self.next_tok, self.pos, self.done, self.ctr)))
        # the next position's input embedding, written by the launch that chooses the words (odic_embed_args)
        self.emb = ops.embed_args(eng.embed, eng.pos_table, self.ycat, L * d, d, math.sqrt(d))


class CaptionerEngine:
    """precision 'fp32': everything exact fp32.  'bf16': the ENCODER's products run on the bf16 MFMA
    GEMM (fp32 accumulation, fp32 residual streams, fp32 normalisations); the decoder stays fp32."""

    def __init__(self, sd: SD, g: Geometry, device, precision: str = "fp32"):
        self.g, self.device, self.precision = g, device, precision
        self.cdt = _CDT[precision]
        # decoder LayerNorms folded into the consuming skinny GEMM (10 fewer launches per step; see
        # odic_gemm_args.ln_colsum).  ODIC_FOLD_LN=0 keeps the separate odic_layernorm launches.
        self.fuse_ln = os.environ.get("ODIC_FOLD_LN", "1") == "1"
        d = g.d_model
        f32 = lambda k: _dev(sd[k], device, torch.float32)          # noqa: E731
        cat = lambda ks: torch.cat([f32(k) for k in ks], 0).contiguous()   # noqa: E731
        cw = lambda t: pack_operand(t, self.cdt)                    # noqa: E731  (encoder GEMM operand, alpha)
        (self.in_w, self.in_a), self.in_b = cw(f32("input_linear.weight")), f32("input_linear.bias")
        self.enc = []
        for i in range(g.N_enc):
            p = f"encoders.{i}"
            s = p + ".stc_exp"
            q, q_a = cw(f32(s + ".query_exp_vectors.weight"))
            key_w, key_a = cw(f32(s + ".key_embed.weight"))
            sel_w, sel_a = cw(f32(s + ".selector_embed.weight"))
            ab_w, ab_a = cw(cat([s + ".class_a_embed.weight", s + ".class_b_embed.weight"]))  # [2d, d]
            f1w, f1a = cw(f32(p + ".ff.linear_1.weight"))
            f2w, f2a = cw(f32(p + ".ff.linear_2.weight"))
            self.enc.append(dict(
                n1w=f32(p + ".norm_1.weight"), n1b=f32(p + ".norm_1.bias"),
                n2w=f32(p + ".norm_2.weight"), n2b=f32(p + ".norm_2.bias"),
                q=q, q_a=q_a,
                bvT=f32(s + ".bias_exp_vectors.weight").t().contiguous(),              # [d, nq] fp32
                key_w=key_w, key_a=key_a, key_b=f32(s + ".key_embed.bias"),
                sel_w=sel_w, sel_a=sel_a, sel_b=f32(s + ".selector_embed.bias"),
                ab_w=ab_w, ab_a=ab_a,
                ab_b=cat([s + ".class_a_embed.bias", s + ".class_b_embed.bias"]),
                f1w=f1w, f1a=f1a, f1b=f32(p + ".ff.linear_1.bias"),
                f2w=f2w, f2a=f2a, f2b=f32(p + ".ff.linear_2.bias")))
        (self.er_w, self.er_a), self.er_b = cw(f32("enc_reduce_group.weight")), f32("enc_reduce_group.bias")
        self.ern_w, self.ern_b = f32("enc_reduce_norm.weight"), f32("enc_reduce_norm.bias")
        self.group_meta = ops.stcexp_group_meta(g.num_exp_enc_list, device)
        self.dec = []
        kvw, kvb = [], []
        for i in range(g.N_dec):
            p = f"decoders.{i}"
            s = p + ".dyn_exp"
            names = ["cond_embed", "key_linear", "class_a_embed", "class_b_embed", "selector_embed"]
            self.dec.append(dict(
                n1w=f32(p + ".norm_1.weight"), n1b=f32(p + ".norm_1.bias"),
                n2w=f32(p + ".norm_2.weight"), n2b=f32(p + ".norm_2.bias"),
                n3w=f32(p + ".norm_3.weight"), n3b=f32(p + ".norm_3.bias"),
                dyn_w=cat([f"{s}.{n}.weight" for n in names]), dyn_b=cat([f"{s}.{n}.bias" for n in names]),
                qexp=f32(s + ".query_exp_vectors.weight"), bexp=f32(s + ".bias_exp_vectors.weight"),
                wq=f32(p + ".mha.Wq.weight"), bq=f32(p + ".mha.Wq.bias"),
                wo=f32(p + ".mha.out_linear.weight"), bo=f32(p + ".mha.out_linear.bias"),
                f1w=f32(p + ".ff.linear_1.weight"), f1b=f32(p + ".ff.linear_1.bias"),
                f2w=f32(p + ".ff.linear_2.weight"), f2b=f32(p + ".ff.linear_2.bias")))
            w = self.dec[-1]
            w["dyn_f"] = ops.fold_layernorm(w["dyn_w"], w["dyn_b"], w["n1w"], w["n1b"])
            w["wq_f"] = ops.fold_layernorm(w["wq"], w["bq"], w["n2w"], w["n2b"])
            w["f1_f"] = ops.fold_layernorm(w["f1w"], w["f1b"], w["n3w"], w["n3b"])
            kvw += [p + ".mha.Wk.weight", p + ".mha.Wv.weight"]
            kvb += [p + ".mha.Wk.bias", p + ".mha.Wv.bias"]
        self.kv_w32, self.kv_b = cat(kvw), cat(kvb)                 # [2·N_dec·d, d]
        self.kv_w, self.kv_a = cw(self.kv_w32)
        self.dr_w, self.dr_b = f32("dec_reduce_group.weight"), f32("dec_reduce_group.bias")
        self.drn_w, self.drn_b = f32("dec_reduce_norm.weight"), f32("dec_reduce_norm.bias")
        self.voc_w, self.voc_b = f32("vocab_linear.weight"), f32("vocab_linear.bias")
        self.voc_f = ops.fold_layernorm(self.voc_w, self.voc_b, self.drn_w, self.drn_b)
        self.embed, self.pos_table = f32("out_embedder.embed.weight"), f32("pos_encoder.weight")

    # ------------------------------------------------------------------------------------------
    def _as_operand(self, x: torch.Tensor, M: int, C_: int, ldx: int) -> torch.Tensor:
        """fp32 activations → the encoder GEMM operand dtype (identity view in fp32 mode)."""
        if x.dtype == self.cdt:
            return x
        if self.cdt == torch.bfloat16 and x.dtype == torch.float32:
            return ops.cast_bf16(x, M=M, C_=C_, ldx=ldx)
        if self.cdt == ops.H2_DTYPE and x.dtype == torch.float32:
            return ops.cast_h2(x, M=M, C_=C_, ldx=ldx)
        raise RuntimeError(f"cannot feed {x.dtype} to a {self.cdt} encoder")

    def encode(self, feats: torch.Tensor, enc_len: torch.Tensor, want_bf16_mem: bool = False):
        """feats [B,S,F] (fp32, or bf16 in bf16 mode) → encoder output fp32 [B,S,d]  (forward_enc after
        the backbone: End_ExpansionNet_v2.py:82-101 / ExpansionNet_v2.py:52-70).  enc_len int32 [B]."""
        g, dv, cdt = self.g, self.device, self.cdt
        B, S, F = feats.shape
        d, L, nq, M = g.d_model, g.N_enc, sum(g.num_exp_enc_list), B * S
        x3 = cdt == ops.H2_DTYPE
        pad = 64 if cdt == torch.bfloat16 else (32 if x3 else 1)   # the bf16 / split-fp16 GEMMs need K % 64 / 32 == 0
        # split fp16: the normalised expansion weights (<= 1, typically 1e-2 forward / 1e-3 backward) are written
        # scaled by a power of two so that their lo halves stay fp16 normals; undone in the consuming product's alpha
        fw_sc, bw_sc = (256.0, 4096.0) if x3 else (1.0, 1.0)
        Sp, nqp = -(-S // pad) * pad, -(-nq // pad) * pad
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dv)      # noqa: E731
        zc = lambda *s: torch.zeros(*s, dtype=cdt, device=dv)               # noqa: E731  (K padding must be finite)
        feats2 = feats.reshape(M, F)
        if not feats2.is_contiguous():
            feats2 = feats2.contiguous()
        x0 = ops.gemm(self._as_operand(feats2, M, F, F), self.in_w, self.in_b, out_dtype=torch.float32,
                      alpha=self.in_a)                                                                   # [M,d]
        xcat = f(M, L * d)
        z = f(B, nq, S)
        # K-padded operands.  odic_stcexp_normalize writes its four outputs out to their leading dimension (zeros in the
        # padding), so they need no clearing; the three GEMM outputs whose padding columns no launch ever writes are kept
        # per (shape, stream) and zeroed ONCE — clearing them inside every captured encode pass was seven fill launches
        ec = lambda *s: torch.empty(*s, dtype=cdt, device=dv)               # noqa: E731
        pf, nf = ec(B, nq, Sp), ec(B, nq, Sp)
        pb, nb = ec(B, S, nqp), ec(B, S, nqp)
        colsum = f(B * len(g.num_exp_enc_list) * 2 * S)
        key = (B, S, torch.cuda.current_stream(dv).cuda_stream)
        pads = self._pad_ws.get(key) if hasattr(self, "_pad_ws") else None
        if pads is None:
            if torch.cuda.is_current_stream_capturing():      # (a first call under capture: plain per-call buffers)
                pads = (zc(B, d, nqp), zc(B, d, nqp), zc(B, 2 * d, Sp))
            else:
                if not hasattr(self, "_pad_ws"):
                    self._pad_ws = {}
                pads = self._pad_ws[key] = (zc(B, d, nqp), zc(B, d, nqp), zc(B, 2 * d, Sp))
        AT, BT, vabT = pads
        A2, B2 = f(B, S, d), f(B, S, d)
        key = torch.empty(M, d, dtype=cdt, device=dv)
        ld = L * d
        for i, w in enumerate(self.enc):
            xin, ldin = (x0, d) if i == 0 else (xcat[:, (i - 1) * d:], ld)
            xo = xcat[:, i * d:]
            x2 = ops.layernorm(xin, w["n1w"], w["n1b"], M=M, C_=d, ldx=ldin, out_dtype=cdt)
            ops.gemm(x2, w["key_w"], w["key_b"], out=key, alpha=w["key_a"])
            sel = ops.gemm(x2, w["sel_w"], w["sel_b"], out_dtype=torch.float32, alpha=w["sel_a"])
            # (class_a | class_b) projections, produced transposed: [B, 2d, S] = W·x2ᵀ + b(row)
            ops.gemm(w["ab_w"], x2, w["ab_b"], out=vabT, bias_axis=1, M=2 * d, N=S, K=d, lda=d, ldw=d, ldc=Sp,
                     batch=B, strideA=0, strideW=S * d, strideC=2 * d * Sp, alpha=w["ab_a"])
            # z = Q·Kᵀ/sqrt(d)
            ops.gemm(w["q"], key, out=z, alpha=w["q_a"] / math.sqrt(d), M=nq, N=S, K=d, lda=d, ldw=d, ldc=S,
                     batch=B, strideA=0, strideW=S * d, strideC=nq * S)
            ops.stcexp_normalize(z, enc_len, self.group_meta, len(g.num_exp_enc_list), pf, nf, pb, nb, colsum,
                                 scale_fw=fw_sc, scale_bw=bw_sc)
            # class_aᵀ [d,nq] = Vaᵀ·pos_fwᵀ + Bvᵀ    (layers.py:63-64, transposed)
            ops.gemm(vabT, pf, residual=w["bvT"], out=AT, M=d, N=nq, K=Sp, lda=Sp, ldw=Sp, ldr=nq, ldc=nqp, batch=B,
                     strideA=2 * d * Sp, strideW=nq * Sp, strideR=0, strideC=d * nqp, alpha=1.0 / fw_sc)
            ops.gemm(vabT[:, d:], nf, residual=w["bvT"], out=BT, M=d, N=nq, K=Sp, lda=Sp, ldw=Sp, ldr=nq, ldc=nqp,
                     batch=B, strideA=2 * d * Sp, strideW=nq * Sp, strideR=0, strideC=d * nqp, alpha=1.0 / fw_sc)
            # backward: [S,nq]·[nq,d]
            ops.gemm(pb, AT, out=A2, M=S, N=d, K=nqp, lda=nqp, ldw=nqp, ldc=d, batch=B, strideA=S * nqp,
                     strideW=d * nqp, strideC=S * d, alpha=1.0 / bw_sc)
            ops.gemm(nb, BT, out=B2, M=S, N=d, K=nqp, lda=nqp, ldw=nqp, ldc=d, batch=B, strideA=S * nqp,
                     strideW=d * nqp, strideC=S * d, alpha=1.0 / bw_sc)
            ops.selector_mix(xin, ldin, sel, d, A2, d, B2, d, xo, ld, M, d)
            x2 = ops.layernorm(xo, w["n2w"], w["n2b"], M=M, C_=d, ldx=ld, out_dtype=cdt)
            h = ops.gemm(x2, w["f1w"], w["f1b"], act=ops.ACT_RELU, alpha=w["f1a"])
            ops.gemm(h, w["f2w"], w["f2b"], residual=xo, out=xo, M=M, N=d, K=g.ff, lda=g.ff, ldw=g.ff, ldr=ld,
                     ldc=ld, alpha=w["f2a"])
        pre = ops.gemm(self._as_operand(xcat, M, ld, ld), self.er_w, self.er_b, residual=xcat[:, (L - 1) * d:], M=M,
                       N=d, K=ld, lda=ld, ldw=ld, ldr=ld, ldc=d, out=f(M, d), alpha=self.er_a)
        mem = ops.layernorm(pre, self.ern_w, self.ern_b).view(B, S, d)
        if want_bf16_mem and cdt != torch.float32:                 # the K/V projection's operand, rounded once
            return mem, ops.layernorm(pre, self.ern_w, self.ern_b, out_dtype=cdt).view(B, S, d)
        return mem

    def project_kv(self, mem: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Cross-attention K/V of every decoder layer, once per image (the reference recomputes them
        every step for every beam, layers.py:274-276): [B,S,d] → [B,S,2·N_dec·d]."""
        B, S, d = mem.shape
        n = self.kv_w.shape[0]
        if out is None:
            out = torch.empty(B, S, n, dtype=torch.float32, device=self.device)
        a = mem.reshape(B * S, d)
        if not a.is_contiguous():
            a = a.contiguous()
        # bf16 mode: the product runs on the bf16 MFMA GEMM whatever dtype the caller holds the memory in, so
        # the direct beam_search call and CaptionPipeline (which feeds the bf16 LayerNorm output) see
        # bit-identical K/V: round-to-nearest-even of the same fp32 values either way.
        if a.dtype != self.kv_w.dtype:
            a = self._as_operand(a, B * S, d, d)
        ops.gemm(a, self.kv_w, self.kv_b, out=out, M=B * S, N=n, K=d, lda=d, ldw=d, ldc=n, alpha=self.kv_a)
        return out

    # ------------------------------------------------------------------------------------------
    def new_state(self, n_img: int, beams: int, T: int, kv: torch.Tensor, enc_len: torch.Tensor) -> DecodeState:
        if T > 128:
            raise RuntimeError("decode length > 128 positions is not supported by odic_dynexp_step")
        if T - 1 > self.pos_table.shape[0]:
            raise RuntimeError(f"{T - 1} decode positions exceed the pos_encoder table ({self.pos_table.shape[0]})")
        return DecodeState(self, n_img, beams, T, kv, enc_len, kv.shape[1])

    def step_logits(self, st: DecodeState, embed: bool = True) -> None:
        """Process position *st.pos for every sequence: next_tok → st.logits [N,V].  embed=False: the input row of
        this position is already in st.ycat (written by the launch that chose the word: beam_reset / beam_search_step
        with st.emb)."""
        g = self.g
        d, L, N = g.d_model, g.N_dec, st.N
        ld = L * d
        if embed:
            ops.dec_embed(st.next_tok, self.embed, self.pos_table, st.pos, st.ycat, ld, N, d, math.sqrt(d))
        # the folded-LayerNorm form lives in the skinny-M kernel (M <= 192 rows, gemm_f32.hip); wider
        # searches (batch 64 x beam 5, decode groups) take the LayerNorm + GEMM pair
        fuse = self.fuse_ln and N <= 192
        eps = 1e-5

        def ln_gemm(x, ldx, nw, nb, W, b, folded, **kw):
            """LayerNorm(x)·Wᵀ + b: one launch (LayerNorm folded into the product) or two."""
            if fuse:
                Wf, bf, cs = folded
                return ops.gemm(x, Wf, bf, M=N, N=Wf.shape[0], K=d, lda=ldx, ldw=d, ldc=Wf.shape[0], ln_fold=(cs, eps), **kw)
            return ops.gemm(ops.layernorm(x, nw, nb, M=N, C_=d, ldx=ldx), W, b, **kw)

        for i, w in enumerate(self.dec):
            c = st.caches[i]
            xin = st.ycat if i == 0 else st.ycat[:, (i - 1) * d:]
            xo = st.ycat[:, i * d:]
            lin = ln_gemm(xin, ld, w["n1w"], w["n1b"], w["dyn_w"], w["dyn_b"], w["dyn_f"])               # [N,5d]
            ops.dynexp_step(lin, 5 * d, w["qexp"], w["bexp"], c["cond"], c["key"], c["va"], c["vb"], c["wfa"],
                            c["wfb"], c["qk"], st.anc, st.row_valid, st.pos, xin, ld, xo, ld, N, st.T, d,
                            g.num_exp_dec)
            q = ln_gemm(xo, ld, w["n2w"], w["n2b"], w["wq"], w["bq"], w["wq_f"])
            att = torch.empty(N, d, dtype=torch.float32, device=self.device)
            ops.cross_attn_step(q, d, st.kv, st.kv.shape[2], 2 * i * d, (2 * i + 1) * d, st.enc_len,
                                st.row_valid, att, d, N, st.n_img, st.S, d, g.num_heads)
            ops.gemm(att, w["wo"], w["bo"], residual=xo, out=xo, M=N, N=d, K=d, lda=d, ldw=d, ldr=ld, ldc=ld)
            h = ln_gemm(xo, ld, w["n3w"], w["n3b"], w["f1w"], w["f1b"], w["f1_f"], act=ops.ACT_RELU)
            ops.gemm(h, w["f2w"], w["f2b"], residual=xo, out=xo, M=N, N=d, K=g.ff, lda=g.ff, ldw=g.ff, ldr=ld,
                     ldc=ld)
        pre = torch.empty(N, d, dtype=torch.float32, device=self.device)
        ops.gemm(st.ycat, self.dr_w, self.dr_b, residual=st.ycat[:, (L - 1) * d:], out=pre, M=N, N=d, K=ld, lda=ld,
                 ldw=ld, ldr=ld, ldc=d)
        if fuse:
            ops.gemm(pre, self.voc_f[0], self.voc_f[1], out=st.logits, M=N, N=g.vocab_size, K=d, lda=d, ldw=d,
                     ldc=g.vocab_size, ln_fold=(self.voc_f[2], eps))
        else:
            ops.gemm(ops.layernorm(pre, self.drn_w, self.drn_b), self.voc_w, self.voc_b, out=st.logits)

    def beam_step(self, st: DecodeState, eos_idx: int) -> None:
        """One full search step: decoder → log-softmax / top-k (one block per row) → beam bookkeeping + the next
        position's input rows (one block per image).  The state must have been armed with
        ops.beam_reset(st.beam_state, ..., emb=st.emb).  (odic_beam_search_step does the two launches in one, but a
        block per image then works through its k rows alone: 20 / 30 us against 14 / 16 at beam 3 / 5,
        tools/topk_bench.py.)"""
        self.step_logits(st, embed=False)
        V = self.g.vocab_size
        if os.environ.get("ODIC_FUSED_SEARCH_STEP", "0") == "1":
            ops.beam_search_step(st.logits, V, V, st.beam_state, st.n_img, st.beams, st.T, eos_idx, emb=st.emb)
            return
        ops.logsoftmax_topk(st.logits, V, None, 0, st.cand_val, st.cand_idx, st.N, V, st.beams)
        ops.beam_step(st.cand_val, st.cand_idx, st.beam_state, st.n_img, st.beams, st.T, eos_idx, emb=st.emb)
