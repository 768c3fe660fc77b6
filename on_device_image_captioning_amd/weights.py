"""Model geometry + deterministic synthetic checkpoint generator.

`rf_model.pth` (the checkpoint demo.py:100-104 loads) is an external download and is
not available offline, so parity and benchmarks run on *generated* weights.  The
generator is keyed by the state-dict key name and uses numpy's Philox counter RNG, so
the same 520-key dict (layout: SURVEY.md §8 row A21, reference
utils/saving_utils.py:66-71 + the module tree of models/End_ExpansionNet_v2.py:11-119)
is reproduced bit-for-bit on any machine / torch version.

Nothing in here depends on the HIP extension; it is shared by the product path
(bench.py synthetic weights), by tests and by oracle/make_golden.py.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class Geometry:
    """Hyper-parameters, named as the constructor kwargs of the reference model
    (demo.py:68-99 / test.py:372-403)."""
    swin_img_size: int = 384
    swin_patch_size: int = 4
    swin_in_chans: int = 3
    swin_embed_dim: int = 192
    swin_depths: Tuple[int, ...] = (2, 2, 18, 2)
    swin_num_heads: Tuple[int, ...] = (6, 12, 24, 48)
    swin_window_size: int = 12
    swin_mlp_ratio: float = 4.0
    final_swin_dim: int = 1536
    d_model: int = 512
    N_enc: int = 3
    N_dec: int = 3
    ff: int = 2048
    num_heads: int = 8
    num_exp_enc_list: Tuple[int, ...] = (32, 64, 128, 256, 512)
    num_exp_dec: int = 16
    vocab_size: int = 10000
    max_seq_len: int = 74

    # ---- derived
    @property
    def grid0(self) -> int:
        return self.swin_img_size // self.swin_patch_size

    def stage_res(self, s: int) -> int:
        return self.grid0 >> s

    def stage_dim(self, s: int) -> int:
        return self.swin_embed_dim << s

    def stage_shift(self, s: int, b: int) -> int:
        """Shift of block b in stage s (reference swin_transformer_mod.py:444 and the
        `min(input_resolution) <= window_size` override at :262-265)."""
        if self.stage_res(s) <= self.swin_window_size:
            return 0
        return 0 if b % 2 == 0 else self.swin_window_size // 2

    def stage_window(self, s: int) -> int:
        return min(self.swin_window_size, self.stage_res(s))

    def model_kwargs(self) -> dict:
        """kwargs accepted verbatim by End_ExpansionNet_v2.__init__ (ours and the reference's)."""
        return dict(
            swin_img_size=self.swin_img_size, swin_patch_size=self.swin_patch_size,
            swin_in_chans=self.swin_in_chans, swin_embed_dim=self.swin_embed_dim,
            swin_depths=list(self.swin_depths), swin_num_heads=list(self.swin_num_heads),
            swin_window_size=self.swin_window_size, swin_mlp_ratio=self.swin_mlp_ratio,
            swin_qkv_bias=True, swin_qk_scale=None, swin_drop_rate=0.0,
            swin_attn_drop_rate=0.0, swin_drop_path_rate=0.0,
            swin_norm_layer=torch.nn.LayerNorm, swin_ape=False, swin_patch_norm=True,
            swin_use_checkpoint=False, final_swin_dim=self.final_swin_dim,
            d_model=self.d_model, N_enc=self.N_enc, N_dec=self.N_dec, ff=self.ff,
            num_heads=self.num_heads, num_exp_enc_list=list(self.num_exp_enc_list),
            num_exp_dec=self.num_exp_dec, max_seq_len=self.max_seq_len)


#: Swin-L/384 + ExpansionNet v2 — the geometry of every BASELINE.json config.
FULL = Geometry()

#: Reduced width/depth (still img 384 / window 12 → N=144, head dim 32) for fast CI.
TINY = Geometry(swin_embed_dim=96, swin_depths=(2, 2, 2, 2), swin_num_heads=(3, 6, 12, 24),
                final_swin_dim=768, d_model=128, ff=256, num_heads=4,
                num_exp_enc_list=(8, 16, 24), num_exp_dec=4, vocab_size=500, max_seq_len=24,
                N_enc=2, N_dec=2)


#: TINY with channel counts that are multiples of 64 (the bf16 MFMA GEMM needs K % 64 == 0, which every
#: Swin-L stage satisfies: 192·2^s).  No golden fixture: compared against the live oracle.
TINY64 = Geometry(swin_embed_dim=128, swin_depths=(2, 2, 2, 2), swin_num_heads=(4, 8, 16, 32),
                  final_swin_dim=1024, d_model=128, ff=256, num_heads=4,
                  num_exp_enc_list=(8, 16, 24), num_exp_dec=4, vocab_size=500, max_seq_len=24,
                  N_enc=2, N_dec=2)


# ----------------------------------------------------------------------------------------------
# constant buffers
# ----------------------------------------------------------------------------------------------
def relative_position_index(ws: int) -> torch.Tensor:
    """(ws², ws²) int64: (Δh + ws-1)·(2ws-1) + (Δw + ws-1)  — reference
    swin_transformer_mod.py:163-173 (legacy numbering)."""
    r = torch.arange(ws)
    hh = r.repeat_interleave(ws)          # row of token i
    ww = r.repeat(ws)                     # col of token i
    dh = hh[:, None] - hh[None, :] + (ws - 1)
    dw = ww[:, None] - ww[None, :] + (ws - 1)
    return (dh * (2 * ws - 1) + dw).to(torch.int64)


def shifted_window_region_ids(res: int, ws: int, shift: int) -> torch.Tensor:
    """(res, res) int region id on the *shifted* grid — the 3×3 slicing of reference
    swin_transformer_mod.py:281-292."""
    edges = torch.zeros(res, dtype=torch.int64)
    edges[res - ws:res - shift] = 1
    edges[res - shift:] = 2
    return edges[:, None] * 3 + edges[None, :]


def shifted_window_attn_mask(res: int, ws: int, shift: int) -> torch.Tensor:
    """(nW, ws², ws²) fp32 in {0,-100} — reference swin_transformer_mod.py:294-297."""
    rid = shifted_window_region_ids(res, ws, shift)
    n = res // ws
    win = rid.view(n, ws, n, ws).permute(0, 2, 1, 3).reshape(n * n, ws * ws)
    diff = win[:, None, :] != win[:, :, None]
    return torch.where(diff, torch.tensor(-100.0), torch.tensor(0.0))


# ----------------------------------------------------------------------------------------------
# state-dict spec
# ----------------------------------------------------------------------------------------------
def _captioner_spec(g: Geometry, feat_dim: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    d, ff = g.d_model, g.ff
    nq = sum(g.num_exp_enc_list)
    out: List[Tuple[str, Tuple[int, ...], str]] = []

    def lin(name, o, i):
        out.append((f"{name}.weight", (o, i), "matrix"))
        out.append((f"{name}.bias", (o,), "bias"))

    def ln(name, c):
        out.append((f"{name}.weight", (c,), "ln_w"))
        out.append((f"{name}.bias", (c,), "ln_b"))

    for i in range(g.N_enc):
        p = f"encoders.{i}"
        ln(f"{p}.norm_1", d)
        ln(f"{p}.norm_2", d)
        out.append((f"{p}.stc_exp.query_exp_vectors.weight", (nq, d), "matrix"))
        out.append((f"{p}.stc_exp.bias_exp_vectors.weight", (nq, d), "matrix"))
        for nm in ("key_embed", "class_a_embed", "class_b_embed", "selector_embed"):
            lin(f"{p}.stc_exp.{nm}", d, d)
        lin(f"{p}.ff.linear_1", ff, d)
        lin(f"{p}.ff.linear_2", d, ff)
    for i in range(g.N_dec):
        p = f"decoders.{i}"
        ln(f"{p}.norm_1", d)
        ln(f"{p}.norm_2", d)
        ln(f"{p}.norm_3", d)
        for nm in ("Wq", "Wk", "Wv", "out_linear"):
            lin(f"{p}.mha.{nm}", d, d)
        lin(f"{p}.dyn_exp.cond_embed", d, d)
        out.append((f"{p}.dyn_exp.query_exp_vectors.weight", (g.num_exp_dec, d), "matrix"))
        out.append((f"{p}.dyn_exp.bias_exp_vectors.weight", (g.num_exp_dec, d), "matrix"))
        for nm in ("key_linear", "class_a_embed", "class_b_embed", "selector_embed"):
            lin(f"{p}.dyn_exp.{nm}", d, d)
        lin(f"{p}.ff.linear_1", ff, d)
        lin(f"{p}.ff.linear_2", d, ff)
    lin("input_linear", d, feat_dim)
    lin("vocab_linear", g.vocab_size, d)
    out.append(("out_embedder.embed.weight", (g.vocab_size, d), "matrix"))
    out.append(("pos_encoder.weight", (g.max_seq_len, d), "matrix"))
    lin("enc_reduce_group", d, d * g.N_enc)
    ln("enc_reduce_norm", d)
    lin("dec_reduce_group", d, d * g.N_dec)
    ln("dec_reduce_norm", d)
    return out


def _swin_spec(g: Geometry) -> List[Tuple[str, Tuple[int, ...], str]]:
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    P = "swin_transf"
    c0, ps = g.swin_embed_dim, g.swin_patch_size
    out.append((f"{P}.patch_embed.proj.weight", (c0, g.swin_in_chans, ps, ps), "matrix"))
    out.append((f"{P}.patch_embed.proj.bias", (c0,), "bias"))
    out.append((f"{P}.patch_embed.norm.weight", (c0,), "ln_w"))
    out.append((f"{P}.patch_embed.norm.bias", (c0,), "ln_b"))
    for s, depth in enumerate(g.swin_depths):
        C, h, res, ws = g.stage_dim(s), g.swin_num_heads[s], g.stage_res(s), g.stage_window(s)
        hid = int(C * g.swin_mlp_ratio)
        for b in range(depth):
            p = f"{P}.layers.{s}.blocks.{b}"
            if g.stage_shift(s, b) > 0:
                out.append((f"{p}.attn_mask", ((res // ws) ** 2, ws * ws, ws * ws), "attn_mask"))
            out.append((f"{p}.norm1.weight", (C,), "ln_w"))
            out.append((f"{p}.norm1.bias", (C,), "ln_b"))
            out.append((f"{p}.attn.relative_position_bias_table", ((2 * ws - 1) ** 2, h), "matrix"))
            out.append((f"{p}.attn.relative_position_index", (ws * ws, ws * ws), "rel_index"))
            out.append((f"{p}.attn.qkv.weight", (3 * C, C), "matrix"))
            out.append((f"{p}.attn.qkv.bias", (3 * C,), "bias"))
            out.append((f"{p}.attn.proj.weight", (C, C), "matrix"))
            out.append((f"{p}.attn.proj.bias", (C,), "bias"))
            out.append((f"{p}.norm2.weight", (C,), "ln_w"))
            out.append((f"{p}.norm2.bias", (C,), "ln_b"))
            out.append((f"{p}.mlp.fc1.weight", (hid, C), "matrix"))
            out.append((f"{p}.mlp.fc1.bias", (hid,), "bias"))
            out.append((f"{p}.mlp.fc2.weight", (C, hid), "matrix"))
            out.append((f"{p}.mlp.fc2.bias", (C,), "bias"))
        if s < len(g.swin_depths) - 1:
            p = f"{P}.layers.{s}.downsample"
            out.append((f"{p}.reduction.weight", (2 * C, 4 * C), "matrix"))
            out.append((f"{p}.norm.weight", (4 * C,), "ln_w"))
            out.append((f"{p}.norm.bias", (4 * C,), "ln_b"))
    cl = g.stage_dim(len(g.swin_depths) - 1)
    out.append((f"{P}.norm.weight", (cl,), "ln_w"))
    out.append((f"{P}.norm.bias", (cl,), "ln_b"))
    return out


def state_dict_spec(g: Geometry = FULL, end_to_end: bool = True,
                    img_feature_dim: int | None = None) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(key, shape, kind)] in the reference's registration order.  `kind` drives the
    generator: matrix | bias | ln_w | ln_b | rel_index | attn_mask."""
    if end_to_end:
        return _swin_spec(g) + _captioner_spec(g, g.final_swin_dim)
    return _captioner_spec(g, img_feature_dim if img_feature_dim is not None else g.final_swin_dim)


# ----------------------------------------------------------------------------------------------
# generator
# ----------------------------------------------------------------------------------------------
def _philox(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = int.from_bytes(h[:16], "little")
    return np.random.Generator(np.random.Philox(key=key))


def _uniform(name: str, seed: int, shape, bound: float) -> torch.Tensor:
    rng = _philox(name, seed)
    a = rng.random(size=int(np.prod(shape)), dtype=np.float32)
    a = (a * np.float32(2.0) - np.float32(1.0)) * np.float32(bound)
    return torch.from_numpy(a.reshape(shape))


def synth_tensor(name: str, shape, kind: str, g: Geometry, seed: int = 0) -> torch.Tensor:
    if kind == "matrix":
        recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * recept, shape[0] * recept
        return _uniform(name, seed, shape, math.sqrt(6.0 / (fan_in + fan_out)))
    if kind == "bias":
        return _uniform(name, seed, shape, 0.05)
    if kind == "ln_w":
        return 1.0 + _uniform(name, seed, shape, 0.1)
    if kind == "ln_b":
        return _uniform(name, seed, shape, 0.05)
    if kind == "rel_index":
        ws = int(round(math.sqrt(shape[0])))
        return relative_position_index(ws)
    if kind == "attn_mask":
        ws = int(round(math.sqrt(shape[1])))
        res = int(round(math.sqrt(shape[0]))) * ws
        return shifted_window_attn_mask(res, ws, ws // 2)
    raise ValueError(kind)


def synth_state_dict(g: Geometry = FULL, seed: int = 0, end_to_end: bool = True,
                     img_feature_dim: int | None = None, variant: str = "xavier",
                     sos_idx: int = 79, eos_idx: int = 77) -> Dict[str, torch.Tensor]:
    """Deterministic checkpoint.

    variant:
      "xavier"  — realistic scale (the reference applies xavier_uniform_ to every
                  parameter with dim>1, models/End_ExpansionNet_v2.py:112-114).
                  Never emits EOS in practice → decode runs to max length.
      "sharp"   — vocab_linear ×12: top-1/top-2 log-prob margins ≫ bf16 noise, so token
                  IDs can be compared exactly across precisions.
      "eos"     — "sharp" plus a vocab_linear.bias bump on `eos_idx` so that beams
                  finish at assorted lengths (exercises the finished-beam bookkeeping of
                  legacy_models/captioning_model.py:178-221).
    """
    sd: Dict[str, torch.Tensor] = {}
    for name, shape, kind in state_dict_spec(g, end_to_end, img_feature_dim):
        sd[name] = synth_tensor(name, shape, kind, g, seed)
    if variant in ("sharp", "eos"):
        sd["vocab_linear.weight"] = sd["vocab_linear.weight"] * 12.0
        sd["vocab_linear.bias"] = sd["vocab_linear.bias"] * 4.0
    if variant == "eos":
        b = sd["vocab_linear.bias"].clone()
        logit_std = float(sd["vocab_linear.weight"].std()) * math.sqrt(g.d_model)
        b[eos_idx] = b[eos_idx] + eos_sigma(g.vocab_size) * logit_std
        sd["vocab_linear.bias"] = b
    elif variant not in ("xavier", "sharp"):
        raise ValueError(variant)
    return sd


def eos_sigma(vocab_size: int) -> float:
    """EOS logit bump of variant="eos" in units of the logit standard deviation; tuned with the
    oracle so that beams end at mixed lengths (2.2 at V=500, 4.5 at V=10000)."""
    return 2.2 if vocab_size <= 1000 else 4.5


def num_parameters(sd: Dict[str, torch.Tensor]) -> int:
    """Trainable-parameter count: everything except the int64 index / mask buffers.
    Known answer for FULL end-to-end: 233,803,076 (reference benchmarking/plotting.py:22)."""
    return sum(v.numel() for k, v in sd.items()
               if not k.endswith("relative_position_index") and not k.endswith("attn_mask"))


def synth_images(batch: int, g: Geometry = FULL, seed: int = 42) -> torch.Tensor:
    """(B,3,H,W) fp32 N(0,1)-like images (the harness shape of reference
    benchmarking/benchmarking.py:86), Philox-generated so both machines agree."""
    rng = _philox("images", seed)
    a = rng.standard_normal(size=(batch, g.swin_in_chans, g.swin_img_size, g.swin_img_size),
                            dtype=np.float32)
    return torch.from_numpy(a)


def synth_features(batch: int, seq: int, dim: int, seed: int = 42) -> torch.Tensor:
    rng = _philox("features", seed)
    return torch.from_numpy(rng.standard_normal(size=(batch, seq, dim), dtype=np.float32))
