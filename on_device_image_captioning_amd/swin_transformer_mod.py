"""Swin Transformer backbone — parameter container + HIP-backed forward.

Mirrors the module tree (and therefore the state-dict keys/shapes) of reference
models/swin_transformer_mod.py: SwinTransformer:670 → PatchEmbed:611, BasicLayer:513 →
SwinTransformerBlock:291 → WindowAttention:160 / Mlp:97, PatchMerging:466.  The sub-modules only
OWN parameters and buffers; all arithmetic happens in engine.SwinEngine (libodic_hip.so).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .weights import Geometry, relative_position_index, shifted_window_attn_mask


def _trunc_normal_(t: torch.Tensor, std: float = 0.02) -> None:
    nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class Mlp(nn.Module):
    def __init__(self, in_features: int, hidden_features: int):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim: int, window_size: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, (window_size, window_size), num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        _trunc_normal_(self.relative_position_bias_table)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim: int, input_resolution, num_heads: int, window_size: int, shift_size: int,
                 mlp_ratio: float, qkv_bias: bool, norm_layer):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(input_resolution) <= window_size:      # window covers the map: no shift, no partition
            self.shift_size = 0
            self.window_size = min(input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        mask = None
        if self.shift_size > 0:
            mask = shifted_window_attn_mask(input_resolution[0], self.window_size, self.shift_size)
        self.register_buffer("attn_mask", mask)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim: int, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, norm_layer,
                 downsample):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias, norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim, norm_layer):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None


class SwinTransformer(nn.Module):
    """Constructor keywords as the reference's (swin_transformer_mod.py:697-705).  Dropout /
    drop-path rates are accepted and ignored (inference: Identity, :354); `ape` and
    `use_checkpoint` must be False (every script of the reference passes False)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False,
                 patch_norm=True, use_checkpoint=False, rank=0):
        super().__init__()
        if ape or use_checkpoint or qk_scale is not None or not patch_norm or not qkv_bias:
            raise NotImplementedError("ape / use_checkpoint / qk_scale / patch_norm=False / qkv_bias=False are "
                                      "outside the accelerated path (the reference's scripts never set them)")
        self.num_layers, self.embed_dim = len(depths), embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, norm_layer)
        res = self.patch_embed.patches_resolution
        self.patches_resolution = res
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), input_resolution=(res[0] // 2 ** i, res[1] // 2 ** i),
                depth=depths[i], num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio,
                qkv_bias=qkv_bias, norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None))
        self.norm = norm_layer(self.num_features)
        self.geometry = Geometry(swin_img_size=img_size, swin_patch_size=patch_size, swin_in_chans=in_chans,
                                 swin_embed_dim=embed_dim, swin_depths=tuple(depths),
                                 swin_num_heads=tuple(num_heads), swin_window_size=window_size,
                                 swin_mlp_ratio=mlp_ratio, final_swin_dim=self.num_features)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
