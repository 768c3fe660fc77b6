"""Checkpoint plumbing for the fork's layer-removed model variants (reference
test.py:38-77,360-365 / train.py:305-341,437-446).

`--param_config 1` builds the model with N_enc = 2, `--param_config 2` with N_enc = N_dec = 2, and a
3-layer checkpoint is folded into it by `load_state_dict_filtered`:
  * every `encoders.2.*` tensor is stored under `encoders.1.*` (the dict is walked in checkpoint order, so
    layer 2 REPLACES layer 1: the pruned model keeps layers 0 and 2);
  * `enc_reduce_group.weight` (d, 3d) keeps its first and last thirds → (d, 2d) — the columns that multiplied
    the outputs of the kept layers;
  * with filter_prefixes == "dec" the same happens to `decoders.2.*` and `dec_reduce_group.weight` too.
Deliberate difference: the reference also swaps `swin_transf.patch_embed.proj.weight` for a freshly
initialised (192, 3, 3, 3) tensor, which cannot be loaded into the patch-4 backbone it is paired with
(size mismatch); the weight is kept here.
"""
from __future__ import annotations

from typing import Dict

import torch


def filter_state_dict(state_dict: Dict[str, torch.Tensor], filter_prefixes: str = "enc") -> Dict[str, torch.Tensor]:
    """3-layer `model_state_dict` → the state dict of the 2-encoder ("enc") or 2-encoder + 2-decoder ("dec")
    model, by the rules above."""
    if filter_prefixes not in ("enc", "dec"):
        raise ValueError("filter_prefixes must be 'enc' or 'dec'")

    def two_thirds(w: torch.Tensor) -> torch.Tensor:
        third = w.shape[-1] // 3
        return torch.hstack((w[:, :third], w[:, -third:]))

    out: Dict[str, torch.Tensor] = {}
    for key, value in state_dict.items():
        if filter_prefixes == "dec":
            if "decoders.2" in key:
                out[key.replace("decoders.2", "decoders.1")] = value
                continue
            if "dec_reduce_group.weight" in key:
                out[key] = two_thirds(value)
                continue
        if "encoders.2" in key:
            out[key.replace("encoders.2", "encoders.1")] = value
        elif "enc_reduce_group.weight" in key:
            out[key] = two_thirds(value)
        else:
            out[key] = value
    return out


def load_state_dict_filtered(model, checkpoint, filter_prefixes: str = "enc"):
    """Reference call shape (test.py:38): `checkpoint` is the dict torch.load returned (its
    'model_state_dict' entry is used); the model must have been built with the reduced layer counts."""
    return model.load_state_dict(filter_state_dict(checkpoint["model_state_dict"], filter_prefixes))
