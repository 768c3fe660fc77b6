"""tests/golden/tiny_long.npz: config-4-shaped searches (beam 5, beam_max_seq_len = the model's max_seq_len)
recorded from the REAL reference on the TINY geometry, plus the oracle cross-check.

    python oracle/make_golden_long.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from on_device_image_captioning_amd import weights as W          # noqa: E402
from oracle import expansionnet_ref as R                          # noqa: E402
from oracle import make_golden as MG                              # noqa: E402


def main():
    torch.set_grad_enabled(False)
    E2E, _, _ = MG._import_reference()
    g = W.TINY
    T = g.max_seq_len
    img = W.synth_images(3, g)
    store = {}
    for variant in ("xavier", "eos"):
        sd = W.synth_state_dict(g, variant=variant, eos_idx=MG.TINY_EOS)
        ref = MG.build_ref_e2e(E2E, g, sd)
        for k in (5, 3):
            toks, lps = ref(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=k, how_many_outputs=2,
                            beam_max_seq_len=T, sample_or_max="max", sos_idx=MG.TINY_SOS, eos_idx=MG.TINY_EOS)
            otoks, olps = R.beam_search(sd, g, img, [0] * 3, MG.TINY_SOS, MG.TINY_EOS, k, 2, T)
            key = f"{variant}.beam{k}_T{T}"
            store[key + ".tokens"] = np.array([[r + [-1] * (T - len(r)) for r in per] for per in toks])
            store[key + ".logprobs"] = lps.numpy()
            print(key, "oracle tokens equal:", toks == otoks, "max |Δlp|", float((lps - olps).abs().max()),
                  "lens", [[len(r) for r in per] for per in toks])
            assert toks == otoks
    np.savez_compressed(os.path.join(MG.OUT, "tiny_long.npz"), **store)
    print("wrote tiny_long.npz")

    # FULL geometry at the reference's demo length (demo.py:21 beam_max_seq_len 74, beam 5): 2 images
    g = W.FULL
    img = W.synth_images(2, g)
    store = {}
    for variant in ("xavier", "eos"):
        sd = W.synth_state_dict(g, variant=variant, eos_idx=MG.EOS)
        ref = MG.build_ref_e2e(E2E, g, sd)
        toks, lps = ref(enc_x=img, enc_x_num_pads=[0] * 2, mode="beam_search", beam_size=5, how_many_outputs=2,
                        beam_max_seq_len=74, sample_or_max="max", sos_idx=MG.SOS, eos_idx=MG.EOS)
        otoks, olps = R.beam_search(sd, g, img, [0] * 2, MG.SOS, MG.EOS, 5, 2, 74)
        key = f"{variant}.beam5_T74"
        store[key + ".tokens"] = np.array([[r + [-1] * (74 - len(r)) for r in per] for per in toks])
        store[key + ".logprobs"] = lps.numpy()
        print("full", key, "oracle tokens equal:", toks == otoks, "max |Δlp|", float((lps - olps).abs().max()),
              "lens", [[len(r) for r in per] for per in toks])
        del ref
    np.savez_compressed(os.path.join(MG.OUT, "full_long.npz"), **store)
    print("wrote full_long.npz")


if __name__ == "__main__":
    main()
