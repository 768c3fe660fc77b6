"""tests/golden/tiny_variants.npz + full_variants.npz: the fork's layer-removed models (`--param_config 1/2`:
N_enc = 2 / N_enc = N_dec = 2, reference test.py:360-365) run by the REAL reference classes on a 3-layer
synthetic checkpoint folded by the rules of test.py:38-77 (applied through
on_device_image_captioning_amd.checkpoint_utils.filter_state_dict; the script asserts that the reference's own
2-layer modules accept the folded dict with strict=True).  Build container only."""
from __future__ import annotations

import os
import sys
from dataclasses import replace

sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden as MG                                            # noqa: E402
from on_device_image_captioning_amd import weights as W              # noqa: E402
from on_device_image_captioning_amd.checkpoint_utils import filter_state_dict   # noqa: E402
from oracle import expansionnet_ref as R                             # noqa: E402

torch.set_grad_enabled(False)


def pad_tokens(toks, width):
    out = np.full((len(toks), len(toks[0]), width), -1, dtype=np.int64)
    for b, per in enumerate(toks):
        for j, row in enumerate(per):
            out[b, j, :len(row)] = row
    return out


def main():
    E2E, FEAT, _ = MG._import_reference()
    # ---- TINY geometry widened to 3 encoder / 3 decoder layers, end to end (Swin + captioner)
    g3 = replace(W.TINY, N_enc=3, N_dec=3)
    sd3 = W.synth_state_dict(g3, variant="eos", eos_idx=MG.TINY_EOS)
    img = W.synth_images(3, g3)
    store = {}
    for cfg, (ne, nd, mode) in {1: (2, 3, "enc"), 2: (2, 2, "dec")}.items():
        g = replace(g3, N_enc=ne, N_dec=nd)
        sd = filter_state_dict(sd3, mode)
        ref = MG.build_ref_e2e(E2E, g, sd)                           # strict load into the reference's 2-layer model
        mem = ref.forward_enc(img, [0] * 3)
        MG.put(store, f"cfg{cfg}.enc_out", mem)
        dev = float((mem - R.forward_enc(sd, g, img, [0] * 3)).abs().max())
        for k, T in ((1, 12), (3, 16)):
            toks, lps = ref(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=k, how_many_outputs=1,
                            beam_max_seq_len=T, sample_or_max="max", sos_idx=MG.TINY_SOS, eos_idx=MG.TINY_EOS)
            store[f"cfg{cfg}.beam{k}_T{T}.tokens"] = pad_tokens(toks, T)
            store[f"cfg{cfg}.beam{k}_T{T}.logprobs"] = lps.numpy()
            otoks, _ = R.beam_search(sd, g, img, [0] * 3, MG.TINY_SOS, MG.TINY_EOS, k, 1, T)
            assert otoks == toks, (cfg, k)
        print(f"tiny cfg{cfg}: oracle vs reference enc_out max diff {dev:.2e}")
    np.savez_compressed(os.path.join(MG.OUT, "tiny_variants.npz"), **store)

    # ---- full captioner geometry, features-only model (the shape train.py folds checkpoints for)
    gf = W.FULL
    sdf = W.synth_state_dict(gf, variant="xavier", end_to_end=False, img_feature_dim=1536)
    feats = W.synth_features(4, 144, 1536, seed=77)
    pads = [0, 5, 0, 17]
    store = {}
    for cfg, (ne, nd, mode) in {1: (2, 3, "enc"), 2: (2, 2, "dec")}.items():
        g = replace(gf, N_enc=ne, N_dec=nd)
        sd = filter_state_dict(sdf, mode)
        ref = MG.build_ref_feat(FEAT, g, sd, 1536)
        toks, lps = ref(enc_x=feats, enc_x_num_pads=pads, mode="beam_search", beam_size=3, how_many_outputs=1,
                        beam_max_seq_len=20, sample_or_max="max", sos_idx=MG.SOS, eos_idx=MG.EOS)
        store[f"cfg{cfg}.beam3_T20.tokens"] = pad_tokens(toks, 20)
        store[f"cfg{cfg}.beam3_T20.logprobs"] = lps.numpy()
        otoks, _ = R.beam_search(sd, g, feats, pads, MG.SOS, MG.EOS, 3, 1, 20, end_to_end=False)
        assert otoks == toks, cfg
        print(f"full features cfg{cfg}: lens {[len(t[0]) for t in toks]}")
    np.savez_compressed(os.path.join(MG.OUT, "full_variants.npz"), **store)


if __name__ == "__main__":
    main()
