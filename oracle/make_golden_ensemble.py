"""tests/golden/tiny_ensemble.npz: the REAL reference's EsembleCaptioningModel (legacy_models/
ensemble_captioning_model.py, imported as in make_golden.py) over two / three TINY end-to-end models with
different synthetic weights → beam-3 token ids, per-token log-probs.  Also checks the oracle's ensemble
search (oracle/expansionnet_ref.beam_search with a list of state dicts) against it.

    python oracle/make_golden_ensemble.py
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


def pad_tokens(pred, T):
    out = np.full((len(pred), len(pred[0]), T), -1, np.int64)
    for b, per in enumerate(pred):
        for j, seq in enumerate(per):
            out[b, j, :len(seq)] = seq
    return out


def main():
    torch.set_grad_enabled(False)
    E2E, _, _ = MG._import_reference()
    from models.ensemble_captioning_model import EsembleCaptioningModel     # type: ignore
    g = W.TINY
    store = {}
    img = W.synth_images(3, g)
    for name, members in (("two", ((0, "sharp"), (1, "sharp"))), ("three", ((0, "eos"), (1, "sharp"), (2, "xavier")))):
        sds = [W.synth_state_dict(g, seed=s, variant=v, eos_idx=MG.TINY_EOS) for s, v in members]
        models = [MG.build_ref_e2e(E2E, g, sd) for sd in sds]
        ens = EsembleCaptioningModel(models, "cpu").eval()
        for beam, T in ((3, 12), (1, 12)):
            pred, lp = ens(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=beam, how_many_outputs=beam,
                           beam_max_seq_len=T, sample_or_max="max", sos_idx=MG.TINY_SOS, eos_idx=MG.TINY_EOS)
            key = f"{name}.beam{beam}_T{T}"
            store[key + ".tokens"] = pad_tokens(pred, T)
            store[key + ".logprobs"] = lp.numpy()
            opred, olp = R.beam_search(sds, g, img, [0] * 3, MG.TINY_SOS, MG.TINY_EOS, beam, beam, T)
            same = opred == pred
            print(key, "oracle tokens equal:", same, "max |Δlogprob|:", float((olp - lp).abs().max()),
                  "lens", [len(p[0]) for p in pred])
            assert same
        store[name + ".seeds"] = np.asarray([s for s, _ in members])
        store[name + ".variants"] = np.asarray([v for _, v in members])
    np.savez_compressed(os.path.join(MG.OUT, "tiny_ensemble.npz"), **store)
    print("wrote tiny_ensemble.npz")


if __name__ == "__main__":
    main()
