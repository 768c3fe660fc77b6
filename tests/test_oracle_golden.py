"""Pin the oracle (oracle/expansionnet_ref.py) against outputs of the REAL reference.

The fixtures in tests/golden/ were produced by oracle/make_golden.py, which imports
/root/reference/legacy_models (as `models`) in the build container, loads the synthetic
checkpoint into it and records samples/checksums of its outputs.  Runs on CPU (`-m "not gpu"`).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, cached_state_dict
from on_device_image_captioning_amd import weights as W
from oracle import expansionnet_ref as R

torch.set_grad_enabled(False)
SOS, EOS = 79, 77
TSOS, TEOS = 3, 2


def check_sample(store, name, t, atol=2e-5, rtol=1e-5):
    meta = store[name + ".meta"]
    stride = int(meta[0])
    shape = [int(v) for v in meta[3:]]
    assert list(t.shape) == shape, (name, t.shape, shape)
    f = t.reshape(-1).double()
    np.testing.assert_allclose(f[::stride].float().numpy(), store[name + ".sample"], atol=atol, rtol=rtol,
                               err_msg=name)
    assert abs(float(f.sum()) - meta[1]) <= 1e-6 * meta[2] + 1e-3, name
    assert abs(float(f.abs().sum()) - meta[2]) <= 1e-5 * meta[2], name


def unpad(tok_arr):
    return [[[int(v) for v in row if v >= 0] for row in per] for per in tok_arr]


# ----------------------------------------------------------------------------------- A21 / A5
def test_state_dict_layout_matches_reference():
    spec = json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))
    for tag, g in (("full", W.FULL), ("tiny", W.TINY)):
        ours = [[k, list(s), "torch.int64" if kind == "rel_index" else "torch.float32"]
                for k, s, kind in W.state_dict_spec(g, True)]
        assert ours == spec[tag + "_e2e"]
        ours_f = [[k, list(s), "torch.float32"] for k, s, _ in W.state_dict_spec(g, False)]
        assert ours_f == spec[tag + "_feat"]
    assert len(spec["full_e2e"]) == 520 and len(spec["full_feat"]) == 158
    # known answer printed by the reference's benchmarking.py:58-66 / plotting.py:22
    assert spec["full_e2e_params"] == 233803076
    assert spec["full_feat_params"] == 38604560
    n = sum(int(np.prod(s)) for k, s, kind in W.state_dict_spec(W.FULL) if kind not in ("rel_index", "attn_mask"))
    assert n == 233803076


def test_constant_buffers_match_reference_hashes():
    spec = json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))
    g = W.FULL
    for k, shape, kind in W.state_dict_spec(g):
        if kind in ("rel_index", "attn_mask"):
            t = W.synth_tensor(k, shape, kind, g)
            assert hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest() == spec["full_buffer_sha256"][k], k


def test_window_token_index_is_roll_partition():
    # independent restatement with torch.roll + reshape on a labelled grid
    for res, ws, shift in ((96, 12, 6), (24, 12, 6), (24, 12, 0), (12, 12, 0)):
        lab = torch.arange(res * res).view(1, res, res, 1)
        sh = torch.roll(lab, shifts=(-shift, -shift), dims=(1, 2)) if shift else lab
        n = res // ws
        win = sh.view(1, n, ws, n, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(n * n, ws * ws)
        assert torch.equal(win, R.window_token_index(res, ws, shift))


# ----------------------------------------------------------------------------------- TINY end-to-end
@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_tiny_backbone_encoder_teacher(variant):
    g = W.TINY
    sd = cached_state_dict("TINY", variant)
    store = np.load(os.path.join(GOLDEN, f"tiny_{variant}.npz"))
    img = W.synth_images(3, g)
    taps = {}
    feats = R.swin_forward(sd, g, img, taps)
    for name in taps:
        check_sample(store, name, taps[name])
    check_sample(store, "swin_out", feats)
    check_sample(store, "enc_out", R.encoder_forward(sd, g, feats, [0] * 3))
    dec = torch.from_numpy(store["teacher.tokens"]).long()
    lg = R.forward_teacher(sd, g, img, dec, [0] * 3, store["teacher.pads"].tolist())
    check_sample(store, "teacher.logits", lg, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("variant", ["xavier", "eos"])
@pytest.mark.parametrize("k,T", [(1, 12), (3, 12), (5, 20), (3, 24)])
def test_tiny_beam_search(variant, k, T):
    g = W.TINY
    sd = cached_state_dict("TINY", variant)
    store = np.load(os.path.join(GOLDEN, f"tiny_{variant}.npz"))
    toks, lps = R.beam_search(sd, g, W.synth_images(3, g), [0] * 3, TSOS, TEOS, k, min(k, 2), T)
    assert toks == unpad(store[f"beam{k}_T{T}.tokens"])
    np.testing.assert_allclose(lps.numpy(), store[f"beam{k}_T{T}.logprobs"], atol=5e-5)
    if variant == "eos" and k == 3:
        lens = {len(r) for per in toks for r in per}
        assert len(lens) > 1, "eos fixture must exercise finished beams"


def test_tiny_features_only_ragged_pads():
    g, fd = W.TINY, 64
    sd = cached_state_dict("TINY", "eos", end_to_end=False, img_feature_dim=fd)
    store = np.load(os.path.join(GOLDEN, "tiny_features.npz"))
    feats = W.synth_features(4, 20, fd)
    epads = [0, 3, 7, 1]
    check_sample(store, "enc_out", R.forward_enc(sd, g, feats, epads, end_to_end=False))
    for k, T in ((1, 10), (3, 16)):
        toks, lps = R.beam_search(sd, g, feats, epads, TSOS, TEOS, k, 1, T, end_to_end=False)
        assert toks == unpad(store[f"beam{k}_T{T}.tokens"])
        np.testing.assert_allclose(lps.numpy(), store[f"beam{k}_T{T}.logprobs"], atol=5e-5)


# ----------------------------------------------------------------------------------- FULL geometry
def test_full_backbone_and_search():
    g = W.FULL
    sd = cached_state_dict("FULL", "xavier")
    store = np.load(os.path.join(GOLDEN, "full_xavier.npz"))
    img = W.synth_images(2, g)
    taps = {}
    feats = R.swin_forward(sd, g, img, taps)
    for name in taps:
        if name + ".meta" in store:
            check_sample(store, name, taps[name])
    check_sample(store, "swin_out", feats)
    mem = R.encoder_forward(sd, g, feats, [0, 0])
    check_sample(store, "enc_out", mem)
    # isolated window attention (A4) of the first shifted block of each stage
    for s in range(4):
        p = f"swin_transf.layers.{s}.blocks.1"
        C, h = g.stage_dim(s), g.swin_num_heads[s]
        nW = (g.stage_res(s) // g.stage_window(s)) ** 2
        x = W.synth_features(nW, 144, C, seed=100 + s)
        qkv = torch.nn.functional.linear(x, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
        qkv = qkv.view(1, nW, 144, 3, h, 32)
        q, k, v = (qkv[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))
        o = R.window_attention_core(q, k, v, R.rel_pos_bias(sd, p + ".attn"), sd.get(p + ".attn_mask"), 32 ** -0.5)
        o = o.permute(0, 1, 3, 2, 4).reshape(nW, 144, C)
        o = torch.nn.functional.linear(o, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
        check_sample(store, f"winattn_s{s}", o)
    dec = torch.from_numpy(store["teacher.tokens"]).long()
    orig = R.forward_enc
    try:
        R.forward_enc = lambda *a, **kw: mem          # reuse the encoder output computed above
        lg = R.forward_teacher(sd, g, img, dec, [0, 0], [0, 3], True)
        check_sample(store, "teacher.logprobs", lg, atol=1e-4, rtol=1e-4)
        for k in (1, 3):
            toks, lps = R.beam_search(sd, g, img, [0, 0], SOS, EOS, k, 1, 20)
            assert toks == unpad(store[f"beam{k}_T20.tokens"])
            np.testing.assert_allclose(lps.numpy(), store[f"beam{k}_T20.logprobs"], atol=5e-5)
    finally:
        R.forward_enc = orig


def test_full_eos_search():
    g = W.FULL
    sd = cached_state_dict("FULL", "eos")
    store = np.load(os.path.join(GOLDEN, "full_eos.npz"))
    img = W.synth_images(2, g)
    mem = R.forward_enc(sd, g, img, [0, 0])
    orig = R.forward_enc
    try:
        R.forward_enc = lambda *a, **kw: mem
        for k in (1, 3, 5):
            toks, lps = R.beam_search(sd, g, img, [0, 0], SOS, EOS, k, 1, 20)
            assert toks == unpad(store[f"beam{k}_T20.tokens"])
            np.testing.assert_allclose(lps.numpy(), store[f"beam{k}_T20.logprobs"], atol=5e-5)
    finally:
        R.forward_enc = orig


# ----------------------------------------------------------------------------------- caller-side helpers
def test_tokens2description_strings():
    helper = json.load(open(os.path.join(GOLDEN, "helpers.json")))
    from on_device_image_captioning_amd.language_utils import load_vocab, tokens2description
    w2i, i2w = load_vocab()
    assert len(i2w) == helper["vocab_size"] == 10000
    assert i2w[SOS] == helper["sos_word"] and i2w[EOS] == helper["eos_word"]
    assert hashlib.sha256("\n".join(i2w).encode()).hexdigest() == helper["vocab_sha256"]
    for toks, want in helper["tokens2description"]:
        assert tokens2description(toks, i2w, SOS, EOS) == want
        assert R.tokens2description(toks, i2w, SOS, EOS) == want


def test_cider_d_matches_reference_scorer():
    """F1: own CIDEr-D scorer vs scores recorded from the reference's eval/cider (oracle/make_golden_cider.py)."""
    from on_device_image_captioning_amd.cider import CiderD
    cases = json.load(open(os.path.join(GOLDEN, "cider.json")))
    assert len(cases) == 4
    for c in cases:
        score, scores = CiderD().compute_score(c["gts"], c["res"])
        assert abs(score - c["score"]) < 1e-9
        np.testing.assert_allclose(scores, c["scores"], atol=1e-9)
    assert max(c["score"] for c in cases) > 1.0       # non-degenerate fixtures


# ----------------------------------------------------------------------------------- F4 ensemble
@pytest.mark.parametrize("name", ["two", "three"])
@pytest.mark.parametrize("beam", [3, 1])
def test_oracle_ensemble_search_matches_reference(name, beam):
    """oracle beam_search over a LIST of checkpoints = the reference's EsembleCaptioningModel
    (fixtures: oracle/make_golden_ensemble.py)."""
    store = np.load(os.path.join(GOLDEN, "tiny_ensemble.npz"))
    g = W.TINY
    sds = [W.synth_state_dict(g, seed=int(s), variant=str(v), eos_idx=2)
           for s, v in zip(store[name + ".seeds"], store[name + ".variants"])]
    img = W.synth_images(3, g)
    pred, lp = R.beam_search(sds, g, img, [0] * 3, 3, 2, beam, beam, 12)
    assert pred == unpad(store[f"{name}.beam{beam}_T12.tokens"])
    np.testing.assert_allclose(lp.numpy(), store[f"{name}.beam{beam}_T12.logprobs"], atol=2e-5)


# ----------------------------------------------------------------------------------- config-4 shape
@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_oracle_long_beam5_search_matches_reference(variant):
    """beam 5 at beam_max_seq_len = max_seq_len (the demo.py / COCO-evaluation shape) on the TINY geometry."""
    store = np.load(os.path.join(GOLDEN, "tiny_long.npz"))
    g = W.TINY
    sd = W.synth_state_dict(g, variant=variant, eos_idx=2)
    pred, lp = R.beam_search(sd, g, W.synth_images(3, g), [0] * 3, 3, 2, 5, 2, g.max_seq_len)
    assert pred == unpad(store[f"{variant}.beam5_T{g.max_seq_len}.tokens"])
    np.testing.assert_allclose(lp.numpy(), store[f"{variant}.beam5_T{g.max_seq_len}.logprobs"], atol=2e-5)


# ----------------------------------------------------------------------------------- F1: caption cleaner
def test_caption_cleaner_matches_reference_strings():
    """utils/language_utils.py:4-72 restated in language_utils.py: every helper reproduces the strings the
    reference's functions returned for the recorded raw captions (oracle/make_golden_text.py)."""
    from on_device_image_captioning_amd import language_utils as L
    g = json.load(open(os.path.join(GOLDEN, "text_cleaner.json")))
    raw = g["raw"]
    lo = L.lowercase_and_clean_trailing_spaces(raw)
    assert lo == g["lowercase_and_clean_trailing_spaces"]
    sp = L.add_space_between_non_alphanumeric_symbols(lo)
    assert sp == g["add_space_between_non_alphanumeric_symbols"]
    rp = L.remove_punctuations(sp)
    assert rp == g["remove_punctuations"]
    tk = L.tokenize(rp)
    assert tk == g["tokenize"]
    assert L.remove_punctuations(raw) == g["remove_punctuations_direct"]
    assert L.tokenize(raw) == g["tokenize_direct"]
    assert L.compute_num_pads(g["compute_num_pads_in"]) == g["compute_num_pads"]
    ids = L.convert_allsentences_word2idx(tk, g["word2idx"])
    assert ids == g["convert_allsentences_word2idx"]
    i2w = sorted(g["word2idx"], key=g["word2idx"].get)
    assert L.convert_allsentences_idx2word(ids, i2w) == g["convert_allsentences_idx2word"]
    assert L.clean_captions(raw) == [" ".join(t) for t in g["tokenize"]]


# ----------------------------------------------------------------------------------- F4: layer-removed variants
def test_layer_removed_checkpoint_folding_and_oracle():
    """`--param_config 1/2` (test.py:360-365): the 3-layer checkpoint folded by the rules of test.py:38-77 and run
    by the oracle's 2-layer model gives the captions the REFERENCE's 2-layer classes gave on the same folded
    dict (tests/golden/tiny_variants.npz, oracle/make_golden_variants.py)."""
    from dataclasses import replace
    from on_device_image_captioning_amd.checkpoint_utils import filter_state_dict
    g3 = replace(W.TINY, N_enc=3, N_dec=3)
    sd3 = W.synth_state_dict(g3, variant="eos", eos_idx=TEOS)
    e = filter_state_dict(sd3, "enc")
    d = filter_state_dict(sd3, "dec")
    assert not any(k.startswith("encoders.2") for k in e) and any(k.startswith("decoders.2") for k in e)
    assert not any(k.startswith(("encoders.2", "decoders.2")) for k in d)
    assert torch.equal(e["encoders.1.ff.linear_1.weight"], sd3["encoders.2.ff.linear_1.weight"])   # layer 2 replaces 1
    assert torch.equal(d["decoders.1.mha.Wq.weight"], sd3["decoders.2.mha.Wq.weight"])
    dm = g3.d_model
    w = sd3["enc_reduce_group.weight"]
    assert torch.equal(e["enc_reduce_group.weight"], torch.cat([w[:, :dm], w[:, 2 * dm:]], 1))
    assert e["dec_reduce_group.weight"].shape == (dm, 3 * dm) and d["dec_reduce_group.weight"].shape == (dm, 2 * dm)
    store = np.load(os.path.join(GOLDEN, "tiny_variants.npz"))
    img = W.synth_images(3, g3)
    for cfg, (ne, nd, sd) in {1: (2, 3, e), 2: (2, 2, d)}.items():
        g = replace(g3, N_enc=ne, N_dec=nd)
        check_sample(store, f"cfg{cfg}.enc_out", R.forward_enc(sd, g, img, [0] * 3))
        toks, lps = R.beam_search(sd, g, img, [0] * 3, TSOS, TEOS, 3, 1, 16)
        assert toks == unpad(store[f"cfg{cfg}.beam3_T16.tokens"])
        np.testing.assert_allclose(lps.numpy(), store[f"cfg{cfg}.beam3_T16.logprobs"], atol=5e-5)
