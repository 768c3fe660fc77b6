"""End-to-end parity of the HIP path (called through the reference's own class API) against the
golden fixtures recorded from the REAL reference (tests/golden/, see oracle/make_golden.py) and
against the CPU oracle.  `-m gpu` only.

Bars
  fp32 mode : activations within 2e-4 of the reference's fp32 CPU outputs (summation order only),
              token ids of greedy AND beam search identical to the reference's, log-probs 1e-3; at the
              bench shape (Swin-L, B=16, beam 3, hipGraph pipeline) the CIDEr-D delta against the direct
              call is 0 — the north star's "within ±0.1 CIDEr-D" bar holds with room to spare.
  bf16 mode : backbone features within 3e-2 relative; teacher-forced log-probs within 0.2 nat and the
              arg-max identical wherever the reference's top-1/top-2 margin exceeds twice the local
              error; the hipGraph pipeline at the bench shape returns EXACTLY the captions of the
              un-pipelined bf16 call.  Free-running bf16 captions are NOT identical to the fp32 ones
              on synthetic (random) weights: about a third diverge after a common prefix, because a
              random-weight decoder has top-1/top-2 margins (median 0.65 nat, p10 0.08 on the sharpened
              checkpoint) inside the reach of a 1e-2 relative feature perturbation.  The measured
              CIDEr-D delta is recorded (gpurun_out/parity_diag.json, bench.py's `parity` object) and
              bounded below as a regression floor; parity on real weights is unpinned (rf_model.pth
              is not available offline).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, cached_state_dict
from on_device_image_captioning_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
SOS, EOS = 79, 77
TSOS, TEOS = 3, 2
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from on_device_image_captioning_amd import _hip
    _hip.load()


_MODELS = {}


def build_model(geom, variant, precision="fp32"):
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import End_ExpansionNet_v2, make_drop_args
    key = (geom, variant)
    if key not in _MODELS:
        g = getattr(W, geom)
        m = End_ExpansionNet_v2(**g.model_kwargs(), output_word2idx={i: i for i in range(g.vocab_size)},
                                output_idx2word=list(range(g.vocab_size)), drop_args=make_drop_args(), rank=DEV)
        m.load_state_dict(cached_state_dict(geom, variant), strict=True)
        _MODELS[key] = m.to(DEV).eval()
    return _MODELS[key].set_precision(precision)


def check_sample(store, name, t, atol, rtol=1e-4):
    meta = store[name + ".meta"]
    stride = int(meta[0])
    shape = [int(v) for v in meta[3:]]
    assert list(t.shape) == shape, (name, t.shape, shape)
    f = t.detach().reshape(-1).double().cpu()
    want = store[name + ".sample"]
    got = f[::stride].float().numpy()
    err = np.abs(got - want).max()
    assert err <= atol + rtol * np.abs(want).max(), f"{name}: max err {err:.3e}"
    assert abs(float(f.abs().sum()) - meta[2]) <= 1e-3 * meta[2], name
    return err


def _diag(key, value):
    """Side-channel for measured error levels (read back from gpurun_out/)."""
    path = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "parity_diag.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        d = json.load(open(path)) if os.path.exists(path) else {}
        d[key] = value
        json.dump(d, open(path, "w"), indent=1)
    except OSError:
        pass


def unpad(tok_arr):
    return [[[int(v) for v in row if v >= 0] for row in per] for per in tok_arr]


# ----------------------------------------------------------------------------------------- TINY fp32
@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_tiny_backbone_encoder_fp32(variant):
    g = W.TINY
    m = build_model("TINY", variant)
    store = np.load(os.path.join(GOLDEN, f"tiny_{variant}.npz"))
    img = W.synth_images(3, g).to(DEV)
    swin, cap = m._engines()
    taps = {}
    feats = swin.forward(img, taps)
    for name, t in taps.items():
        check_sample(store, name, t, 2e-4)
    check_sample(store, "swin_out", feats, 2e-4)
    check_sample(store, "enc_out", m.forward_enc(img, [0] * 3), 2e-4)


@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_tiny_teacher_forced_logits(variant):
    g = W.TINY
    m = build_model("TINY", variant)
    store = np.load(os.path.join(GOLDEN, f"tiny_{variant}.npz"))
    img = W.synth_images(3, g).to(DEV)
    dec = torch.from_numpy(store["teacher.tokens"]).long()
    lg = m(enc_x=img, dec_x=dec.to(DEV), enc_x_num_pads=[0] * 3, dec_x_num_pads=store["teacher.pads"].tolist(),
           apply_log_softmax=False, mode="forward")
    check_sample(store, "teacher.logits", lg, 1e-3 if variant == "eos" else 2e-4)


@pytest.mark.parametrize("variant", ["xavier", "eos"])
@pytest.mark.parametrize("k,T", [(1, 12), (3, 12), (5, 20), (3, 24)])
def test_tiny_beam_search_matches_reference(variant, k, T):
    g = W.TINY
    m = build_model("TINY", variant)
    store = np.load(os.path.join(GOLDEN, f"tiny_{variant}.npz"))
    img = W.synth_images(3, g).to(DEV)
    toks, lps = m(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=k, how_many_outputs=min(k, 2),
                  beam_max_seq_len=T, sample_or_max="max", sos_idx=TSOS, eos_idx=TEOS)
    assert toks == unpad(store[f"beam{k}_T{T}.tokens"])
    np.testing.assert_allclose(lps.cpu().numpy(), store[f"beam{k}_T{T}.logprobs"], atol=1e-3)


def test_tiny_features_only_ragged_pads():
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import make_drop_args
    from on_device_image_captioning_amd.ExpansionNet_v2 import ExpansionNet_v2
    g, fd = W.TINY, 64
    sd = cached_state_dict("TINY", "eos", end_to_end=False, img_feature_dim=fd)
    m = ExpansionNet_v2(d_model=g.d_model, N_enc=g.N_enc, N_dec=g.N_dec, ff=g.ff, num_heads=g.num_heads,
                        num_exp_enc_list=list(g.num_exp_enc_list), num_exp_dec=g.num_exp_dec,
                        output_word2idx={i: i for i in range(g.vocab_size)},
                        output_idx2word=list(range(g.vocab_size)), max_seq_len=g.max_seq_len,
                        drop_args=make_drop_args(), img_feature_dim=fd, rank=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).eval()
    store = np.load(os.path.join(GOLDEN, "tiny_features.npz"))
    feats = W.synth_features(4, 20, fd).to(DEV)
    epads = [0, 3, 7, 1]
    mem = m.forward_enc(feats, epads)
    # padded encoder rows are garbage-by-construction in the reference too; they are compared as well
    check_sample(store, "enc_out", mem, 2e-4)
    for k, T in ((1, 10), (3, 16)):
        toks, lps = m(enc_x=feats, enc_x_num_pads=epads, mode="beam_search", beam_size=k, how_many_outputs=1,
                      beam_max_seq_len=T, sample_or_max="max", sos_idx=TSOS, eos_idx=TEOS)
        assert toks == unpad(store[f"beam{k}_T{T}.tokens"])
        np.testing.assert_allclose(lps.cpu().numpy(), store[f"beam{k}_T{T}.logprobs"], atol=1e-3)


def test_captioner_call_shape_and_errors():
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import E2E_ExpansionNet_Captioner
    g = W.TINY
    m = build_model("TINY", "eos")
    store = np.load(os.path.join(GOLDEN, "tiny_eos.npz"))
    cap = E2E_ExpansionNet_Captioner({"sos_idx": TSOS, "eos_idx": TEOS, "beam_size": 3, "how_many_outputs": 2,
                                      "beam_max_seq_len": 12}, model=m, rank=DEV)
    toks, _ = cap(W.synth_images(3, g).to(DEV), enc_x_num_pads=[0] * 3, mode="beam_search")
    assert toks == unpad(store["beam3_T12.tokens"])
    with pytest.raises(ValueError):
        E2E_ExpansionNet_Captioner({"sos_idx": 1, "eos_idx": 2})
    with pytest.raises(AssertionError):
        m(enc_x=W.synth_images(1, g).to(DEV), enc_x_num_pads=[0], mode="beam_search", beam_size=2,
          how_many_outputs=3, sos_idx=TSOS, eos_idx=TEOS)
    with pytest.raises(AssertionError):
        m(enc_x=W.synth_images(1, g).to(DEV), enc_x_num_pads=[0], mode="beam_search")      # no sos/eos


def test_tiny64_bf16_backbone_close_to_oracle():
    from oracle import expansionnet_ref as R
    g = W.TINY64
    m = build_model("TINY64", "xavier", "bf16")
    img = W.synth_images(3, g)
    want = R.swin_forward(cached_state_dict("TINY64", "xavier"), g, img)
    feats = m._engines()[0].forward(img.to(DEV)).cpu()
    rel = (feats - want).abs().max().item() / want.abs().max().item()
    _diag("tiny64_bf16_swin_rel_err", rel)
    assert rel < 3e-2, rel
    build_model("TINY64", "xavier", "fp32")


# ----------------------------------------------------------------------------------------- FULL (Swin-L/384)
def test_full_fp32_backbone_and_search():
    g = W.FULL
    m = build_model("FULL", "xavier")
    store = np.load(os.path.join(GOLDEN, "full_xavier.npz"))
    img = W.synth_images(2, g).to(DEV)
    swin, cap = m._engines()
    taps = {}
    feats = swin.forward(img, taps)
    worst = 0.0
    for name, t in taps.items():
        if name + ".meta" in store:
            worst = max(worst, check_sample(store, name, t, 5e-4))
    check_sample(store, "swin_out", feats, 5e-4)
    mem = m.forward_enc(img, [0, 0])
    check_sample(store, "enc_out", mem, 5e-4)
    dec = torch.from_numpy(store["teacher.tokens"]).long().to(DEV)
    lp = m.forward_dec(mem, [0, 0], dec, [0, 3], apply_log_softmax=True)
    check_sample(store, "teacher.logprobs", lp, 1e-3)
    for k in (1, 3, 5):
        toks, lps = m._search_from_memory(mem, [0, 0], SOS, EOS, k, 1, 20)
        assert toks == unpad(store[f"beam{k}_T20.tokens"]), f"beam {k}"
        np.testing.assert_allclose(lps.cpu().numpy(), store[f"beam{k}_T20.logprobs"], atol=1e-3)


def test_full_fp32_eos_search():
    g = W.FULL
    m = build_model("FULL", "eos")
    store = np.load(os.path.join(GOLDEN, "full_eos.npz"))
    img = W.synth_images(2, g).to(DEV)
    mem = m.forward_enc(img, [0, 0])
    for k in (1, 3, 5):
        toks, lps = m._search_from_memory(mem, [0, 0], SOS, EOS, k, 1, 20)
        assert toks == unpad(store[f"beam{k}_T20.tokens"]), f"beam {k}"
        np.testing.assert_allclose(lps.cpu().numpy(), store[f"beam{k}_T20.logprobs"], atol=2e-3)


def test_full_bf16_margin_aware_parity():
    """bf16 backbone vs the fp32 reference on the SAME prefixes (the reference's greedy captions,
    teacher forced): log-probs of every position within 0.2 nat, backbone features within 3e-2,
    and the arg-max token identical wherever the reference's top-1/top-2 margin exceeds twice the
    observed log-prob error (with random weights the margins are often smaller than any reduced
    precision can resolve, SURVEY §7 'Hard parts')."""
    from oracle import expansionnet_ref as R
    g = W.FULL
    sd = cached_state_dict("FULL", "eos")
    store = np.load(os.path.join(GOLDEN, "full_eos.npz"))
    img = W.synth_images(2, g)
    ref_tok = unpad(store["beam1_T20.tokens"])
    T = max(len(r[0]) for r in ref_tok)
    dec = torch.full((2, T), EOS, dtype=torch.long)
    pads = []
    for b, r in enumerate(ref_tok):
        dec[b, :len(r[0])] = torch.tensor(r[0])
        pads.append(T - len(r[0]))
    feats_ref = R.swin_forward(sd, g, img)
    mem_ref = R.encoder_forward(sd, g, feats_ref, [0, 0])
    lp_ref = R.decoder_forward(sd, g, mem_ref, [0, 0], dec, pads, True)

    m = build_model("FULL", "eos", "bf16")
    swin, cap = m._engines()
    feats = swin.forward(img.to(DEV))
    rel = (feats.cpu() - feats_ref).abs().max().item() / feats_ref.abs().max().item()
    lp = m.forward_dec(cap.encode(feats, m._enc_lens(2, 144, None)), [0, 0], dec.to(DEV), pads, True).cpu()
    errs, flips, checked = [], 0, 0
    for b in range(2):
        n = T - pads[b]
        e = (lp[b, :n] - lp_ref[b, :n]).abs()
        top2 = torch.topk(lp_ref[b, :n], 2, -1).values
        margin = top2[:, 0] - top2[:, 1]
        errs.append(float(e.max()))
        for t in range(n):
            if margin[t] > 2 * float(e[t].max()) + 1e-3:
                checked += 1
                flips += int(lp[b, t].argmax() != lp_ref[b, t].argmax())
    toks, _ = m(enc_x=img.to(DEV), enc_x_num_pads=[0, 0], mode="beam_search", beam_size=1, how_many_outputs=1,
                beam_max_seq_len=20, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
    agree = [sum(1 for x, y in zip(a[0], r[0]) if x == y) for a, r in zip(toks, ref_tok)]
    _diag("full_bf16", dict(swin_rel_err=rel, max_logprob_err=errs, margin_checked=checked, flips=flips,
                            greedy_prefix_agree=agree, ref_len=[len(r[0]) for r in ref_tok]))
    build_model("FULL", "eos", "fp32")
    assert rel < 3e-2
    assert max(errs) < 0.2
    assert flips == 0 and checked > 0


def test_engine_is_hip_backed_and_host_tensors_are_refused_by_the_ops():
    """There is no CPU arithmetic anywhere: an op handed a host tensor raises.  (A host-resident nn.Module is
    fine — its engines live on the GPU: test_demo_flow_with_a_host_resident_model.)"""
    from on_device_image_captioning_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 64), torch.zeros(8, 64))


# ----------------------------------------------------------------------------------------- graph pipeline
@pytest.mark.parametrize("variant,graphs,poll", [("eos", True, 0), ("eos", False, 4), ("xavier", True, 4)])
def test_pipeline_matches_reference_best_caption(variant, graphs, poll):
    """CaptionPipeline (two streams, hipGraph replay, deferred collection) returns the reference's
    best beam-3 caption for every image, batch after batch."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.TINY
    m = build_model("TINY", variant)
    store = np.load(os.path.join(GOLDEN, f"tiny_{variant}.npz"))
    want = [per[0] for per in unpad(store["beam3_T12.tokens"])]
    img = W.synth_images(3, g).to(DEV)
    pipe = CaptionPipeline(m, 3, 3, 12, TSOS, TEOS, use_graphs=graphs, done_poll=poll)
    pipe.submit(img)
    pipe.submit(img.flip(0).contiguous())
    pipe.submit(img)                                   # three outstanding: encode + two decode lanes
    assert pipe.full()
    assert pipe.collect() == want
    assert pipe.collect() == want[::-1]
    assert pipe.collect() == want
    assert pipe(img) == want


@pytest.mark.parametrize("group,lanes,poll", [(2, 2, 0), (3, 1, 4), (2, 1, 0)])
def test_pipeline_decode_groups_match_reference(group, lanes, poll):
    """`decode_group` batches searched by one chain of step kernels (G·B·k rows) give the per-batch
    captions, including a partially filled group flushed by collect()."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.TINY
    m = build_model("TINY", "eos")
    store = np.load(os.path.join(GOLDEN, "tiny_eos.npz"))
    want = [per[0] for per in unpad(store["beam3_T12.tokens"])]
    img = W.synth_images(3, g).to(DEV)
    flip = img.flip(0).contiguous()
    pipe = CaptionPipeline(m, 3, 3, 12, TSOS, TEOS, done_poll=poll, decode_lanes=lanes, decode_group=group)
    got, sent = [], []
    for i in range(7):                                  # 7 batches: full groups + a partial tail
        while pipe.full():
            got.append(pipe.collect())
        pipe.submit(flip if i % 3 == 1 else img)
        sent.append(want[::-1] if i % 3 == 1 else want)
    while pipe.outstanding():
        got.append(pipe.collect())
    assert got == sent
    assert pipe(flip) == want[::-1]                     # a lone batch: group of one real + repeats


def test_pipeline_on_cu_masked_streams_gives_the_same_captions():
    """decode_cus=R: the decode lanes on compute units 0..R-1 of the mask numbering, the encode stream on the
    others (hipExtStreamCreateWithCUMask; an operating-point option, off by default) — captions unchanged."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.TINY
    m = build_model("TINY", "eos")
    store = np.load(os.path.join(GOLDEN, "tiny_eos.npz"))
    want = [per[0] for per in unpad(store["beam3_T12.tokens"])]
    img = W.synth_images(3, g).to(DEV)
    pipe = CaptionPipeline(m, 3, 3, 12, TSOS, TEOS, decode_cus=32)
    pipe.submit(img)
    pipe.submit(img.flip(0).contiguous())
    pipe.submit(img)
    assert pipe.collect() == want
    assert pipe.collect() == want[::-1]
    assert pipe.collect() == want


# ----------------------------------------------------------------------------------------- F3: sampling
def test_sampling_mode_logprobs_are_teacher_forced_logprobs():
    """mode='sampling' draws tokens with the device RNG (not reproducible against the reference's CPU
    generator), but the log-prob it reports for every drawn token must be the model's log-prob of that
    token given the prefix — checked against an independent teacher-forced pass of the CPU oracle."""
    from oracle import expansionnet_ref as R
    g = W.TINY
    sd = cached_state_dict("TINY", "eos")
    m = build_model("TINY", "eos")
    img = W.synth_images(2, g)
    torch.manual_seed(0)
    toks, lps = m(enc_x=img.to(DEV), enc_x_num_pads=[0, 0], mode="sampling", how_many_outputs=3,
                  sample_max_seq_len=10, sos_idx=TSOS, eos_idx=TEOS)
    assert len(toks) == 2 and all(len(per) == 3 for per in toks)
    mem = R.forward_enc(sd, g, img, [0, 0])
    lens = set()
    for b in range(2):
        for j in range(3):
            seq = toks[b][j]
            lens.add(len(seq))
            assert seq[0] == TSOS and TEOS not in seq[1:-1]
            dec = torch.tensor([seq])
            ref = R.decoder_forward(sd, g, mem[b:b + 1], [0], dec, [0], True)[0]
            want = [0.0] + [float(ref[t, seq[t + 1]]) for t in range(len(seq) - 1)]
            got = lps[b, j, :len(seq)].cpu().tolist()
            np.testing.assert_allclose(got, want, atol=2e-3)
            assert float(lps[b, j, len(seq):].abs().sum()) == 0.0
    # sampled beam search runs and returns well-formed output
    toks2, lps2 = m(enc_x=img.to(DEV), enc_x_num_pads=[0, 0], mode="beam_search", beam_size=3, how_many_outputs=2,
                    beam_max_seq_len=8, sample_or_max="sample", sos_idx=TSOS, eos_idx=TEOS)
    assert len(toks2) == 2 and all(len(per) == 2 and per[0][0] == TSOS for per in toks2)
    assert lps2.shape[:2] == (2, 2)


# ----------------------------------------------------------------------------------------- other BASELINE configs
def _features_model(g, sd, fd, precision="fp32"):
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import make_drop_args
    from on_device_image_captioning_amd.ExpansionNet_v2 import ExpansionNet_v2
    m = ExpansionNet_v2(d_model=g.d_model, N_enc=g.N_enc, N_dec=g.N_dec, ff=g.ff, num_heads=g.num_heads,
                        num_exp_enc_list=list(g.num_exp_enc_list), num_exp_dec=g.num_exp_dec,
                        output_word2idx={i: i for i in range(g.vocab_size)},
                        output_idx2word=list(range(g.vocab_size)), max_seq_len=g.max_seq_len,
                        drop_args=make_drop_args(), img_feature_dim=fd, rank=DEV)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval().set_precision(precision)


def test_config2_features_only_full_geometry_batch48_beam3():
    """BASELINE.json configs[1]: ExpansionNet_v2 features-only (N_enc=3, N_dec=3, d=512), batch 48, beam 3 —
    full captioner geometry, ragged encoder padding, token ids against the live CPU oracle."""
    from oracle import expansionnet_ref as R
    g = W.FULL
    sd = cached_state_dict("FULL", "eos", end_to_end=False, img_feature_dim=1536, eos_idx=EOS)
    m = _features_model(g, sd, 1536)
    feats = W.synth_features(48, 144, 1536)
    epads = [(7 * i) % 23 for i in range(48)]
    toks, lps = m(enc_x=feats.to(DEV), enc_x_num_pads=epads, mode="beam_search", beam_size=3, how_many_outputs=1,
                  beam_max_seq_len=20, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
    want, wlps = R.beam_search(sd, g, feats, epads, SOS, EOS, 3, 1, 20, end_to_end=False)
    assert toks == want
    np.testing.assert_allclose(lps.cpu().numpy(), wlps.numpy(), atol=2e-3)


def test_features_only_bf16_encoder_close():
    from oracle import expansionnet_ref as R
    g, fd = W.TINY, 64
    sd = cached_state_dict("TINY", "eos", end_to_end=False, img_feature_dim=fd)
    m = _features_model(g, sd, fd, "bf16")
    feats = W.synth_features(4, 20, fd)
    epads = [0, 3, 7, 1]
    mem = m.forward_enc(feats.to(DEV), epads).cpu()
    want = R.forward_enc(sd, g, feats, epads, end_to_end=False)
    valid = torch.arange(20)[None, :] < (20 - torch.tensor(epads))[:, None]
    rel = ((mem - want).abs() * valid[..., None]).max().item() / want.abs().max().item()
    _diag("tiny_features_bf16_enc_rel_err", rel)
    assert rel < 4e-2, rel


def test_sharded_evaluation_driver_on_gpu_pipeline():
    """configs[3] shape in miniature: many images, contiguous shards, sub-batches through the graph
    pipeline, ragged tail, gather (single rank here; the two-rank exchange is covered on CPU with gloo)."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline, caption_sharded
    g = W.TINY
    m = build_model("TINY", "eos")
    store = np.load(os.path.join(GOLDEN, "tiny_eos.npz"))
    want3 = [per[0] for per in unpad(store["beam5_T20.tokens"])]
    base = W.synth_images(3, g).to(DEV)
    n_items, batch = 11, 4
    images = base[[i % 3 for i in range(n_items)]]
    pipe = CaptionPipeline(m, batch, 5, 20, TSOS, TEOS, done_poll=4)

    def caption_batch(imgs):
        pipe.submit(imgs.contiguous())
        return pipe.collect_device()

    caps = caption_sharded(caption_batch, n_items, lambda lo, hi: images[lo:hi], batch, pipe.T, TEOS,
                           torch.device(DEV), 0, 1)
    assert caps == [want3[i % 3] for i in range(n_items)]


# ----------------------------------------------------------------------------------------- F4: ensemble
@pytest.mark.parametrize("name", ["two", "three"])
@pytest.mark.parametrize("beam", [3, 1])
def test_ensemble_beam_search_matches_reference(name, beam):
    """EsembleCaptioningModel on the step engines (shared beam state, log-mean-softmax kernel) returns the
    token ids of the reference's ensemble search and its per-token log-probs."""
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import End_ExpansionNet_v2, make_drop_args
    from on_device_image_captioning_amd.ensemble_captioning_model import EsembleCaptioningModel
    store = np.load(os.path.join(GOLDEN, "tiny_ensemble.npz"))
    g = W.TINY
    members = []
    for s_, v_ in zip(store[name + ".seeds"], store[name + ".variants"]):
        m = End_ExpansionNet_v2(**g.model_kwargs(), output_word2idx={i: i for i in range(g.vocab_size)},
                                output_idx2word=list(range(g.vocab_size)), drop_args=make_drop_args(), rank=DEV)
        m.load_state_dict(W.synth_state_dict(g, seed=int(s_), variant=str(v_), eos_idx=TEOS), strict=True)
        members.append(m.to(DEV).eval())
    ens = EsembleCaptioningModel(members, DEV)
    img = W.synth_images(3, g).to(DEV)
    pred, lp = ens(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=beam, how_many_outputs=beam,
                   beam_max_seq_len=12, sample_or_max="max", sos_idx=TSOS, eos_idx=TEOS)
    want_tok = [[[int(v) for v in row if v >= 0] for row in per] for per in store[f"{name}.beam{beam}_T12.tokens"]]
    assert pred == want_tok
    np.testing.assert_allclose(lp.cpu().numpy(), store[f"{name}.beam{beam}_T12.logprobs"], atol=1e-3)
    with pytest.raises(AssertionError):
        ens(enc_x=img, enc_x_num_pads=[0] * 3, mode="forward")


# ----------------------------------------------------------------------------------------- config 4 shape
@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_long_beam5_search_matches_reference_tiny(variant):
    """beam 5 / beam 3 at beam_max_seq_len = max_seq_len (every position of the pos-encoder table is used)."""
    store = np.load(os.path.join(GOLDEN, "tiny_long.npz"))
    g = W.TINY
    m = build_model("TINY", variant)
    img = W.synth_images(3, g).to(DEV)
    for k in (5, 3):
        pred, lp = m(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=k, how_many_outputs=2,
                     beam_max_seq_len=g.max_seq_len, sample_or_max="max", sos_idx=TSOS, eos_idx=TEOS)
        assert pred == unpad(store[f"{variant}.beam{k}_T{g.max_seq_len}.tokens"])
        np.testing.assert_allclose(lp.cpu().numpy(), store[f"{variant}.beam{k}_T{g.max_seq_len}.logprobs"], atol=1e-3)


@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_full_geometry_beam5_T74_matches_reference(variant):
    """BASELINE config 4 per-image shape: Swin-L/384, beam 5, beam_max_seq_len 74 (demo.py:21), fp32 mode —
    direct call and the hipGraph pipeline (early stop through the device `done` flag on the eos checkpoint)."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    store = np.load(os.path.join(GOLDEN, "full_long.npz"))
    g = W.FULL
    m = build_model("FULL", variant)
    img = W.synth_images(2, g).to(DEV)
    pred, lp = m(enc_x=img, enc_x_num_pads=[0] * 2, mode="beam_search", beam_size=5, how_many_outputs=2,
                 beam_max_seq_len=74, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
    want = unpad(store[f"{variant}.beam5_T74.tokens"])
    assert pred == want
    np.testing.assert_allclose(lp.cpu().numpy(), store[f"{variant}.beam5_T74.logprobs"], atol=2e-3)
    pipe = CaptionPipeline(m, 2, 5, 74, SOS, EOS, done_poll=4)
    assert pipe(img) == [per[0] for per in want]
    assert pipe(img.flip(0).contiguous()) == [per[0] for per in want][::-1]


# ----------------------------------------------------------------------------------------- bench shape
def _bench_batches(n, g):
    return [W.synth_images(16, g, seed=3000 + i).to(DEV) for i in range(n)]


def _drain(pipe, batches):
    caps = []
    for b in batches:
        while pipe.full():
            caps += pipe.collect()
        pipe.submit(b)
    while pipe.outstanding():
        caps += pipe.collect()
    return caps


def _direct(m, b, k=3, T=20):
    toks, _ = m(enc_x=b, enc_x_num_pads=[0] * b.shape[0], mode="beam_search", beam_size=k, how_many_outputs=1,
                beam_max_seq_len=T, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
    return [t[0] for t in toks]


@pytest.mark.parametrize("precision", ["bf16", "x3"])
def test_bench_shape_pipeline_logprobs_equal_the_direct_call_bit_for_bit(precision):
    """The hipGraph pipeline (three batches in flight, two decode lanes beside the encode graph) against the un-pipelined
    call on the SAME kernels: tokens AND every per-token log-prob of every caption identical BIT FOR BIT, over 12 sweeps
    of four distinct batches in both orders (3,072 full-length captions, 61,440 log-probs; xavier weights never emit EOS).  Round 2 guarded the wrong-attention-row
    incident with a 150-sweep caption comparison, which sees a 1e-3 log-prob error only where it flips a near-tie; a
    bitwise log-prob comparison sees every occurrence.  The incident itself is closed at its source: the instruction form
    behind it (v_pk_fma_f32 with op_sel source selection) is absent from the library
    (tests/test_isa_lint.py) and reproduced in isolation by tools/pkfma_probe.py."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.FULL
    m = build_model("FULL", "xavier", precision)
    batches = _bench_batches(4, g)
    want = []
    for b in batches:
        toks, lps = m(enc_x=b, enc_x_num_pads=[0] * 16, mode="beam_search", beam_size=3, how_many_outputs=1,
                      beam_max_seq_len=20, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
        want.append(([t[0] for t in toks], [lps[i, 0, :len(t[0])].cpu() for i, t in enumerate(toks)]))
    pipe = CaptionPipeline(m, 16, 3, 20, SOS, EOS, keep_scores=True)
    bad_tok, bad_lp, n_lp = [], [], 0
    for s_ in range(12):
        order = [0, 1, 2, 3] if s_ % 2 == 0 else [3, 2, 1, 0]
        got = []
        for i in order:
            while pipe.full():
                got.append(pipe.collect_scored())
            pipe.submit(batches[i])
        while pipe.outstanding():
            got.append(pipe.collect_scored())
        for i, (caps, lps) in zip(order, got):
            if caps != want[i][0]:
                bad_tok.append((s_, i))
            for r, (a_, b_) in enumerate(zip(lps, want[i][1])):
                n_lp += a_.numel()
                if a_.shape != b_.shape or not torch.equal(a_.view(torch.int32), b_.view(torch.int32)):
                    bad_lp.append((s_, i, r))
    build_model("FULL", "xavier", "fp32")
    assert n_lp == 12 * 64 * 20
    assert not bad_tok and not bad_lp, f"captions differ at {bad_tok[:4]}, log-probs differ at {bad_lp[:8]}"


@pytest.mark.parametrize("variant", ["eos", "xavier"])
def test_bench_shape_bf16_pipeline_equals_direct_call(variant):
    """The configuration bench.py times — Swin-L, bf16, B=16, beam 3, T=20, hipGraphs, two decode lanes, the
    wave-priority encode kernels — with four distinct batches and three of them in flight: every caption equals
    the un-pipelined bf16 call's (exact).  A lane / K-V hand-off / result-ring race would show here."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.FULL
    m = build_model("FULL", variant, "bf16")
    batches = _bench_batches(4, g)
    pipe = CaptionPipeline(m, 16, 3, 20, SOS, EOS)
    assert pipe.D == 2 and pipe.g_enc is not None and all(gs is not None for gs in pipe.g_step)
    for b in batches[:3]:
        pipe.submit(b)
    assert pipe.full() and pipe.outstanding() == 3
    got = pipe.collect()
    pipe.submit(batches[3])
    while pipe.outstanding():
        got += pipe.collect()
    got += _drain(pipe, batches[::-1])                       # a second sweep in another order through the same graphs
    want = [c for b in batches for c in _direct(m, b)]
    want += [c for b in batches[::-1] for c in _direct(m, b)]
    build_model("FULL", variant, "fp32")
    assert got == want


def test_bench_shape_cider_d_bf16_vs_fp32():
    """North-star bar "beam-3 captions within ±0.1 CIDEr-D": scored with cider.CiderD (pinned to the reference's
    scorer) over 256 synthetic images on the eos checkpoint, fp32-mode captions (= the reference's, token for
    token) as ground truth; README units (100 x compute_score).
      fp32 pipeline vs fp32 direct call : delta must be 0 (bar met);
      bf16 pipeline vs fp32             : measured and recorded; asserted only against a regression floor — on
                                          random weights a third of the captions diverge (module docstring)."""
    from on_device_image_captioning_amd.evaluation import caption_agreement
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.FULL
    batches = _bench_batches(16, g)
    m = build_model("FULL", "eos", "fp32")
    ref = _drain(CaptionPipeline(m, 16, 3, 20, SOS, EOS), batches)
    direct = [c for b in batches[:2] for c in _direct(m, b)]
    agree32 = caption_agreement(ref[:32], direct)
    assert agree32["cider_d_delta"] == 0.0 and agree32["identical"] == 1.0
    m = build_model("FULL", "eos", "bf16")
    got = _drain(CaptionPipeline(m, 16, 3, 20, SOS, EOS), batches)
    agree = caption_agreement(got, ref)
    _diag("bench_shape_cider_d_bf16_vs_fp32", agree)
    build_model("FULL", "eos", "fp32")
    assert agree["images"] == 256
    assert agree["identical"] >= 0.5 and agree["mean_prefix"] >= 0.7, agree       # regression floor, not the bar


# ----------------------------------------------------------------------------------------- small parity holes
def test_window_attention_fixtures_of_the_reference_through_hip():
    """`winattn_s{0..3}` in full_xavier.npz are outputs of the REFERENCE's WindowAttention.forward (qkv Linear →
    scaled QKᵀ + relative-position bias + SW-MSA mask → softmax → PV → proj Linear, swin_transformer_mod.py:183-214)
    on the first shifted block of every stage.  Replayed here through odic_gemm + odic_window_attention + odic_gemm
    (fp32): the fixture's window-partitioned input is scattered to the token-major layout the kernels read (the
    roll + window_partition index map), the output gathered back."""
    from on_device_image_captioning_amd import ops
    from oracle import expansionnet_ref as R
    g = W.FULL
    sd = cached_state_dict("FULL", "xavier")
    store = np.load(os.path.join(GOLDEN, "full_xavier.npz"))
    for s in range(4):
        p = f"swin_transf.layers.{s}.blocks.1"
        C, h, res, ws = g.stage_dim(s), g.swin_num_heads[s], g.stage_res(s), g.stage_window(s)
        shift = g.stage_shift(s, 1)
        nW = (res // ws) ** 2
        xin = W.synth_features(nW, 144, C, seed=100 + s)
        tok = R.window_token_index(res, ws, shift).reshape(-1)            # [nW·144] token of (window, slot)
        x_tok = torch.empty(res * res, C)
        x_tok[tok] = xin.reshape(-1, C)
        dv = lambda k: sd[k].to(DEV)                                       # noqa: E731
        qkv = ops.gemm(x_tok.to(DEV), dv(p + ".attn.qkv.weight"), dv(p + ".attn.qkv.bias"))
        att = ops.window_attention(qkv, dv(p + ".attn.relative_position_bias_table"), 1, res, C, h, ws, shift)
        out = ops.gemm(att, dv(p + ".attn.proj.weight"), dv(p + ".attn.proj.bias"))
        check_sample(store, f"winattn_s{s}", out.cpu()[tok].reshape(nW, 144, C), 2e-5)


def test_wide_search_beyond_the_folded_layernorm_limit():
    """B·beam = 200 decoder rows (> 192, the row limit of the folded-LayerNorm skinny GEMM): the step falls back
    to LayerNorm + GEMM and returns the oracle's captions (batch 64 x beam 5 used to raise)."""
    from oracle import expansionnet_ref as R
    g = W.TINY
    sd = cached_state_dict("TINY", "eos")
    m = build_model("TINY", "eos")
    base = W.synth_images(4, g)
    img = base[[i % 4 for i in range(40)]].contiguous()
    toks, lps = m(enc_x=img.to(DEV), enc_x_num_pads=[0] * 40, mode="beam_search", beam_size=5, how_many_outputs=1,
                  beam_max_seq_len=12, sample_or_max="max", sos_idx=TSOS, eos_idx=TEOS)
    want, wlps = R.beam_search(sd, g, base, [0] * 4, TSOS, TEOS, 5, 1, 12)
    assert toks == [want[i % 4] for i in range(40)]
    np.testing.assert_allclose(lps.cpu().numpy()[:4], wlps.numpy(), atol=1e-3)


def test_sampled_beam_search_bookkeeping_with_injected_draws():
    """sample_or_max='sample' (captioning_model.py:128-131,166-168): the candidate words are drawn on the device
    (odic_logsoftmax_sample).  The draws themselves cannot equal the reference's CPU multinomial; injected into the
    oracle's search they must give the SAME captions and log-probs, i.e. everything around the draw is exact."""
    from oracle import expansionnet_ref as R
    g = W.TINY
    sd = cached_state_dict("TINY", "eos")
    m = build_model("TINY", "eos")
    img = W.synth_images(3, g)
    k, T = 3, 12
    for seed in (0, 7):
        m.sampling_seed, m._draw_log = seed, []
        toks, lps = m(enc_x=img.to(DEV), enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=k, how_many_outputs=2,
                      beam_max_seq_len=T, sample_or_max="sample", sos_idx=TSOS, eos_idx=TEOS)
        draws, m._draw_log = m._draw_log, None
        assert all(d.shape == (3 * k, k) for d in draws)
        for d in draws:                                                     # without replacement
            assert all(len(set(row.tolist())) == k for row in d)

        def draw_fn(step, lp):
            d = draws[step].long()
            return d.view(3, k, k)[:, 0] if lp.shape[0] == 3 else d        # seeding uses beam 0's draws

        want, wlps = R.beam_search(sd, g, img, [0] * 3, TSOS, TEOS, k, 2, T, draw_fn=draw_fn)
        assert toks == want
        np.testing.assert_allclose(lps.cpu().numpy(), wlps.numpy(), atol=1e-3)


def test_demo_flow_with_a_host_resident_model(tmp_path):
    """demo.py:57-133 as written: vocabulary → model built with rank='cpu' and never moved → preprocess_image of
    a file → model(enc_x=<host tensor>, ..., mode='beam_search') → tokens2description.  The nn.Module stays on
    the host; the HIP engines run on cuda:0 and results come back on the input's device."""
    from PIL import Image
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import End_ExpansionNet_v2, make_drop_args
    from on_device_image_captioning_amd.image_utils import preprocess_image
    from on_device_image_captioning_amd.language_utils import load_vocab, tokens2description
    from oracle import expansionnet_ref as R
    g = W.TINY
    word2idx, idx2word = load_vocab()
    idx2word = idx2word[:g.vocab_size]
    word2idx = {w: i for i, w in enumerate(idx2word)}
    sos_idx, eos_idx = TSOS, TEOS
    model = End_ExpansionNet_v2(**g.model_kwargs(), output_word2idx=word2idx, output_idx2word=idx2word,
                                drop_args=make_drop_args(), rank="cpu")
    sd = cached_state_dict("TINY", "eos")
    model.load_state_dict(sd)
    rng = np.random.default_rng(5)
    path = str(tmp_path / "synthetic.jpg")
    Image.fromarray(rng.integers(0, 256, size=(300, 420, 3), dtype=np.uint8), "RGB").save(path, quality=95)
    image = preprocess_image(path, g.swin_img_size)
    assert image.device.type == "cpu" and next(model.parameters()).device.type == "cpu"
    kwargs = {"beam_size": 5, "beam_max_seq_len": 20, "sample_or_max": "max", "how_many_outputs": 1,
              "sos_idx": sos_idx, "eos_idx": eos_idx}
    with torch.no_grad():
        pred, lp = model(enc_x=image, enc_x_num_pads=[0], mode="beam_search", **kwargs)
    assert lp.device.type == "cpu"
    want, wlp = R.beam_search(sd, g, image, [0], sos_idx, eos_idx, 5, 1, 20)
    assert pred == want
    text = tokens2description(pred[0][0], idx2word, sos_idx, eos_idx)
    assert text == R.tokens2description(want[0][0], idx2word, sos_idx, eos_idx)
    assert text[0].isupper() and text.endswith(".")


# ----------------------------------------------------------------------------------------- F4: layer-removed variants
@pytest.mark.parametrize("cfg", [1, 2])
def test_layer_removed_variants_match_reference(cfg):
    """`--param_config 1` (N_enc = 2) and `2` (N_enc = N_dec = 2), reference test.py:360-365: model built with the
    reduced layer counts, the 3-layer checkpoint folded through checkpoint_utils.load_state_dict_filtered
    (test.py:38-77), captions against the REFERENCE's 2-layer classes on the same folded dict — end to end at the
    TINY geometry and features-only at the full captioner geometry with ragged encoder pads."""
    from dataclasses import replace
    from on_device_image_captioning_amd.checkpoint_utils import load_state_dict_filtered
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import End_ExpansionNet_v2, make_drop_args
    from on_device_image_captioning_amd.ExpansionNet_v2 import ExpansionNet_v2
    ne, nd, mode = {1: (2, 3, "enc"), 2: (2, 2, "dec")}[cfg]
    g3 = replace(W.TINY, N_enc=3, N_dec=3)
    g = replace(g3, N_enc=ne, N_dec=nd)
    ckpt = {"model_state_dict": W.synth_state_dict(g3, variant="eos", eos_idx=TEOS)}
    m = End_ExpansionNet_v2(**g.model_kwargs(), output_word2idx={i: i for i in range(g.vocab_size)},
                            output_idx2word=list(range(g.vocab_size)), drop_args=make_drop_args(), rank=DEV)
    load_state_dict_filtered(m, ckpt, mode)
    m.to(DEV).eval()
    store = np.load(os.path.join(GOLDEN, "tiny_variants.npz"))
    img = W.synth_images(3, g).to(DEV)
    check_sample(store, f"cfg{cfg}.enc_out", m.forward_enc(img, [0] * 3), 2e-4)
    for k, T in ((1, 12), (3, 16)):
        toks, lps = m(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=k, how_many_outputs=1,
                      beam_max_seq_len=T, sample_or_max="max", sos_idx=TSOS, eos_idx=TEOS)
        assert toks == unpad(store[f"cfg{cfg}.beam{k}_T{T}.tokens"])
        np.testing.assert_allclose(lps.cpu().numpy(), store[f"cfg{cfg}.beam{k}_T{T}.logprobs"], atol=1e-3)
    # features-only, full captioner geometry
    gf = replace(W.FULL, N_enc=ne, N_dec=nd)
    ckpt = {"model_state_dict": W.synth_state_dict(W.FULL, variant="xavier", end_to_end=False, img_feature_dim=1536)}
    mf = ExpansionNet_v2(d_model=gf.d_model, N_enc=ne, N_dec=nd, ff=gf.ff, num_heads=gf.num_heads,
                         num_exp_enc_list=list(gf.num_exp_enc_list), num_exp_dec=gf.num_exp_dec,
                         output_word2idx={i: i for i in range(gf.vocab_size)}, output_idx2word=list(range(gf.vocab_size)),
                         max_seq_len=gf.max_seq_len, drop_args=make_drop_args(), img_feature_dim=1536, rank=DEV)
    load_state_dict_filtered(mf, ckpt, mode)
    mf.to(DEV).eval()
    storef = np.load(os.path.join(GOLDEN, "full_variants.npz"))
    feats = W.synth_features(4, 144, 1536, seed=77).to(DEV)
    toks, lps = mf(enc_x=feats, enc_x_num_pads=[0, 5, 0, 17], mode="beam_search", beam_size=3, how_many_outputs=1,
                   beam_max_seq_len=20, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
    assert toks == unpad(storef[f"cfg{cfg}.beam3_T20.tokens"])
    np.testing.assert_allclose(lps.cpu().numpy(), storef[f"cfg{cfg}.beam3_T20.logprobs"], atol=2e-3)


@pytest.mark.parametrize("name", ["two", "three"])
def test_ensemble_on_the_graph_pipeline_matches_reference(name):
    """EsembleCaptioningModel through CaptionPipeline: every member's encode pass in the encode graph, every member's
    decoder step + odic_ensemble_logprobs + odic_topk_rows + odic_beam_step in ONE step graph per lane, two lanes,
    several batches in flight — best caption per image = the reference ensemble search's (tiny_ensemble.npz)."""
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import End_ExpansionNet_v2, make_drop_args
    from on_device_image_captioning_amd.ensemble_captioning_model import EsembleCaptioningModel
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    store = np.load(os.path.join(GOLDEN, "tiny_ensemble.npz"))
    g = W.TINY
    members = []
    for s_, v_ in zip(store[name + ".seeds"], store[name + ".variants"]):
        m = End_ExpansionNet_v2(**g.model_kwargs(), output_word2idx={i: i for i in range(g.vocab_size)},
                                output_idx2word=list(range(g.vocab_size)), drop_args=make_drop_args(), rank=DEV)
        m.load_state_dict(W.synth_state_dict(g, seed=int(s_), variant=str(v_), eos_idx=TEOS), strict=True)
        members.append(m.to(DEV).eval())
    ens = EsembleCaptioningModel(members, DEV)
    img = W.synth_images(3, g).to(DEV)
    want = [per[0] for per in unpad(store[f"{name}.beam3_T12.tokens"])]
    pipe = CaptionPipeline(ens, 3, 3, 12, TSOS, TEOS, done_poll=4)
    assert pipe.M == len(members)
    pipe.submit(img)
    pipe.submit(img.flip(0).contiguous())
    pipe.submit(img)
    assert pipe.collect() == want
    assert pipe.collect() == want[::-1]
    assert pipe.collect() == want
    assert pipe(img.flip(0).contiguous()) == want[::-1]


# ----------------------------------------------------------------------------------------- configs[4]: fp8 mode
def test_tiny64_fp8_backbone_close_to_oracle():
    """Low-precision backbone mode (fp8 MFMA for qkv / fc1 / fc2 with calibrated static scales, fp16 qkv / attention
    activations): backbone features against the fp32 oracle.  e4m3 keeps 3 mantissa bits (2^-4 relative rounding per
    operand), so the bar is 1.5e-1 of the feature scale — written here, measured value recorded."""
    from oracle import expansionnet_ref as R
    g = W.TINY64
    m = build_model("TINY64", "xavier", "fp8")
    img = W.synth_images(3, g)
    want = R.swin_forward(cached_state_dict("TINY64", "xavier"), g, img)
    swin = m._engines()[0]
    assert swin.fp8_ready
    feats = swin.forward(img.to(DEV)).cpu()
    rel = (feats - want).abs().max().item() / want.abs().max().item()
    _diag("tiny64_fp8_swin_rel_err", rel)
    assert rel < 1.5e-1, rel
    build_model("TINY64", "xavier", "fp32")


def test_full_fp8_margin_aware_parity_and_pipeline():
    """configs[4] at the Swin-L geometry (xavier checkpoint, realistic logit scale): backbone features within 1.5e-1
    of the fp32 reference's (e4m3 operands: 2^-4 relative rounding, ten times bf16's), teacher-forced log-probs on the
    reference's greedy captions within 1 nat, arg-max identical wherever the reference's top-1/top-2 margin exceeds
    twice the local error, and the hipGraph pipeline (batch 16) returns exactly the un-pipelined fp8-mode captions."""
    from oracle import expansionnet_ref as R
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.FULL
    sd = cached_state_dict("FULL", "xavier")
    store = np.load(os.path.join(GOLDEN, "full_xavier.npz"))
    img = W.synth_images(2, g)
    ref_tok = unpad(store["beam1_T20.tokens"])
    T = max(len(r[0]) for r in ref_tok)
    dec = torch.full((2, T), EOS, dtype=torch.long)
    pads = []
    for b, r in enumerate(ref_tok):
        dec[b, :len(r[0])] = torch.tensor(r[0])
        pads.append(T - len(r[0]))
    feats_ref = R.swin_forward(sd, g, img)
    mem_ref = R.encoder_forward(sd, g, feats_ref, [0, 0])
    lp_ref = R.decoder_forward(sd, g, mem_ref, [0, 0], dec, pads, True)
    m = build_model("FULL", "xavier", "fp8")
    swin, cap = m._engines()
    feats = swin.forward(img.to(DEV))
    rel = (feats.cpu() - feats_ref).abs().max().item() / feats_ref.abs().max().item()
    lp = m.forward_dec(cap.encode(feats.to(cap.cdt), m._enc_lens(2, 144, None)), [0, 0], dec.to(DEV), pads, True).cpu()
    errs, flips, checked = [], 0, 0
    for b in range(2):
        n = T - pads[b]
        e = (lp[b, :n] - lp_ref[b, :n]).abs()
        top2 = torch.topk(lp_ref[b, :n], 2, -1).values
        margin = top2[:, 0] - top2[:, 1]
        errs.append(float(e.max()))
        for t in range(n):
            if margin[t] > 2 * float(e[t].max()) + 1e-3:
                checked += 1
                flips += int(lp[b, t].argmax() != lp_ref[b, t].argmax())
    _diag("full_fp8", dict(swin_rel_err=rel, max_logprob_err=errs, margin_checked=checked, flips=flips))
    assert rel < 1.5e-1 and max(errs) < 1.0 and flips == 0, (rel, errs, flips, checked)
    batches = _bench_batches(2, g)
    pipe = CaptionPipeline(m, 16, 3, 20, SOS, EOS)
    got = _drain(pipe, batches)
    want = [c for b in batches for c in _direct(m, b)]
    build_model("FULL", "xavier", "fp32")
    assert got == want


# ----------------------------------------------------------------------------------------- configs[4]: accuracy gate, calibration, B = 64
def _teacher_forced_agreement(m, precision, images, cal=None):
    """Teacher-forced comparison of `precision` against the fp32 mode ON THE fp32 MODE'S OWN greedy captions (= the
    reference's, token for token): per decoding position the arg-max agreement, whether the fp32 top-1 word is among the
    mode's top 3, and the log-prob error of the fp32 top-1 word; plus the backbone feature error."""
    n = images.shape[0]
    m.set_precision("fp32")
    toks, _ = m(enc_x=images, enc_x_num_pads=[0] * n, mode="beam_search", beam_size=1, how_many_outputs=1,
                beam_max_seq_len=20, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
    T = max(len(t[0]) for t in toks)
    dec = torch.full((n, T), EOS, dtype=torch.long)
    pads = []
    for i, t in enumerate(toks):
        dec[i, :len(t[0])] = torch.tensor(t[0])
        pads.append(T - len(t[0]))
    f32 = m._engines()[0].forward(images)
    lp32 = m.forward_dec(m.forward_enc(images, [0] * n), [0] * n, dec.to(DEV), pads, True).cpu()
    if cal is None:
        m.set_precision(precision)
    else:
        m.set_precision(precision, calibration_images=cal)
    fx = m._engines()[0].forward(images)
    lpx = m.forward_dec(m.forward_enc(images, [0] * n), [0] * n, dec.to(DEV), pads, True).cpu()
    agree = top3 = npos = 0
    dlp = []
    for i in range(n):
        for t in range(T - pads[i] - 1):                         # position t predicts token t + 1
            w = int(lp32[i, t].argmax())
            npos += 1
            agree += int(int(lpx[i, t].argmax()) == w)
            top3 += int(w in lpx[i, t].topk(3).indices.tolist())
            dlp.append(abs(float(lpx[i, t, w] - lp32[i, t, w])))
    return dict(positions=npos, argmax_agreement=agree / npos, ref_top1_in_top3=top3 / npos,
                mean_abs_logprob_err_of_ref_word=sum(dlp) / len(dlp), max_abs_logprob_err_of_ref_word=max(dlp),
                feature_rel_err=float((fx.float() - f32).abs().max() / f32.abs().max()))


def test_full_fp8_accuracy_gate():
    """The accuracy gate of the low-precision mode: 16 synthetic images, xavier checkpoint,
    304 teacher-forced positions.  The bars are WRITTEN here and sit just outside the values measured when the block-scaled
    fp8 path was introduced (recorded in gpurun_out/parity_diag.json: `full_fp8_accuracy_gate`); a change that costs the
    mode accuracy fails this test.  (The same numbers for bf16 are recorded for scale, not gated here.)"""
    g = W.FULL
    m = build_model("FULL", "xavier", "fp32")
    images = W.synth_images(16, g, seed=4242).to(DEV)
    r8 = _teacher_forced_agreement(m, "fp8", images)
    r16 = _teacher_forced_agreement(m, "bf16", images)
    _diag("full_fp8_accuracy_gate", dict(fp8=r8, bf16=r16))
    build_model("FULL", "xavier", "fp32")
    assert r8["positions"] == 16 * 19
    # measured at introduction: feature error 8.4e-2, arg-max agreement 0.839, reference word in the top 3 0.977, log-prob
    # error of the reference word 0.018 mean / 0.17 max   (bf16 on the same positions: 8.4e-3, 0.987, 0.997, 0.0016 / 0.11)
    assert r8["feature_rel_err"] <= 1.0e-1, r8
    assert r8["argmax_agreement"] >= 0.78 and r8["ref_top1_in_top3"] >= 0.95, r8
    assert r8["mean_abs_logprob_err_of_ref_word"] <= 0.03 and r8["max_abs_logprob_err_of_ref_word"] <= 0.35, r8
    assert r16["argmax_agreement"] >= 0.96 and r16["feature_rel_err"] <= 1.2e-2, r16


def test_fp8_calibration_images_are_plumbed_and_clipping_is_reported():
    """the static activation scales were always calibrated on two synthetic noise images.  Now
    set_precision('fp8', calibration_images=...) carries the caller's sample to the engine (and invalidates the packed
    weights when it changes), and fp8_saturation_report() says which quantised tensors of a batch would clip.  Checked
    with a calibration set of ANOTHER distribution than the evaluation images: calibrated on flat grey images (LayerNorm
    outputs collapse to their bias), held-out noise images clip in most quantised tensors — the report says so and the
    feature error shows it; calibrated on images of the evaluation distribution nothing clips (on the calibration set
    itself observed amax / range = 1 / 1.25 by construction)."""
    g = W.FULL
    m = build_model("FULL", "xavier", "fp8")
    noise_cal, noise_eval = W.synth_images(2, g, seed=7), W.synth_images(2, g, seed=99)
    flat = torch.full((2, 3, g.swin_img_size, g.swin_img_size), 0.1)
    rep_default = m.fp8_saturation_report(noise_eval)             # default calibration = noise images
    assert rep_default["tensors"] == 24 * 3 and rep_default["clipping_tensors"] <= 2, rep_default
    m.set_precision("fp8", calibration_images=flat)
    assert m._eng_cache is None                                   # packed scales were invalidated
    rep_flat = m.fp8_saturation_report(noise_eval)
    f_flat = m._engines()[0].forward(noise_eval.to(DEV))
    m.set_precision("fp8", calibration_images=noise_cal)
    rep_own = m.fp8_saturation_report(noise_cal)
    assert rep_own["clipping_tensors"] == 0 and abs(rep_own["worst_ratio"] - 0.8) < 1e-3, rep_own
    rep_held = m.fp8_saturation_report(noise_eval)
    f_cal = m._engines()[0].forward(noise_eval.to(DEV))
    m.set_precision("fp32")
    f32 = m._engines()[0].forward(noise_eval.to(DEV))
    e_flat = float((f_flat - f32).abs().max() / f32.abs().max())
    e_cal = float((f_cal - f32).abs().max() / f32.abs().max())
    _diag("fp8_calibration", dict(default_calibration=rep_default, flat_calibration_on_noise=rep_flat,
                                  noise_calibration_held_out=rep_held, feature_err_flat_calibration=e_flat,
                                  feature_err_noise_calibration=e_cal))
    assert rep_flat["clipping_tensors"] >= 24 and rep_flat["worst_ratio"] > 1.5, rep_flat
    assert rep_held["clipping_tensors"] <= 2 and rep_held["worst_ratio"] < 1.25, rep_held
    # (clipping at 1.2-1.7x the range only touches the tails: the max feature error moves from 8.2e-2 to 8.5e-2 here; on
    #  data whose ranges differ by more than these synthetic sets the cast saturates further — the report is the guard)
    assert e_cal <= 1.0e-1 and e_flat <= 2.0e-1, (e_cal, e_flat)
    with pytest.raises(ValueError):
        m.set_precision("bf16", calibration_images=flat)
    build_model("FULL", "xavier", "fp32")


def test_fp8_b64_pipeline_equals_direct_call():
    """BASELINE.json configs[4] at its own batch: 64 images x beam 3 = 192 decoder rows — exactly the folded-LayerNorm
    skinny-GEMM limit (engine.step_logits) — through the hipGraph pipeline in the fp8 mode: captions equal the direct call."""
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g = W.FULL
    m = build_model("FULL", "xavier", "fp8")
    batches = [W.synth_images(64, g, seed=7000 + i).to(DEV) for i in range(2)]
    pipe = CaptionPipeline(m, 64, 3, 20, SOS, EOS)
    assert pipe.states[0].N == 192
    got = _drain(pipe, batches + batches[::-1])
    want = [c for b in batches + batches[::-1] for c in _direct(m, b)]
    build_model("FULL", "xavier", "fp32")
    assert got == want and len(got) == 256


@pytest.mark.parametrize("precision", ["bf16", "x3", "fp32"])
def test_image_chunked_stages_give_the_same_features_bit_for_bit(precision):
    """`SwinEngine.stage_chunks` (ODIC_SWIN_CHUNKS) runs a stage's blocks over B/n images at a time: every kernel is
    per-row / per-window, so the bf16 / x3 features must not change by a bit — whatever tile the tuner picks for the smaller M."""
    from on_device_image_captioning_amd.engine import SwinEngine
    g = W.FULL
    eng = SwinEngine(cached_state_dict("FULL", "xavier"), g, torch.device(DEV), precision)
    img = W.synth_images(4, g, seed=77).to(DEV)
    eng.stage_chunks = [1]
    ref = eng.forward(img).clone()
    for chunks in ([2, 2], [4, 2, 2], [4, 4, 4, 4], [3, 1]):        # 3 does not divide 4: that stage stays whole
        eng.stage_chunks = chunks
        out = eng.forward(img)
        if precision == "fp32":      # the fp32 MFMA GEMM's tile configurations differ in their K order: summation-order noise only
            assert float((out - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (chunks, float((out - ref).abs().max()))
        else:
            assert torch.equal(out, ref), (precision, chunks, float((out - ref).abs().max()))


def test_fused_stage0_launches_leave_the_backbone_features_unchanged():
    """The round-3 fusions of the width-192 stage at the engine level (Swin-L, B = 2): norm1 → qkv → attention core as ONE
    launch must not change the bf16 backbone features by a bit against LayerNorm-while-reading product + attention core
    (the kernels agree bit for bit: tests/test_hip_ops.py::test_swin_qkv_attention_fused); LayerNorm-while-reading against
    the separate odic_layernorm launches changes the rounding pattern of the normalised rows only (gamma folded into the
    bf16 weights): features within the bf16 mode's own error against fp32."""
    from on_device_image_captioning_amd.engine import SwinEngine
    g = W.FULL
    sd = cached_state_dict("FULL", "xavier")
    img = W.synth_images(2, g, seed=91).to(DEV)
    eng = SwinEngine(sd, g, torch.device(DEV), "bf16")
    assert eng.ln_read and eng.fuse_qkv_attn and "qkv_lnr" in eng.stages[0][0][0] and "qkv_lnr" not in eng.stages[1][0][0]
    fused = eng.forward(img).clone()
    eng.fuse_qkv_attn = False
    two = eng.forward(img).clone()
    assert torch.equal(fused, two), float((fused - two).abs().max())
    saved = [(w.pop("qkv_lnr"), w.pop("fc1_lnr")) for w in eng.stages[0][0]]
    plain = eng.forward(img).clone()
    for w, (a, b) in zip(eng.stages[0][0], saved):
        w["qkv_lnr"], w["fc1_lnr"] = a, b
    ref = SwinEngine(sd, g, torch.device(DEV), "fp32").forward(img)
    scale = float(ref.abs().max())
    e_fused, e_plain = float((fused - ref).abs().max()) / scale, float((plain - ref).abs().max()) / scale
    _diag("stage0_fusion_feature_err_vs_fp32", {"fused": e_fused, "separate_layernorm": e_plain})
    assert e_fused <= 1.5 * e_plain + 1e-3 and e_fused <= 3e-2, (e_fused, e_plain)
