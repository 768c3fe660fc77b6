"""The near-exact fast mode (`precision='x3'`): split-fp16 operands (hi + lo fp16 pairs, 22 significand bits), three
fp16 MFMAs per product, fp32 accumulation — kernel parity against fp64 and end-to-end parity against the fp32 mode
(whose token ids equal the reference's, tests/golden).

Tolerances (written where they are asserted): a split operand carries 2^-22 relative error per element and the dropped
lo·lo term is 2^-22 of the product, so a length-K contraction of O(1)·O(w) terms is expected within a few 1e-6 of the
fp64 result relative to sqrt(K)·|a|·|w| — the same class as the fp32 MFMA path's summation-order noise (2e-5 bound).
"""
import math

import pytest
import torch

from on_device_image_captioning_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from on_device_image_captioning_amd import _hip, ops as o
    _hip.load()
    return o


def dev(t):
    return t.to("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel_err(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-300))


# ------------------------------------------------------------------------------------------ the format itself
def test_h2_pack_roundtrip_and_device_cast(ops):
    x = rnd(37, 96, seed=1, scale=3.0)
    x[0, :8] = torch.tensor([0.0, 1e-7, -1e-7, 65504.0, -70000.0, 1.0, 2.0 ** -14, 3.14159265])
    h = ops.h2_from_f32(x)
    assert h.dtype == ops.H2_DTYPE and h.shape == x.shape
    back = ops.h2_to_f32(h)
    xc = x.clamp(-65504, 65504)
    # 22 significand bits: |x − (hi + lo)| <= 2^-22·|x| (+ one fp16 subnormal quantum 2^-25 for tiny lo parts)
    assert float(((back - xc).abs() - xc.abs() * 2.0 ** -22).max()) <= 2.0 ** -25
    got = ops.cast_h2(dev(x))                                       # the device kernel writes the same bytes
    assert torch.equal(got.cpu(), h)
    assert torch.equal(ops.h2_from_f32(torch.zeros(4, 16)), torch.zeros(4, 16, dtype=torch.int32))   # zero bytes = 0


def test_layernorm_and_patch_merge_write_h2(ops):
    M, C = 37, 768
    x, g, b = rnd(M, C, seed=1, scale=3.0) + 0.7, 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    ref32 = ops.layernorm(dev(x), dev(g), dev(b))                    # the fp32 kernel's values, split afterwards
    got = ops.layernorm(dev(x), dev(g), dev(b), out_dtype=ops.H2_DTYPE)
    assert got.dtype == ops.H2_DTYPE
    assert torch.equal(got.cpu(), ops.h2_from_f32(ref32.cpu()))
    B, res, Cin = 2, 24, 96
    xm = rnd(B, res * res, Cin, seed=5)
    gm, bm = 1 + 0.1 * rnd(4 * Cin, seed=2), 0.1 * rnd(4 * Cin, seed=3)
    r32 = ops.patch_merge_layernorm(dev(xm), dev(gm), dev(bm), B, res, Cin)
    rh2 = ops.patch_merge_layernorm(dev(xm), dev(gm), dev(bm), B, res, Cin, out_dtype=ops.H2_DTYPE)
    assert torch.equal(rh2.cpu(), ops.h2_from_f32(r32.cpu()))


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_gemm_x3_identity_asymmetric(ops, cfg):
    """A = I with an ASYMMETRIC integer W (values whose hi AND lo halves are non-zero: |w| up to 40000 needs more than
    fp16's 11 bits): the output must equal Wᵀ exactly — catches any fragment, plane-order, swizzle or C-layout slip."""
    K = 256
    A = torch.eye(K)
    Wt = ((torch.arange(192 * K).reshape(192, K) * 7919) % 80001 - 40000).float()
    got = ops.gemm(ops.h2_from_f32(A).cuda(), ops.h2_from_f32(Wt).cuda(), out_dtype=torch.float32, tile_cfg=cfg)
    assert torch.equal(got.cpu(), Wt.T.contiguous())
    got2 = ops.gemm(ops.h2_from_f32(Wt).cuda(), ops.h2_from_f32(A).cuda(), out_dtype=torch.float32, tile_cfg=cfg)
    assert torch.equal(got2.cpu(), Wt)


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 192, 192), (144, 576, 160), (2304, 1536, 1536), (517, 264, 992)])
def test_gemm_x3_random_against_fp64(ops, M, N, K, cfg):
    A, Wt = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    b, r = rnd(N, seed=3), rnd(M, N, seed=4)
    want = A.double() @ Wt.double().T + b.double() + r.double()
    sc = ops.pow2_scale_for_h2(Wt)
    got = ops.gemm(ops.cast_h2(dev(A)), ops.h2_from_f32(Wt * sc).cuda(), dev(b), dev(r), out_dtype=torch.float32,
                   alpha=1.0 / sc, tile_cfg=cfg)
    # error budget: 3 roundings of 2^-22 per term, random signs → ~2^-22·sqrt(K)·|a||w| ≈ 1e-7·sqrt(K)·0.05; the bound
    # below (1e-6 of the output scale) is ~10x that and 20x tighter than the fp32 path's 2e-5
    assert rel_err(got, want) <= 1e-6, rel_err(got, want)
    # h2 output: the same values split again
    got_h2 = ops.gemm(ops.cast_h2(dev(A)), ops.h2_from_f32(Wt * sc).cuda(), dev(b), dev(r), alpha=1.0 / sc, tile_cfg=cfg)
    assert got_h2.dtype == ops.H2_DTYPE
    assert torch.equal(got_h2.cpu(), ops.h2_from_f32(got.cpu()))


@pytest.mark.parametrize("cfg,bm", [(20, 128), (21, 64)])
def test_gemm_x3_a_resident_kernels_and_layernorm_while_reading(ops, cfg, bm):
    """Tile configs 20 / 21 (gemm_x3_apanel_kernel, K = 192): bit-for-bit the tiled kernel's results on split-fp16 A (same
    MFMAs in the same order), every epilogue of the stage-0 products, fp32 and h2 outputs; then the LayerNorm-while-reading
    form (a_ln = fp32 rows, A = None) against fp64 LayerNorm → Linear at the fp32-class error of this mode, on rows with
    mean >> spread and a constant row; shapes that are not whole tiles are refused."""
    g = torch.Generator().manual_seed(cfg)
    K = 192
    for M, N in ((bm * 3, 32 * 5), (bm * 9, 32 * 18)):
        A, Wt = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
        b = torch.randn(N, generator=g)
        sc = ops.pow2_scale_for_h2(Wt)
        Ah, Wh = ops.cast_h2(dev(A)), ops.h2_from_f32(Wt * sc).cuda()
        for kw in (dict(out_dtype=torch.float32), dict(bias=dev(b), act=1), dict(bias=dev(b), out_dtype=torch.float32, act=2)):
            got = ops.gemm(Ah, Wh, alpha=1.0 / sc, tile_cfg=cfg, **kw)
            ref = ops.gemm(Ah, Wh, alpha=1.0 / sc, tile_cfg=0, **kw)
            assert got.dtype == ref.dtype and torch.equal(got, ref), (cfg, M, N, kw.keys())
    M, N = bm * 5, 32 * 12
    # (fp32 LayerNorm: the mean carries ~1 ulp of |mean|, so the normalised row is good to ulp(mean) / spread — rows with
    #  mean / spread up to ~10 keep the mode's 1e-6 class; the constant row must come out as exactly beta·Wᵀ + b)
    x = torch.randn(M, K, generator=g) * (0.5 + 3.0 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g) * 4.0
    x[3] = 5.0
    W, b = torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    want = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double()
    Wg, b2, _ = ops.fold_layernorm(dev(W), dev(b), dev(gamma), dev(beta))
    sc = ops.pow2_scale_for_h2(Wg)
    Wh = ops.h2_from_f32(Wg.cpu() * sc).cuda()
    got = ops.gemm(None, Wh, b2, a_ln=dev(x), alpha=1.0 / sc, tile_cfg=cfg, out_dtype=torch.float32)
    # the normalised rows carry the fp32 rounding of mean / rstd (x has mean 40, spread 1: ~1e-6 relative), then 22-bit operands
    assert rel_err(got, want) <= 4e-6, rel_err(got, want)
    assert torch.isfinite(got).all()
    got_h2 = ops.gemm(None, Wh, b2, a_ln=dev(x), alpha=1.0 / sc, tile_cfg=cfg, act=1)
    assert got_h2.dtype == ops.H2_DTYPE
    assert rel_err(ops.h2_to_f32(got_h2.cpu()), torch.nn.functional.gelu(want)) <= 4e-6
    with pytest.raises(RuntimeError):
        ops.gemm(None, Wh, b2, a_ln=dev(x[: bm + 16].contiguous()), alpha=1.0 / sc, tile_cfg=cfg)
    with pytest.raises(RuntimeError):
        ops.gemm(None, Wh, b2, a_ln=dev(x), alpha=1.0 / sc, tile_cfg=1)


def test_gemm_x3_beats_bf16_and_matches_fp32_class(ops):
    """The point of the mode, measured: on the same operands the split-fp16 product is > 1000x closer to fp64 than the
    bf16 product and within a factor of a few of the exact fp32 MFMA product."""
    M, N, K = 512, 512, 1536
    A, Wt = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.03)
    want = A.double() @ Wt.double().T
    e32 = rel_err(ops.gemm(dev(A), dev(Wt)), want)
    e16 = rel_err(ops.gemm(dev(A).bfloat16(), dev(Wt).bfloat16(), out_dtype=torch.float32), want)
    sc = ops.pow2_scale_for_h2(Wt)
    ex3 = rel_err(ops.gemm(ops.cast_h2(dev(A)), ops.h2_from_f32(Wt * sc).cuda(), out_dtype=torch.float32, alpha=1.0 / sc), want)
    assert ex3 * 1000 < e16, (ex3, e16)
    assert ex3 < 8 * e32 + 1e-7, (ex3, e32)


def test_gemm_x3_small_magnitudes_keep_their_low_halves(ops):
    """fp16 subnormal lo halves must not be flushed by the MFMA: operands of magnitude 2^-6 (lo halves < 2^-17, all
    subnormal) still give the 22-bit result (error far below the 2^-11 of a hi-only product)."""
    M, N, K = 128, 128, 256
    A, Wt = rnd(M, K, seed=1, scale=2.0 ** -6), rnd(N, K, seed=2, scale=2.0 ** -6)
    want = A.double() @ Wt.double().T
    got = ops.gemm(ops.cast_h2(dev(A)), ops.cast_h2(dev(Wt)), out_dtype=torch.float32)
    hi_only = (A.half().double() @ Wt.half().double().T)
    assert rel_err(hi_only, want) > 5e-5                              # what a flushed lo half would leave
    assert rel_err(got, want) < 5e-6, rel_err(got, want)


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_x3_epilogues_and_batched_strided(ops, act):
    M, N, K = 130, 96, 64
    A, Wt, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    pre = 0.5 * (A.double() @ Wt.double().T) + b.double()
    f = [lambda v: v, lambda v: torch.nn.functional.gelu(v), torch.relu, torch.sigmoid][act]
    want = f(pre) + r.double()
    got = ops.gemm(ops.cast_h2(dev(A)), ops.cast_h2(dev(Wt)), dev(b), dev(r), act=act, alpha=0.5, out_dtype=torch.float32)
    assert rel_err(got, want) <= 2e-6, (act, rel_err(got, want))     # exact-erf GELU, not the bf16 mode's polynomial
    # out[b] = A·W[b]ᵀ + bias[m]: A shared, W a strided sub-matrix (pointer 8 columns in), h2 output with padded rows
    Bn, M2, N2, K2, ldw, ldc = 3, 70, 48, 64, 128, 64
    A2, Wfull, bias = rnd(M2, K2, seed=1), rnd(Bn, N2, ldw, seed=2), rnd(M2, seed=3)
    Wh = ops.cast_h2(dev(Wfull).view(-1, ldw)).view(Bn, N2, ldw)
    out = torch.zeros(Bn, M2, ldc, dtype=ops.H2_DTYPE, device="cuda")
    ops.gemm(ops.cast_h2(dev(A2)), Wh[:, :, 8:], dev(bias), out=out, bias_axis=1, M=M2, N=N2, K=K2, lda=K2, ldw=ldw,
             ldc=ldc, batch=Bn, strideA=0, strideW=N2 * ldw, strideC=M2 * ldc)
    want2 = torch.einsum("mk,bnk->bmn", A2.double(), Wfull[:, :, 8:8 + K2].double()) + bias.double()[None, :, None]
    got2 = ops.h2_to_f32(out.cpu())
    assert rel_err(got2[:, :, :N2], want2) <= 2e-6
    assert float(got2[:, :, N2:].abs().max()) == 0.0, "padding columns must stay untouched"


def test_gemm_x3_rejects_what_it_cannot_do(ops):
    a = torch.zeros(64, 48, dtype=ops.H2_DTYPE, device="cuda")       # K = 48 is not a multiple of 32
    with pytest.raises(RuntimeError):
        ops.gemm(a, a.clone(), out_dtype=torch.float32)
    a = torch.zeros(64, 64, dtype=ops.H2_DTYPE, device="cuda")
    with pytest.raises(RuntimeError):                                 # bf16 outputs do not exist in this mode
        ops.gemm(a, a.clone(), out_dtype=torch.bfloat16)


# ------------------------------------------------------------------------------------------ window attention
@pytest.mark.parametrize("res,heads,shift", [(96, 3, 0), (96, 3, 6), (48, 6, 6), (24, 12, 6), (24, 12, 0), (12, 48, 0)])
def test_window_attention_x3(ops, res, heads, shift):
    from tests.test_hip_ops import _win_ref
    B, ws = 2, 12
    C = heads * 32
    qkv = rnd(B * res * res, 3 * C, seed=res + shift, scale=1.5)
    table = rnd(529, heads, seed=9, scale=0.5)
    want = _win_ref(qkv, table, B, res, C, heads, ws, shift)
    got = ops.window_attention(ops.cast_h2(dev(qkv)), dev(table), B, res, C, heads, ws, shift,
                               bias_shifted_prescaled=ops.shifted_bias_prescaled(dev(table), ws, 32 ** -0.5))
    assert got.dtype == ops.H2_DTYPE
    f32 = ops.window_attention(dev(qkv), dev(table), B, res, C, heads, ws, shift)      # the exact-fp32 kernel
    e_x3, e_32 = rel_err(ops.h2_to_f32(got.cpu()), want), rel_err(f32, want)
    # fp32-class: v_exp_f32 (1 ulp) and the packed-bias pre-division dominate both kernels
    assert e_x3 <= 4e-6, (e_x3, e_32)


# ------------------------------------------------------------------------------------------ encoder glue
def test_stcexp_normalize_h2_outputs_with_scales(ops):
    B, S, groups = 3, 20, (8, 16, 24)
    nq = sum(groups)
    z = rnd(B, nq, S, seed=3)
    lens = torch.tensor([20, 13, 17], dtype=torch.int32)
    d = "cuda"
    meta = ops.stcexp_group_meta(groups, d)
    ws = torch.empty(B * len(groups) * 2 * S, device=d)
    o32 = [torch.empty(B, nq, S, device=d), torch.empty(B, nq, S, device=d), torch.empty(B, S, nq, device=d),
           torch.empty(B, S, nq, device=d)]
    ops.stcexp_normalize(dev(z), dev(lens), meta, len(groups), *o32, ws)
    oh = [torch.full((B, nq, 32), 7, device=d, dtype=ops.H2_DTYPE) for _ in range(2)] + \
         [torch.full((B, S, 64), 7, device=d, dtype=ops.H2_DTYPE) for _ in range(2)]
    ops.stcexp_normalize(dev(z), dev(lens), meta, len(groups), *oh, ws, scale_fw=256.0, scale_bw=4096.0)
    for i, (sc, n) in enumerate(((256.0, S), (256.0, S), (4096.0, nq), (4096.0, nq))):
        got = ops.h2_to_f32(oh[i].cpu())
        assert rel_err(got[..., :n] / sc, o32[i]) <= 3e-7, i        # power-of-two scale: the same values to 2^-22
        assert float(got[..., n:].abs().max()) == 0.0               # zero-filled K padding


# ========================================================================================== end to end
import os  # noqa: E402

import numpy as np  # noqa: E402

from conftest import GOLDEN  # noqa: E402

SOS, EOS = 79, 77
DEV = "cuda:0"


def _e2e():
    import tests.test_e2e_gpu as E
    return E


@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_tiny_x3_matches_reference_fixtures(variant):
    """TINY geometry through the split-fp16 path: every backbone tap, the Swin output and the encoder output within the
    SAME 2e-4 bound the exact-fp32 mode is held to against the real reference's recorded activations, and greedy /
    beam-3 / beam-5 token ids identical to the reference's."""
    E = _e2e()
    g = W.TINY
    m = E.build_model("TINY", variant, "x3")
    store = np.load(os.path.join(GOLDEN, f"tiny_{variant}.npz"))
    img = W.synth_images(3, g).to(DEV)
    swin, cap = m._engines()
    assert swin.precision == "x3" and cap.precision == "x3"
    taps = {}
    feats = swin.forward(img, taps)
    for name, t in taps.items():
        E.check_sample(store, name, t, 2e-4)
    E.check_sample(store, "swin_out", feats, 2e-4)
    E.check_sample(store, "enc_out", m.forward_enc(img, [0] * 3), 2e-4)
    for k, T in ((1, 12), (3, 12), (5, 20)):
        toks, lps = m(enc_x=img, enc_x_num_pads=[0] * 3, mode="beam_search", beam_size=k, how_many_outputs=min(k, 2),
                      beam_max_seq_len=T, sample_or_max="max", sos_idx=E.TSOS, eos_idx=E.TEOS)
        assert toks == E.unpad(store[f"beam{k}_T{T}.tokens"]), (k, T)
        np.testing.assert_allclose(lps.cpu().numpy(), store[f"beam{k}_T{T}.logprobs"], atol=1e-3)
    E.build_model("TINY", variant, "fp32")


@pytest.mark.parametrize("variant", ["xavier", "eos"])
def test_full_x3_matches_reference_fixtures(variant):
    """Swin-L/384 geometry: taps / features / encoder output against the reference's recorded activations at the fp32
    mode's 5e-4 bound, teacher-forced log-probs 1e-3, greedy + beam-3 + beam-5 token ids identical to the reference."""
    E = _e2e()
    g = W.FULL
    m = E.build_model("FULL", variant, "x3")
    store = np.load(os.path.join(GOLDEN, f"full_{variant}.npz"))
    img = W.synth_images(2, g).to(DEV)
    swin, cap = m._engines()
    if variant == "xavier":
        taps = {}
        feats = swin.forward(img, taps)
        for name, t in taps.items():
            if name + ".meta" in store:
                E.check_sample(store, name, t, 5e-4)
        E.check_sample(store, "swin_out", feats, 5e-4)
    mem = m.forward_enc(img, [0, 0])
    if variant == "xavier":
        E.check_sample(store, "enc_out", mem, 5e-4)
    for k in (1, 3, 5):
        toks, lps = m._search_from_memory(mem, [0, 0], SOS, EOS, k, 1, 20)
        assert toks == E.unpad(store[f"beam{k}_T20.tokens"]), f"beam {k}"
        np.testing.assert_allclose(lps.cpu().numpy(), store[f"beam{k}_T20.logprobs"], atol=2e-3)
    E.build_model("FULL", variant, "fp32")


def test_full_x3_feature_error_is_fp32_class():
    """Backbone features of the split-fp16 mode against the exact-fp32 mode on the same images: relative error below
    2e-5 (bf16: 7e-3) — measured and recorded in gpurun_out/parity_diag.json."""
    E = _e2e()
    g = W.FULL
    img = W.synth_images(2, g, seed=11).to(DEV)
    f32 = E.build_model("FULL", "xavier", "fp32")._engines()[0].forward(img)
    fx3 = E.build_model("FULL", "xavier", "x3")._engines()[0].forward(img)
    rel = float((fx3 - f32).abs().max() / f32.abs().max())
    E._diag("full_x3_feature_rel_err_vs_fp32", rel)
    E.build_model("FULL", "xavier", "fp32")
    assert rel < 2e-5, rel


@pytest.mark.parametrize("variant", ["eos", "xavier"])
def test_bench_shape_x3_captions_equal_fp32_captions(variant):
    """The parity bar of the north star at the bench shape (Swin-L, B=16, beam 3, T=20, hipGraph pipeline, 256 synthetic
    images), on BOTH synthetic checkpoints: the split-fp16 mode's captions against the fp32 mode's (= the reference's,
    token for token): >= 99 % identical and CIDEr-D delta <= 0.1 (README units); the pipelined captions also equal the
    un-pipelined x3 call's."""
    from on_device_image_captioning_amd.evaluation import caption_agreement
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    E = _e2e()
    g = W.FULL
    batches = E._bench_batches(16, g)
    m = E.build_model("FULL", variant, "fp32")
    ref = E._drain(CaptionPipeline(m, 16, 3, 20, SOS, EOS), batches)
    m = E.build_model("FULL", variant, "x3")
    pipe = CaptionPipeline(m, 16, 3, 20, SOS, EOS)
    got = E._drain(pipe, batches)
    direct = [c for b in batches[:2] for c in E._direct(m, b)]
    agree = caption_agreement(got, ref)
    E._diag(f"bench_shape_x3_vs_fp32_{variant}", agree)
    E.build_model("FULL", variant, "fp32")
    assert got[:32] == direct
    assert agree["images"] == 256
    assert agree["identical"] >= 0.99 and agree["cider_d_delta"] <= 0.1, agree


# ========================================================================================== device-side *pos guards
def test_step_kernels_replayed_past_the_last_position_change_nothing():
    """A C-ABI caller that replays a decoder step once too often (the recorded fault's class: *pos is device memory no
    host check can see): with *pos == T every kernel that indexes by it returns without touching caches, state or the
    input row — and with *pos beyond the position table odic_dec_embed writes nothing."""
    from on_device_image_captioning_amd import ops
    E = _e2e()
    g = W.TINY
    m = E.build_model("TINY", "eos", "fp32")
    cap = m._captioner_engine()
    img = W.synth_images(2, g).to(DEV)
    mem = m.forward_enc(img, [0, 0])
    kv = cap.project_kv(mem)
    T, k = 8, 3
    st = cap.new_state(2, k, T, kv, torch.full((2,), mem.shape[1], dtype=torch.int32, device=DEV))
    ops.beam_reset(st.beam_state, 2, k, T, E.TSOS, emb=st.emb)
    for _ in range(T - 1):
        cap.beam_step(st, E.TEOS)
    torch.cuda.synchronize()
    assert int(st.pos.item()) == T - 1

    def snapshot():
        torch.cuda.synchronize()
        c = [t.clone() for layer in st.caches for t in layer.values()]
        return c + [st.tokens.clone(), st.logprobs.clone(), st.anc.clone(), st.cumul.clone(), st.n_elem.clone(),
                    st.has_eos.clone(), st.row_valid.clone(), st.next_tok.clone(), st.pos.clone(), st.ycat.clone()]

    ncache = sum(len(layer) for layer in st.caches)
    before = snapshot()
    for _ in range(2):                      # two full steps past the last position (*pos stays T - 1: beam_step guards)
        cap.beam_step(st, E.TEOS)
    after = snapshot()
    # the beam state is frozen (the prefix is full); position T-1 is still a legal cache row, so the decoder kernels
    # may fill it — inside their buffers, nothing faults
    assert all(torch.equal(a, b) for a, b in zip(before[ncache:-1], after[ncache:-1]))
    before = after
    st.pos.fill_(T)                         # one past the caches: dynexp_step / dec_embed must return untouched
    poison = st.ycat.clone()
    cap.step_logits(st, embed=True)
    after2 = snapshot()
    assert all(torch.equal(a, b) for a, b in zip(before[:-2], after2[:-2]))      # caches and beam state
    st.pos.fill_(cap.pos_table.shape[0] + 5)
    st.ycat.copy_(poison)
    ops.dec_embed(st.next_tok, cap.embed, cap.pos_table, st.pos, st.ycat, st.ycat.shape[1], st.N, g.d_model, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(st.ycat, poison)
    st.pos.fill_(-3)                        # a corrupted counter: same story on the low side
    ops.dec_embed(st.next_tok, cap.embed, cap.pos_table, st.pos, st.ycat, st.ycat.shape[1], st.N, g.d_model, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(st.ycat, poison)
    cap.step_logits(st, embed=False)
    after3 = snapshot()
    assert all(torch.equal(a, b) for a, b in zip(before[:-2], after3[:-2]))
