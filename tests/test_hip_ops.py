"""Per-kernel parity: every C-ABI entry point against the CPU oracle / plain fp32 torch formulas.

`-m gpu` only.  Tolerances: fp32 kernels 2e-5 relative to the output scale (summation order only);
bf16 kernels are compared against an fp64 evaluation of the SAME bf16-rounded inputs, so the
tolerance covers accumulation order + the final bf16 rounding (2^-8 relative).
"""
import math

import numpy as np
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
    _hip.load()          # fail loudly if the extension is missing
    return o


def dev(t):
    return t.to("cuda")


DEFAULT_BF16_TILE_CFGS = [0, 1, 2, 7, 10, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49]        # what the default build of gemm_bf16.hip carries
                                                                          # (50-53, whole tiles only: tested on their own)


def _need_experimental_gemm():
    """Tile configurations 3-6, 8, 9, 11-33 and the LayerNorm fold exist in -DODIC_EXPERIMENTAL_GEMM builds only."""
    from on_device_image_captioning_amd import _hip
    if b"experimental-gemm" not in _hip.load().odic_build_info():
        pytest.skip("default build: experimental GEMM configurations are compiled out (make EXTRA=-DODIC_EXPERIMENTAL_GEMM)")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def assert_close(got, want, rtol, name=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= rtol * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e} (rtol {rtol})"


# ------------------------------------------------------------------------------------------ GEMM fp32
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (48, 512, 512), (300, 190, 52), (144, 1000, 37), (1, 10, 3),
                                   (2304, 512, 1536)])
def test_gemm_f32_shapes(ops, M, N, K):
    A, Wt, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    want = A.double() @ Wt.double().T + b.double()
    got = ops.gemm(dev(A), dev(Wt), dev(b))
    assert_close(got, want, 2e-5, "gemm_f32")


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_f32_epilogues(ops, act):
    M, N, K = 130, 96, 64
    A, Wt, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    pre = 0.5 * (A.double() @ Wt.double().T) + b.double()
    f = [lambda v: v, lambda v: torch.nn.functional.gelu(v), torch.relu, torch.sigmoid][act]
    want = f(pre) + r.double()
    got = ops.gemm(dev(A), dev(Wt), dev(b), dev(r), act=act, alpha=0.5)
    assert_close(got, want, 2e-5, f"gemm_f32 act{act}")


def test_gemm_f32_row_bias_strided_batched(ops):
    # out[b] = A · W[b]ᵀ + bias[m]  with A shared, W a strided sub-matrix, padded output rows
    Bn, M, N, K, ldw, ldc = 3, 70, 50, 40, 100, 64
    A, Wfull, bias = rnd(M, K, seed=1), rnd(Bn, N, ldw, seed=2), rnd(M, seed=3)
    out = torch.full((Bn, M, ldc), 7.0)
    dout = dev(out)
    ops.gemm(dev(A), dev(Wfull)[:, :, 8:], dev(bias), out=dout, bias_axis=1, M=M, N=N, K=K, lda=K, ldw=ldw,
             ldc=ldc, batch=Bn, strideA=0, strideW=N * ldw, strideC=M * ldc)
    # note: the W view starts 8 columns in → pointer offset handled by passing a view's data_ptr
    want = torch.einsum("mk,bnk->bmn", A.double(), Wfull[:, :, 8:8 + K].double()) + bias.double()[None, :, None]
    got = dout.cpu()
    assert_close(got[:, :, :N], want, 2e-5, "gemm_f32 batched")
    assert torch.all(got[:, :, N:] == 7.0), "padding columns must stay untouched"


@pytest.mark.parametrize("shape", [-1, 0, 1, 2])
@pytest.mark.parametrize("M,N,K,act", [(48, 200, 512, 0), (48, 2048, 512, 2), (144, 512, 512, 0), (17, 100, 64, 0),
                                       (100, 1000, 512, 0)])
def test_gemm_f32_folded_layernorm(ops, M, N, K, act, shape):
    # LayerNorm(A)·Wᵀ + b with A a strided slice of a wider buffer (the decoder's residual-stream layout),
    # LayerNorm folded into the product: W·diag(gamma), bias + W·beta, row sums (ops.fold_layernorm)
    lda = 3 * K
    buf, Wt, b = rnd(M, lda, seed=1, scale=2.0) + 0.3, rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    g, be = 1 + 0.1 * rnd(K, seed=4), 0.1 * rnd(K, seed=5)
    A = buf[:, K:2 * K]
    want = torch.nn.functional.layer_norm(A.double(), (K,), g.double(), be.double(), 1e-5) @ Wt.double().T + b.double()
    if act == 2:
        want = torch.relu(want)
    Wf, bf, cs = ops.fold_layernorm(dev(Wt), dev(b), dev(g), dev(be))
    dbuf = dev(buf)
    got = ops.gemm(dbuf[:, K:], Wf, bf, M=M, N=N, K=K, lda=lda, ldw=K, ldc=N, act=act, ln_fold=(cs, 1e-5),
                   tile_cfg=shape)
    assert_close(got, want, 3e-5, "folded LN gemm")


def test_gemm_f32_folded_layernorm_rejects_wide_M(ops):
    with pytest.raises(RuntimeError):          # not available outside the skinny fp32 path
        Wf, bf, cs = ops.fold_layernorm(dev(rnd(64, 64, seed=2)), None, dev(rnd(64)), dev(rnd(64)))
        ops.gemm(dev(rnd(400, 64, seed=1)), Wf, bf, ln_fold=(cs, 1e-5))


@pytest.mark.parametrize("shape", [-1, 0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(48, 512, 2048), (48, 512, 1536), (96, 512, 512), (150, 10000, 512), (192, 64, 4096),
                                   (40, 10000, 512), (3, 512, 80)])
def test_gemm_f32_skinny_wave_splits(ops, M, N, K, shape):
    # the three decompositions of the skinny kernel (K-slices x row tiles over waves / one row tile per block /
    # three row tiles per wave), with residual + relu epilogue
    A, Wt, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(M, N, seed=4)
    want = torch.relu(A.double() @ Wt.double().T + b.double()) + r.double()
    got = ops.gemm(dev(A), dev(Wt), dev(b), dev(r), act=2, tile_cfg=shape)
    assert_close(got, want, 2e-5, "gemm_f32 skinny")


def test_copy_kernel(ops):
    # odic_copy: the pipeline's K/V hand-off (any dtype, 16-byte multiples; misuse is rejected, not truncated)
    src = torch.randn(3, 144, 3072, device="cuda")
    dst = torch.zeros_like(src)
    assert ops.copy(src, dst) is dst and torch.equal(src, dst)
    s16 = torch.arange(4096, device="cuda", dtype=torch.int32).bfloat16()
    d16 = torch.zeros_like(s16)
    ops.copy(s16, d16)
    assert torch.equal(s16, d16)
    with pytest.raises(RuntimeError):
        ops.copy(src, torch.zeros(3, 144, 3071, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.copy(torch.zeros(6, device="cuda"), torch.zeros(6, device="cuda"))      # 24 bytes


# ------------------------------------------------------------------------------------------ GEMM bf16
def test_gemm_bf16_identity_asymmetric(ops):
    # A = I, asymmetric small-integer W: out must equal Wᵀ exactly (catches any fragment / C-layout swap)
    K = 128
    A = torch.eye(K)
    Wt = (torch.arange(192 * K).reshape(192, K) % 251 - 125).float()
    got = ops.gemm(dev(A).bfloat16(), dev(Wt).bfloat16(), out_dtype=torch.float32)
    assert torch.equal(got.cpu(), Wt.T.contiguous())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 192, 192), (144, 576, 192), (2304, 1536, 6144),
                                   (1000, 3072, 768)])
def test_gemm_bf16_random(ops, M, N, K):
    A, Wt, b, r = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16(), rnd(N, seed=3), rnd(M, N, seed=4)
    want = A.double() @ Wt.double().T + b.double()
    got32 = ops.gemm(dev(A), dev(Wt), dev(b), dev(r), out_dtype=torch.float32)
    assert_close(got32, want + r.double(), 2e-4, "gemm_bf16→f32")
    got16 = ops.gemm(dev(A), dev(Wt), dev(b), act=1, out_dtype=torch.bfloat16)
    assert_close(got16, torch.nn.functional.gelu(want), 6e-3, "gemm_bf16 gelu→bf16")


def test_gelu_poly_accuracy(ops):
    # the bf16 GEMM epilogue evaluates GELU as relu(x) - u·(0.5 - u·P(u²)) (odic_common.h::gelu_poly4);
    # feed exact values through an identity product and compare with the erf form over [-12, 12]
    K = 64
    x = torch.linspace(-12, 12, 64 * 4096).bfloat16().float().reshape(-1, K)
    got = ops.gemm(dev(x).bfloat16(), dev(torch.eye(K)).bfloat16(), act=1, out_dtype=torch.float32).cpu()
    want = torch.nn.functional.gelu(x.double())
    err = (got.double() - want).abs().max().item()
    assert err <= 6e-5, err


# ------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("C", [96, 192, 512, 768, 1536, 3072, 6144])
@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_layernorm(ops, C, odt):
    M = 37
    x, g, b = rnd(M, C, seed=1, scale=3.0) + 0.7, 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    want = torch.nn.functional.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)
    got = ops.layernorm(dev(x), dev(g), dev(b), out_dtype=odt)
    assert got.dtype == odt
    assert_close(got, want, 2e-5 if odt == torch.float32 else 5e-3, "layernorm")


def test_patch_merge_layernorm(ops):
    from oracle import expansionnet_ref as R
    B, res, C = 2, 24, 96
    x = rnd(B, res * res, C, seed=5)
    g, b = 1 + 0.1 * rnd(4 * C, seed=2), 0.1 * rnd(4 * C, seed=3)
    sd = {"m.norm.weight": g, "m.norm.bias": b, "m.reduction.weight": torch.eye(4 * C)}
    want = R.patch_merging(sd, "m", x, res)            # identity reduction → exposes gather + LN
    got = ops.patch_merge_layernorm(dev(x), dev(g), dev(b), B, res, C)
    assert_close(got, want, 2e-5, "patch_merge_ln")


def test_patch_embed(ops):
    from oracle import expansionnet_ref as R
    g = W.TINY
    sd = {k: W.synth_tensor(k, s, kind, g) for k, s, kind in W.state_dict_spec(g) if "patch_embed" in k}
    img = W.synth_images(2, g)
    want = R.patch_embed(sd, g, img)
    P = "swin_transf.patch_embed"
    got = ops.patch_embed(dev(img), dev(sd[P + ".proj.weight"].reshape(g.swin_embed_dim, -1).contiguous()),
                          dev(sd[P + ".proj.bias"]), dev(sd[P + ".norm.weight"]), dev(sd[P + ".norm.bias"]), 4)
    assert_close(got, want, 2e-5, "patch_embed")


# ------------------------------------------------------------------------------------------ window attention
def _win_ref(qkv, table, B, res, C, heads, ws, shift):
    from oracle import expansionnet_ref as R
    tok = R.window_token_index(res, ws, shift)
    nW, N = tok.shape
    x = qkv.view(B, res * res, 3, heads, 32)[:, tok.reshape(-1)].view(B, nW, N, 3, heads, 32)
    q, k, v = (x[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))
    idx = W.relative_position_index(ws)
    bias = table[idx.reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    mask = W.shifted_window_attn_mask(res, ws, shift) if shift > 0 else None
    o = R.window_attention_core(q.double(), k.double(), v.double(), bias.double(),
                                None if mask is None else mask.double(), 32 ** -0.5)
    o = o.permute(0, 1, 3, 2, 4).reshape(B, nW * N, C)
    out = torch.empty(B, res * res, C, dtype=torch.float64)
    out[:, tok.reshape(-1)] = o
    return out.reshape(B * res * res, C)


@pytest.mark.parametrize("res,heads,shift", [(96, 3, 0), (96, 3, 6), (48, 6, 6), (24, 12, 6), (24, 12, 0), (12, 48, 0)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_window_attention(ops, res, heads, shift, dt):
    B, ws = 2, 12
    C = heads * 32
    qkv = rnd(B * res * res, 3 * C, seed=res + shift, scale=1.5).to(dt)
    table = rnd(529, heads, seed=9, scale=0.5)
    want = _win_ref(qkv.float(), table, B, res, C, heads, ws, shift)
    got = ops.window_attention(dev(qkv), dev(table), B, res, C, heads, ws, shift)
    assert got.dtype == dt
    assert_close(got, want, 2e-5 if dt == torch.float32 else 1.2e-2, f"window_attention {dt}")
    if dt == torch.bfloat16:          # fast path: dense pre-gathered bias, LDS-DMA gathers, transposed LDS reads
        got2 = ops.window_attention(dev(qkv), dev(table), B, res, C, heads, ws, shift,
                                    bias_shifted_prescaled=ops.shifted_bias_prescaled(dev(table), ws, 32 ** -0.5))
        assert_close(got2, want, 1.2e-2, "window_attention bf16 v2")


# ------------------------------------------------------------------------------------------ encoder glue
def test_stcexp_normalize_and_mix(ops):
    B, S, groups = 3, 20, (8, 16, 24)
    nq = sum(groups)
    z = rnd(B, nq, S, seed=3)
    lens = torch.tensor([20, 13, 17], dtype=torch.int32)
    valid = (torch.arange(S)[None, :] < lens[:, None]).double()[:, None, :]
    zd = z.double()
    pf, nf = torch.relu(zd) * valid, torch.relu(-zd) * valid
    pf, nf = pf / (pf.sum(-1, keepdim=True) + 1e-9), nf / (nf.sum(-1, keepdim=True) + 1e-9)
    zt = zd.transpose(1, 2)
    pb, nb = torch.relu(zt).clone(), torch.relu(-zt).clone()
    lo = 0
    for n in groups:
        pb[..., lo:lo + n] = pb[..., lo:lo + n] / (pb[..., lo:lo + n].sum(-1, keepdim=True) + 1e-9)
        nb[..., lo:lo + n] = nb[..., lo:lo + n] / (nb[..., lo:lo + n].sum(-1, keepdim=True) + 1e-9)
        lo += n
    d = "cuda"
    o = [torch.empty(B, nq, S, device=d), torch.empty(B, nq, S, device=d), torch.empty(B, S, nq, device=d),
         torch.empty(B, S, nq, device=d)]
    ws = torch.empty(B * len(groups) * 2 * S, device=d)
    ops.stcexp_normalize(dev(z), dev(lens), ops.stcexp_group_meta(groups, d), len(groups), *o, ws)
    assert_close(o[0], pf, 2e-5, "pos_fw"); assert_close(o[1], nf, 2e-5, "neg_fw")
    assert_close(o[2], pb / len(groups), 2e-5, "pos_bw"); assert_close(o[3], nb / len(groups), 2e-5, "neg_bw")
    # bf16 outputs with zero-filled K padding (operands of the bf16 GEMM)
    ob = [torch.full((B, nq, 64), 7.0, device=d, dtype=torch.bfloat16) for _ in range(2)] + \
         [torch.full((B, S, 64), 7.0, device=d, dtype=torch.bfloat16) for _ in range(2)]
    ops.stcexp_normalize(dev(z), dev(lens), ops.stcexp_group_meta(groups, d), len(groups), *ob, ws)
    assert_close(ob[0][:, :, :S], pf, 5e-3, "pos_fw bf16"); assert_close(ob[3][:, :, :nq], nb / len(groups), 5e-3, "neg_bw bf16")
    assert float(ob[0][:, :, S:].abs().max()) == 0.0 and float(ob[2][:, :, nq:].abs().max()) == 0.0
    xc = rnd(37, 96, seed=8)
    assert torch.equal(ops.cast_bf16(dev(xc)).cpu(), xc.bfloat16())
    M, dm = 50, 128
    x, s, a, b = (rnd(M, dm, seed=i) for i in range(4))
    out = torch.empty(M, dm, device=d)
    ops.selector_mix(dev(x), dm, dev(s), dm, dev(a), dm, dev(b), dm, out, dm, M, dm)
    sg = torch.sigmoid(s.double())
    assert_close(out, x.double() + sg * a.double() + (1 - sg) * b.double(), 2e-5, "selector_mix")


# ------------------------------------------------------------------------------------------ decoder steps
def test_logsoftmax_topk(ops):
    N, V, k = 7, 10000, 5
    x = rnd(N, V, seed=1, scale=3.0)
    want = torch.log_softmax(x.double(), -1)
    wv, wi = torch.topk(want, k, -1)
    lp = torch.empty(N, V, device="cuda")
    tv = torch.empty(N, k, device="cuda")
    ti = torch.empty(N, k, dtype=torch.int32, device="cuda")
    ops.logsoftmax_topk(dev(x), V, lp, V, tv, ti, N, V, k)
    assert_close(lp, want, 2e-6, "log_softmax")
    assert torch.equal(ti.cpu().long(), wi)
    assert_close(tv, wv, 2e-6, "topk values")


@pytest.mark.parametrize("V,k", [(10000, 3), (10000, 8), (10000, 12), (37, 5), (12000, 4), (513, 16)])
def test_logsoftmax_topk_threshold_selection_edge_cases(ops, V, k):
    """The top-k is chosen among the elements >= the k-th largest wave maximum (decoder_ops.hip); rows on which that
    candidate list overflows — all-equal rows, hundreds of exact ties at the top, k above the wave count — and rows
    shorter than a block take the block-wide path: indices and values must still be exactly torch's stable order
    (value descending, lower index first), for streamed rows (V > 10240) as well."""
    g = torch.Generator().manual_seed(5)
    rows = [torch.randn(V, generator=g) * 2.0,                                   # ordinary
            torch.zeros(V),                                                      # all equal → indices 0..k-1
            torch.randn(V, generator=g).round(),                                 # a coarse grid: many ties everywhere
            torch.where(torch.rand(V, generator=g) < 0.5, torch.tensor(4.0), torch.randn(V, generator=g))]  # half the row tied at the top
    rows[0][V - 1] = rows[0][0] = rows[0].max() + 1.0                            # tie between the first and the last element
    x = torch.stack(rows)
    N = x.shape[0]
    want = torch.log_softmax(x.double(), -1)
    order = torch.argsort(x.double(), dim=-1, descending=True, stable=True)[:, :k]
    tv = torch.empty(N, k, device="cuda")
    ti = torch.empty(N, k, dtype=torch.int32, device="cuda")
    ops.logsoftmax_topk(dev(x), V, None, 0, tv, ti, N, V, k)
    assert torch.equal(ti.cpu().long(), order), (ti.cpu(), order)
    assert_close(tv, torch.gather(want, 1, order), 3e-6, "topk values")


def test_ensemble_logprobs_and_topk_rows(ops):
    # log(mean_m softmax(logits_m)) and its row-wise top-k (ties → lower index), the two ensemble-step kernels
    N, V, M, k = 7, 10000, 3, 5
    logits = [rnd(N, V, seed=10 + m, scale=3.0) for m in range(M)]
    want = torch.stack([torch.softmax(l.double(), -1) for l in logits]).mean(0).log()
    out = torch.empty(N, V, device="cuda")
    ops.ensemble_logprobs([dev(l) for l in logits], out)
    assert_close(out, want, 2e-6, "ensemble_logprobs")
    out[2, 17] = out[2, 4000] = out[2].max() + 1.0          # an exact tie: the lower index must win
    tv = torch.empty(N, k, device="cuda")
    ti = torch.empty(N, k, dtype=torch.int32, device="cuda")
    ops.topk_rows(out, tv, ti, k)
    ref_v, ref_i = torch.topk(out.cpu(), k, dim=-1)
    assert torch.equal(tv.cpu(), ref_v)
    assert ti[2, 0].item() == 17 and ti[2, 1].item() == 4000
    rows = [r for r in range(N) if r != 2]
    assert torch.equal(ti.cpu()[rows].long(), ref_i[rows])


@pytest.mark.parametrize("beams,S,d,heads", [(2, 20, 128, 4), (1, 144, 512, 8), (3, 144, 512, 8), (5, 144, 512, 8),
                                             (7, 50, 128, 8), (11, 33, 64, 4)])
def test_cross_attn_step(ops, beams, S, d, heads):
    n_img = 3
    N = n_img * beams
    q = rnd(N, d, seed=1)
    kv = rnd(n_img, S, 3 * d, seed=2)          # K at col 16.. is not contiguous with V on purpose
    koff, voff = d, 2 * d
    lens = torch.tensor([S, max(1, S // 2 + 1), max(1, S - 4)], dtype=torch.int32)
    valid = torch.ones(N, dtype=torch.int32)
    valid[min(3, N - 1)] = 0
    out = torch.empty(N, d, device="cuda")
    ops.cross_attn_step(dev(q), d, dev(kv), 3 * d, koff, voff, dev(lens), dev(valid), out, d, N, n_img, S, d, heads)
    dk = d // heads
    want = torch.empty(N, d, dtype=torch.float64)
    for n in range(N):
        i = n // beams
        K = kv[i, :, koff:koff + d].double().view(S, heads, dk)
        Vv = kv[i, :, voff:voff + d].double().view(S, heads, dk)
        s = torch.einsum("hc,shc->hs", q[n].double().view(heads, dk), K) / math.sqrt(dk)
        allow = (torch.arange(S) < lens[i]) & bool(valid[n])
        s = s.masked_fill(~allow[None, :], -1e4)
        want[n] = torch.einsum("hs,shc->hc", torch.softmax(s, -1), Vv).reshape(d)
    assert_close(out, want, 2e-5, "cross_attn_step")


def test_dynexp_step_matches_full_recompute(ops):
    """Feed T positions one at a time (identity ancestry) and compare every row with the oracle's
    full-prefix DynamicExpansionBlock."""
    from oracle import expansionnet_ref as R
    N, T, d, E = 3, 9, 128, 4
    names = ["cond_embed", "key_linear", "class_a_embed", "class_b_embed", "selector_embed"]
    sd = {}
    for i, nm in enumerate(names):
        sd[f"p.{nm}.weight"] = rnd(d, d, seed=10 + i, scale=d ** -0.5)
        sd[f"p.{nm}.bias"] = rnd(d, seed=20 + i, scale=0.1)
    sd["p.query_exp_vectors.weight"] = rnd(E, d, seed=30, scale=0.3)
    sd["p.bias_exp_vectors.weight"] = rnd(E, d, seed=31, scale=0.3)
    x = rnd(N, T, d, seed=40)
    pads = [0, 2, 4]
    t = torch.arange(T)
    ok = t[None, :] < (T - torch.tensor(pads))[:, None]
    causal = ((t[None, :, None] >= t[None, None, :]) & ok[:, :, None] & ok[:, None, :]).float()
    want = R.dynamic_expansion(sd, "p", x, E, causal)                     # (N,T,d)

    dv = "cuda"
    Wcat = torch.cat([sd[f"p.{nm}.weight"] for nm in names], 0)
    bcat = torch.cat([sd[f"p.{nm}.bias"] for nm in names], 0)
    caches = [torch.zeros(T, N, d, device=dv) for _ in range(4)] + [torch.zeros(T, N, T, E, device=dv) for _ in range(2)]
    qk = torch.zeros(T, N, E, device=dv)
    anc = torch.arange(N, dtype=torch.int32, device=dv)[:, None].repeat(1, T).contiguous()
    pos = torch.zeros(1, dtype=torch.int32, device=dv)
    got = torch.empty(N, T, d)
    for step in range(T):
        pos.fill_(step)
        xs = dev(x[:, step].contiguous())
        lin = ops.gemm(xs, dev(Wcat), dev(bcat))
        y = torch.zeros(N, d, device=dv)
        valid = dev(ok[:, step].to(torch.int32))
        ops.dynexp_step(lin, 5 * d, dev(sd["p.query_exp_vectors.weight"]), dev(sd["p.bias_exp_vectors.weight"]),
                        *caches, qk, anc, valid, pos, y, d, y, d, N, T, d, E)
        got[:, step] = y.cpu()
    assert_close(got, want, 5e-5, "dynexp_step")


def test_gemm_bf16_256sq_phase_pipeline(ops):
    """Config 12 (256x256 tile, four phases per K-tile, counted LDS-DMA waits): ragged M/N, 2..24 K-tiles,
    every epilogue feature, and run-to-run identical results (a pipeline race shows up as flicker)."""
    _need_experimental_gemm()
    for (M, N, K) in ((300, 328, 128), (517, 260, 384), (1024, 768, 768), (2304, 1536, 1536), (700, 3072, 256)):
        A, Wt = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16()
        b, r = rnd(N, seed=3), rnd(M, N, seed=4)
        want = torch.relu(0.5 * (A.double() @ Wt.double().T) + b.double()) + r.double()
        dA, dW, db, dr = dev(A), dev(Wt), dev(b), dev(r)
        got = ops.gemm(dA, dW, db, dr, act=2, alpha=0.5, out_dtype=torch.float32, tile_cfg=12)
        assert_close(got, want, 2e-4, f"cfg12 {M}x{N}x{K}")
        for _ in range(20):
            again = ops.gemm(dA, dW, db, dr, act=2, alpha=0.5, out_dtype=torch.float32, tile_cfg=12)
            assert torch.equal(again, got), f"cfg12 {M}x{N}x{K}: results differ between launches"
        got16 = ops.gemm(dA, dW, db, act=1, out_dtype=torch.bfloat16, tile_cfg=12)
        assert_close(got16, torch.nn.functional.gelu(A.double() @ Wt.double().T + b.double()), 6e-3, "cfg12 gelu→bf16")
    with pytest.raises(RuntimeError):                 # K-tiles are consumed in pairs
        ops.gemm(dev(rnd(256, 192, seed=1)).bfloat16(), dev(rnd(256, 192, seed=2)).bfloat16(), tile_cfg=12)


PERSISTENT_CFGS = [16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27]


@pytest.mark.parametrize("cfg", list(range(12)) + [13, 14, 15, 28, 29, 30, 31, 32, 33, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49] + PERSISTENT_CFGS)
def test_gemm_bf16_every_tile_config(ops, cfg):
    """Each tile / pipeline-depth / BK instantiation — one block per tile (0..11) and persistent with dynamic tile
    scheduling (16 + c) — against fp64 on ragged shapes (M, N not multiples of any tile) with every epilogue
    feature on."""
    if cfg not in DEFAULT_BF16_TILE_CFGS:
        _need_experimental_gemm()
    for (M, N, K) in ((300, 328, 192), (517, 260, 320)):
        A, Wt = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16()
        b, r = rnd(N, seed=3), rnd(M, N, seed=4)
        want = torch.relu(0.5 * (A.double() @ Wt.double().T) + b.double()) + r.double()
        got = ops.gemm(dev(A), dev(Wt), dev(b), dev(r), act=2, alpha=0.5, out_dtype=torch.float32, tile_cfg=cfg)
        assert_close(got, want, 2e-4, f"cfg{cfg} {M}x{N}x{K}")
    # bf16 output through the vector epilogues (N % 8 == 0 / N % 64 == 0), whole and ragged row tiles, a long K
    for (M, N, K) in ((576, 576, 256), (300, 320, 192), (290, 192, 1536)):
        A, Wt, b = rnd(M, K, seed=5).bfloat16(), rnd(N, K, seed=6, scale=0.05).bfloat16(), rnd(N, seed=7)
        want = torch.nn.functional.gelu(A.double() @ Wt.double().T + b.double())
        got = ops.gemm(dev(A), dev(Wt), dev(b), act=1, out_dtype=torch.bfloat16, tile_cfg=cfg)
        assert_close(got, want, 6e-3, f"cfg{cfg} bf16 out {M}x{N}x{K}")
        if cfg != 47 and cfg in DEFAULT_BF16_TILE_CFGS:            # (47 sums K in another order; every other default tile: same bits)
            assert torch.equal(got, ops.gemm(dev(A), dev(Wt), dev(b), act=1, out_dtype=torch.bfloat16, tile_cfg=0)), (cfg, M, N, K)


@pytest.mark.parametrize("cfg,K,bm,bnc", [(50, 192, 256, 64), (52, 192, 128, 64), (51, 384, 128, 32), (53, 384, 128, 64)])
def test_gemm_bf16_a_resident_streaming_kernels(ops, cfg, K, bm, bnc):
    """tile_cfg 50-53 (gemm_bf16_apanel_kernel: A rows in registers over the whole K, W chunks streamed through LDS, counted
    store waits) — against fp64 AND bit-for-bit against the tiled kernel (same MFMA, same K order), over every epilogue
    the Swin stage-0/1 products use, panel / chunk counts that exercise every column-range split (1 chunk per block, uneven
    ranges, several panels per XCD), and a strided output; shapes that are not whole tiles are refused."""
    g = torch.Generator().manual_seed(cfg)
    for M, N in ((bm * 3, bnc * 5), (bm * 17, bnc * 12), (bm * 9, bnc * 1)):
        A = dev(torch.randn(M, K, generator=g)).bfloat16()
        Wt = dev(torch.randn(N, K, generator=g) * 0.05).bfloat16()
        b, r = dev(torch.randn(N, generator=g)), dev(torch.randn(M, N, generator=g))
        lin = A.double() @ Wt.double().T
        for name, kw, want in (
                ("plain", dict(out_dtype=torch.bfloat16), lin),
                ("bias", dict(bias=b, out_dtype=torch.bfloat16), lin + b.double()),
                ("bias+gelu", dict(bias=b, act=ops.ACT_GELU, out_dtype=torch.bfloat16),
                 torch.nn.functional.gelu(lin + b.double())),
                ("alpha+bias f32", dict(bias=b, alpha=0.25, out_dtype=torch.float32), 0.25 * lin + b.double()),
                ("bias+residual f32", dict(bias=b, residual=r, out_dtype=torch.float32), lin + b.double() + r.double())):
            if "residual" in kw and cfg == 50:
                with pytest.raises(RuntimeError):          # the 64-row-per-wave form has no residual instantiation
                    ops.gemm(A, Wt, tile_cfg=cfg, **kw)
                continue
            got = ops.gemm(A, Wt, tile_cfg=cfg, **kw)
            ref = ops.gemm(A, Wt, tile_cfg=0, **kw)
            assert_close(got, want, 6e-3 if kw["out_dtype"] == torch.bfloat16 else 3e-4, f"cfg{cfg} {name} {M}x{N}x{K}")
            assert torch.equal(got, ref), (cfg, name, M, N, float((got.float() - ref.float()).abs().max()))
    # residual aliasing the output (x += proj(...)) and a strided fp32 output, as engine.SwinEngine calls it
    if cfg != 50:
        M, N = bm * 4, bnc * 6
        A = dev(torch.randn(M, K, generator=g)).bfloat16()
        Wt = dev(torch.randn(N, K, generator=g) * 0.05).bfloat16()
        wide = dev(torch.randn(M, 2 * N, generator=g))
        x = wide[:, N:]
        want = (A.double() @ Wt.double().T).float() + x.clone()
        ops.gemm(A, Wt, residual=x, out=x, M=M, N=N, K=K, lda=K, ldw=K, ldr=2 * N, ldc=2 * N, tile_cfg=cfg)
        assert_close(x, want, 3e-4, f"cfg{cfg} in-place residual")
    for M, N, Kx in ((bm * 2 + 16, bnc * 4, K), (bm * 2, bnc * 4 + 8, K), (bm * 2, bnc * 4, K + 64)):
        with pytest.raises(RuntimeError):
            ops.gemm(dev(torch.zeros(M, Kx)).bfloat16(), dev(torch.zeros(N, Kx)).bfloat16(), tile_cfg=cfg)


@pytest.mark.parametrize("cfg", [1, 2, 7, 10])
def test_gemm_bf16_whole_line_store_epilogue(ops, cfg):
    """bf16 outputs with N % 64 == 0 leave the tiled kernel through the lane-exchange epilogue (two column groups
    swapped between lanes frow and frow ^ 8, 8 rows x 128 bytes per store): ragged M (rows 8-15 of the last 16 absent,
    a lone row), every epilogue feature, a strided output — against fp64 and bit-for-bit against tile_cfg 0 (NI = 2:
    the plain 16-byte stores)."""
    g = torch.Generator().manual_seed(100 + cfg)
    for M, N, K in ((300, 320, 192), (517, 256, 320), (1, 64, 64), (264, 192, 128)):
        A = dev(torch.randn(M, K, generator=g)).bfloat16()
        Wt = dev(torch.randn(N, K, generator=g) * 0.05).bfloat16()
        b, r = dev(torch.randn(N, generator=g)), dev(torch.randn(M, N, generator=g))
        lin = A.double() @ Wt.double().T
        for name, kw, want in (("plain", {}, lin), ("bias+gelu", dict(bias=b, act=ops.ACT_GELU), torch.nn.functional.gelu(lin + b.double())),
                               ("alpha+bias+res", dict(bias=b, residual=r, alpha=0.5), 0.5 * lin + b.double() + r.double())):
            got = ops.gemm(A, Wt, out_dtype=torch.bfloat16, tile_cfg=cfg, **kw)
            ref = ops.gemm(A, Wt, out_dtype=torch.bfloat16, tile_cfg=0, **kw)
            assert_close(got, want, 6e-3, f"cfg{cfg} {name} {M}x{N}x{K}")
            assert torch.equal(got, ref), (cfg, name, M, N, K)
        wide = torch.full((M, N + 64), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, Wt, b, out=wide, M=M, N=N, K=K, lda=K, ldw=K, ldc=N + 64, tile_cfg=cfg)
        assert_close(wide[:, :N], lin + b.double(), 6e-3, f"cfg{cfg} strided {M}x{N}x{K}")
        assert torch.all(wide[:, N:] == 7.0)


@pytest.mark.parametrize("cfg,K,bm,bnc", [(50, 192, 256, 64), (52, 192, 128, 64), (51, 384, 128, 32), (53, 384, 128, 64)])
def test_gemm_bf16_layernorm_while_reading(ops, cfg, K, bm, bnc):
    """odic_gemm_args.a_ln: the A-resident kernels normalise the fp32 rows while they read them (norm → linear in one
    launch, gamma / beta folded into W / bias).  Against fp64 LayerNorm → Linear, against the two-launch path
    (odic_layernorm → bf16, then the tiled product with the same folded weights), on rows with a large common offset
    (mean >> spread: what the one-pass moments of the K = 384 form must survive) and constant rows (variance 0)."""
    g = torch.Generator().manual_seed(7 * cfg)
    M, N = bm * 5, bnc * 6
    x = torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3.0 + torch.randn(M, 1, generator=g) * 40.0
    x[3] = 5.0                                            # a constant row: (x - mean) = 0, rstd = 1/sqrt(eps)
    x[4] = 0.0
    W = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    want = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double()
    Wf, bf, _ = ops.fold_layernorm_bf16(dev(W), dev(b), dev(gamma), dev(beta))
    xd = dev(x)
    for name, kw, ref in (("linear", {}, want), ("gelu", dict(act=ops.ACT_GELU), torch.nn.functional.gelu(want))):
        got = ops.gemm(None, Wf, bf, a_ln=xd, tile_cfg=cfg, out_dtype=torch.bfloat16, **kw)
        assert got.shape == (M, N) and got.dtype == torch.bfloat16
        assert_close(got, ref, 1.2e-2, f"cfg{cfg} ln+{name} vs fp64")
        xn = ops.layernorm(xd, dev(torch.ones(K)), dev(torch.zeros(K)), out_dtype=torch.bfloat16)
        two = ops.gemm(xn, Wf, bf, tile_cfg=0, out_dtype=torch.bfloat16, **kw)
        # same bf16 operands up to the rounding of the normalised rows (different summation order of the moments)
        diff = (got.float() - two.float()).abs()
        assert float(diff.max()) <= 2e-2 * float(two.float().abs().max()), (cfg, name, float(diff.max()))
        assert float((diff > 0).float().mean()) < 0.02, (cfg, name, float((diff > 0).float().mean()))
    got32 = ops.gemm(None, Wf, bf, a_ln=xd, tile_cfg=cfg, out_dtype=torch.float32)
    assert_close(got32, want, 8e-3, f"cfg{cfg} ln fp32 out")
    assert torch.isfinite(got32).all()
    # refused: a residual, a tiled configuration, a row count that is not whole panels
    with pytest.raises(RuntimeError):
        ops.gemm(None, Wf, bf, a_ln=xd, tile_cfg=1)
    with pytest.raises(RuntimeError):
        ops.gemm(None, Wf, bf, a_ln=xd[: bm + 16].contiguous(), tile_cfg=cfg)


@pytest.mark.parametrize("shift", [0, 6])
def test_swin_qkv_attention_fused(ops, shift):
    """odic_swin_qkv_attention (norm1 → qkv → attention core of a width-192 Swin block in one launch) against the two
    launches it replaces — the LayerNorm-while-reading product + the bf16 window-attention kernel: the same MFMAs in the
    same order and the same bf16 roundings of q / k / v, so the outputs must agree bit for bit — and against an fp64
    restatement of swin_transformer_mod.py:309-334 / :222-263 (roll, partition, bias, SW-MSA mask, softmax, reverse)."""
    B, res, C_, heads, ws = 2, 24, 192, 6, 12
    g = torch.Generator().manual_seed(11 + shift)
    L = res * res
    x = torch.randn(B * L, C_, generator=g) * 1.5 + torch.randn(B * L, 1, generator=g) * 3.0
    W = torch.randn(3 * C_, C_, generator=g) * 0.06
    bq = torch.randn(3 * C_, generator=g) * 0.2
    gamma, beta = 1.0 + 0.2 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.3
    Wf, bf, _ = ops.fold_layernorm_bf16(dev(W), dev(bq), dev(gamma), dev(beta))
    dense = ops.shifted_bias_prescaled(dev(table), ws, 32 ** -0.5)
    xd = dev(x)
    got = ops.swin_qkv_attention(xd, Wf, bf, dense, B, res, C_, heads, ws, shift)
    qkv = ops.gemm(None, Wf, bf, a_ln=xd, out_dtype=torch.bfloat16)
    two = ops.window_attention(qkv, dev(table), B, res, C_, heads, ws, shift, bias_shifted_prescaled=dense)
    assert got.dtype == torch.bfloat16 and got.shape == two.shape
    assert torch.equal(got, two), float((got.float() - two.float()).abs().max())

    # fp64 restatement of the reference block (to the tolerance of bf16 q / k / v / P)
    xn = torch.nn.functional.layer_norm(x.double(), (C_,), gamma.double(), beta.double(), 1e-5)
    qkv64 = (xn @ W.double().T + bq.double()).view(B, res, res, 3 * C_)
    if shift:
        qkv64 = torch.roll(qkv64, (-shift, -shift), (1, 2))
    nw = res // ws
    win = qkv64.view(B, nw, ws, nw, ws, 3, heads, 32).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(B * nw * nw, 3, heads, ws * ws, 32)
    q, k, v = win[:, 0] * 32 ** -0.5, win[:, 1], win[:, 2]
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (ws - 1)
    idx = rel[..., 0] * (2 * ws - 1) + rel[..., 1]
    att = q @ k.transpose(-1, -2) + table.double()[idx.view(-1)].view(ws * ws, ws * ws, heads).permute(2, 0, 1)[None]
    if shift:
        img = torch.zeros(res, res)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[hs, wsl] = cnt
                cnt += 1
        mw = img.view(nw, ws, nw, ws).permute(0, 2, 1, 3).reshape(nw * nw, ws * ws)
        mask = (mw[:, None, :] - mw[:, :, None] != 0).double() * -100.0
        att = att.view(B, nw * nw, heads, ws * ws, ws * ws) + mask[None, :, None]
        att = att.view(B * nw * nw, heads, ws * ws, ws * ws)
    o = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(B, nw, nw, ws, ws, C_).permute(0, 1, 3, 2, 4, 5).reshape(B, res, res, C_)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    assert_close(got, o.reshape(B * L, C_), 2.5e-2, f"fused qkv+attention shift {shift} vs fp64")
    with pytest.raises(RuntimeError):                               # only width 192
        ops.swin_qkv_attention(dev(torch.zeros(B * L, 384)), dev(torch.zeros(1152, 384)).bfloat16(), dev(torch.zeros(1152)),
                               ops.shifted_bias_prescaled(dev(torch.zeros(529, 12)), ws, 32 ** -0.5), B, res, 384, 12, ws, 0)


def test_gemm_bf16_default_build_rejects_compiled_out_configurations(ops):
    from on_device_image_captioning_amd import _hip
    if b"experimental-gemm" in _hip.load().odic_build_info():
        pytest.skip("experimental build")
    A, Wt = dev(rnd(256, 128, seed=1)).bfloat16(), dev(rnd(256, 128, seed=2)).bfloat16()
    for cfg in (3, 12, 16, 33):
        with pytest.raises(RuntimeError):
            ops.gemm(A, Wt, tile_cfg=cfg)
    with pytest.raises(RuntimeError):                  # the LayerNorm fold across two products is compiled out too
        ops.gemm(A, Wt, out_dtype=torch.float32, out16=torch.empty(256, 256, dtype=torch.bfloat16, device="cuda"),
                 stats_out=torch.empty(256, 8, 2, device="cuda"))


@pytest.mark.parametrize("cfg", [16, 17, 23, 26])
def test_gemm_bf16_persistent_many_tiles_per_block(ops, cfg):
    """Persistent launches where every block walks MANY tiles (more tiles than resident slots, tile stealing
    across XCD partitions at the end), all epilogue forms, the workspace re-armed launch after launch, and
    bit-identical results to the one-block-per-tile kernel of the same tile configuration (same MFMA order)."""
    _need_experimental_gemm()
    for (M, N, K, act, odt) in ((9216, 3072, 768, 1, torch.bfloat16), (36864, 384, 384, 0, torch.float32),
                                (20000, 1100, 128, 2, torch.float32)):
        A, Wt = rnd(M, K, seed=5).bfloat16(), rnd(N, K, seed=6, scale=0.05).bfloat16()
        b = rnd(N, seed=7)
        r = rnd(M, N, seed=8) if odt == torch.float32 else None
        dA, dW, db, dr = dev(A), dev(Wt), dev(b), (dev(r) if r is not None else None)
        base = ops.gemm(dA, dW, db, dr, act=act, out_dtype=odt, tile_cfg=cfg - 16)
        for _ in range(4):
            got = ops.gemm(dA, dW, db, dr, act=act, out_dtype=odt, tile_cfg=cfg)
            assert torch.equal(got, base), f"cfg{cfg} {M}x{N}x{K}"
        ws = ops._gemm_workspace(dA.device)
        torch.cuda.synchronize()
        assert int(ws.abs().sum()) == 0                    # counters left at zero


# ------------------------------------------------------------------------------------------ beam bookkeeping, directly
def _beam_state(ops, n_img, k, T):
    from on_device_image_captioning_amd import _hip
    N = n_img * k
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")      # noqa: E731
    t = dict(tokens=z(n_img, k, T, dt=torch.int64), logprobs=z(n_img, k, T), anc=z(N, T, dt=torch.int32), cumul=z(N),
             n_elem=z(N, dt=torch.int32), has_eos=z(N, dt=torch.int32), row_valid=z(N, dt=torch.int32),
             next_tok=z(N, dt=torch.int64), pos=z(1, dt=torch.int32), done=z(1, dt=torch.int32), ctr=z(1, dt=torch.int32))
    st = _hip.BeamState(*(t[n].data_ptr() for n in ("tokens", "logprobs", "anc", "cumul", "n_elem", "has_eos",
                                                    "row_valid", "next_tok", "pos", "done", "ctr")))
    return t, st


@pytest.mark.parametrize("k,V", [(3, 1000), (5, 10000), (2, 12000), (9, 3000)])
def test_beam_search_step_fused_equals_the_three_launches(ops, k, V):
    """odic_beam_search_step (log-softmax + top-k + beam update + next input embedding in one launch) against
    odic_logsoftmax_topk → odic_beam_step → odic_dec_embed on the same logits, bit for bit, over a whole search with
    beams finishing on the way; a second search re-uses both states (counter re-armed, idempotent past the end)."""
    n_img, T, sos, eos, d = 5, 11, 3, 4, 64
    N = n_img * k
    g = torch.Generator().manual_seed(11)
    embed = dev(torch.randn(V, d, generator=g))
    pos_table = dev(torch.randn(T, d, generator=g))
    ta, sa = _beam_state(ops, n_img, k, T)
    tb, sb = _beam_state(ops, n_img, k, T)
    ya, yb = torch.zeros(N, 2 * d, device="cuda"), torch.zeros(N, 2 * d, device="cuda")
    emb = ops.embed_args(embed, pos_table, yb, 2 * d, d, 8.0)
    cv, ci = torch.zeros(N, k, device="cuda"), torch.zeros(N, k, dtype=torch.int32, device="cuda")
    for search in range(2):
        ops.beam_reset(sa, n_img, k, T, sos)
        ops.dec_embed(ta["next_tok"], embed, pos_table, ta["pos"], ya, 2 * d, N, d, 8.0)
        ops.beam_reset(sb, n_img, k, T, sos, emb=emb)
        assert torch.equal(ya, yb)
        for step in range(T + 1):                      # two launches past the last position: nothing may change
            logits = torch.randn(N, V, generator=g) * 3.0
            if step >= 2:
                logits[torch.rand(N, generator=g) < 0.4, eos] = 12.0
            lg = dev(logits)
            ops.logsoftmax_topk(lg, V, None, 0, cv, ci, N, V, k)
            ops.beam_step(cv, ci, sa, n_img, k, T, eos)
            if step < T - 2:                           # (position T-1 is never an input)
                ops.dec_embed(ta["next_tok"], embed, pos_table, ta["pos"], ya, 2 * d, N, d, 8.0)
            ops.beam_search_step(lg, V, V, sb, n_img, k, T, eos, emb=emb)
            for name in ta:
                assert torch.equal(ta[name], tb[name]), f"search {search} step {step}: {name}"
            assert torch.equal(ya, yb), f"search {search} step {step}: next input rows"
        assert int(ta["pos"].item()) == T - 1 and int(ta["has_eos"].sum().item()) > 0


def test_beam_step_direct_against_the_search_loop(ops):
    """odic_beam_reset / odic_beam_step / odic_beam_finalize(_best) on synthetic candidate tables, step by step
    against a plain restatement of captioning_model.py:117-241: EXACT ties in the k·k selection (lowest flat
    index wins, the scan order of torch.topk on sorted input), finished beams (0.0 / −999 masking, frozen length),
    re-summed cumulative scores, the ancestor table, the all-finished fixed point over the remaining steps,
    `done`, and the arrival counter re-armed after every step and across a second search on the same state."""
    n_img, k, T, sos, eos = 6, 3, 14, 5, 7
    N = n_img * k
    rng = np.random.default_rng(3)
    t, st = _beam_state(ops, n_img, k, T)
    for search in range(2):
        ops.beam_reset(st, n_img, k, T, sos)
        toks = [[[sos] for _ in range(k)] for _ in range(n_img)]
        lps = [[[np.float32(0.0)] for _ in range(k)] for _ in range(n_img)]
        n_elem = [[1] * k for _ in range(n_img)]
        anc = np.zeros((N, T), dtype=np.int64)
        all_done_at = None
        for step in range(T - 1):
            # candidates: log-probs on a coarse grid (→ exact ties across beams), sorted descending per row;
            # EOS is frequent from step 3 on so that every beam finishes well before T
            cv = -np.sort(rng.integers(1, 6, size=(N, k)).astype(np.float32) * 0.25, axis=1)
            cw = rng.integers(10, 40, size=(N, k)).astype(np.int32)
            if step >= 3:
                cw[rng.random((N, k)) < 0.45] = eos
            ops.beam_step(dev(torch.from_numpy(cv)), dev(torch.from_numpy(cw)), st, n_img, k, T, eos)
            new_toks, new_lps, new_ne, valid, nxt = [], [], [], [], []
            new_anc = anc.copy()
            for b in range(n_img):
                if step == 0:
                    sel = [(0, c) for c in range(k)]
                    val = {(0, c): cv[b * k, c] for c in range(k)}
                else:
                    tot, val = [], {}
                    for j in range(k):
                        dn = eos in toks[b][j]
                        cu = np.float32(0.0)
                        for x in lps[b][j]:
                            cu = np.float32(cu + x)
                        for c in range(k):
                            v = (np.float32(0.0) if c == 0 else np.float32(-999.0)) if dn else cv[b * k + j, c]
                            val[(j, c)] = v
                            tot.append((np.float32(cu + v), j, c))
                    sel = []
                    for _ in range(k):
                        bi = max(range(len(tot)), key=lambda i: (tot[i][0], -i))        # ties → lowest flat index
                        sel.append((tot[bi][1], tot[bi][2]))
                        tot[bi] = (np.float32(-np.inf), 0, 0)
                rt, rl, rn = [], [], []
                for r, (j, c) in enumerate(sel):
                    had = (eos in toks[b][j]) if step > 0 else False
                    w = int(cw[b * k + j, c])
                    rt.append(toks[b][j] + [w])
                    rl.append(lps[b][j] + [val[(j, c)]])
                    rn.append((n_elem[b][j] if step > 0 else 1) + (0 if had else 1))
                    valid.append(0 if had else 1)
                    nxt.append(w)
                    new_anc[b * k + r, :step] = anc[b * k + j, :step]
                    new_anc[b * k + r, step] = b * k + j
                new_toks.append(rt); new_lps.append(rl); new_ne.append(rn)
            toks, lps, n_elem, anc = new_toks, new_lps, new_ne, new_anc
            L = step + 2
            got_tok = t["tokens"].cpu().numpy()[:, :, :L]
            assert got_tok.tolist() == toks, f"search {search} step {step}: prefixes"
            np.testing.assert_array_equal(t["logprobs"].cpu().numpy()[:, :, :L], np.array(lps, dtype=np.float32))
            cum = [[np.float32(0.0)] * k for _ in range(n_img)]
            for b in range(n_img):
                for j in range(k):
                    c = np.float32(0.0)
                    for x in lps[b][j]:
                        c = np.float32(c + x)
                    cum[b][j] = c
            np.testing.assert_array_equal(t["cumul"].cpu().numpy().reshape(n_img, k), np.array(cum, dtype=np.float32))
            assert t["n_elem"].cpu().numpy().reshape(n_img, k).tolist() == n_elem
            assert t["has_eos"].cpu().numpy().reshape(n_img, k).tolist() == [[int(eos in s_) for s_ in row] for row in toks]
            assert t["row_valid"].cpu().tolist() == valid and t["next_tok"].cpu().tolist() == nxt
            np.testing.assert_array_equal(t["anc"].cpu().numpy()[:, :step + 1], anc[:, :step + 1])
            assert int(t["pos"]) == step + 1 and int(t["ctr"]) == 0
            nothing_grew = all(v == 0 for v in valid)
            if nothing_grew and all_done_at is None:
                all_done_at = step
                frozen = ([row[:] for row in n_elem], [row[:] for row in cum])
            assert int(t["done"]) == int(all_done_at is not None)
            if all_done_at is not None:                         # fixed point: lengths and scores no longer move
                assert n_elem == frozen[0] and cum == frozen[1]
        assert all_done_at is not None and all_done_at < T - 3   # the fixture does exercise the fixed point
        order = torch.empty(n_img, k, dtype=torch.int32, device="cuda")
        score = torch.empty(n_img, k, dtype=torch.float32, device="cuda")
        out_tok = torch.empty(n_img, T, dtype=torch.int32, device="cuda")
        out_len = torch.empty(n_img, dtype=torch.int32, device="cuda")
        ops.beam_finalize_best(st, order, score, out_tok, out_len, n_img, k, T, eos)
        order2, score2 = torch.empty_like(order), torch.empty_like(score)
        ops.beam_finalize(st, order2, score2, n_img, k)
        assert torch.equal(order, order2) and torch.equal(score, score2)
        for b in range(n_img):
            sc = [np.float32(cum[b][j] / np.float32(n_elem[b][j])) for j in range(k)]
            want_order = sorted(range(k), key=lambda j: (-sc[j], j))
            assert order[b].tolist() == want_order
            best = want_order[0]
            n = n_elem[b][best]
            assert int(out_len[b]) == n
            assert out_tok[b].tolist() == toks[b][best][:n] + [eos] * (T - n)


def test_logsoftmax_sample_draws_follow_the_distribution(ops):
    """odic_logsoftmax_sample: k draws without replacement per row (Gumbel-top-k on the device).  Over 40,000 rows
    of one 12-word distribution: no duplicates in a row, reported values = log-probs of the drawn words, the first
    draw's frequencies match softmax(x) and the second draw's match the exact without-replacement marginal (4 sigma);
    same seed → same draws, another seed or position → different draws."""
    V, k, N = 12, 3, 40000
    x = rnd(V, seed=4, scale=1.3)
    logits = dev(x[None, :].expand(N, V).contiguous())
    val = torch.empty(N, k, device="cuda")
    idx = torch.empty(N, k, dtype=torch.int32, device="cuda")
    pos = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.logsoftmax_sample(logits, V, None, 0, val, idx, N, V, k, 1234, pos)
    I = idx.cpu().long()
    logp = torch.log_softmax(x.double(), 0)
    assert_close(val, logp[I].float(), 2e-6, "log-probs of the drawn words")
    assert bool(((I[:, 0] != I[:, 1]) & (I[:, 0] != I[:, 2]) & (I[:, 1] != I[:, 2])).all())
    p = logp.exp()
    f1 = torch.bincount(I[:, 0], minlength=V).double() / N
    assert float(((f1 - p).abs() / (p * (1 - p) / N).sqrt()).max()) < 4.0
    p2 = torch.stack([sum(p[a] * p[b] / (1 - p[a]) for a in range(V) if a != b) for b in range(V)])
    f2 = torch.bincount(I[:, 1], minlength=V).double() / N
    assert float(((f2 - p2).abs() / (p2 * (1 - p2) / N).sqrt()).max()) < 4.0
    idx2 = torch.empty_like(idx)
    ops.logsoftmax_sample(logits, V, None, 0, val, idx2, N, V, k, 1234, pos)
    assert torch.equal(idx, idx2)
    ops.logsoftmax_sample(logits, V, None, 0, val, idx2, N, V, k, 1235, pos)
    assert not torch.equal(idx, idx2)
    pos.fill_(3)
    ops.logsoftmax_sample(logits, V, None, 0, val, idx2, N, V, k, 1234, pos)
    assert not torch.equal(idx, idx2)
    full = torch.empty(4, 10000, device="cuda")                    # the vocabulary-sized row + the log-prob output
    lg = dev(rnd(4, 10000, seed=8))
    ops.logsoftmax_sample(lg, 10000, full, 10000, val[:4], idx[:4], 4, 10000, k, 5, None)
    assert_close(full, torch.log_softmax(lg.double().cpu(), -1), 2e-6, "logp_out")


# ------------------------------------------------------------------------------------------ fp8 / fp16 GEMM (configs[4])
def _to_fp8(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_gemm_fp8_identity_and_random(ops, cfg):
    """odic_gemm with fp8 (OCP e4m3) operands: exact on integer data with A = I and an ASYMMETRIC W (layout check of
    the 16x16x32 fp8 MFMA fragments, the two-MFMAs-per-16-byte-read K assignment, the permuted W rows), then random
    data against fp64 of the same fp8 values with every epilogue feature (per-column scale, bias, GELU, out_scale,
    residual) and the three output types.  Tile configurations 5..9 are 0..4 on the BLOCK-SCALED fp8 MFMA
    (v_mfma_scale_f32_16x16x128_f8f6f4, unit E8M0 scales — the instruction that carries gfx950's 5 PFLOP/s fp8 rate);
    they need K % 128 == 0 and are what the built-in choice (-1) takes when K allows."""
    n = 256
    eye = _to_fp8(torch.eye(n))
    Wt = _to_fp8(((torch.arange(n)[:, None] * 3 + torch.arange(n)[None, :] * 5) % 17 - 8).float())
    got = ops.gemm(dev(eye), dev(Wt), out_dtype=torch.float32, tile_cfg=cfg)
    assert torch.equal(got.cpu(), Wt.float().T.contiguous())
    for (M, N, K) in ((300, 328, 256), (517, 264, 384), (2304, 768, 1024), (1000, 576, 192)):   # K = 192: 64-byte rows
        if cfg >= 5 and K % 128:
            with pytest.raises(RuntimeError):
                ops.gemm(dev(_to_fp8(rnd(M, K, seed=1))), dev(_to_fp8(rnd(N, K, seed=2))), out_dtype=torch.float32, tile_cfg=cfg)
            continue
        A, Wq = _to_fp8(rnd(M, K, seed=1, scale=2.0)), _to_fp8(rnd(N, K, seed=2, scale=3.0))
        cs, b, r = rnd(N, seed=3).abs() * 0.01 + 0.005, rnd(N, seed=4), rnd(M, N, seed=5)
        lin = (A.double() @ Wq.double().T) * cs.double() + b.double()
        got = ops.gemm(dev(A), dev(Wq), dev(b), dev(r), col_scale=dev(cs), out_dtype=torch.float32, tile_cfg=cfg)
        assert_close(got, lin + r.double(), 1e-4, f"fp8 cfg{cfg} {M}x{N}x{K} → f32")     # fp32 accumulation inside the fp8 MFMA
        got16 = ops.gemm(dev(A), dev(Wq), dev(b), col_scale=dev(cs), out_dtype=torch.float16, tile_cfg=cfg)
        assert_close(got16, lin, 1e-3, "fp8 → f16")
        got8 = ops.gemm(dev(A), dev(Wq), dev(b), act=1, col_scale=dev(cs), out_scale=7.0, out_dtype=torch.float8_e4m3fn,
                        tile_cfg=cfg)
        want8 = torch.nn.functional.gelu(lin) * 7.0
        err = (got8.float().cpu().double() - want8).abs()
        assert float((err - want8.abs() * 2 ** -4).max()) <= 2 ** -9 + 1e-6, "fp8 → fp8: half an e4m3 ulp (3 mantissa bits)"


@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_gemm_f16(ops, cfg):
    for (M, N, K) in ((256, 256, 64), (517, 264, 320)):
        A, Wt = rnd(M, K, seed=1).half(), rnd(N, K, seed=2, scale=0.05).half()
        b, r = rnd(N, seed=3), rnd(M, N, seed=4)
        want = (A.double() @ Wt.double().T) + b.double() + r.double()
        got = ops.gemm(dev(A), dev(Wt), dev(b), dev(r), out_dtype=torch.float32, tile_cfg=cfg)
        assert_close(got, want, 2e-5, f"f16 cfg{cfg} {M}x{N}x{K}")


def test_layernorm_fp8_output(ops):
    x, g, b = rnd(300, 768, seed=1, scale=2.0), 1 + rnd(768, seed=2, scale=0.1), rnd(768, seed=3, scale=0.05)
    s_ = 0.02                                                    # consumer's quantisation scale, folded into gamma / beta
    got = ops.layernorm(dev(x), dev(g / s_), dev(b / s_), out_dtype=torch.float8_e4m3fn)
    want = torch.nn.functional.layer_norm(x.double(), (768,), g.double(), b.double()) / s_
    err = (got.float().cpu().double() - want.clamp(-448, 448)).abs()
    assert float((err - want.abs().clamp(max=448) * 2 ** -4).max()) <= 2 ** -9 + 1e-6


@pytest.mark.parametrize("res,heads,shift", [(48, 6, 6), (24, 12, 0), (12, 48, 0)])
def test_window_attention_fp16(ops, res, heads, shift):
    B, ws = 2, 12
    C = heads * 32
    qkv = rnd(B * res * res, 3 * C, seed=res + shift, scale=1.5).half()
    table = rnd(529, heads, seed=9, scale=0.5)
    want = _win_ref(qkv.float(), table, B, res, C, heads, ws, shift)
    got = ops.window_attention(dev(qkv), dev(table), B, res, C, heads, ws, shift,
                               bias_shifted_prescaled=ops.shifted_bias_prescaled(dev(table), ws, 32 ** -0.5))
    assert got.dtype == torch.float16
    assert_close(got, want, 2.5e-3, "window_attention fp16")


# ------------------------------------------------------------------------------------------ LayerNorm folded across two bf16 products
@pytest.mark.parametrize("cfg", [0, 1, 7, 10])
def test_gemm_bf16_layernorm_fold_producer_consumer(ops, cfg):
    """Producer: a residual product (fp32 out) that also leaves the bf16 copy of its rows and their per-32-column
    moments; consumer: the next product normalises those rows in its epilogue.  Checked: the copy is the rounded
    output bit for bit, the moments are those of the bf16 values, and consumer == LayerNorm(copy)·Wᵀ + b in fp64
    (with the packed bf16 W·diag(gamma)), for rows with a large common offset too (the centred combination)."""
    _need_experimental_gemm()
    M, C, N2 = 777, 384, 328
    A0, W0 = rnd(M, 256, seed=1).bfloat16(), rnd(C, 256, seed=2, scale=0.08).bfloat16()
    b0 = rnd(C, seed=3)
    r = rnd(M, C, seed=4, scale=2.0) + 5.0 * rnd(M, 1, seed=5)          # per-row offset up to several sigma
    x16 = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(M, C // 32, 2, device="cuda")
    x = ops.gemm(dev(A0), dev(W0), dev(b0), dev(r), out_dtype=torch.float32, out16=x16, stats_out=stats, tile_cfg=cfg)
    want_x = A0.double() @ W0.double().T + b0.double() + r.double()
    assert_close(x, want_x, 2e-5, "producer fp32 output")
    assert torch.equal(x16.cpu(), x.cpu().bfloat16())
    a = x16.cpu().double().view(M, C // 32, 32)
    gm = a.mean(-1)
    gm2 = ((a - gm[..., None]) ** 2).sum(-1)
    assert_close(stats[..., 0], gm, 1e-5, "group means")
    assert_close(stats[..., 1], gm2, 1e-4, "group centred sums of squares")
    g, b = 1 + rnd(C, seed=6, scale=0.1), rnd(C, seed=7, scale=0.05)
    W1, b1 = rnd(N2, C, seed=8, scale=0.05), rnd(N2, seed=9)
    Wf, bf, cs = ops.fold_layernorm_bf16(dev(W1), dev(b1), dev(g), dev(b))
    got = ops.gemm(x16, Wf, bf, act=1, ln_fold=(cs, 1e-5), ln_stats=stats, tile_cfg=cfg)
    ad = x16.cpu().double()
    norm = (ad - ad.mean(-1, keepdim=True)) / torch.sqrt(ad.var(-1, unbiased=False, keepdim=True) + 1e-5)
    want = torch.nn.functional.gelu(norm @ Wf.cpu().double().T + bf.cpu().double())
    assert_close(got, want, 8e-3, "consumer (bf16 out)")
    ln = torch.nn.functional.layer_norm(ad, (C,), g.double(), b.double())
    assert_close(got, torch.nn.functional.gelu(ln @ W1.double().T + b1.double()), 1.5e-2, "consumer vs LayerNorm + Linear")
