"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32, plain ATen ops) restatement of the ExpansionNet v2 inference path of
nighting0le01/On_Device_Image_Captioning, written clean-room from SURVEY.md §8 and a reading of
the reference sources.  It is the checker for the HIP path and the `cpu_baseline` of bench.py.

  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
  * The product package (on_device_image_captioning_amd/) never imports it and has no CPU fallback.

Pinning: this restatement is checked against outputs of the *real* reference
(/root/reference/legacy_models imported as `models`) by oracle/make_golden.py, which writes the
fixtures under tests/golden/; tests/test_oracle_golden.py replays them.  The reference itself has
no tests or golden vectors for this path, so those fixtures are the pin.

Every function takes the flat state dict (layout: SURVEY §8 A21) — there is no nn.Module here.
Reference line numbers refer to /root/reference/legacy_models/*.py unless a path is given.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# =================================================================================================
# small helpers
# =================================================================================================
def _linear(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    w = sd[name + ".weight"]
    return F.layer_norm(x, (w.numel(),), w, sd[name + ".bias"], 1e-5)


# =================================================================================================
# Swin backbone  (swin_transformer_mod.py)
# =================================================================================================
def window_token_index(res: int, ws: int, shift: int) -> torch.Tensor:
    """(nW, ws²) int64: for window w and in-window slot n, the *un-shifted* flat token index
    h*res+w that roll(-shift) + window_partition (:314-320) places there."""
    n = res // ws
    a = torch.arange(res)
    src = (a + shift) % res                       # shifted[p] = x[(p+shift) % res]
    hh = src.view(n, ws)                          # [window-row, in-window row] -> source row
    tok = hh[:, None, :, None] * res + hh[None, :, None, :]      # (n, n, ws, ws)
    return tok.reshape(n * n, ws * ws)


def window_attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: torch.Tensor,
                          mask: torch.Tensor | None, scale: float) -> torch.Tensor:
    """q,k,v: (B, nW, h, N, hd); bias (h,N,N); mask (nW,N,N) or None → (B,nW,h,N,hd).
    WindowAttention.forward :193-211 without the two Linear layers."""
    s = torch.matmul(q * scale, k.transpose(-1, -2)) + bias[None, None]
    if mask is not None:
        s = s + mask[None, :, None]
    return torch.matmul(torch.softmax(s, dim=-1), v)


def rel_pos_bias(sd: SD, prefix: str) -> torch.Tensor:
    """(h, N, N) bias gathered from the (529,h) table by the stored index buffer (:196-198)."""
    table = sd[prefix + ".relative_position_bias_table"]
    idx = sd[prefix + ".relative_position_index"]
    n = idx.shape[0]
    return table[idx.reshape(-1)].reshape(n, n, -1).permute(2, 0, 1).contiguous()


def swin_block(sd: SD, prefix: str, x: torch.Tensor, res: int, ws: int, shift: int, heads: int
               ) -> torch.Tensor:
    """SwinTransformerBlock.forward :303-340 expressed with an explicit token gather/scatter."""
    B, L, C = x.shape
    hd = C // heads
    tok = window_token_index(res, ws, shift)                       # (nW, N)
    nW, N = tok.shape
    y = _ln(sd, prefix + ".norm1", x)
    yw = y[:, tok.reshape(-1)]                                     # (B, nW*N, C) window order
    qkv = _linear(sd, prefix + ".attn.qkv", yw).view(B, nW, N, 3, heads, hd)
    q, k, v = (qkv[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))   # (B,nW,h,N,hd)
    mask = sd.get(prefix + ".attn_mask") if shift > 0 else None
    o = window_attention_core(q, k, v, rel_pos_bias(sd, prefix + ".attn"), mask, hd ** -0.5)
    o = o.permute(0, 1, 3, 2, 4).reshape(B, nW * N, C)
    o = _linear(sd, prefix + ".attn.proj", o)
    a = torch.empty_like(x)
    a[:, tok.reshape(-1)] = o                                      # window_reverse + roll back
    x = x + a
    hdn = _linear(sd, prefix + ".mlp.fc1", _ln(sd, prefix + ".norm2", x))
    hdn = F.gelu(hdn)                                              # exact erf GELU (nn.GELU :87)
    return x + _linear(sd, prefix + ".mlp.fc2", hdn)


def patch_merging(sd: SD, prefix: str, x: torch.Tensor, res: int) -> torch.Tensor:
    """PatchMerging.forward :377-398 — 2×2 neighbours in order (0,0),(1,0),(0,1),(1,1)."""
    B, L, C = x.shape
    g = x.view(B, res // 2, 2, res // 2, 2, C)
    cat = torch.cat([g[:, :, 0, :, 0], g[:, :, 1, :, 0], g[:, :, 0, :, 1], g[:, :, 1, :, 1]], -1)
    cat = cat.reshape(B, (res // 2) ** 2, 4 * C)
    return F.linear(_ln(sd, prefix + ".norm", cat), sd[prefix + ".reduction.weight"])


def patch_embed(sd: SD, g, img: torch.Tensor) -> torch.Tensor:
    """PatchEmbed.forward :511-519: conv k=s=patch → (B, L, C) → LayerNorm."""
    P = "swin_transf.patch_embed"
    y = F.conv2d(img, sd[P + ".proj.weight"], sd[P + ".proj.bias"], stride=g.swin_patch_size)
    return _ln(sd, P + ".norm", y.flatten(2).transpose(1, 2))


def swin_forward(sd: SD, g, img: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    """SwinTransformer.forward_features :630-642 → (B, (res_last)², C_last)."""
    x = patch_embed(sd, g, img)
    if taps is not None:
        taps["patch_embed"] = x
    for s, depth in enumerate(g.swin_depths):
        res, ws = g.stage_res(s), g.stage_window(s)
        for b in range(depth):
            x = swin_block(sd, f"swin_transf.layers.{s}.blocks.{b}", x, res, ws,
                           g.stage_shift(s, b), g.swin_num_heads[s])
            if taps is not None:
                taps[f"s{s}b{b}"] = x
        if s < len(g.swin_depths) - 1:
            x = patch_merging(sd, f"swin_transf.layers.{s}.downsample", x, res)
            if taps is not None:
                taps[f"merge{s}"] = x
    return _ln(sd, "swin_transf.norm", x)


# =================================================================================================
# Expansion encoder (layers.py StaticExpansionBlock :45-102, EncoderLayer :118-123)
# =================================================================================================
def static_expansion(sd: SD, prefix: str, x: torch.Tensor, groups: Sequence[int],
                     key_valid: torch.Tensor, eps: float = 1e-9) -> torch.Tensor:
    """x (B,S,d); key_valid (B,S) bool — False on padded (trailing) encoder positions."""
    d = x.shape[-1]
    Q = sd[prefix + ".query_exp_vectors.weight"]                   # (nq, d)
    Bv = sd[prefix + ".bias_exp_vectors.weight"]
    z = torch.matmul(Q, _linear(sd, prefix + ".key_embed", x).transpose(-1, -2)) / math.sqrt(d)
    kv = key_valid[:, None, :].to(z.dtype)                         # (B,1,S)
    pos = torch.relu(z) * kv
    neg = torch.relu(-z) * kv
    pos = pos / (pos.sum(-1, keepdim=True) + eps)
    neg = neg / (neg.sum(-1, keepdim=True) + eps)
    A = torch.matmul(pos, _linear(sd, prefix + ".class_a_embed", x)) + Bv
    Bm = torch.matmul(neg, _linear(sd, prefix + ".class_b_embed", x)) + Bv
    zt = z.transpose(-1, -2)                                       # (B,S,nq)  — not masked (:67-68)
    pb, nb = torch.relu(zt), torch.relu(-zt)
    lo = 0
    pbn, nbn = torch.empty_like(pb), torch.empty_like(nb)
    for n in groups:                                               # per-group L1 norm (:70-79)
        sl = slice(lo, lo + n)
        pbn[..., sl] = pb[..., sl] / (pb[..., sl].sum(-1, keepdim=True) + eps)
        nbn[..., sl] = nb[..., sl] / (nb[..., sl].sum(-1, keepdim=True) + eps)
        lo += n
    A2 = torch.matmul(pbn, A) / len(groups)
    B2 = torch.matmul(nbn, Bm) / len(groups)
    sel = torch.sigmoid(_linear(sd, prefix + ".selector_embed", x))
    return sel * A2 + (1 - sel) * B2


def feed_forward(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward.forward :305-308."""
    return _linear(sd, prefix + ".linear_2", torch.relu(_linear(sd, prefix + ".linear_1", x)))


def encoder_forward(sd: SD, g, feats: torch.Tensor, enc_num_pads: Sequence[int]) -> torch.Tensor:
    """forward_enc after the backbone: End_ExpansionNet_v2.py:82-101 /
    ExpansionNet_v2.py:52-70.  feats (B,S,feat_dim) → (B,S,d)."""
    B, S, _ = feats.shape
    valid = torch.arange(S)[None, :] < (S - torch.as_tensor(list(enc_num_pads)))[:, None]
    x = _linear(sd, "input_linear", feats)
    outs = []
    for i in range(g.N_enc):
        p = f"encoders.{i}"
        x = x + static_expansion(sd, p + ".stc_exp", _ln(sd, p + ".norm_1", x),
                                 g.num_exp_enc_list, valid)
        x = x + feed_forward(sd, p + ".ff", _ln(sd, p + ".norm_2", x))
        outs.append(x)
    x = x + _linear(sd, "enc_reduce_group", torch.cat(outs, -1))
    return _ln(sd, "enc_reduce_norm", x)


# =================================================================================================
# Decoder, full-prefix recompute exactly as the reference does it
# (layers.py DynamicExpansionBlock :152-204, MultiHeadAttention :266-295, DecoderLayer :222-248)
# =================================================================================================
def dynamic_expansion(sd: SD, prefix: str, x: torch.Tensor, n_exp: int, causal: torch.Tensor,
                      eps: float = 1e-9) -> torch.Tensor:
    """x (N,T,d); causal (N,T,T) 0/1 mask = tril with padded rows AND columns zeroed."""
    N, T, d = x.shape
    cond = _linear(sd, prefix + ".cond_embed", x)[:, :, None, :]           # (N,T,1,d)
    Q = (sd[prefix + ".query_exp_vectors.weight"][None, None] + cond).reshape(N, T * n_exp, d)
    Bv = (sd[prefix + ".bias_exp_vectors.weight"][None, None] + cond).reshape(N, T * n_exp, d)
    K = _linear(sd, prefix + ".key_linear", x)
    z = torch.matmul(Q, K.transpose(-1, -2)) / math.sqrt(d)                # (N, T*e, T)
    m1 = causal[:, :, None, :].expand(N, T, n_exp, T).reshape(N, T * n_exp, T)
    pos, neg = torch.relu(z) * m1, torch.relu(-z) * m1
    pos = pos / (pos.sum(-1, keepdim=True) + eps)
    neg = neg / (neg.sum(-1, keepdim=True) + eps)
    A = torch.matmul(pos, _linear(sd, prefix + ".class_a_embed", x))
    Bm = torch.matmul(neg, _linear(sd, prefix + ".class_b_embed", x))
    m2 = causal[:, :, :, None].expand(N, T, T, n_exp).reshape(N, T, T * n_exp)
    zt = z.transpose(-1, -2)
    pb, nb = torch.relu(zt) * m2, torch.relu(-zt) * m2
    pb = pb / (pb.sum(-1, keepdim=True) + eps)
    nb = nb / (nb.sum(-1, keepdim=True) + eps)
    A2 = torch.matmul(pb, A + Bv)
    B2 = torch.matmul(nb, Bm + Bv)
    sel = torch.sigmoid(_linear(sd, prefix + ".selector_embed", x))
    return sel * A2 + (1 - sel) * B2


def cross_attention(sd: SD, prefix: str, q_in: torch.Tensor, mem: torch.Tensor, heads: int,
                    allow: torch.Tensor) -> torch.Tensor:
    """q_in (N,T,d), mem (N,S,d), allow (N,T,S) 0/1; masked entries get -1e4 (:285-287)."""
    N, T, d = q_in.shape
    S = mem.shape[1]
    dk = d // heads
    q = _linear(sd, prefix + ".Wq", q_in).view(N, T, heads, dk).transpose(1, 2)
    k = _linear(sd, prefix + ".Wk", mem).view(N, S, heads, dk).transpose(1, 2)
    v = _linear(sd, prefix + ".Wv", mem).view(N, S, heads, dk).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dk)
    s = s.masked_fill(allow[:, None] == 0, -1e4)
    o = torch.matmul(torch.softmax(s, -1), v).transpose(1, 2).reshape(N, T, d)
    return _linear(sd, prefix + ".out_linear", o)


def _dec_masks(N: int, T: int, S: int, dec_pads: Sequence[int], enc_pads: Sequence[int]
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils/masking.py:22-47 expressed with lengths."""
    dlen = T - torch.as_tensor(list(dec_pads))
    slen = S - torch.as_tensor(list(enc_pads))
    t = torch.arange(T)
    row_ok = (t[None, :] < dlen[:, None])                                   # (N,T)
    causal = (t[None, :, None] >= t[None, None, :]) & row_ok[:, :, None] & row_ok[:, None, :]
    cross = row_ok[:, :, None] & (torch.arange(S)[None, None, :] < slen[:, None, None])
    return causal.float(), cross.float()


def decoder_forward(sd: SD, g, cross: torch.Tensor, enc_num_pads: Sequence[int],
                    tokens: torch.Tensor, dec_num_pads: Sequence[int], log_softmax: bool,
                    end_to_end: bool = True) -> torch.Tensor:
    """forward_dec: End_ExpansionNet_v2.py:103-138 (end_to_end: encoder pads forced to 0 at :107)
    / ExpansionNet_v2.py:72-107.  tokens (N,T) int64 → (N,T,V)."""
    N, T = tokens.shape
    S = cross.shape[1]
    if end_to_end:
        enc_num_pads = [0] * N
    causal, allow = _dec_masks(N, T, S, dec_num_pads, enc_num_pads)
    y = sd["out_embedder.embed.weight"][tokens] * math.sqrt(g.d_model) + sd["pos_encoder.weight"][:T]
    outs = []
    for i in range(g.N_dec):
        p = f"decoders.{i}"
        y = y + dynamic_expansion(sd, p + ".dyn_exp", _ln(sd, p + ".norm_1", y), g.num_exp_dec, causal)
        y = y + cross_attention(sd, p + ".mha", _ln(sd, p + ".norm_2", y), cross, g.num_heads, allow)
        y = y + feed_forward(sd, p + ".ff", _ln(sd, p + ".norm_3", y))
        outs.append(y)
    y = y + _linear(sd, "dec_reduce_group", torch.cat(outs, -1))
    y = _linear(sd, "vocab_linear", _ln(sd, "dec_reduce_norm", y))
    return torch.log_softmax(y, -1) if log_softmax else y


# =================================================================================================
# Whole-model entry points  (captioning_model.py)
# =================================================================================================
def forward_enc(sd: SD, g, enc_x: torch.Tensor, enc_num_pads: Sequence[int], end_to_end: bool = True
                ) -> torch.Tensor:
    if end_to_end:
        feats = swin_forward(sd, g, enc_x)
        return encoder_forward(sd, g, feats, [0] * enc_x.shape[0])
    return encoder_forward(sd, g, enc_x, enc_num_pads)


def forward_teacher(sd: SD, g, enc_x, dec_x, enc_num_pads, dec_num_pads, log_softmax=False,
                    end_to_end=True) -> torch.Tensor:
    """CaptioningModel.forward mode='forward' :27-30."""
    mem = forward_enc(sd, g, enc_x, enc_num_pads, end_to_end)
    return decoder_forward(sd, g, mem, enc_num_pads, dec_x, dec_num_pads, log_softmax, end_to_end)


def beam_search(sd, g, enc_x: torch.Tensor, enc_num_pads: Sequence[int], sos_idx: int,
                eos_idx: int, beam_size: int = 3, how_many_outputs: int = 1, max_seq_len: int = 20,
                end_to_end: bool = True, trace: list | None = None, draw_fn=None
                ) -> Tuple[List[List[List[int]]], torch.Tensor]:
    """captioning_model.py:111-241, with plain tensors.  `draw_fn=None` is the 'max' branch (top-k
    candidates); `draw_fn(step, log_probs (rows, V)) -> LongTensor (rows, k)` supplies the candidate words of
    the 'sample' branch (:128-131,166-168, where the reference calls multinomial) — tests inject the draws the
    device made, so the bookkeeping around them can be compared exactly.

    `sd` may also be a LIST of state dicts: the ensemble search of ensemble_captioning_model.py:48-83 —
    every model encodes and decodes on its own, the per-step distribution is log(mean_m softmax(logits_m))
    and the search itself (:87-291, a copy of the single-model loop) is unchanged.

    State per image b and beam j:  toks[b,j,:t], lps[b,j,:t] (per-token log-probs, slot 0 = 0),
    n_elem[b,j] (length incl. SOS and EOS).  `trace`, if given, receives per-step
    (top-1 − top-2) log-prob margins of the k·k selection for parity diagnostics."""
    assert how_many_outputs <= beam_size
    B, k = enc_x.shape[0], beam_size
    enc_num_pads = list(enc_num_pads)
    sds = list(sd) if isinstance(sd, (list, tuple)) else [sd]
    mems = [forward_enc(s_, g, enc_x, enc_num_pads, end_to_end) for s_ in sds]

    def logprobs(mem_list, pads, toks_, dec_pads):
        if len(sds) == 1:
            return decoder_forward(sds[0], g, mem_list[0], pads, toks_, dec_pads, True, end_to_end)
        probs = [torch.softmax(decoder_forward(s_, g, m_, pads, toks_, dec_pads, False, end_to_end), dim=-1)
                 for s_, m_ in zip(sds, mem_list)]
        return torch.stack(probs, 0).mean(0).log()                          # ensemble_captioning_model.py:66-83

    # ---- first step: one distribution per image, its top-k seed the beams (:117-140)
    lp0 = logprobs(mems, enc_num_pads, torch.full((B, 1), sos_idx, dtype=torch.long), [0] * B)[:, 0]
    if draw_fn is None:
        v0, w0 = torch.topk(lp0, k, dim=-1)
    else:
        w0 = draw_fn(0, lp0).long()
        v0 = lp0.gather(-1, w0)
    toks = torch.stack([torch.full((B, k), sos_idx, dtype=torch.long), w0], -1)      # (B,k,2)
    lps = torch.stack([torch.zeros(B, k), v0], -1)
    cumul = lps.sum(-1)
    n_elem = torch.full((B, k), 2, dtype=torch.long)
    mems_k = [m_[:, None].expand(B, k, *m_.shape[1:]).reshape(B * k, *m_.shape[1:]) for m_ in mems]
    pads_k = [p for p in enc_num_pads for _ in range(k)]
    bidx = torch.arange(B)[:, None]

    for t in range(2, max_seq_len):
        lp = logprobs(mems_k, pads_k, toks.reshape(B * k, t), (t - n_elem).reshape(-1).tolist())[:, t - 1]
        if draw_fn is None:
            cv, cw = torch.topk(lp, k, dim=-1)                              # (B*k, k)
        else:
            cw = draw_fn(t - 1, lp).long()
            cv = lp.gather(-1, cw)
        cv, cw = cv.view(B, k, k).clone(), cw.view(B, k, k)
        done = (toks == eos_idx).any(-1)                                    # (B,k)
        cv[:, :, 0] = torch.where(done, torch.zeros(()), cv[:, :, 0])       # keep score (:193)
        cv[:, :, 1:] = torch.where(done[:, :, None], torch.full((), -999.0), cv[:, :, 1:])
        total = (cumul[:, :, None] + cv).reshape(B, k * k)
        tv, ti = torch.topk(total, k, dim=-1)
        if trace is not None:
            srt = torch.sort(total, dim=-1, descending=True).values
            trace.append((srt[:, k - 1] - srt[:, k]).min().item() if k * k > k else float("inf"))
        parent, word = ti // k, ti % k
        new_w = cw[bidx, parent, word]
        new_lp = cv[bidx, parent, word]
        had_eos = done[bidx, parent]
        toks = torch.cat([toks[bidx, parent], new_w[..., None]], -1)
        lps = torch.cat([lps[bidx, parent], new_lp[..., None]], -1)
        cumul = lps.sum(-1)
        n_elem = n_elem[bidx, parent] + (~had_eos).long()
        if bool((n_elem != t + 1).all()):
            break

    score = cumul / n_elem                                                  # (:226)
    _, order = torch.topk(score, k, dim=-1)
    out_tok: List[List[List[int]]] = []
    out_lp = []
    for b in range(B):
        row = []
        for j in range(how_many_outputs):
            i = int(order[b, j])
            n = int(n_elem[b, i])
            row.append(toks[b, i, :n].tolist())
            out_lp.append(lps[b, i, :n])
        out_tok.append(row)
    out_lp = torch.nn.utils.rnn.pad_sequence(out_lp, batch_first=True).view(B, how_many_outputs, -1)
    return out_tok, out_lp


def greedy(sd: SD, g, enc_x, enc_num_pads, sos_idx, eos_idx, max_seq_len=20, end_to_end=True):
    """Greedy decoding = beam_size 1 through the same search."""
    return beam_search(sd, g, enc_x, enc_num_pads, sos_idx, eos_idx, 1, 1, max_seq_len, end_to_end)


# =================================================================================================
# caller-side helpers (utils/image_utils.py:5-23, utils/language_utils.py:82-93)
# =================================================================================================
def preprocess_image(path: str, img_size: int = 384) -> torch.Tensor:
    """PIL bilinear resize → /255 → ImageNet normalise.  Non-RGB inputs become an all-black
    RGB image (the reference calls PIL_Image.new, image_utils.py:18-19)."""
    import numpy as np
    from PIL import Image
    im = Image.open(path)
    if im.mode != "RGB":
        im = Image.new("RGB", im.size)
    im = im.resize((img_size, img_size), Image.BILINEAR)
    a = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    mean = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    return ((a - mean) / std)[None]


def tokens2description(tokens: Sequence[int], idx2word: Sequence[str], sos_idx: int, eos_idx: int
                       ) -> str:
    words = []
    for t in tokens:
        if t == sos_idx:
            continue
        if t == eos_idx:
            break
        words.append(idx2word[t])
    words[-1] = words[-1] + "."
    return " ".join(words).capitalize()
