"""Search / API layer.

Two call shapes of the reference are served:
  * legacy:      model(enc_x, dec_x, enc_x_num_pads, dec_x_num_pads, apply_log_softmax, mode=..., **kw)
                 and model.beam_search(...)            — legacy_models/captioning_model.py:24-57,111-241
                 (what demo.py:124-129 and test.py:209-214 call)
  * refactored:  Captioner(beam_search_args, model=...)(enc_x, enc_x_num_pads=..., mode="beam_search")
                                                       — models/captioning_model.py:40-110

The search itself runs on the GPU: one incremental decoder step per new token (engine.py) and the
beam bookkeeping of captioning_model.py:172-223 in odic_beam_step.  The host loop only enqueues
steps; it looks at the device-side `done` flag every few steps (the all-beams-finished state is a
fixed point of the step, so running a few extra steps cannot change the result).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import engine as _engine
from . import ops

_DONE_POLL = 4      # host looks at the device `done` flag every this many steps


class CaptioningModel(nn.Module):
    """Base class: mode dispatch + GPU search.  Subclasses provide `_engines()`, `forward_enc`,
    `_cross_kv(mem)` and set `self.rank` (legacy_models/captioning_model.py:11-16)."""

    def __init__(self, apply_log_softmax: bool = False):
        super().__init__()
        self.rank = None
        self.apply_log_softmax = apply_log_softmax
        self._eng_cache = None
        self.precision = "fp32"
        self.encoder_precision = None       # None = follow `precision`
        self.calibration_images = None      # fp8 mode: images the static activation scales are calibrated on
        self.sampling_seed = 0              # Philox key of the device-side draws ('sample' / 'sampling' modes)
        self._sampling_calls = 0
        self._draw_log = None               # list → every sampled-beam-search draw is appended (tests)

    # ------------------------------------------------------------------ engine cache plumbing
    def check_required_attributes(self):
        if self.rank is None:
            raise NotImplementedError("Subclass must assign the rank integer according to the GPU group")

    def set_precision(self, precision: str, encoder_precision: Optional[str] = None,
                      calibration_images: Optional[torch.Tensor] = None) -> "CaptioningModel":
        """'fp32' (default; exact-fp32 MFMA, parity mode), 'bf16' (backbone GEMMs + window attention in bf16 with
        fp32 accumulation and an fp32 residual stream; expansion-encoder products in bf16 too unless
        `encoder_precision='fp32'`) or 'fp8' (BASELINE.json configs[4]: Swin-block GEMMs qkv / fc1 / fc2 on the fp8
        MFMA with statically calibrated scales, fp16 qkv / attention activations, everything else as 'bf16'), or
        'x3' — the near-exact fast mode: every backbone / encoder contraction on split-fp16 operands (hi + lo pairs,
        22 significand bits, three fp16 MFMAs per product with fp32 accumulation; 'bf16x3' is accepted as an alias
        for the name the technique usually goes by), fp32 residual streams, exact-erf GELU — the mode that reproduces
        the fp32 (= reference) captions at several times the exact-fp32 MFMA rate.  The decoder is always fp32.
        `calibration_images` (fp8 only; fp32 [n,3,H,W], preprocessed like the inputs): the images the static per-tensor
        activation scales are measured on (amax x 1.25 → the e4m3 maximum).  Default: two synthetic noise images —
        fine for synthetic benchmarks, NOT for real photographs, whose LayerNorm / GELU ranges differ; the fp8 casts
        saturate silently at ±448, so calibrate on a sample of the deployment distribution and check
        `fp8_saturation_report(images)` on held-out images."""
        if precision == "bf16x3":
            precision = "x3"
        if encoder_precision == "bf16x3":
            encoder_precision = "x3"
        if precision not in ("fp32", "bf16", "fp8", "x3") or (encoder_precision or "bf16") not in ("fp32", "bf16", "x3"):
            raise ValueError("precision must be 'fp32', 'bf16', 'fp8' or 'x3' (encoder_precision 'fp32', 'bf16' or 'x3')")
        if calibration_images is not None and precision != "fp8":
            raise ValueError("calibration_images only applies to precision='fp8'")
        same_cal = (calibration_images is None and self.calibration_images is None) or \
            (calibration_images is not None and self.calibration_images is not None and
             calibration_images.shape == self.calibration_images.shape and
             bool(torch.equal(calibration_images.detach().cpu(), self.calibration_images)))
        if (precision, encoder_precision) != (self.precision, self.encoder_precision) or not same_cal:
            self.precision, self.encoder_precision = precision, encoder_precision
            self.calibration_images = None if calibration_images is None else calibration_images.detach().float().cpu().clone()
            self._eng_cache = None
        return self

    def fp8_saturation_report(self, images: torch.Tensor) -> dict:
        """fp8 mode: how the activation ranges of `images` compare with the calibrated scales — per quantised tensor
        (LayerNorm outputs and GELU hidden of every Swin block) the ratio observed amax / representable range; a ratio
        above 1 means the static cast clips there (End_ExpansionNet_v2 only)."""
        if self.precision != "fp8":
            raise RuntimeError("fp8_saturation_report needs set_precision('fp8')")
        swin = self._engines()[0]
        return swin.fp8_saturation(images.to(swin.device, torch.float32))

    def _next_sampling_seed(self) -> int:
        """A fresh Philox key per sampling call (so repeated calls differ), reproducible from `sampling_seed`."""
        self._sampling_calls += 1
        return (int(self.sampling_seed) * 0x9E3779B97F4A7C15 + self._sampling_calls) & 0xFFFFFFFFFFFFFFFF

    def _apply(self, fn, *a, **k):
        self._eng_cache = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._eng_cache = None
        return super().load_state_dict(*a, **k)

    def _device(self) -> torch.device:
        """GPU the HIP engines of this model live on.  A module moved with `.to('cuda:N')` runs there.  A module
        left on the host — demo.py:68-104 builds the model with rank='cpu', loads the checkpoint and never calls
        `.to()` — keeps its nn.Parameters on the host and the engines pack their own copy of the weights onto
        `rank` if that names a GPU, else cuda:0; inputs are moved there, results come back on the input's
        device.  Arithmetic is HIP in every case: without a GPU this raises (there is no CPU fallback)."""
        dv = next(self.parameters()).device
        if dv.type == "cuda":
            return dv
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP path needs a GPU and none is visible; there is no CPU fallback in this package")
        r = self.rank
        if isinstance(r, int) and not isinstance(r, bool):
            return torch.device("cuda", r)
        if isinstance(r, torch.device) and r.type == "cuda":
            return r
        if isinstance(r, str) and r.startswith("cuda"):
            return torch.device(r)
        return torch.device("cuda", 0)

    # ------------------------------------------------------------------ to be provided
    def forward_enc(self, enc_input, enc_input_num_pads):
        raise NotImplementedError

    def _captioner_engine(self) -> "_engine.CaptionerEngine":
        raise NotImplementedError

    def _enc_lens(self, n: int, S: int, enc_input_num_pads) -> torch.Tensor:
        raise NotImplementedError

    # ------------------------------------------------------------------ decoder API
    def forward_dec(self, cross_input, enc_input_num_pads, dec_input, dec_input_num_pads,
                    apply_log_softmax: bool = False):
        """(N,T) token ids → (N,T,V) logits / log-probs, teacher forced (End_ExpansionNet_v2.py:103-138).
        Runs the incremental step T times; padded positions reproduce the reference's masked rows."""
        eng = self._captioner_engine()
        dv = eng.device
        cross_input = cross_input.to(dv, torch.float32)
        dec_input = dec_input.to(dv, torch.int64)
        N, T = dec_input.shape
        S = cross_input.shape[1]
        enc_len = self._enc_lens(N, S, enc_input_num_pads)
        kv = eng.project_kv(cross_input)
        st = eng.new_state(N, 1, T + 1, kv, enc_len)
        st.anc.copy_(torch.arange(N, dtype=torch.int32, device=dv)[:, None].expand(N, T + 1))
        dec_len = torch.as_tensor([T - int(p) for p in _as_list(dec_input_num_pads, N)], device=dv)
        V = eng.g.vocab_size
        out = torch.empty(N, T, V, dtype=torch.float32, device=dv)
        for t in range(T):
            st.pos.fill_(t)
            st.next_tok.copy_(dec_input[:, t])
            st.row_valid.copy_((dec_len > t).to(torch.int32))
            eng.step_logits(st)
            if apply_log_softmax:
                ops.logsoftmax_topk(st.logits, V, out[:, t], T * V, st.cand_val, st.cand_idx, N, V, 1)
            else:
                out[:, t].copy_(st.logits)
        return out

    def forward(self, enc_x, dec_x=None, enc_x_num_pads=[0], dec_x_num_pads=[0], apply_log_softmax=False,
                mode="forward", **kwargs):
        if mode == "forward":
            x = self.forward_enc(enc_x, enc_x_num_pads)
            y = self.forward_dec(x, enc_x_num_pads, dec_x, dec_x_num_pads, apply_log_softmax)
            return y.to(enc_x.device) if isinstance(enc_x, torch.Tensor) else y
        assert ("sos_idx" in kwargs.keys() or "eos_idx" in kwargs.keys()), \
            "sos and eos must be provided in case of batch sampling or beam search"
        sos_idx = kwargs.get("sos_idx", -999)
        eos_idx = kwargs.get("eos_idx", -999)
        if mode == "beam_search":
            return self.beam_search(enc_x, enc_x_num_pads, sos_idx=sos_idx, eos_idx=eos_idx,
                                    beam_size=kwargs.get("beam_size", 5),
                                    how_many_outputs=kwargs.get("how_many_outputs", 1),
                                    max_seq_len=kwargs.get("beam_max_seq_len", 20),
                                    sample_or_max=kwargs.get("sample_or_max", "max"))
        if mode == "sampling":
            return self.get_batch_multiple_sampled_prediction(
                enc_x, enc_x_num_pads, num_outputs=kwargs.get("how_many_outputs", 1), sos_idx=sos_idx,
                eos_idx=eos_idx, max_seq_len=kwargs.get("sample_max_seq_len", 20))
        raise ValueError(f"unknown mode {mode!r}")

    def get_batch_multiple_sampled_prediction(self, enc_input, enc_input_num_pads, num_outputs, sos_idx, eos_idx,
                                              max_seq_len):
        """mode='sampling' (legacy_models/captioning_model.py:59-109): `num_outputs` ancestral samples per
        image.  Each step runs the incremental decoder and ONE kernel that normalises the row and draws the
        next word on the device (odic_logsoftmax_sample, Philox noise keyed by `self.sampling_seed`, the row and
        the position) — no host round trip, no ATen launch.  The draws cannot be those of the reference's CPU
        generator; what IS checkable — and tested — is that every reported log-prob equals the teacher-forced
        log-prob of the returned sequence, and that the draw frequencies follow the distribution."""
        eng = self._captioner_engine()
        dv = eng.device
        mem = self.forward_enc(enc_input, enc_input_num_pads)
        bs, S, _ = mem.shape
        N = bs * num_outputs
        V = eng.g.vocab_size
        enc_len = self._enc_lens(bs, S, enc_input_num_pads)
        st = eng.new_state(bs, num_outputs, max_seq_len + 1, eng.project_kv(mem), enc_len)
        st.anc.copy_(torch.arange(N, dtype=torch.int32, device=dv)[:, None].expand(N, max_seq_len + 1))
        st.next_tok.fill_(sos_idx)
        st.row_valid.fill_(1)
        draw_val = torch.empty(N, 1, dtype=torch.float32, device=dv)
        draw_idx = torch.empty(N, 1, dtype=torch.int32, device=dv)
        toks = torch.full((N, max_seq_len + 1), sos_idx, dtype=torch.int64, device=dv)
        lps = torch.zeros(N, max_seq_len + 1, dtype=torch.float32, device=dv)
        where_eos = torch.full((N,), max_seq_len, dtype=torch.int64, device=dv)
        finished = torch.zeros(N, dtype=torch.bool, device=dv)
        seed = self._next_sampling_seed()
        t = 0
        while t < max_seq_len:
            st.pos.fill_(t)
            eng.step_logits(st)
            ops.logsoftmax_sample(st.logits, V, None, 0, draw_val, draw_idx, N, V, 1, seed, st.pos)
            nxt = draw_idx[:, 0].long()
            toks[:, t + 1] = nxt
            lps[:, t + 1] = draw_val[:, 0]
            t += 1
            hit = nxt == eos_idx
            where_eos = torch.minimum(where_eos, torch.where(hit, torch.full_like(where_eos, t), where_eos))
            finished |= hit
            st.next_tok.copy_(nxt)
            if t % _DONE_POLL == 0 and bool(finished.all()):
                break
        toks_h, eos_h = toks.cpu(), where_eos.cpu()
        res = [[toks_h[i * num_outputs + j, :int(eos_h[i * num_outputs + j]) + 1].tolist()
                for j in range(num_outputs)] for i in range(bs)]
        # the reference stops at the step every sequence has finished (:96-97) and pads to the longest one: the
        # host only looks every _DONE_POLL steps, so trim the columns the extra steps added
        width = int(eos_h.max()) + 1
        ar = torch.arange(width, device=dv)[None, :]
        probs = lps[:, :width].masked_fill(ar > where_eos[:, None], 0.0).reshape(bs, num_outputs, -1)
        return res, probs.to(enc_input.device) if isinstance(enc_input, torch.Tensor) else probs

    # ------------------------------------------------------------------ search
    def beam_search(self, enc_input, enc_input_num_pads, sos_idx, eos_idx, beam_size=3, how_many_outputs=1,
                    max_seq_len=20, sample_or_max="max"):
        assert (how_many_outputs <= beam_size), "requested output per sequence must be lower than beam width"
        assert (sample_or_max == "max" or sample_or_max == "sample"), \
            "argument must be chosen between 'max' and 'sample'"
        mem = self.forward_enc(enc_input, enc_input_num_pads)
        toks, lp = self._search_from_memory(mem, enc_input_num_pads, sos_idx, eos_idx, beam_size, how_many_outputs,
                                            max_seq_len, sample=(sample_or_max == "sample"))
        return toks, (lp.to(enc_input.device) if isinstance(enc_input, torch.Tensor) else lp)

    def _search_from_memory(self, mem, enc_input_num_pads, sos_idx, eos_idx, beam_size, how_many_outputs,
                            max_seq_len, sample: bool = False) -> Tuple[List[List[List[int]]], torch.Tensor]:
        eng = self._captioner_engine()
        dv = eng.device
        B, S, _ = mem.shape
        k = beam_size
        steps = max(1, max_seq_len - 1)            # positions 0 .. steps-1 are fed; prefixes reach steps+1
        T = steps + 1
        enc_len = self._enc_lens(B, S, enc_input_num_pads)
        st = eng.new_state(B, k, T, eng.project_kv(mem), enc_len)
        # start state; for the deterministic search also the embedded start token as the input of position 0
        ops.beam_reset(st.beam_state, B, k, T, sos_idx, emb=None if sample else st.emb)
        V = eng.g.vocab_size
        seed = self._next_sampling_seed() if sample else 0
        for t in range(steps):
            if sample:
                # 'sample' variant (captioning_model.py:128-131,166-168): the k candidates of every beam are
                # drawn without replacement from its distribution instead of being its top-k — on the device
                eng.step_logits(st)
                ops.logsoftmax_sample(st.logits, V, None, 0, st.cand_val, st.cand_idx, st.N, V, k, seed, st.pos)
                if self._draw_log is not None:                    # test hook: the draws, for replay in the oracle
                    self._draw_log.append(st.cand_idx.cpu().clone())
                ops.beam_step(st.cand_val, st.cand_idx, st.beam_state, st.n_img, st.beams, st.T, eos_idx)
            else:
                eng.beam_step(st, eos_idx)
            if t >= 1 and (t + 1) % _DONE_POLL == 0 and t + 1 < steps and int(st.done.item()):
                break
        order = torch.empty(B, k, dtype=torch.int32, device=dv)
        score = torch.empty(B, k, dtype=torch.float32, device=dv)
        ops.beam_finalize(st.beam_state, order, score, B, k)
        order_h = order.cpu()
        n_elem_h = st.n_elem.view(B, k).cpu()
        tokens_h = st.tokens.cpu()
        res_tok: List[List[List[int]]] = []
        lp_rows = []
        for b in range(B):
            per = []
            for j in range(how_many_outputs):
                i = int(order_h[b, j])
                n = int(n_elem_h[b, i])
                per.append(tokens_h[b, i, :n].tolist())
                lp_rows.append(st.logprobs[b, i, :n])
            res_tok.append(per)
        lp = torch.nn.utils.rnn.pad_sequence(lp_rows, batch_first=True).view(B, how_many_outputs, -1)
        return res_tok, lp


def _as_list(pads, n: int) -> List[int]:
    if isinstance(pads, torch.Tensor):
        pads = pads.tolist()
    pads = list(pads)
    if len(pads) != n:
        raise RuntimeError(f"expected {n} pad counts, got {len(pads)}")
    return [int(p) for p in pads]


# =================================================================================================
# refactored API  (models/captioning_model.py:40-110, models/End_ExpansionNet_v2.py:311-354)
# =================================================================================================
class Captioner:
    def __init__(self, beam_search_args, model=None, split_encoder=False, apply_log_softmax=False, encoder=None,
                 decoder=None):
        self.rank = None
        self.split_encoder = split_encoder
        if self.split_encoder:
            self.encoder, self.decoder = encoder, decoder
            if self.encoder is None or self.decoder is None:
                raise ValueError("Both encoder and decoder must be supplied in Split Encoder mode")
            raise NotImplementedError("split encoder/decoder modules exist for the reference's FX int8 "
                                      "quantisation route, which is out of scope (SURVEY §2 row 21)")
        self.model = model
        if self.model is None:
            raise ValueError("An Encoder-Decoder model must be provided")
        self.beam_search_args = beam_search_args
        self.apply_log_softmax = apply_log_softmax

    def __call__(self, enc_x, dec_x=None, enc_x_num_pads=[0], dec_x_num_pads=[0], mode="beam_search"):
        assert ("sos_idx" in self.beam_search_args.keys() or "eos_idx" in self.beam_search_args.keys()), \
            "sos and eos must be provided in case of batch sampling or beam search"
        sos_idx = self.beam_search_args["sos_idx"]
        eos_idx = self.beam_search_args["eos_idx"]
        a = self.beam_search_args
        if mode == "beam_search":
            self.apply_log_softmax = True
            return self.model.beam_search(enc_x, enc_x_num_pads, sos_idx=sos_idx, eos_idx=eos_idx,
                                          beam_size=a.get("beam_size", 5),
                                          how_many_outputs=a.get("how_many_outputs", 1),
                                          max_seq_len=a.get("beam_max_seq_len", 20),
                                          sample_or_max=a.get("sample_or_max", "max"))
        if mode == "sampling":
            self.apply_log_softmax = True
            return self.model.get_batch_multiple_sampled_prediction(
                enc_x, enc_x_num_pads, num_outputs=a.get("how_many_outputs", 1), sos_idx=sos_idx, eos_idx=eos_idx,
                max_seq_len=a.get("sample_max_seq_len", 20))
        raise ValueError(f"unknown mode {mode!r}")

    def forward_enc(self, enc_input, enc_input_num_pads):
        return self.model.forward_enc(enc_input, enc_input_num_pads)

    def forward_dec(self, cross_input, enc_input_num_pads, dec_input, dec_input_num_pads):
        return self.model.forward_dec(cross_input, enc_input_num_pads, dec_input, dec_input_num_pads,
                                      self.apply_log_softmax)

    def beam_search(self, *a, **k):
        return self.model.beam_search(*a, **k)
