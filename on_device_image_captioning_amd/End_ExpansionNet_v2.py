"""End_ExpansionNet_v2 — drop-in for the reference class of the same name
(legacy_models/End_ExpansionNet_v2.py:10-138 = the API demo.py/test.py use; constructor keywords
also accept the refactored models/End_ExpansionNet_v2.py:11-47 form with `apply_log_softmax`).

Same constructor kwargs, same 520-key state dict (strict load of rf_model.pth's
'model_state_dict'), same forward()/forward_enc()/forward_dec()/beam_search() signatures — but the
arithmetic runs in hand-written HIP kernels on an MI355X.
"""
from __future__ import annotations

from argparse import Namespace

import torch
import torch.nn as nn

from . import engine as _engine
from .captioning_model import Captioner, CaptioningModel, _as_list
from .layers import DecoderLayer, EmbeddingLayer, EncoderLayer
from .swin_transformer_mod import SwinTransformer
from .weights import Geometry


def make_drop_args(enc=0.0, dec=0.0, enc_input=0.0, dec_input=0.0, other=0.0) -> Namespace:
    """The `drop_args` Namespace the reference scripts build (demo.py:61-66); inference ignores it."""
    return Namespace(enc=enc, dec=dec, enc_input=enc_input, dec_input=dec_input, other=other)


class End_ExpansionNet_v2(CaptioningModel):
    def __init__(self,
                 swin_img_size, swin_patch_size, swin_in_chans, swin_embed_dim, swin_depths, swin_num_heads,
                 swin_window_size, swin_mlp_ratio, swin_qkv_bias, swin_qk_scale, swin_drop_rate,
                 swin_attn_drop_rate, swin_drop_path_rate, swin_norm_layer, swin_ape, swin_patch_norm,
                 swin_use_checkpoint,
                 final_swin_dim,
                 d_model, N_enc, N_dec, ff, num_heads, num_exp_enc_list, num_exp_dec,
                 output_word2idx, output_idx2word, max_seq_len, drop_args, rank=0, apply_log_softmax=False):
        super().__init__(apply_log_softmax)
        self.swin_transf = SwinTransformer(
            img_size=swin_img_size, patch_size=swin_patch_size, in_chans=swin_in_chans, embed_dim=swin_embed_dim,
            depths=swin_depths, num_heads=swin_num_heads, window_size=swin_window_size, mlp_ratio=swin_mlp_ratio,
            qkv_bias=swin_qkv_bias, qk_scale=swin_qk_scale, drop_rate=swin_drop_rate,
            attn_drop_rate=swin_attn_drop_rate, drop_path_rate=swin_drop_path_rate, norm_layer=swin_norm_layer,
            ape=swin_ape, patch_norm=swin_patch_norm, use_checkpoint=swin_use_checkpoint)
        self.output_word2idx, self.output_idx2word = output_word2idx, output_idx2word
        self.max_seq_len = max_seq_len
        self.num_exp_dec, self.num_exp_enc_list = num_exp_dec, num_exp_enc_list
        self.N_enc, self.N_dec, self.d_model = N_enc, N_dec, d_model
        V = len(output_word2idx)

        self.encoders = nn.ModuleList([EncoderLayer(d_model, ff, num_exp_enc_list, drop_args.enc)
                                       for _ in range(N_enc)])
        self.decoders = nn.ModuleList([DecoderLayer(d_model, num_heads, ff, num_exp_dec, drop_args.dec)
                                       for _ in range(N_dec)])
        self.input_linear = nn.Linear(final_swin_dim, d_model)
        self.vocab_linear = nn.Linear(d_model, V)
        self.out_embedder = EmbeddingLayer(V, d_model, drop_args.dec_input)
        self.pos_encoder = nn.Embedding(max_seq_len, d_model)
        self.enc_reduce_group = nn.Linear(d_model * N_enc, d_model)
        self.enc_reduce_norm = nn.LayerNorm(d_model)
        self.dec_reduce_group = nn.Linear(d_model * N_dec, d_model)
        self.dec_reduce_norm = nn.LayerNorm(d_model)
        for p in self.parameters():                      # reference :112-114
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

        self.trained_steps = 0
        self.rank = rank
        self.check_required_attributes()
        self.geometry = Geometry(
            swin_img_size=swin_img_size, swin_patch_size=swin_patch_size, swin_in_chans=swin_in_chans,
            swin_embed_dim=swin_embed_dim, swin_depths=tuple(swin_depths), swin_num_heads=tuple(swin_num_heads),
            swin_window_size=swin_window_size, swin_mlp_ratio=swin_mlp_ratio, final_swin_dim=final_swin_dim,
            d_model=d_model, N_enc=N_enc, N_dec=N_dec, ff=ff, num_heads=num_heads,
            num_exp_enc_list=tuple(num_exp_enc_list), num_exp_dec=num_exp_dec, vocab_size=V,
            max_seq_len=max_seq_len)

    # ------------------------------------------------------------------ engines
    def _engines(self):
        if self._eng_cache is None:
            dv = self._device()
            sd = self.state_dict()
            self._eng_cache = (_engine.SwinEngine(sd, self.geometry, dv, self.precision,
                                                  calibration_images=self.calibration_images),
                               _engine.CaptionerEngine(sd, self.geometry, dv, self.encoder_precision or ("bf16" if self.precision == "fp8" else self.precision)))
        return self._eng_cache

    def _captioner_engine(self):
        return self._engines()[1]

    def _enc_lens(self, n, S, enc_input_num_pads):
        # end-to-end: the encoder never has padding (End_ExpansionNet_v2.py:107)
        return torch.full((n,), S, dtype=torch.int32, device=self._device())

    # ------------------------------------------------------------------ reference API
    def forward_enc(self, enc_input, enc_input_num_pads):
        assert (enc_input_num_pads is None or list(enc_input_num_pads) == ([0] * enc_input.size(0))), \
            "End to End case have no padding"
        swin, cap = self._engines()
        img = enc_input.to(swin.device, torch.float32)
        feats = swin.forward(img, out_dtype=cap.cdt)
        B, S, _ = feats.shape
        return cap.encode(feats, self._enc_lens(B, S, None))

    def forward_dec(self, cross_input, enc_input_num_pads, dec_input, dec_input_num_pads, apply_log_softmax=False):
        assert (enc_input_num_pads is None or list(enc_input_num_pads) == ([0] * cross_input.size(0))), \
            "enc_input_num_pads should be no None"
        return super().forward_dec(cross_input, enc_input_num_pads, dec_input, dec_input_num_pads,
                                   apply_log_softmax or self.apply_log_softmax)


class E2E_ExpansionNet_Captioner(Captioner):
    """models/End_ExpansionNet_v2.py:311-354."""

    def __init__(self, beam_search_args, model=None, split_encoder=False, apply_log_softmax=False, encoder=None,
                 decoder=None, rank=0, N_enc=3, N_dec=3, num_exp_dec=16, num_exp_enc_list=[32, 64, 128, 256, 512]):
        super().__init__(beam_search_args, model, split_encoder, apply_log_softmax, encoder, decoder)
        self.rank = rank
        self.N_enc, self.N_dec = N_enc, N_dec
        self.num_exp_dec, self.num_exp_enc_list = num_exp_dec, num_exp_enc_list
