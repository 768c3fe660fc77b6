"""ExpansionNet_v2 — features-only captioner (input = precomputed backbone features (B, S<=144, F)
with trailing padding), drop-in for legacy_models/ExpansionNet_v2.py:9-107 (BASELINE config 2)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as _engine
from .captioning_model import CaptioningModel, _as_list
from .layers import DecoderLayer, EmbeddingLayer, EncoderLayer
from .weights import Geometry


class ExpansionNet_v2(CaptioningModel):
    def __init__(self, d_model, N_enc, N_dec, ff, num_heads, num_exp_enc_list, num_exp_dec, output_word2idx,
                 output_idx2word, max_seq_len, drop_args, img_feature_dim=2048, rank=0):
        super().__init__()
        self.output_word2idx, self.output_idx2word = output_word2idx, output_idx2word
        self.max_seq_len = max_seq_len
        self.num_exp_dec, self.num_exp_enc_list = num_exp_dec, num_exp_enc_list
        self.N_enc, self.N_dec, self.d_model = N_enc, N_dec, d_model
        V = len(output_word2idx)
        self.encoders = nn.ModuleList([EncoderLayer(d_model, ff, num_exp_enc_list, drop_args.enc)
                                       for _ in range(N_enc)])
        self.decoders = nn.ModuleList([DecoderLayer(d_model, num_heads, ff, num_exp_dec, drop_args.dec)
                                       for _ in range(N_dec)])
        self.input_linear = nn.Linear(img_feature_dim, d_model)
        self.vocab_linear = nn.Linear(d_model, V)
        self.out_embedder = EmbeddingLayer(V, d_model, drop_args.dec_input)
        self.pos_encoder = nn.Embedding(max_seq_len, d_model)
        self.enc_reduce_group = nn.Linear(d_model * N_enc, d_model)
        self.enc_reduce_norm = nn.LayerNorm(d_model)
        self.dec_reduce_group = nn.Linear(d_model * N_dec, d_model)
        self.dec_reduce_norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.trained_steps = 0
        self.rank = rank
        self.geometry = Geometry(final_swin_dim=img_feature_dim, d_model=d_model, N_enc=N_enc, N_dec=N_dec, ff=ff,
                                 num_heads=num_heads, num_exp_enc_list=tuple(num_exp_enc_list),
                                 num_exp_dec=num_exp_dec, vocab_size=V, max_seq_len=max_seq_len)

    def _captioner_engine(self):
        if self._eng_cache is None:
            self._eng_cache = _engine.CaptionerEngine(self.state_dict(), self.geometry, self._device(),
                                                          self.encoder_precision or ("bf16" if self.precision == "fp8" else self.precision))
        return self._eng_cache

    def _enc_lens(self, n, S, enc_input_num_pads):
        pads = _as_list(enc_input_num_pads, n)
        return torch.tensor([S - p for p in pads], dtype=torch.int32, device=self._device())

    def forward_enc(self, enc_input, enc_input_num_pads):
        eng = self._captioner_engine()
        feats = enc_input.to(eng.device, torch.float32)
        B, S, _ = feats.shape
        return eng.encode(feats, self._enc_lens(B, S, enc_input_num_pads))
