"""ExpansionNet v2 blocks — parameter containers mirroring reference models/layers.py
(EmbeddingLayer:9, StaticExpansionBlock:20, EncoderLayer:105, DynamicExpansionBlock:126,
DecoderLayer:207, MultiHeadAttention:251, FeedForward:298).  They own the parameters under the
reference's names; engine.CaptionerEngine does the arithmetic on the GPU."""
from __future__ import annotations

import torch.nn as nn


class EmbeddingLayer(nn.Module):
    def __init__(self, vocab_size, d_model, dropout_perc=0.0):
        super().__init__()
        self.embed = nn.Embedding(vocab_size, d_model)
        self.d_model = d_model


class FeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dropout_perc=0.0):
        super().__init__()
        self.linear_1 = nn.Linear(d_model, d_ff)
        self.linear_2 = nn.Linear(d_ff, d_model)


class StaticExpansionBlock(nn.Module):
    def __init__(self, d_model, num_enc_exp_list, dropout_perc=0.0, eps=1e-9):
        super().__init__()
        self.d_model, self.num_enc_exp_list, self.eps = d_model, num_enc_exp_list, eps
        self.query_exp_vectors = nn.Embedding(sum(num_enc_exp_list), d_model)
        self.bias_exp_vectors = nn.Embedding(sum(num_enc_exp_list), d_model)
        self.key_embed = nn.Linear(d_model, d_model)
        self.class_a_embed = nn.Linear(d_model, d_model)
        self.class_b_embed = nn.Linear(d_model, d_model)
        self.selector_embed = nn.Linear(d_model, d_model)


class EncoderLayer(nn.Module):
    def __init__(self, d_model, d_ff, num_enc_exp_list, dropout_perc=0.0, eps=1e-9):
        super().__init__()
        self.norm_1 = nn.LayerNorm(d_model)
        self.norm_2 = nn.LayerNorm(d_model)
        self.stc_exp = StaticExpansionBlock(d_model, num_enc_exp_list, dropout_perc, eps)
        self.ff = FeedForward(d_model, d_ff, dropout_perc)


class MultiHeadAttention(nn.Module):
    def __init__(self, d_model, num_heads, dropout_perc=0.0):
        super().__init__()
        assert d_model % num_heads == 0, "num heads must be multiple of d_model"
        self.d_model, self.num_heads, self.d_k = d_model, num_heads, d_model // num_heads
        self.Wq = nn.Linear(d_model, d_model)
        self.Wk = nn.Linear(d_model, d_model)
        self.Wv = nn.Linear(d_model, d_model)
        self.out_linear = nn.Linear(d_model, d_model)


class DynamicExpansionBlock(nn.Module):
    def __init__(self, d_model, num_exp, dropout_perc=0.0, eps=1e-9):
        super().__init__()
        self.d_model, self.num_exp, self.eps = d_model, num_exp, eps
        self.cond_embed = nn.Linear(d_model, d_model)
        self.query_exp_vectors = nn.Embedding(num_exp, d_model)
        self.bias_exp_vectors = nn.Embedding(num_exp, d_model)
        self.key_linear = nn.Linear(d_model, d_model)
        self.class_a_embed = nn.Linear(d_model, d_model)
        self.class_b_embed = nn.Linear(d_model, d_model)
        self.selector_embed = nn.Linear(d_model, d_model)


class DecoderLayer(nn.Module):
    def __init__(self, d_model, num_heads, d_ff, num_exp, dropout_perc=0.0, eps=1e-9):
        super().__init__()
        self.norm_1 = nn.LayerNorm(d_model)
        self.norm_2 = nn.LayerNorm(d_model)
        self.norm_3 = nn.LayerNorm(d_model)
        self.mha = MultiHeadAttention(d_model, num_heads, dropout_perc)
        self.dyn_exp = DynamicExpansionBlock(d_model, num_exp, dropout_perc, eps)
        self.ff = FeedForward(d_model, d_ff, dropout_perc)
