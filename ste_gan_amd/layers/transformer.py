"""Transformer encoder layer of the frozen EMG encoder, host side -- mirrors ``ste_gan/layers/transformer.py``.

Parameter names / shapes equal the reference's (``self_attn.w_q|w_k|w_v (H,F,A)``, ``self_attn.w_o (H,A,F)``,
``self_attn.relative_positional.embeddings (H,2R-1,A,1)``, ``linear1|linear2``, ``norm1|norm2``), so its
checkpoints load unchanged.  Execution (eval mode only -- the GAN keeps the encoder frozen,
ste_gan/train.py:58-60) works on the (B, C, T) layout: one K=1 conv launch for the fused q/k/v
projection, one attention kernel, K=1 convs for w_o / FFN (ReLU fused) and two add+LayerNorm kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_NONE, ACT_RELU


class LearnedRelativePositionalEmbedding(nn.Module):
    """Holds the (H, 2R-1, A, 1) table of transformer.py:134-166 (unmasked, per-head embeddings)."""

    def __init__(self, max_relative_pos: int, num_heads: int, embedding_dim: int, unmasked: bool = False,
                 heads_share_embeddings: bool = False, add_to_values: bool = False):
        super().__init__()
        if not unmasked or heads_share_embeddings or add_to_values:
            raise ValueError("ste_gan_amd supports the encoder configuration of the reference: unmasked, per-head "
                             "embeddings added to keys only (transformer.py:83)")
        self.max_relative_pos = max_relative_pos
        self.num_heads = num_heads
        self.embedding_dim = embedding_dim
        self.embeddings = nn.Parameter(torch.zeros(num_heads, 2 * max_relative_pos - 1, embedding_dim, 1))
        nn.init.normal_(self.embeddings, mean=0.0, std=embedding_dim ** (-0.5))


class MultiHeadAttention(nn.Module):
    """transformer.py:63-113."""

    def __init__(self, d_model=256, n_head=4, dropout=0.1, relative_positional=True, relative_positional_distance=100):
        super().__init__()
        if not relative_positional:
            raise ValueError("ste_gan_amd.MultiHeadAttention implements the relative-positional variant only")
        self.d_model, self.n_head = d_model, n_head
        d_qkv = d_model // n_head
        assert d_qkv * n_head == d_model, 'd_model must be divisible by n_head'
        self.d_qkv = d_qkv
        self.w_q = nn.Parameter(torch.empty(n_head, d_model, d_qkv))
        self.w_k = nn.Parameter(torch.empty(n_head, d_model, d_qkv))
        self.w_v = nn.Parameter(torch.empty(n_head, d_model, d_qkv))
        self.w_o = nn.Parameter(torch.empty(n_head, d_qkv, d_model))
        for w in (self.w_q, self.w_k, self.w_v, self.w_o):
            nn.init.xavier_normal_(w)
        self.relative_positional = LearnedRelativePositionalEmbedding(relative_positional_distance, n_head, d_qkv, True)
        self.batch_first = False
        self._packed = None

    def packed(self):
        """(Cout,Cin,1) conv views of the einsum weights, rebuilt when a parameter changes (frozen in practice)."""
        key = tuple((p.data_ptr(), p._version) for p in (self.w_q, self.w_k, self.w_v, self.w_o,
                                                          self.relative_positional.embeddings))
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                H, F, A = self.w_q.shape
                qkv = torch.cat([w.permute(0, 2, 1).reshape(H * A, F) for w in (self.w_q, self.w_k, self.w_v)], 0)
                wo = self.w_o.reshape(H * A, F).t()
                emb = self.relative_positional.embeddings[..., 0]
                self._packed = (key, qkv.contiguous().unsqueeze(-1), wo.contiguous().unsqueeze(-1), emb.contiguous())
        return self._packed[1:]

    def forward(self, x_bct, hand=None):
        """x: (B, d_model, T) -> (B, d_model, T) (the reference works on (T,B,d); same numbers)."""
        w_qkv, w_o, emb = self.packed()
        C = self.d_model
        qkv = ops.frozen_conv1d(x_bct, w_qkv, None, None, None, ops.ConvCfg(C, 3 * C, 1), hand)
        o = ops.relpos_attention(qkv, emb, self.n_head, self.relative_positional.max_relative_pos)
        return ops.frozen_conv1d(o, w_o, None, None, None, ops.ConvCfg(C, C, 1))


class TransformerEncoderLayer(nn.Module):
    """Post-LN encoder layer -- transformer.py:8-61 (eval mode: the three dropouts are identities)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, relative_positional=True,
                 relative_positional_distance=100):
        super().__init__()
        self.self_attn = MultiHeadAttention(d_model, nhead, dropout=dropout, relative_positional=relative_positional,
                                            relative_positional_distance=relative_positional_distance)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.d_model, self.dim_feedforward = d_model, dim_feedforward

    def forward(self, src_bct):
        d, f = self.d_model, self.dim_feedforward
        # the two residuals: src_bct and src each feed a conv and a LayerNorm.  With gradients flowing, the LayerNorm's
        # backward (which runs first) parks its gradient for that tensor and the conv's input-gradient kernel adds it
        # (ops.GradHandover) -- two autograd adds per layer less.  The FFN's ReLU rides on linear2's input staging, so
        # its mask is applied in linear2's input-gradient epilogue instead of by an activation-backward launch.
        h1 = h2 = None
        if ops.HANDOVER and torch.is_grad_enabled() and src_bct.requires_grad:
            h1, h2 = ops.GradHandover(), ops.GradHandover()
        a = self.self_attn(src_bct, (ops.TAKE, h1) if h1 is not None else None)
        src = ops.add_layernorm(src_bct, a, self.norm1.weight, self.norm1.bias, self.norm1.eps,
                                (ops.PARK_DX, h1) if h1 is not None else None)
        h = ops.frozen_conv1d(src, self.linear1.weight.unsqueeze(-1), None, self.linear1.bias, None,
                              ops.ConvCfg(d, f, 1), (ops.TAKE, h2) if h2 is not None else None)
        h = ops.frozen_conv1d(h, self.linear2.weight.unsqueeze(-1), None, self.linear2.bias, None,
                              ops.ConvCfg(f, d, 1, in_act=ACT_RELU))
        return ops.add_layernorm(src, h, self.norm2.weight, self.norm2.bias, self.norm2.eps,
                                 (ops.PARK_DX, h2) if h2 is not None else None)
