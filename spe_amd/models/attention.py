"""Projection-free multi-head attention (mirror of reference models/attention.py:55-385): the
module owns only out_proj; q/k may have a different per-head width than v; the scale comes from the
QUERY head dim.  Batch-first [B, L, E] inputs."""
import torch
from torch import nn

from .. import ops
from .layers import Linear


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True, kdim=None, vdim=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.num_heads = num_heads
        self.dropout = float(dropout)
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.out_proj = Linear(self.vdim, self.vdim)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def core(self, query, key, value, key_padding_mask=None):
        """query [B,Lq,E], key [B,Lk,E], value [B,Lk,vdim] -> [B,Lq,vdim] (before out_proj)."""
        B, Lq, E = query.shape
        Lk, H = key.shape[1], self.num_heads
        o, _ = ops.attention(query.view(B, Lq, H, E // H), key.view(B, Lk, H, E // H),
                             value.view(B, Lk, H, self.vdim // H), key_padding_mask, float(self.head_dim) ** -0.5,
                             self.dropout if self.training else 0.0)
        return o

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False, attn_mask=None):
        assert attn_mask is None, "attn_mask is never used on the SPE path"
        return self.out_proj(self.core(query, key, value, key_padding_mask)), None
