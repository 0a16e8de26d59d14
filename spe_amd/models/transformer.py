"""Conditional-DETR transformer on libspe_hip.so kernels (mirror of reference
models/transformer.py: Transformer.forward_refine 122-160, TransformerEncoderLayer.forward_post
275-288, TransformerDecoder.forward 206-250, TransformerDecoderLayer.forward_post 355-427,
gen_sineembed_for_position 35-49).  Same parameter names; activations are batch-first [B, L, d].

MI355X-first differences from the reference's execution (results identical):
  * the 1+num_refines proposal stages share the decoder weights and the memory, so they run through the
    decoder in ONE pass, stacked along the query axis (the reference runs the decoder once per stage and
    recomputes the query-independent memory-side projections each time);
  * attention never forms head-averaged weights (the reference computes and discards them).
"""
import copy
import math

import torch
from torch import nn

from .. import ops
from .attention import MultiheadAttention
from .layers import MLP, Dropout, LayerNorm, Linear


MEMSIDE_BATCH = True


def gen_sineembed_for_position(pos_tensor, d_model=256):
    """[.., 2] normalised (x, y) -> [.., d_model] sine embedding.  The exponent divisor is the
    reference's hard-coded 128 (transformer.py:41), NOT d_model/2.  Tiny [B,Q,d] tensor ops."""
    n_steps = d_model // 2
    scale = 2 * math.pi
    dim_t = torch.arange(n_steps, dtype=torch.float32, device=pos_tensor.device)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / 128)
    pos_x = pos_tensor[..., 0, None] * scale / dim_t
    pos_y = pos_tensor[..., 1, None] * scale / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=-1).flatten(-2)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=-1).flatten(-2)
    return torch.cat((pos_y, pos_x), dim=-1)


class PackedSelfAttention(nn.Module):
    """Parameter layout of torch nn.MultiheadAttention (in_proj_weight [3d,d], in_proj_bias, out_proj)
    used by the reference's encoder layer (transformer.py:258)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, float(dropout)
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, qk_in, v_in, key_padding_mask):
        B, S, d = qk_in.shape
        H = self.num_heads
        qk = ops.linear(qk_in, self.in_proj_weight[:2 * d], self.in_proj_bias[:2 * d]).view(B, S, 2, H, d // H)
        v = ops.linear(v_in, self.in_proj_weight[2 * d:], self.in_proj_bias[2 * d:]).view(B, S, H, d // H)
        o, _ = ops.attention(qk[:, :, 0], qk[:, :, 1], v, key_padding_mask, float(d // H) ** -0.5,
                             self.dropout if self.training else 0.0)
        return self.out_proj(o)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if normalize_before:
            raise NotImplementedError("pre-norm is not reachable in the reference (transformer.py:461-462)")
        self.self_attn = PackedSelfAttention(d_model, nhead, dropout=dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.dropout1 = Dropout(dropout)
        self.dropout2 = Dropout(dropout)

    def forward(self, src, src_key_padding_mask=None, pos=None):
        qk = src if pos is None else ops.add(src, pos)
        src2 = self.self_attn(qk, src, src_key_padding_mask)
        src, src_skip = self.norm1.residual2(src, src2, self.dropout1)       # two consumers: linear1 and the next residual site
        src2 = self.linear2(self.dropout(self.linear1(src, ops.ACT_RELU)))
        return self.norm2.residual(src_skip, src2, self.dropout2)


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer if num_layers != 0 else nn.Identity(), max(num_layers, 1))
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, src_key_padding_mask=None, pos=None):
        out = src
        if self.num_layers:
            for layer in self.layers:
                out = layer(out, src_key_padding_mask=src_key_padding_mask, pos=pos)
        return out if self.norm is None else self.norm(out)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.sa_qcontent_proj = Linear(d_model, d_model)
        self.sa_qpos_proj = Linear(d_model, d_model)
        self.sa_kcontent_proj = Linear(d_model, d_model)
        self.sa_kpos_proj = Linear(d_model, d_model)
        self.sa_v_proj = Linear(d_model, d_model)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout, vdim=d_model)
        self.ca_qcontent_proj = Linear(d_model, d_model)
        self.ca_qpos_proj = Linear(d_model, d_model)
        self.ca_kcontent_proj = Linear(d_model, d_model)
        self.ca_kpos_proj = Linear(d_model, d_model)
        self.ca_v_proj = Linear(d_model, d_model)
        self.ca_qpos_sine_proj = Linear(d_model, d_model)
        self.cross_attn = MultiheadAttention(d_model * 2, nhead, dropout=dropout, vdim=d_model)
        self.nhead = nhead
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.norm3 = LayerNorm(d_model)
        self.dropout1 = Dropout(dropout)
        self.dropout2 = Dropout(dropout)
        self.dropout3 = Dropout(dropout)

    def memory_side(self, memory, pos, is_first):
        """Query-independent half of the cross attention: per-head [k_content(+k_pos) | k_pos] keys
        [B,S,H,2*dh] and values [B,S,d] (transformer.py:390-419)."""
        B, S, d = memory.shape
        H, dh = self.nhead, d // self.nhead
        k_content = self.ca_kcontent_proj(memory)
        v = self.ca_v_proj(memory)
        k_pos = self.ca_kpos_proj(pos)
        k = ops.add(k_content, k_pos) if is_first else k_content
        k = torch.cat([k.view(B, S, H, dh), k_pos.view(B, S, H, dh)], dim=3)
        return k.view(B, S, 2 * d), v

    def forward(self, tgt, mem_kv, memory_key_padding_mask, query_pos, query_sine_embed, is_first, n_stages=1, qp=None):
        """tgt [B, R*Q, d]: the R proposal stages are stacked along the query axis.  Cross attention, projections,
        FFN and LayerNorms are per-query, so they run once on all R*Q rows; only the query self-attention must not
        mix stages - it sees the same buffer as [B*R, Q, d]."""
        B, RQ, d = tgt.shape
        H, dh = self.nhead, d // self.nhead
        Q = RQ // n_stages
        # ---- self attention over the queries of each stage
        # qp: this layer's projections of query_pos (sa_qpos, sa_kpos, ca_qpos or None), evaluated for all layers at once by the decoder
        q_pos, k_pos, ca_pos = qp if qp is not None else (self.sa_qpos_proj(query_pos), self.sa_kpos_proj(query_pos), None)
        content = (self.sa_qcontent_proj, self.sa_kcontent_proj, self.sa_v_proj)
        if ops.group_linear_ok(tgt, content):       # the three projections of tgt AND q + q_pos, k + k_pos: one launch each way
            q, k, v = ops.group_linear(tgt, content, adds=(q_pos, k_pos, None))
        else:
            q, k, v = (m(tgt) for m in content)
            q = q + q_pos
            k = k + k_pos
        sa = lambda t: t.view(B * n_stages, Q, d)
        tgt2 = self.self_attn(sa(q), sa(k), sa(v))[0].view(B, RQ, d)
        # (post-norm: the norm's output feeds the next projection AND the next residual site - two results of one node, so that their gradients are
        # added inside its backward kernel)
        tgt, tgt_skip = self.norm1.residual2(tgt, tgt2, self.dropout1)
        # ---- conditional cross attention
        q = self.ca_qcontent_proj(tgt)
        if is_first:
            q = q + (ca_pos if ca_pos is not None else self.ca_qpos_proj(query_pos))
        qs = self.ca_qpos_sine_proj(query_sine_embed)
        if isinstance(mem_kv[0], ops.MemoryKV):
            # keys / values are operand fragments of the flash MHA kernels, key layout [k_content | k_pos] for EVERY layer.  The first
            # layer's k_content + k_pos (transformer.py:400-406) is carried by the query instead:
            # q_c (k_c + k_p) + q_s k_p = q_c k_c + (q_s + q_c) k_p
            holder, tok, layer_id = mem_kv
            if is_first:
                qs = qs + q
            q = torch.cat([q.view(B, RQ, H, dh), qs.view(B, RQ, H, dh)], dim=3)
            ca = self.cross_attn
            o = ops.cross_attention_kv(q, tok, holder, layer_id, memory_key_padding_mask, float(ca.head_dim) ** -0.5,
                                       ca.dropout if ca.training else 0.0)
            tgt2 = ca.out_proj(o)
        else:
            q = torch.cat([q.view(B, RQ, H, dh), qs.view(B, RQ, H, dh)], dim=3).view(B, RQ, 2 * d)
            k, v = mem_kv
            tgt2 = self.cross_attn(q, k, v, key_padding_mask=memory_key_padding_mask)[0]
        tgt, tgt_skip = self.norm2.residual2(tgt_skip, tgt2, self.dropout2)
        # ---- FFN
        tgt2 = self.linear2(self.dropout(self.linear1(tgt, ops.ACT_RELU)))
        return self.norm3.residual(tgt_skip, tgt2, self.dropout3)


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False, d_model=256):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate
        self.d_model = d_model
        self.query_scale = MLP(d_model, d_model, d_model, 2)
        self.ref_point_head = MLP(d_model, d_model, 2, 2)
        for layer_id in range(num_layers - 1):
            self.layers[layer_id + 1].ca_qpos_proj = None      # only the first layer adds query_pos (transformer.py:203-204)

    def _memory_side_all(self, memory, pos, mem_cache):
        """The memory-side keys / values of EVERY layer from two GEMMs: memory x [ca_kcontent | ca_v of all layers] and
        pos x [ca_kpos of all layers] (they depend on neither the queries nor the previous layer; the reference evaluates the
        3 * num_layers projections one by one, per decoder pass: transformer.py:389-396).  Same arithmetic per output column."""
        mem_w = [m for layer in self.layers for m in (layer.ca_kcontent_proj, layer.ca_v_proj)]
        pos_w = [layer.ca_kpos_proj for layer in self.layers]
        Wm, bm = [m.weight for m in mem_w], [m.bias for m in mem_w]
        Wp, bp = [m.weight for m in pos_w], [m.bias for m in pos_w]
        if not (MEMSIDE_BATCH and memory.is_cuda and ops.multi_linear_ok(memory, Wm, bm) and ops.multi_linear_ok(pos, Wp, bp)):
            return
        B, S, d = memory.shape
        H, dh = self.layers[0].nhead, d // self.layers[0].nhead
        if ops.memory_side_kv_ok(memory, Wm, Wp, H):
            # fragments straight from two fp16 GEMMs and one fragment launch: no fp32 keys / values, no concatenation, no per-layer pack
            holder, toks = ops.memory_side_kv(memory, pos, H, Wm, Wp, bm, bp)
            for l in range(len(self.layers)):
                mem_cache[l] = (holder, toks[l], l)
            return
        ym = ops.multi_linear(memory, Wm, bm)                 # 2*L column blocks [B, S, d] of one [B, S, 2*L*d] buffer
        yp = ops.multi_linear(pos, Wp, bp)                    # L column blocks
        for l in range(len(self.layers)):
            kc, v, kp = ym[2 * l], ym[2 * l + 1], yp[l]
            k = kc + kp if l == 0 else kc
            k = torch.cat([k.view(B, S, H, dh), kp.view(B, S, H, dh)], dim=3)
            mem_cache[l] = (k.view(B, S, 2 * d), v)

    def _query_pos_all(self, query_pos):
        """query_pos is the same tensor in every layer (transformer.py:186-204): its 2 * num_layers + 1 projections (sa_qpos_proj and
        sa_kpos_proj of every layer, ca_qpos_proj of the first) come from one group launch each way (<= 16 Linears per launch) instead
        of 13 + 13 launches and 12 gradient accumulations.  -> per layer (sa_qpos, sa_kpos, ca_qpos or None), or None."""
        mods = [m for layer in self.layers for m in (layer.sa_qpos_proj, layer.sa_kpos_proj)] + [self.layers[0].ca_qpos_proj]
        if not ops.group_linear_ok(query_pos, mods[:2]):
            return None
        outs = []
        for i in range(0, len(mods), 16):
            chunk = mods[i:i + 16]
            outs += list(ops.group_linear(query_pos, chunk)) if len(chunk) > 1 else [chunk[0](query_pos)]
        return [(outs[2 * l], outs[2 * l + 1], outs[-1] if l == 0 else None) for l in range(len(self.layers))]

    def forward(self, tgt, memory, memory_key_padding_mask, pos, query_pos, mem_cache=None, n_stages=1):
        """tgt/query_pos [B, R*Q, d] (R stages stacked along the query axis); memory/pos [B,S,d].
        -> (hs [L,B,R*Q,d], reference_points [B,R*Q,2])."""
        mem_cache = {} if mem_cache is None else mem_cache
        output = tgt
        reference_points = self.ref_point_head(query_pos).sigmoid()            # [B,RQ,2]
        intermediate = []
        # the reference points do not change from layer to layer (transformer.py:187-193 recomputes the embedding in every
        # layer): one evaluation, ~25 small launches of sin / cos / stack / cat and their backward saved per further layer
        sine0 = gen_sineembed_for_position(reference_points[..., :2], self.d_model)
        if not mem_cache:
            self._memory_side_all(memory, pos, mem_cache)
        qp_all = self._query_pos_all(query_pos)
        for layer_id, layer in enumerate(self.layers):
            if layer_id not in mem_cache:
                mem_cache[layer_id] = layer.memory_side(memory, pos, layer_id == 0)
            sine = sine0
            if layer_id > 0:
                sine = sine0 * self.query_scale(output)
            output = layer(output, mem_cache[layer_id], memory_key_padding_mask, query_pos, sine, layer_id == 0,
                           n_stages=n_stages, qp=None if qp_all is None else qp_all[layer_id])
            intermediate.append(self.norm(output))
        return torch.stack(intermediate), reference_points


class Transformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_queries=300, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False,
                 return_intermediate_dec=False, args=None, num_refines=1, drloc=False):
        super().__init__()
        enc_layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.encoder = TransformerEncoder(enc_layer, num_encoder_layers, None)
        dec_layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers, LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec, d_model=d_model)
        self.H = self.W = None
        self._reset_parameters()
        self.d_model, self.nhead = d_model, nhead
        self.dec_layers, self.num_queries, self.num_refines = num_decoder_layers, num_queries, num_refines

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, mask, query_embed, pos_embed, queries_embed_refine=None):
        """src/pos_embed [B,d,h,w] (views of [B,hw,d] buffers), mask [B,h,w], query_embed [Q,d].
        -> (list of hs [L,B,Q,d], list of reference points [B,Q,2]), one entry per decoder pass."""
        B = src.shape[0]
        memory = src.flatten(2).transpose(1, 2).contiguous()       # no copy when src is the backbone's view
        pos = pos_embed.flatten(2).transpose(1, 2).contiguous()
        mask = mask.flatten(1)
        memory = self.encoder(memory, src_key_padding_mask=mask, pos=pos)
        # All proposal stages go through the decoder in ONE pass: the stages share the decoder weights and the
        # memory, and differ only in their query embeddings (reference transformer.py:147-155 runs the decoder
        # 1 + num_refines times).  Stacking the stages along the query axis is exact - every decoder op is per
        # query except the query self-attention, which is evaluated per stage - and halves the kernel launches.
        queries = [query_embed] + [qe.weight for qe in (queries_embed_refine or [])]
        R, Q = len(queries), query_embed.shape[0]
        query_pos = torch.stack(queries).reshape(1, R * Q, -1).expand(B, -1, -1).contiguous()
        h, r = self.decoder(torch.zeros_like(query_pos), memory, mask, pos, query_pos, n_stages=R)
        hs = [h[:, :, i * Q:(i + 1) * Q] for i in range(R)]
        refs = [r[:, i * Q:(i + 1) * Q] for i in range(R)]
        return hs, refs


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def build_transformer(args):
    return Transformer(d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads, num_queries=args.num_queries,
                       dim_feedforward=args.dim_feedforward, num_encoder_layers=args.enc_layers,
                       num_decoder_layers=args.dec_layers, normalize_before=args.pre_norm, return_intermediate_dec=True)
