"""ORACLE - test infrastructure only.  A plain PyTorch fp32, CPU, functional restatement of the
SPE hot path (CaiT/TSCAM backbone -> conditional-DETR transformer -> heads -> Hungarian matcher
-> set criterion), written from the reference's behaviour; every function cites the reference
file:line it follows (paths relative to MingXiangL/SPE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (spe_amd/) never does; it has no CPU path.

Parity pinning: the reference has no tests or golden vectors of its own,
so this restatement is pinned against outputs of the reference itself, generated in the build
container by tools/gen_golden.py (reference imported from /root/reference) and committed as
tests/golden/*.pt; tests/test_oracle_golden.py checks every tensor.  Third-party pieces the
reference calls (timm 0.4.x Mlp/PatchEmbed/DropPath, torch nn.MultiheadAttention, bicubic
F.interpolate, scipy linear_sum_assignment) are restated from their published semantics.

All functions operate on a flat `sd` (state_dict with the reference's parameter names) so that no
nn.Module structure is shared with either the reference or the product.  Dropout is not modelled
(parity is defined with every drop rate = 0, or in eval mode).
"""
import copy
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# config
# ------------------------------------------------------------------------------------------------
def make_cfg(**kw):
    """Model hyper-parameters (not learned).  Defaults = cfg1 of BASELINE.json (TSCAM_cait_XXS24)."""
    c = dict(embed_dim=192, depth=24, num_heads=4, patch_size=16, num_cls_tokens=20, layer_to_det=23,
             two_branch=False, pos_grid=(50, 84), ln_eps=1e-6,
             nheads=8, enc_layers=0, dec_layers=1, dim_feedforward=2048, num_queries=10, num_refines=1,
             num_det_classes=21, aux_loss=True, focal_gamma=2.0)       # focal_gamma: main.py --focal_gamma (scripts/run_voc0712.py: 0.5)
    c.update(kw)
    return SimpleNamespace(**c)


# ------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------
def lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def mlp_gelu(x, sd, p):
    """timm 0.4.x Mlp: fc1 -> GELU(erf) -> fc2 (drops omitted).  Used at cait.py:409."""
    return lin(F.gelu(lin(x, sd, p + ".fc1")), sd, p + ".fc2")


def mlp_relu(x, sd, p, n):
    """transformer.py:21-33 / conditional_detr.py:626-638: n Linear layers, ReLU between."""
    for i in range(n):
        x = lin(x, sd, f"{p}.layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def inverse_sigmoid(x, eps=1e-5):
    """util/misc.py:477-481."""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


# ------------------------------------------------------------------------------------------------
# backbone (models/cait.py)
# ------------------------------------------------------------------------------------------------
def talking_heads_attention(x, sd, p, H):
    """cait.py:374-393.  Heads are mixed by proj_l before and proj_w after the key softmax."""
    B, N, C = x.shape
    dh = C // H
    qkv = lin(x, sd, p + ".qkv").reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * dh ** -0.5, qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)                                             # [B,H,N,N]
    s = torch.einsum("bhqk,gh->bgqk", s, sd[p + ".proj_l.weight"]) + sd[p + ".proj_l.bias"][None, :, None, None]
    a = s.softmax(-1)
    a = torch.einsum("bhqk,gh->bgqk", a, sd[p + ".proj_w.weight"]) + sd[p + ".proj_w.bias"][None, :, None, None]
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return lin(o, sd, p + ".proj")


def layerscale_block(x, sd, p, H, eps):
    """cait.py:413-416 (DropPath omitted)."""
    x = x + sd[p + ".gamma_1"] * talking_heads_attention(ln(x, sd, p + ".norm1", eps), sd, p + ".attn", H)
    x = x + sd[p + ".gamma_2"] * mlp_gelu(ln(x, sd, p + ".norm2", eps), sd, p + ".mlp")
    return x


def multi_class_attention(u, sd, p, H, n_tok):
    """cait.py:111-139: queries are the first n_tok (=K+1) tokens, keys/values all tokens.
    Returns (x_cls, attention map [B,H,n_tok,N])."""
    B, N, C = u.shape
    dh = C // H
    q = lin(u[:, :n_tok], sd, p + ".q").reshape(B, n_tok, H, dh).permute(0, 2, 1, 3) * dh ** -0.5
    k = lin(u, sd, p + ".k").reshape(B, N, H, dh).permute(0, 2, 1, 3)
    v = lin(u, sd, p + ".v").reshape(B, N, H, dh).permute(0, 2, 1, 3)
    a = (q @ k.transpose(-2, -1)).softmax(-1)
    o = (a @ v).transpose(1, 2).reshape(B, n_tok, C)
    return lin(o, sd, p + ".proj"), a


def class_attention_block(x, cls, sd, p, H, eps):
    """cait.py:322-328."""
    u = torch.cat((cls, x), dim=1)
    y, amap = multi_class_attention(ln(u, sd, p + ".norm1", eps), sd, p + ".attn", H, cls.shape[1])
    cls = cls + sd[p + ".gamma_1"] * y
    cls = cls + sd[p + ".gamma_2"] * mlp_gelu(ln(cls, sd, p + ".norm2", eps), sd, p + ".mlp")
    return cls, amap


def interpolate_pos_embed(pos_embed, grid, hw):
    """cait.py:598-613: bicubic (align_corners=False) resize of the stored [1, gh*gw, C] grid."""
    C = pos_embed.shape[-1]
    pe = pos_embed.transpose(1, 2).reshape(1, C, grid[0], grid[1])
    pe = F.interpolate(pe, size=hw, mode="bicubic", align_corners=False)
    return pe.flatten(2).transpose(1, 2)


def backbone_forward(sd, cfg, img, p="backbone.0.body."):
    """TSCAM_cait.forward (cait.py:615-670) / TSCAM_cait_two_branch.forward (cait.py:761-831).
    Returns dict(x_logits, x_cls_logits, cams_cls, x_patch [B,C,h,w]).  The padding mask is ignored
    inside the backbone exactly as in the reference."""
    B, _, Hi, Wi = img.shape
    P, C, H, K, eps = cfg.patch_size, cfg.embed_dim, cfg.num_heads, cfg.num_cls_tokens, cfg.ln_eps
    h, w = Hi // P, Wi // P
    x = F.conv2d(img, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=P)
    x = x.flatten(2).transpose(1, 2)                                         # cait.py:526
    x = x + interpolate_pos_embed(sd[p + "pos_embed"], cfg.pos_grid, (h, w))
    cls = torch.cat((sd[p + "cls_token"].expand(B, -1, -1), sd[p + "extra_cls_token"].expand(B, -1, -1)), dim=1)
    x_feat = None
    for i in range(cfg.depth):
        x = layerscale_block(x, sd, f"{p}blocks.{i}", H, eps)
        if not cfg.two_branch and i == cfg.layer_to_det:                    # cait.py:629-630
            x_feat = ln(x, sd, p + "norm_to_det", eps)
        if cfg.two_branch and i + 1 == cfg.layer_to_det:                    # cait.py:778-779
            x_feat = x
    if cfg.two_branch:
        for j in range(cfg.depth - cfg.layer_to_det):                       # cait.py:781-784
            x_feat = layerscale_block(x_feat, sd, f"{p}blocks_det.{j}", H, eps)
        x_feat = ln(x_feat, sd, p + "norm_det", eps)
    amap0 = None
    for i in range(2):
        cls, amap = class_attention_block(x, cls, sd, f"{p}blocks_token_only.{i}", H, eps)
        if i == 0:
            amap0 = amap
    xa = ln(torch.cat((cls, x), dim=1), sd, p + "norm", eps)
    x_logits = lin(xa[:, 1:1 + K], sd, p + "cls_head").squeeze(-1)          # cait.py:653
    x_cls_logits = lin(xa[:, 0], sd, p + "cls_head_multi_cls")             # cait.py:654
    cam = amap0[:, :, 1:1 + K, 1 + K:]                                       # [B,H,K,N]
    if cfg.two_branch:                                                       # std_reweighting cait.py:801-806
        std = torch.std(cam, dim=-1, keepdim=True)
        std = std - std.min(dim=1, keepdim=True)[0]
        std = std / std.max(dim=1, keepdim=True)[0]
        cam = (cam * std).sum(1)
    else:                                                                    # head mean, cait.py:658-667
        cam = cam.mean(1)
    return {"x_logits": x_logits, "x_cls_logits": x_cls_logits, "cams_cls": cam.reshape(B, K, h, w),
            "x_patch": x_feat.transpose(1, 2).reshape(B, C, h, w)}


# ------------------------------------------------------------------------------------------------
# positional encodings
# ------------------------------------------------------------------------------------------------
def position_embedding_sine(mask, num_pos_feats, temperature=10000.0):
    """position_encoding.py:37-57 with normalize=True, scale=2*pi.  mask [B,h,w] bool -> [B,2*npf,h,w]."""
    nm = ~mask
    y = nm.cumsum(1, dtype=torch.float32)
    x = nm.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[..., None] / dim_t, y[..., None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def gen_sineembed_for_position(pos, d_model):
    """transformer.py:35-49.  NOTE the hard-coded /128 in the exponent (not d_model-scaled)."""
    n = d_model // 2
    i = torch.arange(n, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(i, 2, rounding_mode="floor") / 128)
    px = pos[:, :, 0, None] * (2 * math.pi) / dim_t
    py = pos[:, :, 1, None] * (2 * math.pi) / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2)


# ------------------------------------------------------------------------------------------------
# transformer (models/transformer.py, models/attention.py); tensors are [L, B, D] as in the reference
# ------------------------------------------------------------------------------------------------
def mha_core(q, k, v, H, key_padding_mask, out_w, out_b):
    """attention.py:269-383: projection-free MHA; scale from the *query* head dim; q/k head dim may
    differ from the v head dim; -inf on padded keys; out_proj at the end."""
    Lq, B, E = q.shape
    Lk, dv = k.shape[0], v.shape[2] // H
    dq = E // H
    qh = (q * dq ** -0.5).reshape(Lq, B * H, dq).transpose(0, 1)
    kh = k.reshape(Lk, B * H, dq).transpose(0, 1)
    vh = v.reshape(Lk, B * H, dv).transpose(0, 1)
    s = torch.bmm(qh, kh.transpose(1, 2))
    if key_padding_mask is not None:
        s = s.view(B, H, Lq, Lk).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(B * H, Lq, Lk)
    o = torch.bmm(s.softmax(-1), vh).transpose(0, 1).reshape(Lq, B, H * dv)
    return F.linear(o, out_w, out_b)


def encoder_layer(src, mask, pos, sd, p, H):
    """transformer.py:275-288 (post-norm; torch nn.MultiheadAttention with packed in_proj)."""
    d = src.shape[-1]
    Wi, bi = sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"]
    qk = src + pos
    q = F.linear(qk, Wi[:d], bi[:d])
    k = F.linear(qk, Wi[d:2 * d], bi[d:2 * d])
    v = F.linear(src, Wi[2 * d:], bi[2 * d:])
    a = mha_core(q, k, v, H, mask, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
    src = ln(src + a, sd, p + ".norm1", 1e-5)
    f = lin(F.relu(lin(src, sd, p + ".linear1")), sd, p + ".linear2")
    return ln(src + f, sd, p + ".norm2", 1e-5)


def decoder_layer(tgt, memory, mask, pos, query_pos, query_sine, sd, p, H, is_first):
    """transformer.py:355-427."""
    Q, B, d = tgt.shape
    S = memory.shape[0]
    q = lin(tgt, sd, p + ".sa_qcontent_proj") + lin(query_pos, sd, p + ".sa_qpos_proj")
    k = lin(tgt, sd, p + ".sa_kcontent_proj") + lin(query_pos, sd, p + ".sa_kpos_proj")
    v = lin(tgt, sd, p + ".sa_v_proj")
    a = mha_core(q, k, v, H, None, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
    tgt = ln(tgt + a, sd, p + ".norm1", 1e-5)
    qc = lin(tgt, sd, p + ".ca_qcontent_proj")
    kc = lin(memory, sd, p + ".ca_kcontent_proj")
    v = lin(memory, sd, p + ".ca_v_proj")
    kp = lin(pos, sd, p + ".ca_kpos_proj")
    if is_first:                                                            # transformer.py:400-406
        q = qc + lin(query_pos, sd, p + ".ca_qpos_proj")
        k = kc + kp
    else:
        q, k = qc, kc
    qs = lin(query_sine, sd, p + ".ca_qpos_sine_proj")
    dh = d // H
    q = torch.cat([q.view(Q, B, H, dh), qs.view(Q, B, H, dh)], dim=3).view(Q, B, 2 * d)
    k = torch.cat([k.view(S, B, H, dh), kp.view(S, B, H, dh)], dim=3).view(S, B, 2 * d)
    a = mha_core(q, k, v, H, mask, sd[p + ".cross_attn.out_proj.weight"], sd[p + ".cross_attn.out_proj.bias"])
    tgt = ln(tgt + a, sd, p + ".norm2", 1e-5)
    f = lin(F.relu(lin(tgt, sd, p + ".linear1")), sd, p + ".linear2")
    return ln(tgt + f, sd, p + ".norm3", 1e-5)


def decoder(memory, mask, pos, query_pos, sd, cfg, p="transformer.decoder"):
    """transformer.py:206-250 with return_intermediate=True -> (hs [L,B,Q,d], reference_points [B,Q,2])."""
    d = memory.shape[-1]
    ref = mlp_relu(query_pos, sd, p + ".ref_point_head", 2).sigmoid().transpose(0, 1)   # [B,Q,2]
    out = torch.zeros_like(query_pos)
    inter = []
    for l in range(cfg.dec_layers):
        center = ref[..., :2].transpose(0, 1)
        sine = gen_sineembed_for_position(center, d)
        if l > 0:
            sine = sine * mlp_relu(out, sd, p + ".query_scale", 2)
        out = decoder_layer(out, memory, mask, pos, query_pos, sine, sd, f"{p}.layers.{l}", cfg.nheads, l == 0)
        inter.append(ln(out, sd, p + ".norm", 1e-5))
    return torch.stack(inter).transpose(1, 2), ref


def transformer_forward(src, mask, pos, sd, cfg):
    """Transformer.forward_refine, transformer.py:122-160.  src/pos [B,d,h,w], mask [B,h,w]."""
    B = src.shape[0]
    mem = src.flatten(2).permute(2, 0, 1)
    posf = pos.flatten(2).permute(2, 0, 1)
    m = mask.flatten(1)
    for l in range(cfg.enc_layers):
        mem = encoder_layer(mem, m, posf, sd, f"transformer.encoder.layers.{l}", cfg.nheads)
    hs, refs = [], []
    qnames = ["query_embed.weight"] + [f"queries_embed_refine.{i}.weight" for i in range(cfg.num_refines)]
    for qn in qnames:
        qp = sd[qn].unsqueeze(1).repeat(1, B, 1)
        h, r = decoder(mem, m, posf, qp, sd, cfg)
        hs.append(h)
        refs.append(r)
    return hs, refs


def downsample_mask(mask, hw):
    """cait_backbone.py:92: nearest-neighbour resize of the bool padding mask."""
    return F.interpolate(mask[None].float(), size=hw).to(torch.bool)[0]


def model_forward(sd, cfg, img, mask):
    """ConditionalDETR_Refine.forward, conditional_detr.py:68-116.  img [B,3,H,W] (already padded),
    mask [B,H,W] bool (True = padding).  Returns {stage: {...}} with the reference's keys
    (x_patch is returned as (tensor, mask) instead of a NestedTensor)."""
    feats = backbone_forward(sd, cfg, img)
    x = feats["x_patch"]
    m = downsample_mask(mask, x.shape[-2:])
    pos = position_embedding_sine(m, x.shape[1] // 2)
    hs_all, refs = transformer_forward(x, m, pos, sd, cfg)
    out = {}
    for r in range(cfg.num_refines + 1):
        hs = hs_all[r]
        rb = inverse_sigmoid(refs[r])
        coords = []
        for l in range(hs.shape[0]):
            t = mlp_relu(hs[l], sd, f"bbox_embed.{r}", 3)
            t = torch.cat([t[..., :2] + rb, t[..., 2:]], dim=-1)
            coords.append(t.sigmoid())
        coords = torch.stack(coords)
        logits = lin(hs, sd, f"class_embed.{r}")
        o = {"pred_logits": logits[-1], "pred_boxes": coords[-1], "x_logits": feats["x_logits"],
             "x_cls_logits": feats["x_cls_logits"], "cams_cls": feats["cams_cls"], "x_patch": (x, m)}
        if cfg.aux_loss:
            o["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b} for a, b in zip(logits[:-1], coords[:-1])]
        out[r] = o
    return out


# ------------------------------------------------------------------------------------------------
# box math (util/box_ops.py)
# ------------------------------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_iou(a, b):
    """box_ops.py:33-46 -> (iou [N,M], union [N,M])."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b - inter
    return inter / union, union


def generalized_box_iou(a, b):
    """box_ops.py:49-74 (including its degenerate-box asserts)."""
    assert (a[:, 2:] >= a[:, :2]).all() and (b[:, 2:] >= b[:, :2]).all()
    iou, union = box_iou(a, b)
    wh = (torch.max(a[:, None, 2:], b[:, 2:]) - torch.min(a[:, None, :2], b[:, :2])).clamp(min=0)
    area = wh[..., 0] * wh[..., 1]
    return iou - (area - union) / area


# ------------------------------------------------------------------------------------------------
# matcher (models/matcher.py)
# ------------------------------------------------------------------------------------------------
def matcher_cost(logits, boxes, tgt_ids, tgt_boxes, w_class=2.0, w_bbox=5.0, w_giou=2.0):
    """matcher.py:62-83 for one image: logits [Q,Kc], boxes [Q,4] -> C [Q,M]."""
    p = logits.sigmoid()
    neg = 0.75 * (p ** 2.0) * (-(1 - p + 1e-8).log())
    pos = 0.25 * ((1 - p) ** 2.0) * (-(p + 1e-8).log())
    c_class = pos[:, tgt_ids] - neg[:, tgt_ids]
    c_bbox = torch.cdist(boxes, tgt_boxes, p=1)
    c_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_boxes))
    return w_bbox * c_bbox + w_class * c_class + w_giou * c_giou


def hungarian(outputs, targets, w_class=2.0, w_bbox=5.0, w_giou=2.0):
    """HungarianMatcher.forward, matcher.py:41-87 (per image; the reference's cross-image blocks of
    the cost matrix are never used)."""
    from scipy.optimize import linear_sum_assignment
    res = []
    with torch.no_grad():
        for b, t in enumerate(targets):
            C = matcher_cost(outputs["pred_logits"][b], outputs["pred_boxes"][b], t["labels"], t["boxes"],
                             w_class, w_bbox, w_giou)
            i, j = linear_sum_assignment(C.cpu())
            res.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
    return res


# ------------------------------------------------------------------------------------------------
# criterion (models/conditional_detr.py)
# ------------------------------------------------------------------------------------------------
def weighted_sigmoid_focal_loss(x, t, num_boxes, w, alpha, gamma):
    """conditional_detr.py:468-494: p_t clamped to [1e-5, 1-1e-5]; loss.mean(1).sum()/num_boxes."""
    p = x.sigmoid()
    ce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
    pt = (p * t + (1 - p) * (1 - t)).clamp(1e-5, 1 - 1e-5)
    loss = w * ce * (1 - pt) ** gamma
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return loss.mean(1).sum() / num_boxes


def _src_idx(indices):
    return (torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)]), torch.cat([s for s, _ in indices]))


def loss_labels(out, targets, indices, num_boxes, alpha, gamma, refine, log=True):
    """conditional_detr.py:237-265 (SetCriterion) / 504-535 (SetCriterionRefine: score weights)."""
    x = out["pred_logits"]
    B, Q, Kc = x.shape
    idx = _src_idx(indices)
    tco = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
    tc = torch.full((B, Q), Kc, dtype=torch.int64)
    tc[idx] = tco
    onehot = F.one_hot(tc, Kc + 1)[..., :Kc].to(x.dtype)
    w = torch.ones_like(onehot)
    if refine:
        avg = torch.tensor([t["scores"].mean() for t in targets]).reshape(-1, 1, 1)
        w = w * avg
        for b, (R, Cc) in enumerate(indices):
            w[b, R, :] = (targets[b]["scores"][Cc].unsqueeze(-1).repeat(1, Kc) * 3).clamp(max=1.0)
    losses = {"loss_ce": weighted_sigmoid_focal_loss(x, onehot, num_boxes, w, alpha, gamma) * Q}
    if log:                                                                 # util/misc.py:439-455 top-1
        if tco.numel() == 0:
            acc = torch.zeros([])
        else:
            acc = (x[idx].argmax(-1) == tco).float().sum() * (100.0 / tco.numel())
        losses["class_error"] = 100 - acc
    return losses


def loss_boxes(out, targets, indices, num_boxes, refine):
    """conditional_detr.py:300-319 / 537-560."""
    idx = _src_idx(indices)
    s = out["pred_boxes"][idx]
    t = torch.cat([tt["boxes"][i] for tt, (_, i) in zip(targets, indices)], dim=0)
    l1 = (s - t).abs()
    gi = 1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(s), box_cxcywh_to_xyxy(t)))
    if refine:
        w = torch.cat([tt["scores"][i] for tt, (_, i) in zip(targets, indices)], dim=0)
        l1 = l1 * w.reshape(-1, 1)
        gi = gi * w
    return {"loss_bbox": l1.sum() / num_boxes, "loss_giou": gi.sum() / num_boxes}


def loss_cardinality(out, targets):
    """conditional_detr.py:286-298 (logging only)."""
    x = out["pred_logits"]
    n = torch.as_tensor([len(t["labels"]) for t in targets], dtype=torch.float32)
    card = (x.argmax(-1) != x.shape[-1] - 1).sum(1).float()
    return {"cardinality_error": F.l1_loss(card, n)}


def loss_img_label(out, targets):
    """conditional_detr.py:225-235."""
    t = torch.stack([tt["img_label"] for tt in targets]).float()
    return {"img_label_logits": F.binary_cross_entropy_with_logits(out["x_logits"], t),
            "img_label_logits_tokens": F.binary_cross_entropy_with_logits(out["x_cls_logits"], t)}


def jitter_targets(targets, ratio, jitter, generator=None):
    """SetCriterion.forward training branch, conditional_detr.py:409-431: every GT box is replicated
    `ratio` times; the first ratio-1 copies are multiplicatively jittered candidates (1000 drawn,
    IoU>0.7 kept, first ratio-1 taken), the original stays last."""
    out = copy.deepcopy(targets)
    for t in out:
        reps = []
        for j in range(len(t["labels"])):
            box = t["boxes"][j].reshape(1, 4)
            scale = torch.cat([torch.empty((1000, 1)).uniform_(1 - jitter, 1 + jitter, generator=generator)
                               for _ in range(4)], dim=1)
            cand = scale * box
            iou, _ = box_iou(box_cxcywh_to_xyxy(cand), box_cxcywh_to_xyxy(box))
            keep = torch.where(iou.reshape(-1) > 0.7)[0]
            n = min(ratio - 1, keep.numel())
            rep = box.repeat(ratio, 1)
            rep[:n] = cand[keep[:n]]
            reps.append(rep)
        if reps:
            t["boxes"] = torch.cat(reps)
        t["labels"] = t["labels"].unsqueeze(1).repeat(1, ratio).reshape(-1)
        if "scores" in t:
            t["scores"] = t["scores"].unsqueeze(1).repeat(1, ratio).reshape(-1)
    return out


def set_criterion(outputs, targets, refine=False, alpha=0.25, gamma=2.0, world_size=1,
                  costs=(2.0, 5.0, 2.0), training=False, ratio=5, jitter=0.1, targets_cp=None, indices_out=None):
    """SetCriterion.forward (conditional_detr.py:399-466) / SetCriterionRefine.  `targets_cp`
    injects already-jittered targets (the train branch is RNG-stream dependent); otherwise they are
    drawn here when training=True.  losses = labels, boxes, cardinality (+ image_label if not
    refine), aux layers suffixed _{i}."""
    if targets_cp is None:
        targets_cp = jitter_targets(targets, ratio, jitter) if training else copy.deepcopy(targets)
    indices = hungarian(outputs, targets_cp, *costs)
    if indices_out is not None:
        indices_out.append(indices)
    num_boxes = max(float(sum(len(t["labels"]) for t in targets_cp)) / world_size, 1.0)
    losses = {}
    losses.update(loss_labels(outputs, targets_cp, indices, num_boxes, alpha, gamma, refine))
    losses.update(loss_boxes(outputs, targets_cp, indices, num_boxes, refine))
    losses.update(loss_cardinality(outputs, targets_cp))
    if not refine:
        losses.update(loss_img_label(outputs, targets_cp))
    for i, aux in enumerate(outputs.get("aux_outputs", [])):
        ind = hungarian(aux, targets_cp, *costs)
        if indices_out is not None:
            indices_out.append(ind)
        d = {}
        d.update(loss_labels(aux, targets_cp, ind, num_boxes, alpha, gamma, refine, log=False))
        d.update(loss_boxes(aux, targets_cp, ind, num_boxes, refine))
        d.update(loss_cardinality(aux, targets_cp))
        losses.update({k + f"_{i}": v for k, v in d.items()})
    return losses


# ------------------------------------------------------------------------------------------------
# post-processing (stage-0 detections -> stage-1 targets)
# ------------------------------------------------------------------------------------------------
def postprocess_refine(outputs, targets):
    """PostProcessRefine.forward, conditional_detr.py:641-677: for every class id present in the
    image's labels (ascending), the best query's score and NORMALISED cxcywh box."""
    prob = outputs["pred_logits"].sigmoid()
    top_v, top_i = prob.max(dim=1)                                          # [B,Kc]
    res = []
    for b, t in enumerate(targets):
        present = sorted(set(int(c) for c in t["labels"].tolist()))
        present = [c for c in present if c < prob.shape[2]]
        lab = torch.tensor(present, dtype=torch.int64)
        res.append({"scores": top_v[b, lab], "labels": lab, "boxes": outputs["pred_boxes"][b, top_i[b, lab]]})
    return res


def postprocess(outputs, target_sizes, keep=100):
    """PostProcess.forward, conditional_detr.py:592-623."""
    logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
    B, Q, Kc = logits.shape
    v, i = torch.topk(logits.sigmoid().view(B, -1), keep, dim=1)
    qi = torch.div(i, Kc, rounding_mode="floor")
    xyxy = box_cxcywh_to_xyxy(boxes).clamp(min=0)
    xyxy = torch.gather(xyxy, 1, qi.unsqueeze(-1).repeat(1, 1, 4))
    h, w = target_sizes.unbind(1)
    xyxy = xyxy * torch.stack([w, h, w, h], dim=1)[:, None, :]
    return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(v, i % Kc, xyxy)]


def weight_dict(cfg, coef=None):
    """conditional_detr.py:765-778."""
    coef = coef or dict(loss_ce=2, loss_bbox=2, img_label_logits=1, img_label_logits_tokens=1, loss_giou=2)
    wd = dict(coef)
    if cfg.aux_loss:
        for i in range(cfg.dec_layers - 1):
            wd.update({k + f"_{i}": v for k, v in coef.items()})
    return wd


def total_loss(sd, cfg, img, mask, targets, targets_refine_scores=None, pseudo=None):
    """One reference training iteration's scalar (engine.py:116-144 without the cv2 CAM step and the
    epoch curriculum): stage-0 criterion on `targets`, stage-1 (refine) criterion on PostProcessRefine
    pseudo labels of stage 0.  Eval-mode criterion (no jitter).  `pseudo`: stage-1 targets given by the caller
    (they are detached inputs of criterion_refine, engine.py:122-130) instead of this function's own."""
    out = model_forward(sd, cfg, img, mask)
    fg = float(getattr(cfg, "focal_gamma", 2.0))
    l0 = set_criterion(out[0], targets, refine=False, gamma=fg)
    if pseudo is None:
        with torch.no_grad():
            pseudo = postprocess_refine(out[0], targets)
    else:
        pseudo = [dict(p_) for p_ in pseudo]
    for p_, t_ in zip(pseudo, targets):
        p_["img_label"] = t_["img_label"]
    l1 = set_criterion(out[1], pseudo, refine=True, gamma=fg)
    wd = weight_dict(cfg)
    tot = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
    return tot, out, l0, l1


# ---- inference post-processing (engine_loc.py:99-124, 150-174) -------------------------------------
def nms_greedy(boxes, scores, iou_threshold):
    """torchvision.ops.nms semantics (the package is absent here; published algorithm): visit boxes by decreasing
    score, keep a box unless an already kept box overlaps it with IoU > threshold; returns kept indices in that order."""
    order = torch.argsort(scores, descending=True, stable=True).tolist()
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep = []
    for i in order:
        ok = True
        for j in keep:
            w = (torch.min(boxes[i, 2], boxes[j, 2]) - torch.max(boxes[i, 0], boxes[j, 0])).clamp(min=0)
            h = (torch.min(boxes[i, 3], boxes[j, 3]) - torch.max(boxes[i, 1], boxes[j, 1])).clamp(min=0)
            inter = w * h
            if inter / (area[i] + area[j] - inter) > iou_threshold:
                ok = False
                break
        if ok:
            keep.append(i)
    return torch.tensor(keep, dtype=torch.long)


def per_class_nms(result, iou_threshold=0.5):
    """engine_loc.py:154-174 for one image's PostProcess result."""
    kb, ks, kl = [], [], []
    for pc in result["labels"].unique().tolist():
        idx = (result["labels"] == pc).nonzero().reshape(-1)
        b, s, l = result["boxes"][idx], result["scores"][idx], result["labels"][idx]
        k = nms_greedy(b, s, iou_threshold)
        kb.append(b[k]); ks.append(s[k]); kl.append(l[k])
    return {"boxes": torch.cat(kb), "scores": torch.cat(ks), "labels": torch.cat(kl)}


def decouple_output(output, bs=2):
    """engine_loc.py:99-124."""
    out = {}
    for k, v in output.items():
        if k == "aux_outputs":
            out[k] = [decouple_output(a, bs) for a in v]
        elif k in ("x_logits", "x_cls_logits"):
            out[k] = torch.maximum(v[:bs], v[bs:2 * bs])
        elif k in ("pred_logits", "pred_boxes", "cams_cls"):
            pos = v[bs:2 * bs].clone()
            if k == "pred_boxes":
                pos[:, :, 0] = 1 - pos[:, :, 0]
            out[k] = torch.cat((v[:bs], pos), dim=1)
        else:
            out[k] = v
    return out
