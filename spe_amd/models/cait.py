"""CaiT / TSCAM backbone on libspe_hip.so kernels.

Host-side mirror of reference models/cait.py for the classes the hot path uses
(Attention_talking_head 344-393, LayerScale_Block 396-416, Multi_Class_Attention 91-139,
LayerScale_Block_CA_MultiClass 311-328, TSCAM_cait 531-670, TSCAM_cait_two_branch 674-831 and the
TSCAM factories 1465-1664), with identical parameter names/shapes (checkpoint compatible) and
batch-first [B, N, C] activations.  Every matmul/softmax/LayerNorm/residual is a HIP kernel launch
(spe_amd/ops.py); timm's Mlp / PatchEmbed / DropPath are restated here (timm is not a dependency).
"""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from .. import ops
from .layers import Dropout, LayerNorm, Linear, trunc_normal_

_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def create_model(name, pretrained=False, **kwargs):
    """timm.create_model stand-in: look the factory up, drop None kwargs."""
    if name not in _REGISTRY:
        raise RuntimeError(f"Unknown backbone {name!r}; available: {sorted(_REGISTRY)}")
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _REGISTRY[name](pretrained=pretrained, **kwargs)


class Mlp(nn.Module):
    """timm 0.4.x Mlp: fc1 -> GELU -> drop -> fc2 -> drop (GELU fused in the fc1 GEMM epilogue)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = Dropout(drop)

    def forward(self, x):
        if not (self.training and self.drop.p > 0.0):
            return ops.mlp_gelu(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        x = self.drop(self.fc1(x, ops.ACT_GELU))
        return self.drop(self.fc2(x))


class PatchEmbedMine(nn.Module):
    """timm PatchEmbed (Conv2d(in,embed,16,16)) with the size check relaxed (cait.py:518-528)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.img_size = img_size
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size[1] // patch_size) * (img_size[0] // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)  # parameters only

    def forward(self, x):
        return ops.patch_embed(x, self.proj.weight, self.proj.bias, self.patch_size[0])


class Attention_talking_head(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = float(attn_drop)
        self.proj = Linear(dim, dim)
        self.proj_l = nn.Linear(num_heads, num_heads)   # parameters only: mixed inside the score kernel
        self.proj_w = nn.Linear(num_heads, num_heads)
        self.proj_drop = Dropout(proj_drop)

    def context(self, x):
        """Attention output before the output projection."""
        if ops.qkv_talking_attention_ok(x, self.qkv.weight, self.qkv.bias, self.num_heads):
            # qkv Linear + attention as one autograd node: the gradient w.r.t. qkv only exists as the bf16 GEMM operand
            return ops.qkv_talking_attention(x, self.qkv.weight, self.qkv.bias, self.proj_l.weight, self.proj_l.bias, self.proj_w.weight,
                                             self.proj_w.bias, self.num_heads, self.scale, self.attn_drop if self.training else 0.0)
        qkv = self.qkv(x)
        return ops.talking_heads_attention(qkv, self.proj_l.weight, self.proj_l.bias, self.proj_w.weight,
                                           self.proj_w.bias, self.num_heads, self.scale,
                                           self.attn_drop if self.training else 0.0)

    def forward(self, x):
        return self.proj_drop(self.proj(self.context(x)))


class LayerScale_Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=None, norm_layer=LayerNorm, Attention_block=Attention_talking_head,
                 Mlp_block=Mlp, init_values=1e-4):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention_block(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                    attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp_block(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x, single_out=False):
        """single_out: the caller promises that the block's output feeds ONE LayerNorm-skip node (the next block's norm1) and nothing else - its
        LayerScale backward then rides on that norm's backward (ops._LayerNormSkip); the attention branch's output always does (norm2 below)."""
        B = x.shape[0]
        # norm.skip(x) -> (LN(x), x): the residual operand comes back through the LayerNorm node, whose backward kernel
        # adds the skip-path gradient to the normalisation's (one launch less per branch than autograd's sum)
        skip = isinstance(self.norm1, LayerNorm) and isinstance(self.norm2, LayerNorm)
        ss1 = ops.drop_path_scale(B, self.drop_path, self.training, x.device)
        y, xs = self.norm1.skip(x) if skip else (self.norm1(x), x)
        if isinstance(self.attn, Attention_talking_head):
            # output projection (+ proj_drop) + DropPath + LayerScale residual as one node (same arithmetic, same dropout stream)
            x = ops.linear_residual(self.attn.context(y), self.attn.proj.weight, self.attn.proj.bias, xs, self.gamma_1, ss1,
                                    self.attn.proj_drop.p if self.training else 0.0, single=skip)
        else:
            x = ops.layerscale_residual(xs, self.attn(y), self.gamma_1, ss1)
        ss = ops.drop_path_scale(B, self.drop_path, self.training, x.device)
        # the MLP of precision mode bf16s runs its forward products on fp16 operands: its LayerNorm emits the fp16 copy instead of the low part
        mlp16 = isinstance(self.mlp, Mlp) and x.is_cuda and K.mlp_f16_ok(x.numel() // x.shape[-1], x.shape[-1], self.mlp.fc1.weight.shape[0],
                                                                        self.mlp.fc2.weight.shape[0])
        y, xs = self.norm2.skip(x, mlp16) if skip else (self.norm2(x), x)
        if isinstance(self.mlp, Mlp):
            return ops.mlp_gelu_residual(y, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight,
                                         self.mlp.fc2.bias, xs, self.gamma_2, ss, self.mlp.drop.p if self.training else 0.0, single=single_out)
        return ops.layerscale_residual(xs, self.mlp(y), self.gamma_2, ss)


def _skip_norms(blk):
    """The block starts with a LayerNorm-skip node (both norms are spe_amd LayerNorms): what `single_out` of the block before it relies on."""
    return isinstance(blk, LayerScale_Block) and isinstance(blk.norm1, LayerNorm) and isinstance(blk.norm2, LayerNorm)


class Multi_Class_Attention(nn.Module):
    """Class attention with K+1 class tokens as queries; the softmax map is kept (it becomes the CAMs)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, num_classes=20):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.q = Linear(dim, dim, bias=qkv_bias)
        self.k = Linear(dim, dim, bias=qkv_bias)
        self.v = Linear(dim, dim, bias=qkv_bias)
        self.attn_drop = float(attn_drop)
        self.proj = Linear(dim, dim)
        self.proj_drop = Dropout(proj_drop)
        self.num_classes = num_classes
        self.attention_map = None

    def get_attention_map(self):
        return self.attention_map

    def forward(self, x, **kwargs):
        B, N, C = x.shape
        H, n_tok = self.num_heads, self.num_classes + 1
        q = self.q(x[:, :n_tok]).view(B, n_tok, H, C // H)
        k = self.k(x).view(B, N, H, C // H)
        v = self.v(x).view(B, N, H, C // H)
        o, amap = ops.attention(q, k, v, None, self.scale, self.attn_drop if self.training else 0.0, need_map=True)
        self.attention_map = amap
        return self.proj_drop(self.proj(o))


class LayerScale_Block_CA_MultiClass(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=None, norm_layer=LayerNorm, Attention_block=Multi_Class_Attention,
                 Mlp_block=Mlp, init_values=1e-4, num_classes=20):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention_block(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                    attn_drop=attn_drop, proj_drop=drop, num_classes=num_classes)
        self.drop_path = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp_block(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        self.num_classes = num_classes

    def forward(self, x, x_cls, **kwargs):
        B = x.shape[0]
        u = torch.cat((x_cls, x), dim=1)
        x_cls = ops.layerscale_residual(x_cls, self.attn(self.norm1(u)), self.gamma_1,
                                        ops.drop_path_scale(B, self.drop_path, self.training, x.device))
        x_cls = ops.layerscale_residual(x_cls, self.mlp(self.norm2(x_cls)), self.gamma_2,
                                        ops.drop_path_scale(B, self.drop_path, self.training, x.device))
        return x_cls


class _TSCAMBase(nn.Module):
    """Shared construction of cait_models (cait.py:421-516) + the TSCAM additions."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=LayerNorm, init_scale=1e-4, depth_token_only=2, mlp_ratio_clstk=4.0,
                 layer_to_det=23, **unused):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.patch_embed = PatchEmbedMine(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.pos_drop = Dropout(drop_rate)
        self._blk = dict(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         drop=drop_rate, attn_drop=attn_drop_rate, drop_path=drop_path_rate, norm_layer=norm_layer,
                         init_values=init_scale)
        self.blocks = nn.ModuleList([LayerScale_Block(**self._blk) for _ in range(depth)])
        self.blocks_token_only = nn.ModuleList([
            LayerScale_Block_CA_MultiClass(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio_clstk, qkv_bias=qkv_bias,
                                           qk_scale=qk_scale, drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=norm_layer,
                                           init_values=init_scale, num_classes=num_classes)
            for _ in range(depth_token_only)])
        self.norm = norm_layer(embed_dim)
        self.head = Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()   # unused, kept for checkpoints
        self.extra_cls_token = nn.Parameter(torch.zeros(1, num_classes, embed_dim))
        self.cls_head = Linear(embed_dim, 1)
        self.cls_head_multi_cls = Linear(embed_dim, num_classes)
        self.patch_size = patch_size
        self.layer_to_det = layer_to_det
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.cls_token, std=0.02)
        trunc_normal_(self.extra_cls_token, std=0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _resize_grid(self, pos_embed, size_hw):
        """Bicubic (align_corners=False) resize of the learned grid (cait.py:572-613)."""
        B, Q, E = pos_embed.shape
        gh, gw = self.img_size[0] // self.patch_size, self.img_size[1] // self.patch_size
        if pos_embed.is_cuda:
            return ops.bicubic_grid(pos_embed, gh, gw, size_hw[0], size_hw[1])
        # construction-time only (finetune_det runs on the CPU before the model is moved): parameter re-gridding
        pe = pos_embed.transpose(1, 2).reshape(B, E, gh, gw)
        pe = F.interpolate(pe, size=size_hw, mode="bicubic", align_corners=False)
        return pe.flatten(2).transpose(1, 2)

    def finetune_det(self, img_size=(800, 1344), use_checkpoint=False):
        """Re-grid the stored pos_embed Parameter to the detection resolution (50x84 by default)."""
        H, W = img_size
        with torch.no_grad():
            pe = self._resize_grid(self.pos_embed.data, (H // self.patch_size, W // self.patch_size))
        self.pos_embed = nn.Parameter(pe.contiguous())
        self.img_size = tuple(img_size)

    def InterpolateInitPosEmbed(self, pos_embed, img_size=(800, 1344)):
        H, W = img_size
        return self._resize_grid(pos_embed, (H // self.patch_size, W // self.patch_size))

    def _embed(self, x):
        B, _, H, W = x.shape
        tok = self.patch_embed(x)
        pe = self.InterpolateInitPosEmbed(self.pos_embed, img_size=(H, W))
        tok = self.pos_drop(ops.add_rows(tok, pe[0].contiguous()))
        cls = torch.cat((self.cls_token.expand(B, -1, -1), self.extra_cls_token.expand(B, -1, -1)), dim=1)
        return tok, cls.contiguous()

    def _heads(self, x, cls_tokens):
        for blk in self.blocks_token_only:
            cls_tokens = blk(x, cls_tokens)
        cls_n = self.norm(cls_tokens)      # LayerNorm is per token: only the class tokens are consumed downstream
        K = self.num_classes
        x_logits = self.cls_head(cls_n[:, 1:1 + K]).squeeze(-1)
        x_cls_logits = self.cls_head_multi_cls(cls_n[:, 0])
        return x_logits, x_cls_logits

    def _pack(self, x_logits, x_cls_logits, cams, x_feat, hw):
        B, N, C = x_feat.shape
        # x_patch is exposed as the reference's [B,C,h,w] *view* of the contiguous [B,N,C] features
        return {"x_logits": x_logits, "x_cls_logits": x_cls_logits, "cams_cls": cams,
                "x_patch": x_feat.transpose(1, 2).view(B, C, hw[0], hw[1])}


class TSCAM_cait(_TSCAMBase):
    """Single-branch TSCAM (cait.py:531-670): detection features tapped after block `layer_to_det`
    through norm_to_det; CAMs = head-mean of the first class-attention map."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.norm_to_det = kw.get("norm_layer", LayerNorm)(self.embed_dim)
        nn.init.constant_(self.norm_to_det.bias, 0)
        nn.init.constant_(self.norm_to_det.weight, 1.0)

    def forward(self, tensor_list):
        x, _ = tensor_list.decompose()
        B, _, H, W = x.shape
        hw = (H // self.patch_size, W // self.patch_size)
        x, cls_tokens = self._embed(x)
        x_feat = None
        nb = len(self.blocks)
        for i, blk in enumerate(self.blocks):
            # (a block output that only the next block's norm1 reads: not the tapped one, not the last)
            x = blk(x, single_out=(i != self.layer_to_det and i + 1 < nb and _skip_norms(self.blocks[i + 1])))
            if i == self.layer_to_det:
                x_feat = self.norm_to_det(x)
        x_logits, x_cls_logits = self._heads(x, cls_tokens)
        K = self.num_classes
        with torch.no_grad():
            amap = self.blocks_token_only[0].attn.get_attention_map()          # [B,H,K+1,K+1+N]
            cams = amap[:, :, 1:1 + K, 1 + K:].mean(1).reshape(B, K, hw[0], hw[1])
        return self._pack(x_logits, x_cls_logits, cams, x_feat, hw)


class TSCAM_cait_two_branch(_TSCAMBase):
    """Two-branch TSCAM (cait.py:674-831): x is cloned before block `layer_to_det`, runs through
    blocks_det + norm_det; CAMs are std-reweighted over heads (cait.py:801-806)."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.blocks_det = nn.ModuleList([LayerScale_Block(**self._blk) for _ in range(self.layer_to_det, self.depth)])
        self.norm_det = kw.get("norm_layer", LayerNorm)(self.embed_dim)
        self.blocks_det.apply(self._init_weights)
        self.norm_det.apply(self._init_weights)

    def init_blocks_det_weight(self):
        for i in range(1, 1 + len(self.blocks_det)):
            self.blocks_det[-i].load_state_dict(self.blocks[-i].state_dict(), strict=True)

    def forward(self, tensor_list):
        x, _ = tensor_list.decompose()
        B, _, H, W = x.shape
        hw = (H // self.patch_size, W // self.patch_size)
        x, cls_tokens = self._embed(x)
        x_feat = None
        nb, nd = len(self.blocks), len(self.blocks_det)
        for i, blk in enumerate(self.blocks):
            x = blk(x, single_out=(i + 1 != self.layer_to_det and i + 1 < nb and _skip_norms(self.blocks[i + 1])))
            if i + 1 == self.layer_to_det:
                x_feat = x
        for i, blk in enumerate(self.blocks_det):
            x_feat = blk(x_feat, single_out=(i + 1 < nd and _skip_norms(self.blocks_det[i + 1])))
        x_feat = self.norm_det(x_feat)
        x_logits, x_cls_logits = self._heads(x, cls_tokens)
        K = self.num_classes
        with torch.no_grad():
            cam = self.blocks_token_only[0].attn.get_attention_map()[..., 1:1 + K, 1 + K:]
            std = torch.std(cam, dim=-1, keepdim=True)
            std = std - std.min(dim=1, keepdim=True)[0]
            std = std / std.max(dim=1, keepdim=True)[0]
            cams = (cam * std).sum(1).reshape(B, K, hw[0], hw[1])
        return self._pack(x_logits, x_cls_logits, cams, x_feat, hw)


# ------------------------------------------------------------------------------------------------
# factories.  XXS* mirror cait.py:1465-1664; S24/S36 combine the plain-classifier dims at
# cait.py:1860-1866 / 1882-1888 with the TSCAM classes (BASELINE.json configs 2-5).
# ------------------------------------------------------------------------------------------------
_LN6 = partial(LayerNorm, eps=1e-6)


def load_deit_checkpoint(model, path, strict=False):
    """DeiT CaiT checkpoints store 'module.'-prefixed keys under 'model' (cait.py:1639-1663)."""
    ck = torch.load(path, map_location="cpu")
    ck = ck.get("model", ck)
    own = model.state_dict()
    sel = {}
    for k in own:
        src = ck.get("module." + k, ck.get(k))
        if src is not None and src.shape == own[k].shape:
            sel[k] = src
    return model.load_state_dict(sel, strict=strict)


def _make(cls, width, depth, heads, init_scale, pretrained, ckpt=None, img_size=384, **kwargs):
    ckpt = kwargs.pop("checkpoint_path", ckpt)
    model = cls(img_size=img_size, patch_size=16, embed_dim=width, depth=depth, num_heads=heads, mlp_ratio=4, qkv_bias=True,
                norm_layer=_LN6, init_scale=init_scale, depth_token_only=2, **kwargs)
    if pretrained and ckpt:   # no network on this path: weights come from a local file or not at all
        load_deit_checkpoint(model, ckpt)
        if hasattr(model, "init_blocks_det_weight"):
            model.init_blocks_det_weight()
    return model, width


@register_model
def TSCAM_cait_XXS24(pretrained=False, **kw):
    return _make(TSCAM_cait, 192, 24, 4, 1e-5, pretrained, **kw)


@register_model
def TSCAM_cait_XXS36(pretrained=False, **kw):
    return _make(TSCAM_cait, 192, 36, 4, 1e-5, pretrained, **kw)


@register_model
def TSCAM_cait_XXS36_Two_Branch(pretrained=False, **kw):
    return _make(TSCAM_cait_two_branch, 192, 36, 4, 1e-5, pretrained, **kw)


@register_model
def TSCAM_cait_S24(pretrained=False, **kw):
    return _make(TSCAM_cait, 384, 24, 8, 1e-5, pretrained, **kw)


@register_model
def TSCAM_cait_S24_Two_Branch(pretrained=False, **kw):
    return _make(TSCAM_cait_two_branch, 384, 24, 8, 1e-5, pretrained, **kw)


@register_model
def TSCAM_cait_S36(pretrained=False, **kw):
    return _make(TSCAM_cait, 384, 36, 8, 1e-6, pretrained, **kw)


@register_model
def TSCAM_cait_S36_Two_Branch(pretrained=False, **kw):
    return _make(TSCAM_cait_two_branch, 384, 36, 8, 1e-6, pretrained, **kw)


@register_model
def TSCAM_cait_tiny(pretrained=False, **kw):
    """C=32, depth 3, 4 heads: the shape of the committed end-to-end golden (tests/golden/e2e_single.pt)."""
    return _make(TSCAM_cait, 32, 3, 4, 1e-5, pretrained, img_size=64, **kw)


@register_model
def TSCAM_cait_tiny_Two_Branch(pretrained=False, **kw):
    return _make(TSCAM_cait_two_branch, 32, 4, 4, 1e-5, pretrained, img_size=64, **kw)
