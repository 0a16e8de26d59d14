"""Import the reference (MingXiangL/SPE, read-only at /root/reference) in THIS container so its
outputs can be captured as golden vectors.  Harness-only shims; nothing from the reference is
copied into the repo and this module is never imported on the GPU box.

Shims:
  1. torch.nn.modules.linear._LinearWithBias (removed in torch>=1.9) -> NonDynamicallyQuantizableLinear
  2. stub `torchvision` (box_area, _is_tracing, __version__, ops.misc.interpolate, IntermediateLayerGetter)
  3. stub `timm` with 0.4.x semantics: Mlp, PatchEmbed, _cfg, register_model, trunc_normal_, DropPath,
     create_model (drops None kwargs, forces pretrained=False - no network here)
  4. Tensor.get_device() patched to return the device object on CPU tensors (the reference uses
     get_device() as a device handle, conditional_detr.py:230,654)
"""
import argparse
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"
_installed = False


def install_shims(ref=REF):
    global _installed
    if _installed:
        return
    import torch.nn.modules.linear as tl
    if not hasattr(tl, "_LinearWithBias"):
        tl._LinearWithBias = tl.NonDynamicallyQuantizableLinear

    # ---- torchvision stub
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.9.0"
    tv._is_tracing = lambda: False
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")
    boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    misc = types.ModuleType("torchvision.ops.misc")
    misc.interpolate = torch.nn.functional.interpolate
    ops.boxes, ops.misc = boxes, misc
    models = types.ModuleType("torchvision.models")
    mutils = types.ModuleType("torchvision.models._utils")
    mutils.IntermediateLayerGetter = type("IntermediateLayerGetter", (nn.ModuleDict,), {})
    models._utils = mutils
    tv.ops, tv.models = ops, models
    for n, m in [("torchvision", tv), ("torchvision.ops", ops), ("torchvision.ops.boxes", boxes),
                 ("torchvision.ops.misc", misc), ("torchvision.models", models),
                 ("torchvision.models._utils", mutils)]:
        sys.modules[n] = m

    # ---- timm stub (0.4.x semantics)
    registry = {}

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
            patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
            self.img_size, self.patch_size = img_size, patch_size
            self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class DropPath(nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            rt = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
            rt.floor_()
            return x.div(keep) * rt

    def register_model(fn):
        registry[fn.__name__] = fn
        return fn

    def create_model(name, pretrained=False, **kw):
        kw = {k: v for k, v in kw.items() if v is not None}
        return registry[name](pretrained=False, **kw)

    def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
        return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tvt = types.ModuleType("timm.models.vision_transformer")
    tvt.Mlp, tvt.PatchEmbed, tvt._cfg = Mlp, PatchEmbed, (lambda **kw: {})
    treg = types.ModuleType("timm.models.registry")
    treg.register_model = register_model
    tl_ = types.ModuleType("timm.models.layers")
    tl_.trunc_normal_, tl_.DropPath = trunc_normal_, DropPath
    tm.vision_transformer, tm.registry, tm.layers, tm.create_model = tvt, treg, tl_, create_model
    timm.models, timm.create_model = tm, create_model
    for n, m in [("timm", timm), ("timm.models", tm), ("timm.models.vision_transformer", tvt),
                 ("timm.models.registry", treg), ("timm.models.layers", tl_)]:
        sys.modules[n] = m

    # ---- get_device as a device handle on CPU
    _orig = torch.Tensor.get_device

    def get_device(self):
        return self.device if not self.is_cuda else _orig(self)
    torch.Tensor.get_device = get_device

    if ref not in sys.path:
        sys.path.insert(0, ref)
    _installed = True
    return registry


def default_args(**over):
    """Namespace mirroring reference main.py:37-146 defaults (only what build_model reads)."""
    a = argparse.Namespace(
        dataset_file="voc", device="cpu", backbone="TSCAM_cait_XXS24", backbone_drop_rate=0.0,
        drop_path_rate=0.0, drop_block_rate=0.0, drop_attn_rate=0.0, layer_to_det=23, lr_backbone=1e-5,
        masks=False, dilation=False, position_embedding="sine", hidden_dim=256, dropout=0.1, nheads=8,
        num_queries=300, dim_feedforward=2048, enc_layers=6, dec_layers=6, pre_norm=False, aux_loss=True,
        num_refines=1, frozen_weights=None, set_cost_class=2, set_cost_bbox=5, set_cost_giou=2,
        hung_match_ratio=5, hungarian_multi=False, box_jitter=0.1, cls_loss_coef=2, bbox_loss_coef=2,
        giou_loss_coef=2, img_label_loss_coef=1, img_label_tokens_loss_coef=1, mask_loss_coef=1,
        dice_loss_coef=1, focal_alpha=0.25, focal_gamma=2, drloc=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def register_tiny_backbones():
    """Tiny TSCAM factories registered through the reference's own decorator so that
    build_model(args) constructs them exactly as it constructs TSCAM_cait_XXS24."""
    install_shims()
    from functools import partial
    import models.cait as rc
    from timm.models.registry import register_model

    @register_model
    def TSCAM_cait_tiny(pretrained=False, **kwargs):
        m = rc.TSCAM_cait(img_size=64, patch_size=16, embed_dim=32, depth=3, num_heads=4, mlp_ratio=4, qkv_bias=True,
                          norm_layer=partial(nn.LayerNorm, eps=1e-6), init_scale=1e-5, depth_token_only=2, **kwargs)
        return m, 32

    @register_model
    def TSCAM_cait_tiny_Two_Branch(pretrained=False, **kwargs):
        m = rc.TSCAM_cait_two_branch(img_size=64, patch_size=16, embed_dim=32, depth=4, num_heads=4, mlp_ratio=4,
                                     qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), init_scale=1e-5,
                                     depth_token_only=2, **kwargs)
        return m, 32


if __name__ == "__main__":
    install_shims()
    import contextlib, io
    from models import build_model
    args = default_args(enc_layers=0, dec_layers=1, num_queries=10)
    with contextlib.redirect_stdout(io.StringIO()):
        model, crit, crit_r, pp, rpp = build_model(args)
    print(type(model).__name__, sum(p.numel() for p in model.parameters()), args.hidden_dim)
