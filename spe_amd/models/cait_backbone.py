"""Backbone wrapper (mirror of reference models/cait_backbone.py:67-121)."""
import torch
import torch.nn.functional as F
from torch import nn

from ..util.misc import NestedTensor
from .cait import create_model
from .position_encoding import build_position_encoding


class Backbone(nn.Module):
    def __init__(self, name, train_backbone, return_interm_layers, dilation, args=None):
        super().__init__()
        if args.dataset_file == "coco":
            num_classes = 90
        elif "voc" in args.dataset_file:
            num_classes = 20
        else:
            raise ValueError(f"dataset_file {args.dataset_file!r}: expected 'coco' or '*voc*'")
        # The reference hard-codes pretrained=True (a torch.hub download, cait_backbone.py:76); there is
        # no network on this path, so weights load only from args.backbone_checkpoint when given.
        ckpt = getattr(args, "backbone_checkpoint", None)
        self.body, num_channels = create_model(
            args.backbone, pretrained=ckpt is not None, checkpoint_path=ckpt, num_classes=num_classes,
            drop_rate=args.backbone_drop_rate, drop_path_rate=args.drop_path_rate, drop_block_rate=None,
            attn_drop_rate=args.drop_attn_rate, layer_to_det=args.layer_to_det)
        self.num_channels = num_channels
        args.hidden_dim = num_channels          # the reference overwrites --hidden_dim (cait_backbone.py:85)

    def forward(self, tensor_list: NestedTensor):
        out = self.body(tensor_list)
        x = out["x_patch"]
        m = tensor_list.mask
        assert m is not None
        mask = F.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]
        out["x_patch"] = NestedTensor(x, mask)
        return out


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list: NestedTensor):
        out = self[0](tensor_list)
        x = out["x_patch"]
        pos = [self[1](x).to(x.tensors.dtype)]
        return out, pos


def build_backbone(args):
    train_backbone = args.lr_backbone > 0
    backbone = Backbone(args.backbone, train_backbone, args.masks, args.dilation, args=args)
    backbone.body.finetune_det()
    model = Joiner(backbone, build_position_encoding(args))
    model.num_channels = backbone.num_channels
    return model
