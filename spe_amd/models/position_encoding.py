"""Sine position embedding (mirror of reference models/position_encoding.py:21-57, 88-92)."""
import math

import torch
from torch import nn

from ..util.misc import NestedTensor


class PositionEmbeddingSine(nn.Module):
    """normalize=True, scale=2*pi variant the reference builds.  No parameters, no gradient; the
    [B,h,w] mask cumsums and sin/cos are tiny and stay as device-side tensor ops (plumbing)."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.scale = 2 * math.pi if scale is None else scale

    @torch.no_grad()
    def forward(self, tensor_list: NestedTensor):
        x, mask = tensor_list.tensors, tensor_list.mask
        assert mask is not None
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / self.num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
        # [B,h,w,2*npf] is the layout the transformer consumes; expose the reference's [B,d,h,w] view
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def build_position_encoding(args):
    n_steps = args.hidden_dim // 2
    if args.position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(n_steps, normalize=True)
    raise ValueError(f"not supported {args.position_embedding}")
