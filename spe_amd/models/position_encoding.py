"""Sine position embedding of the (padded) patch grid - the `position_embedding` the reference attaches to the backbone
(models/position_encoding.py:21-57, built at :88-92 with N_steps = hidden_dim // 2, normalize=True).  Evaluated by one
HIP launch (csrc/misc.hip: pos_sine_kernel) on the [B,h,w] padding mask; no parameters, no gradient."""
import math

import torch
from torch import nn

from .. import kernels as K


class PositionEmbeddingSine(nn.Module):
    EPS = 1e-6

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        # per-feature wavelengths temperature^(2*(k//2)/n): a constant table (not part of the state dict)
        k = torch.arange(num_pos_feats, dtype=torch.float32)
        self.register_buffer("_dim_t", temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / num_pos_feats),
                             persistent=False)

    @torch.no_grad()
    def forward(self, tensor_list):
        mask = tensor_list.mask
        assert mask is not None
        if self._dim_t.device != mask.device:
            self._dim_t = self._dim_t.to(mask.device)
        feats = K.pos_sine(mask, self._dim_t, self.num_pos_feats, self.scale, self.EPS, self.normalize)
        return feats.permute(0, 3, 1, 2)        # the reference's [B,d,h,w]; the transformer consumes the [B,hw,d] buffer


def build_position_encoding(args):
    if args.position_embedding not in ("v2", "sine"):
        raise ValueError(f"not supported {args.position_embedding}")
    return PositionEmbeddingSine(args.hidden_dim // 2, normalize=True)
