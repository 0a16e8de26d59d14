"""Parameter containers with the reference's (torch.nn) names and initialisation whose forward is
a libspe_hip.so launch.  Same state_dict keys as nn.Linear / nn.LayerNorm."""
import torch
import torch.nn as nn

from .. import ops


class Linear(nn.Linear):
    """nn.Linear parameters; forward = MFMA GEMM with fused bias (+ReLU/GELU)."""

    def forward(self, x, act=ops.ACT_NONE):
        return ops.linear(x, self.weight, self.bias, act)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def residual(self, x, z, drop):
        """norm(x + drop(z)) as one kernel (drop: the layer's Dropout module)."""
        return ops.res_drop_layer_norm(x, z, self.weight, self.bias, self.eps, drop.p, drop.training)

    def residual2(self, x, z, drop):
        """(y, y_skip) with y = norm(x + drop(z)): the same values twice, for an output with two consumers (ops._ResDropLayerNorm2)."""
        return ops.res_drop_layer_norm2(x, z, self.weight, self.bias, self.eps, drop.p, drop.training)

    def skip(self, x, f16=False):
        """(LN(x), x) for a pre-norm residual branch: pass the second result to the residual add (ops._LayerNormSkip).  f16: the branch
        starts with a single-term fp16 product (the backbone MLP in precision mode bf16s)."""
        return ops.layer_norm_skip(x, self.weight, self.bias, self.eps, f16)


class Dropout(nn.Module):
    """nn.Dropout semantics with a counter-based (Philox) mask regenerated in backward."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        return ops.dropout(x, self.p, self.training)


class MLP(nn.Module):
    """Reference models/transformer.py:21-33 / conditional_detr.py:626-638: Linear stack, ReLU between."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x, ops.ACT_RELU if i < self.num_layers - 1 else ops.ACT_NONE)
        return x


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """timm trunc_normal_ (reference models/layers/weight_init.py:6-60)."""
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)
