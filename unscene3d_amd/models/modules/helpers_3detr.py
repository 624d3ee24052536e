"""GenericMLP (reference models/modules/helpers_3detr.py:45-112): a Conv1d/Linear stack kept in
`self.layers` (nn.Sequential) so that checkpoint keys (`layers.0.weight`, …) match."""
from functools import partial

import torch.nn as nn

_ACT = {"relu": nn.ReLU, "gelu": nn.GELU, "leakyrelu": partial(nn.LeakyReLU, negative_slope=0.1)}


class GenericMLP(nn.Module):
    def __init__(self, input_dim, hidden_dims, output_dim, norm_fn_name=None, activation="relu", use_conv=False,
                 dropout=None, hidden_use_bias=False, output_use_bias=True, output_use_activation=False,
                 output_use_norm=False, weight_init_name=None):
        super().__init__()
        if norm_fn_name is not None:
            raise NotImplementedError("normalised GenericMLP variants are not used by Mask3D (mask3d.py:74-81)")
        act = _ACT[activation]
        make = (lambda i, o, b: nn.Conv1d(i, o, 1, bias=b)) if use_conv else (lambda i, o, b: nn.Linear(i, o, bias=b))
        if dropout is not None and not isinstance(dropout, list):
            dropout = [dropout] * len(hidden_dims)
        mods, prev = [], input_dim
        for j, width in enumerate(hidden_dims):
            mods += [make(prev, width, hidden_use_bias), act()]
            if dropout is not None:
                mods.append(nn.Dropout(p=dropout[j]))
            prev = width
        mods.append(make(prev, output_dim, output_use_bias))
        if output_use_activation:
            mods.append(act())
        self.layers = nn.Sequential(*mods)
        if weight_init_name == "xavier_uniform":
            for p in self.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)

    def forward(self, x):
        return self.layers(x)
