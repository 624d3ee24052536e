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
        fast = self._linear_chain(x)
        return self.layers(x) if fast is None else fast

    def _linear_chain(self, x):
        """Conv1d(kernel 1) over [B, C, L] == a linear layer over the L*B rows of [L, B, C]: on the device the
        stack runs on the few-row linear kernels (ReLU in the same launch) instead of the library's convolution —
        whose backward for this 128x128x1 filter is MIOpen's naive direct kernel.  None when the stack is anything
        other than Conv1d(k=1) / ReLU on f32 HIP tensors (the stock modules run then)."""
        import torch
        from ... import ops
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3):
            return None
        mods = list(self.layers)
        plan, i = [], 0
        while i < len(mods):
            m = mods[i]
            if not (isinstance(m, nn.Conv1d) and m.kernel_size == (1,) and m.stride == (1,) and m.groups == 1
                    and m.padding == (0,) and m.in_channels % 32 == 0 and m.out_channels % 32 == 0):
                return None
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            plan.append((m, relu))
            i += 2 if relu else 1
        h = x.permute(0, 2, 1)                                          # [B, L, C] rows
        for m, relu in plan:
            h = ops.linear(h, m.weight.view(m.out_channels, m.in_channels), m.bias, relu=relu)
        return h.permute(0, 2, 1)
