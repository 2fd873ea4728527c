"""Plain ReLU MLP, the MIMIC `static` tower (reference: modules/mlp.py:4-27): 5 -> 64 -> 64 -> 64,
~0.03 MFLOP/sample.  Left on torch ops; module_list indices follow the reference
(Linear at 3*i, output Linear at 3*num_blocks) so checkpoints load."""
from __future__ import annotations

from torch import nn


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, num_blocks, output_dim=None, dropout=0., **kwargs):
        super().__init__()
        self.output_dim = output_dim
        layers = []
        for i in range(num_blocks):
            layers += [nn.Linear(input_dim if i == 0 else hidden_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout)]
        if output_dim is not None:
            layers.append(nn.Linear(hidden_dim, output_dim))
        self.module_list = nn.ModuleList(layers)

    def forward(self, x):
        for layer in self.module_list:
            x = layer(x)
        return x
