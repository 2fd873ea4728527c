"""Classification heads (reference: modules/classification.py).

StandardClassifier is the head of every BASELINE config: mean over all tokens, then Linear.  Inside
the fused training engine the three heads, the cross-entropies and their backward are ONE HIP launch
(csrc/heads.hip); this module is the stand-alone form with the reference's (misspelled) state-dict
key `classifer`.
"""
from __future__ import annotations

import torch
from torch import nn


class StandardClassifier(nn.Module):
    def __init__(self, input_shape, num_classes: int, **kwargs):
        super().__init__()
        self.classifer = nn.Linear(input_shape[-1], num_classes)   # sic: reference key, classification.py:87

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        pooled = inputs.reshape(inputs.shape[0], -1, inputs.shape[-1]).mean(dim=1)
        return self.classifer(pooled)
