"""String -> class registry, the reference's plugin API (reference: modules/__init__.py:12-26).

`get_block_by_name(**cfg)` looks `cfg['block_type']` up in this namespace and calls it with the WHOLE
cfg dict (so every block swallows unknown kwargs); same for fusions (`fusion_function`) and
classifiers (`classifier`).
"""
import sys

from .mixer import FeedForward, MixerBlock, FusionMixer, MLPMixer, MLPMixerNoPatching
from .fusion import (ConcatFusion, ConcatDynaFusion, MaxFusion, SumFusion, MeanFusion, ExtraConcatFusion,
                     BiModalGatedUnit)
from .classification import StandardClassifier
from .mlp import MLP


def _lookup(key_name, kwargs):
    try:
        return getattr(sys.modules[__name__], kwargs[key_name])
    except AttributeError as e:
        raise AttributeError(f"m2_mixer_amd.modules has no {key_name} '{kwargs[key_name]}' "
                             "(only the M2-Mixer hot-path blocks are provided)") from e


def get_block_by_name(**kwargs):
    return _lookup('block_type', kwargs)(**kwargs)


def get_fusion_by_name(**kwargs):
    return _lookup('fusion_function', kwargs)(**kwargs)


def get_classifier_by_name(**kwargs):
    return _lookup('classifier', kwargs)(**kwargs)
