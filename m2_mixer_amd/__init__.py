"""m2_mixer_amd: MI355X-native training hot path of M2-Mixer (towers of MixerBlocks + fusion mixer +
multi-head loss) behind the reference's module / registry / LightningModule-shaped surface."""
from .config import set_precision, get_precision, set_dropout_seed  # noqa: F401
from . import modules  # noqa: F401
