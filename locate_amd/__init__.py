"""locate_amd: the LocAtE generator/discriminator training step on MI355X (gfx950).

Public names follow the reference's `libs/__init__.py:1-10` for the hot path: NonLinear, BlockBlock,
Generator, Discriminator, SpectralNorm, get_model, hinge-based step (`TrainStep`), parameter_count, Nadam."""
from .config import NetConfig, get_default, set_default  # noqa: F401
from .nn import (ActivatedBaseConv, Block, BlockBlock, CatModule, DeepResidualConv, Expand, FeaturePooling,  # noqa: F401
                 InPlaceNorm, LinearModule, NonLinear, Norm, ResModule, RootTanhModule, Scale, SelfAttention,
                 SpectralNorm, feature_attention)
from .models import Discriminator, Generator, get_model, init, parameter_count  # noqa: F401
from .optim import Nadam  # noqa: F401
from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401
from .train import TrainLoop, TrainStep  # noqa: F401
