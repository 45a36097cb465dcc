"""Network / training configuration as constructor arguments.

The reference binds these as module-level constants at import time (libs/config.py:1-78, so changing one means
patching the module before the others import it); the values are inputs of the hot path, the mechanism is not
reproduced.  `Generator()` / `Discriminator()` with no argument use `get_default()`; set another default with
`set_default(NetConfig(image_size=64))` or pass `cfg=` explicitly."""
import math
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class NetConfig:
    image_size: int = 128            # IMAGE_SIZE            config.py:37
    base_feature_factor: int = 8     # BASE_FEATURE_FACTOR   config.py:58
    factor: int = 2                  # FACTOR                config.py:44
    g_stride: int = 2                # G_STRIDE              config.py:48
    d_stride: int = 2                # D_STRIDE              config.py:49
    bottleneck: int = 4              # BOTTLENECK            config.py:62
    min_attention_size: int = 8      # MIN_ATTENTION_SIZE    config.py:63
    attention_every_nth_layer: int = 2   # ATTENTION_EVERY_NTH_LAYER config.py:64
    depth: int = 1                   # DEPTH                 config.py:68
    feature_multiplier: int = 1      # FEATURE_MULTIPLIER    config.py:55
    separable: bool = False          # SEPARABLE             config.py:53 (depthwise k x k convs, grouped feature attention)
    glr: float = 5e-4                # GLR                   config.py:70
    dlr: float = 2e-3                # DLR                   config.py:71
    beta1: float = 0.5               # BETA_1                config.py:72
    beta2: float = 0.9               # BETA_2                config.py:73
    seed: int = 999                  # config.py:7

    @property
    def layers(self):                # LAYERS  config.py:50
        return int(math.log(self.image_size, 2))

    @property
    def input_vector_z(self):        # INPUT_VECTOR_Z  config.py:65
        return self.image_size

    @property
    def gen_features(self):          # GEN_FEATURES  config.py:60
        return self.factor ** int(math.log(self.image_size, self.g_stride)) * self.base_feature_factor * 3

    @property
    def dis_features(self):          # DIS_FEATURES  config.py:61
        return self.factor ** int(math.log(self.image_size, self.d_stride)) * self.base_feature_factor

    def with_(self, **kw):
        return replace(self, **kw)


_default = NetConfig()


def get_default():
    return _default


def set_default(cfg):
    global _default
    if not isinstance(cfg, NetConfig):
        raise TypeError("expected a NetConfig")
    _default = cfg
