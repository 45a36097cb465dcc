"""Container-only loader for the *real* reference (ClashLuke/LocAtE at /root/reference).

TEST INFRASTRUCTURE - never imported by the product (`locate_amd/`), by `bench.py`'s timed
region or on the GPU box (the reference tree does not exist there).  It is used by
`oracle/gen_golden.py` alone, to emit the fixtures under `tests/golden/` (the oracle itself is then checked against
those fixtures by `tests/test_oracle_golden.py`, which needs no reference tree).

What it does:
  * never writes bytecode into the read-only reference tree;
  * registers empty `torchvision*` modules (the hot path never touches torchvision, which is not
    installed here; `libs/utils.py:8-10` imports it at module import time);
  * registers a bare `libs` package object so `libs/__init__.py` is skipped and `libs.config` can be
    patched *before* the modules that bind its constants are imported (`libs/models.py:5-6`);
  * never imports `main.py` (it downloads a dataset at import, `main.py:27-36`).

One process = one configuration (the constants are bound at import time).
"""
import importlib
import math
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("LOCATE_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "libs"))


def load_reference(image_size=128, base_feature_factor=8, quiet=True, **constants):
    """Import the reference's `libs.*` modules for one configuration; returns a namespace.  `constants` overrides further
    libs/config.py switches by name (DEPTH, FEATURE_MULTIPLIER, SEPARABLE) before the modules that bind them are imported."""
    if "libs" in sys.modules:
        raise RuntimeError("reference already imported in this process (one process = one config)")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    names = ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils")
    for name in names:
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = sys.modules["torchvision"]
    for sub in ("datasets", "transforms", "utils"):
        setattr(tv, sub, sys.modules["torchvision." + sub])
    import matplotlib
    matplotlib.use("Agg")

    pkg = types.ModuleType("libs")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "libs")]
    sys.modules["libs"] = pkg

    saved_stdout = sys.stdout
    if quiet:
        sys.stdout = open(os.devnull, "w")
    try:
        cfg = importlib.import_module("libs.config")  # seeds torch with 999 and prints
        cfg.IMAGE_SIZE = image_size
        cfg.BASE_FEATURE_FACTOR = base_feature_factor
        # derived constants, same arithmetic as libs/config.py:50,60-61,65
        cfg.LAYERS = int(math.log(cfg.IMAGE_SIZE, 2))
        cfg.GEN_FEATURES = cfg.FACTOR ** int(math.log(cfg.IMAGE_SIZE, cfg.G_STRIDE)) * cfg.BASE_FEATURE_FACTOR * 3
        cfg.DIS_FEATURES = cfg.FACTOR ** int(math.log(cfg.IMAGE_SIZE, cfg.D_STRIDE)) * cfg.BASE_FEATURE_FACTOR
        cfg.INPUT_VECTOR_Z = cfg.IMAGE_SIZE
        for key, value in constants.items():
            if not hasattr(cfg, key):
                raise KeyError("libs/config.py has no constant %r" % key)
            setattr(cfg, key, value)
        ns = types.SimpleNamespace(config=cfg)
        for mod in ("activation", "inplace_norm", "merge", "spectral_norm", "conv", "attention", "scale",
                    "linear", "util_modules", "block", "models", "utils", "nadam", "grad_penalty"):
            setattr(ns, mod, importlib.import_module("libs." + mod))
    finally:
        if quiet:
            sys.stdout.close()
            sys.stdout = saved_stdout
    return ns


class quiet_stdout:
    """The reference prints at construction (`libs/block.py:69`); silence it."""

    def __enter__(self):
        self._saved = sys.stdout
        sys.stdout = open(os.devnull, "w")

    def __exit__(self, *exc):
        sys.stdout.close()
        sys.stdout = self._saved
        return False
