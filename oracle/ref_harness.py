"""Import harness for the *Python reference* (only usable where /root/reference exists).

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_goldens.py to run the reference on CPU in the
build container and capture golden vectors into tests/golden/.  Nothing here ships to the GPU
box's product path and the reference never travels.

The reference files are imported untouched.  Obstacles handled harness-side:
  * `import cv2`, `from torchvision import io, transforms`, `import wandb` (functions.py:5,18,30)
    -> empty stub modules pre-seeded in sys.modules (never called on the hot path).
  * `torch.set_default_device('cuda')` (functions.py:52) -> no-op during import.
  * `device = torch.device('cuda')` (functions.py:49) -> rebound to cpu in every star-importing module.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("GNGF_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models.py"))


def load_reference():
    """Returns (functions, utils, models, params) reference modules running on CPU."""
    import torch
    import matplotlib
    matplotlib.use("Agg")

    for name in ("cv2", "wandb"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.io = types.ModuleType("torchvision.io")
        tv.transforms = types.ModuleType("torchvision.transforms")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.io"] = tv.io
        sys.modules["torchvision.transforms"] = tv.transforms

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    real_set_default_device = torch.set_default_device
    torch.set_default_device = lambda *_a, **_k: None
    try:
        functions = importlib.import_module("functions")
        utils = importlib.import_module("utils")
        models = importlib.import_module("models")
        params = importlib.import_module("params")
    finally:
        torch.set_default_device = real_set_default_device

    cpu = torch.device("cpu")
    for m in (functions, utils, models):
        m.device = cpu
    return functions, utils, models, params


def set_flag(mods, name, value):
    """Behaviour switches are module globals read at call time (models.py:181,212,342,412)."""
    for m in mods:
        setattr(m, name, value)
