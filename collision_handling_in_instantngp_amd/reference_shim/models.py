"""Zero-edit replacement for the reference's `models.py`.

The reference's entry point star-imports four flat modules (`main.py:1-4`: functions, utils, models, params) and hands the
model class to its training loop.  Put THIS directory in front of the reference on the module search path and run the
reference's `main.py` as it is:

    PYTHONPATH=<repo>/collision_handling_in_instantngp_amd/reference_shim:<repo>:<reference> python main.py -f strawberry.jpeg ...

`from models import *` then yields the four class names of the reference's models.py — backed by the gfx950 kernels — and
nothing else (`__all__`).  The behaviour switches stay where the reference keeps them: module globals of `params.py`
(`should_use_hash_function`, `should_softmax_topk_features`, `should_leaky_relu`, `should_batchnorm_data`,
`should_inplace_scatter`; `params.py:1-23`).  They are read from the caller's `params` module at call time (constructor and
every forward), so flipping one in `params` — the reference's own way of selecting "GNGF off" — takes effect here as well.
"""
import sys

from collision_handling_in_instantngp_amd import models as _hip

__all__ = ["DifferentiableTopk", "HashProbDistribution", "MultiResHashEncoding", "GeneralNeuralGaugeFields"]

_FLAGS = ("should_use_hash_function", "should_softmax_topk_features", "should_leaky_relu", "should_batchnorm_data",
          "should_inplace_scatter")


def _sync_flags():
    """params.<flag> -> the package's module globals (the reference reads them as globals of its models module)."""
    params = sys.modules.get("params")
    if params is None:
        try:
            import params  # noqa: F401  (the reference's params.py, when it is on the path)
            params = sys.modules["params"]
        except ImportError:
            return
    for name in _FLAGS:
        if hasattr(params, name):
            setattr(_hip, name, getattr(params, name))


_sync_flags()

DifferentiableTopk = _hip.DifferentiableTopk


class HashProbDistribution(_hip.HashProbDistribution):
    __doc__ = _hip.HashProbDistribution.__doc__

    def forward(self, *args, **kwargs):
        _sync_flags()
        return super().forward(*args, **kwargs)


class MultiResHashEncoding(_hip.MultiResHashEncoding):
    __doc__ = _hip.MultiResHashEncoding.__doc__

    def forward(self, *args, **kwargs):
        _sync_flags()
        return super().forward(*args, **kwargs)


class GeneralNeuralGaugeFields(_hip.GeneralNeuralGaugeFields):
    __doc__ = _hip.GeneralNeuralGaugeFields.__doc__

    def __init__(self, *args, **kwargs):
        _sync_flags()
        super().__init__(*args, **kwargs)

    def forward(self, *args, **kwargs):
        _sync_flags()
        return super().forward(*args, **kwargs)
