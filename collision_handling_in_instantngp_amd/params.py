"""Behaviour switches and default sizes of the hot path.

The reference keeps its switches as module globals that every file star-imports and reads AT CALL TIME (reference
params.py:1-31; models.py:181,212-217,342,412).  The same names and defaults are published here so that
`from .params import *` gives `models` the same globals: flipping a switch at run time means setting it on the `models`
module (e.g. `models.should_use_hash_function = True`), exactly as with the reference.
"""
_SWITCHES = {
    "should_batchnorm_data": False,        # BatchNorm1d on the coordinates instead of the /max(w,h) normalisation
    "should_inplace_scatter": True,        # no effect here: the top-K backward never materialises the dense tensor
    "should_softmax_topk_features": True,  # blend of the K looked-up rows: True softmax | None raw | False normalised
    "should_leaky_relu": False,            # LeakyReLU instead of ReLU in the decoder
    "should_use_hash_function": False,     # spatial hash instead of the learned HPD indices
    "should_log_allocated_memory": False,
}
_SIZES = {
    "exp": 8, "hash_table_size": 2 ** 8, "num_levels": 4, "n_min": 8, "n_max": 32, "feature_dim": 2,
    "MLP_hidden_layers_widths": [64, 64], "HPD_hidden_layers_widths": [32, 64, 128], "HPD_out_features": 2 ** 8,
}
globals().update(_SWITCHES)
globals().update(_SIZES)
__all__ = list(_SWITCHES) + list(_SIZES)
