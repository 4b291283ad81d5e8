"""Behaviour switches of the hot path, mirroring the reference's params.py:1-31 names and defaults.

As in the reference, they are MODULE GLOBALS read at call time by `models` (reference models.py:181,212-217,
342,412): `from .params import *` copies them into `models`' namespace, so flipping one at run time means
setting it on the `models` module (e.g. `models.should_use_hash_function = True`), exactly like the reference.
"""
should_batchnorm_data = False          # params.py:5
should_inplace_scatter = True          # params.py:11  (no effect here: top-K backward never builds the dense tensor)
should_softmax_topk_features = True    # params.py:14  True: softmax | None: raw | False: normalised
should_leaky_relu = False              # params.py:17
should_use_hash_function = False       # params.py:20
should_log_allocated_memory = False    # params.py:23

exp = 8                                # params.py:26-31
hash_table_size = 2 ** exp
num_levels = 4
n_min = 8
n_max = 32
feature_dim = 2
MLP_hidden_layers_widths = [64, 64]    # params.py:33-35
HPD_hidden_layers_widths = [32, 64, 128]
HPD_out_features = hash_table_size
