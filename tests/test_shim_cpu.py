"""CPU: the zero-edit swap of the reference's models.py (collision_handling_in_instantngp_amd/reference_shim/models.py):
with the shim directory in front of a `params` module on the path, `from params import *; from models import *` — the
reference's own import lines (main.py:3-4) — yields exactly the reference's four class names, and the behaviour switches are
read from the caller's params at call time."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "collision_handling_in_instantngp_amd", "reference_shim")

PROBE = textwrap.dedent("""
    import sys
    from params import *
    from models import *
    import models, params
    names = sorted(n for n in dir() if n in ("DifferentiableTopk", "HashProbDistribution", "MultiResHashEncoding", "GeneralNeuralGaugeFields"))
    assert names == ["DifferentiableTopk", "GeneralNeuralGaugeFields", "HashProbDistribution", "MultiResHashEncoding"], names
    assert sorted(models.__all__) == names
    assert "torch" not in dir() and "ops" not in dir()            # the star import brings the four names and nothing else
    import inspect
    sig = inspect.signature(GeneralNeuralGaugeFields.__init__)
    want = ["self", "input_dim", "hash_table_size", "num_levels", "n_min", "n_max", "MLP_hidden_layers_widths",
            "HPD_hidden_layers_widths", "HPD_out_features", "feature_dim", "topk_k", "should_keep_topk_only", "should_bw",
            "should_log", "HPD_weights_path", "encoding_weights_path"]
    base = [p for p in inspect.signature(GeneralNeuralGaugeFields.__mro__[1].__init__).parameters]
    assert base[:len(want)] == want, base                         # the reference's constructor signature (models.py:240-257)
    kw = dict(input_dim=2, hash_table_size=64, num_levels=3, n_min=4, n_max=16, MLP_hidden_layers_widths=[64, 64],
              HPD_hidden_layers_widths=[32, 64, 128], HPD_out_features=64, feature_dim=2, topk_k=2)
    net = GeneralNeuralGaugeFields(**kw)                          # params says GNGF on
    assert hasattr(net, "HPD") and not net._hash_mode
    assert sorted(k for k in net.state_dict() if k.startswith("encoding")) == [f"encoding._hash_tables.{l}.weight" for l in range(3)]
    params.should_use_hash_function = True                        # the reference's way of switching GNGF off
    net2 = GeneralNeuralGaugeFields(**kw)
    assert net2._hash_mode and not hasattr(net2, "HPD") and "_prime_numbers" in net2.state_dict()
    params.should_leaky_relu = True
    net3 = GeneralNeuralGaugeFields(**kw)
    import torch
    assert isinstance(net3.mlp[0][1], torch.nn.LeakyReLU)
    print("SHIM OK")
""")

PARAMS = textwrap.dedent("""
    # a stand-in with the reference's flag names (params.py:1-23): values are this test's own
    should_use_hash_function = False
    should_softmax_topk_features = True
    should_leaky_relu = False
    should_batchnorm_data = False
    should_inplace_scatter = True
""")


def _run(extra_path, tmp_path):
    (tmp_path / "probe.py").write_text(PROBE)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SHIM, ROOT, extra_path]))
    return subprocess.run([sys.executable, str(tmp_path / "probe.py")], env=env, capture_output=True, text=True, timeout=300,
                          cwd=str(tmp_path))


def test_star_import_of_the_shim_yields_the_reference_names(tmp_path):
    (tmp_path / "params.py").write_text(PARAMS)
    r = _run(str(tmp_path), tmp_path)
    assert r.returncode == 0 and "SHIM OK" in r.stdout, r.stderr[-2000:]


def test_shim_next_to_the_reference_params_when_present(tmp_path):
    """build container only: the reference's own params.py on the path (read, never copied)"""
    ref = os.environ.get("GNGF_REFERENCE_ROOT", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "params.py")):
        import pytest
        pytest.skip("reference not present (GPU box)")
    r = _run(ref, tmp_path)
    assert r.returncode == 0 and "SHIM OK" in r.stdout, r.stderr[-2000:]
