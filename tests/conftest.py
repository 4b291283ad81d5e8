import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) when no GPU is visible, e.g. a bare `pytest tests/`."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]
    return load


class _EncodePath:
    """what the `encode_path` fixture hands to a test: .path ("auto" | "tiled") and .assert_chain(...)"""

    def __init__(self, path, trace):
        self.path, self.trace = path, trace

    def assert_chain(self, min_launches=1, fixed_point=True):
        """"tiled": the pixel stage of every backward pass ran through gngf_encode_tiled_bwd on the level-interleaved kernel with
        a bound on |d enc| — i.e. through the fixed-point vertex grid (dG64) and the vertex stages that read it
        (vertex_bwd_hash64 / vertex_bwd_sorted<FROM64> / dg64_to_float): the chain bench.py times.  "auto": nothing to assert
        (small batches take the direct form)."""
        if self.path != "tiled":
            return
        assert len(self.trace) >= min_launches, f"expected >= {min_launches} tiled pixel-stage backward launches, saw {len(self.trace)}"
        for r in self.trace:
            assert r["Ls"] > 0 and r["interleaved"], r
            if fixed_point:        # (hash source since round 5: the same fixed-point sums, added to the table rows by the store pass itself)
                assert r["bound"] and (r["dG64"] or r.get("direct_hash")), r


@pytest.fixture(params=["auto", "tiled"])
def encode_path(request):
    """Runs a golden test twice: on the dispatch the product picks by itself (P <= 4096: the direct form) and FORCED onto the tiled
    form with every level staged — binning, tiled_fwd_il / tiled_bwd_il, the fixed-point vertex grid and the vertex stages that
    read it: the kernels that make up 100 % of the encoder time of the headline benchmark."""
    from collision_handling_in_instantngp_amd import ops
    prev = (ops.ENCODE_PATH, ops.TILED_CELLS_PER_PIXEL, ops.PIXEL_BWD_TRACE)
    trace = []
    if request.param == "tiled":
        ops.ENCODE_PATH = "tiled"
        ops.TILED_CELLS_PER_PIXEL = 1e12          # a handful of pixels on a 512^2 grid: stage the level anyway
    ops.PIXEL_BWD_TRACE = trace
    try:
        yield _EncodePath(request.param, trace)
    finally:
        ops.ENCODE_PATH, ops.TILED_CELLS_PER_PIXEL, ops.PIXEL_BWD_TRACE = prev


# ------------------------------------------------------------------------------------------------ which kernel chain did each test run?
STEP_CHAINS_BY_TEST = {}       # nodeid -> sorted list of ops.StepConfig.chain() strings its forward passes took (this process + reported children)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """Every GPU test runs with ops.SEEN_STEP_CONFIGS collecting the kernel chains (ops.StepConfig) of its forward passes; the
    session writes gpurun_out/step_configs_gpu.json = {chain: [tests that ran it]} — the evidence behind the coverage table
    tests/step_config_coverage.json that tests/test_step_config_cpu.py checks the reachable configurations against."""
    ops = None
    if "gpu" in item.keywords:
        try:
            from collision_handling_in_instantngp_amd import ops
            ops.SEEN_STEP_CONFIGS = set()
        except Exception:  # pragma: no cover
            ops = None
    yield
    if ops is not None:
        seen, ops.SEEN_STEP_CONFIGS = ops.SEEN_STEP_CONFIGS, None
        if seen:
            STEP_CHAINS_BY_TEST[item.nodeid] = sorted(seen)


def _write_step_chains():
    if not STEP_CHAINS_BY_TEST:
        return
    import json
    by_chain = {}
    for nodeid, chains in STEP_CHAINS_BY_TEST.items():
        for c in chains:
            by_chain.setdefault(c, []).append(nodeid)
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "step_configs_gpu.json"), "w") as f:
            json.dump({c: sorted(v) for c, v in sorted(by_chain.items())}, f, indent=1)
    except OSError:  # pragma: no cover
        pass


# ------------------------------------------------------------------------------------------------ achieved-error report
class ParityRecorder:
    """Every tolerance check of the parity tests also records the ACHIEVED error, so that the numbers behind
    "matches to <= X" are on file (PARITY.md is generated from the GPU run's report by tools/make_parity_md.py)."""

    def __init__(self):
        self.rows = []

    def record(self, quantity, got, want, rtol, atol):
        got = np.asarray(got, np.float64)
        want = np.asarray(want, np.float64)
        if got.shape != want.shape or got.size == 0:
            return
        err = np.abs(got - want)
        ref_max = float(np.abs(want).max())
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = err / np.abs(want)
        big = np.abs(want) > 1e-3 * ref_max if ref_max > 0 else np.zeros(want.shape, bool)
        self.rows.append({
            "test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0],
            "quantity": quantity, "n": int(got.size),
            "max_abs_err": float(err.max()), "ref_max_abs": ref_max,
            "max_err_over_ref_max": float(err.max() / ref_max) if ref_max > 0 else 0.0,
            "max_rel_err_significant": float(rel[big].max()) if big.any() else 0.0,
            "rtol": float(rtol), "atol": float(atol),
        })


PARITY = ParityRecorder()


def parity_close(got, want, rtol, atol, quantity=""):
    """np.testing.assert_allclose(got, want, rtol, atol) + a row in the achieved-error report."""
    try:
        import torch
        if isinstance(got, torch.Tensor):
            got = got.detach().cpu().numpy()
        if isinstance(want, torch.Tensor):
            want = want.detach().cpu().numpy()
    except ImportError:  # pragma: no cover
        pass
    PARITY.record(quantity, got, want, rtol, atol)
    np.testing.assert_allclose(np.asarray(got, np.float64), np.asarray(want, np.float64), rtol=rtol, atol=atol, err_msg=quantity)


def pytest_sessionfinish(session, exitstatus):
    _write_step_chains()
    if not PARITY.rows:
        return
    import json
    try:
        import torch
        kind = "gpu" if torch.cuda.is_available() else "cpu"
    except Exception:  # pragma: no cover
        kind = "cpu"
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"parity_{kind}.json"), "w") as f:
            json.dump({"kind": kind, "exitstatus": int(exitstatus), "rows": PARITY.rows}, f, indent=0)
    except OSError:  # pragma: no cover
        pass
