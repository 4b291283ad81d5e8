"""CPU: ops.StepConfig.choose — the ONE place that decides which kernel chain a forward / backward pair of the fused encoder takes
(VERDICT r4 item 7).  Enumerates the configurations the shapes, index sources, decoder forms and data-parallel states of this
package can reach (the library's own host-side predicates are used: no GPU), and asserts that every reachable kernel chain
(ops.StepConfig.chain()) was RUN by at least one GPU test: tests/step_config_coverage.json is the record a full `pytest -m gpu` run
writes (tests/conftest.py collects the chains of every test's forward passes; tests/test_zz_gpu_step_config_coverage.py checks the
committed record against what the running session sees).  A new switch or branch that creates a chain nobody runs fails here first."""
import itertools
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COVERAGE_FILE = os.path.join(ROOT, "tests", "step_config_coverage.json")     # {chain: [GPU tests that ran it]} — written by a GPU run
# (tests/conftest.py records ops.StepConfig.chain() of every forward pass per test -> gpurun_out/step_configs_gpu.json; copied here)


def _coverage():
    import json
    return json.load(open(COVERAGE_FILE))


SHAPES = {      # name: (L, F, T, n_min, n_max, fp32 tables, P) — BASELINE.json's configs and the shapes of the test models
    "cfg1": (4, 2, 256, 8, 32, True, 57404), "cfg1_small": (4, 2, 256, 8, 32, True, 4096),
    "cfg2": (16, 2, 2 ** 19, 16, 512, True, 2 ** 20), "cfg4": (16, 2, 2 ** 22, 16, 4096, True, 2 ** 20),
    "cfg5": (16, 4, 2 ** 24, 16, 8192, False, 2 ** 20),
    "partly_staged": (8, 2, 2 ** 14, 16, 1024, True, 2 ** 15), "partly_staged_fp16": (8, 2, 2 ** 14, 16, 1024, False, 2 ** 15),
    "staged_fp16": (8, 2, 2 ** 14, 16, 128, False, 2 ** 15), "four_features": (8, 4, 2 ** 16, 16, 4096, True, 2 ** 17),
}


def _reachable():
    """{chain: [(shape, source, decoder form, data-parallel state), ...]} — shared with tests/test_gpu_step_config_matrix.py, which RUNS
    one case of every chain on the GPU"""
    from collision_handling_in_instantngp_amd import ops
    from oracle import gngf_oracle as orc
    shapes = SHAPES
    out = {}
    for (name, (L, F, T, n_min, n_max, fp32, P)), mode, decoder, dpstate in itertools.product(
            shapes.items(), (ops.MODE_HASH, ops.MODE_VERTEX_TABLE), ("fused_loss", "plain"), ("single", "single+persist_ok", "exchange")):
        if not fp32 and mode == ops.MODE_VERTEX_TABLE:
            continue                                       # (fp16 tables: hash source only in this package's configs and tests)
        n_host = [int(n) for n in orc.level_resolutions(n_min, n_max, L)]
        plan = ops.EncodePlan(P, n_host, F)
        defer = decoder != "plain"
        hidden = defer and L * F == 32                     # the one-launch training kernel exists at 32 encoder features
        sc = ops.StepConfig.choose(plan, L, T, F, P, mode, fp32, True, defer, hidden, dpstate == "exchange",
                                   dpstate == "single+persist_ok", True, True)
        out.setdefault(sc.chain(), []).append((name, "hash" if mode == ops.MODE_HASH else "vertex_table", decoder, dpstate))
    return out


def test_every_reachable_step_configuration_is_named_next_to_a_gpu_parity_test():
    reach, cov = _reachable(), _coverage()
    missing = {c: cases for c, cases in reach.items() if not cov.get(c)}
    assert not missing, "kernel chains no GPU test ran:\n" + "\n".join(f"  {c}   <- e.g. {cases[0]}" for c, cases in missing.items())
    assert len(reach) >= 15, sorted(reach)


def test_the_named_gpu_tests_exist():
    seen = {}
    for chain, tests in _coverage().items():
        assert tests, chain
        for t in tests:
            fname, func = t.split("::")[0], t.split("::")[1].split("[")[0]
            fname = os.path.basename(fname)
            if fname not in seen:
                seen[fname] = open(os.path.join(ROOT, "tests", fname)).read()
            assert re.search(rf"^def {re.escape(func)}\(", seen[fname], re.M), f"{t} (named for {chain!r}) does not exist"
            assert "pytest.mark.gpu" in seen[fname], fname


def test_choose_is_a_pure_function_of_its_arguments_and_the_tuning_object():
    """same facts -> same frozen config; a switch flipped in ops.TUNING -> another one (and back)"""
    import dataclasses
    from collision_handling_in_instantngp_amd import ops
    from oracle import gngf_oracle as orc
    n_host = [int(n) for n in orc.level_resolutions(16, 512, 16)]
    plan = ops.EncodePlan(2 ** 20, n_host, 2)
    args = (plan, 16, 2 ** 19, 2, 2 ** 20, ops.MODE_HASH, True, True, True, True, False, True, True, True)
    a, b = ops.StepConfig.choose(*args), ops.StepConfig.choose(*args)
    assert a == b and hash(a) == hash(b) and a.grad_sink == "table_rows" and a.vertex_fwd == "fused"
    with pytest.raises(dataclasses.FrozenInstanceError):
        a.grad_sink = "dG64"
    prev = ops.TUNING
    try:
        ops.HASH_DIRECT_SCATTER = False                    # (the old upper-case names forward to the frozen object)
        assert ops.TUNING is not prev and ops.TUNING.hash_direct_scatter is False
        c = ops.StepConfig.choose(*args)
        assert c.grad_sink == "dG64" and c != a
    finally:
        ops.TUNING = prev
    assert ops.StepConfig.choose(*args) == a
    # no gradient asked for: nothing is allocated for a backward pass
    ng = ops.StepConfig.choose(plan, 16, 2 ** 19, 2, 2 ** 20, ops.MODE_HASH, True, False, False, False, False, False, True, True)
    assert ng.grad_sink == "none" and ng.table_grad == "none"
