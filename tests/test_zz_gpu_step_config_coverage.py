"""GPU, runs LAST (file name): the committed record tests/step_config_coverage.json — {kernel chain: [GPU tests that ran it]}, what
tests/test_step_config_cpu.py checks the reachable configurations against — is compared with what THIS session observed: a test the
record names for a chain, and that ran in this session, must have taken that chain.  (Partial runs check the tests they ran.)"""
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_step_config_coverage_matches_this_session():
    from conftest import STEP_CHAINS_BY_TEST
    path = os.path.join(ROOT, "tests", "step_config_coverage.json")
    if not os.path.isfile(path):
        pytest.skip("no committed record yet")
    record = json.load(open(path))
    ran = {nodeid.split("/")[-1]: set(chains) for nodeid, chains in STEP_CHAINS_BY_TEST.items()}
    checked, wrong = 0, []
    for chain, tests in record.items():
        for t in tests:
            key = t.split("/")[-1]
            if key in ran:
                checked += 1
                if chain not in ran[key]:
                    wrong.append((key, chain, sorted(ran[key])))
    assert not wrong, wrong[:5]
    if checked == 0:
        pytest.skip("none of the recorded tests ran in this session")
    print(f"[step-config coverage] {checked} (test, chain) pairs of the committed record confirmed in this session")
