"""A few cases of the randomised parity campaign (tests/fuzz_parity.py: every perturbation of the randomised tests drawn together,
every door of the C-ABI against the binary128 oracle) in the suite, so that the campaign's generator keeps working; the campaign
itself (hundreds of cases) is run by hand on the GPU box: profiles/r05_fuzz_parity.txt.  Likewise tests/fuzz_sequence.py (random
walks of state changes on one context against fresh contexts): profiles/r05_fuzz_sequence.txt."""
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from tests.fuzz_parity import run_case
from tests.test_golden import compare_records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [0, 1, 2, 3, 5, 6, 11, 16])
def test_fuzz_case(case):
    from oracle.oracle import Oracle
    label, frac = run_case(case, 20_000, sa, L, Oracle, compare_records)
    assert 0.0 <= frac <= 1.0, label


@pytest.mark.parametrize("walk", [0, 1, 2, 3, 6, 12])
def test_fuzz_sequence_walk(walk):
    """tests/fuzz_sequence.py: a long-lived context through a random walk of state changes, every trace against a fresh context
    configured to the walk's current state (stale caches: tile position, hoisted tables, zones, variant, quanta)."""
    from tests.fuzz_sequence import run_walk
    name, log = run_walk(walk, 10, sa, L)
    assert len(log) == 20
