"""The static solver schedule, checked on the CPU: `relmc_debug_symbolic` builds the pass program of a case without a device,
tests/schedule_interp.py executes it with numpy in the kernel's data layout, and the result must equal numpy.linalg.solve on the dense
bus-pair system.  Covers the elimination orders (primary + the two further orders of the retry path), every pass form, both tiles, and
random networks up to the tiles' limits.  (The reference delegates this solve to MATLAB's `\\` inside MIPS, mc_simulation.m:41.)"""
import numpy as np
import pytest

from powersystemsreliabilityassessment_amd import case24, case96
from tests import schedule_interp as si
from tests.test_random_cases import random_case


def _check(case, variant, seed, outages=0):
    s = si.symbolic(case, variant)
    rng = np.random.default_rng(seed)
    line_on = None
    if outages:
        line_on = np.ones(case.nl, bool); line_on[rng.choice(case.nl, outages, replace=False)] = False
    W, A, rhs = si.random_system(s, rng, line_on)
    x = si.solve(s, W)
    ref = np.linalg.solve(A, rhs)
    assert np.abs(x - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (variant, np.abs(x - ref).max())
    return s


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_rts24_schedule_solves_the_block_system(variant):
    s = _check(case24.rts24(), variant, 1)
    assert s.tile == 0 and s.rw == 16 and s.nb == 24
    _check(case24.rts24(), variant, 2, outages=3)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_rts96_schedule_solves_the_block_system(variant):
    s = _check(case96.rts96(), variant, 3)
    assert s.tile == 1 and s.rw == 64 and s.nb == 73
    _check(case96.rts96(), variant, 4, outages=5)


def test_schedule_sizes_do_not_regress():
    """Dependent passes per Newton step (update + inversion + back substitution): what the kernel's solver time is proportional to."""
    s24 = si.symbolic(case24.rts24()); s96 = si.symbolic(case96.rts96())
    assert s24.npass <= 21 and s96.npass <= 32


@pytest.mark.parametrize("seed,nb,chords,ng,loads", [(1, 2, 0, 2, 1), (2, 7, 3, 4, 4), (3, 19, 8, 9, 10), (4, 30, 14, 12, 14), (5, 60, 25, 30, 35),
                                                     (6, 100, 26, 40, 50), (7, 120, 6, 40, 60)])
def test_random_networks(seed, nb, chords, ng, loads):
    rng = np.random.default_rng(100 + seed)
    case = random_case(rng, nb, chords, ng, loads)
    for variant in (0, 1, 2):
        try:
            _check(case, variant, seed)
        except RuntimeError as e:                       # a further order may not fit the tile (fill, passes): then it is simply unavailable
            assert variant != 0 and "relmc_debug_symbolic failed (-4)" in str(e), e
