"""A short budget of the randomised parity run (tests/fuzz_parity.py) inside the GPU suite: random shapes, k, metrics, table orders,
validity patterns, both layouts and both entry points against the CPU oracle, bit for bit.  ANNLITE_FUZZ_SECONDS / ANNLITE_FUZZ_SEED
lengthen or re-seed it (profiles/r05/fuzz_parity_final.txt: 2 x 140 s, 17 924 calls, no mismatch)."""
import os

import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]


def test_random_cases_equal_the_oracle(oracle):
    import fuzz_parity

    seconds = float(os.environ.get('ANNLITE_FUZZ_SECONDS', '12'))
    seed = int(os.environ.get('ANNLITE_FUZZ_SEED', '3'))
    n_cases, n_calls, n_bad, by_m = fuzz_parity.run(seconds, seed)
    assert n_bad == 0, (n_cases, n_calls, n_bad)
    assert n_cases >= 10 and n_calls >= 2 * n_cases, (n_cases, n_calls, by_m)


def test_random_pruned_searches_over_cells_equal_the_oracle(oracle):
    """annlite_ivf_search_topk (byte-table cell tiles) on random cell sizes / probes / validity / layouts / table kinds, and the promises of
    annlite_ivf_search_candidates' lists on the same inputs (fuzz_parity.check_candidate_lists)"""
    import fuzz_parity

    seconds = float(os.environ.get('ANNLITE_FUZZ_SECONDS', '12'))
    seed = int(os.environ.get('ANNLITE_FUZZ_SEED', '3'))
    n_cases, n_calls, n_bad = fuzz_parity.run_cells(seconds, seed + 100)
    assert n_bad == 0, (n_cases, n_calls, n_bad)
    assert n_cases >= 5 and n_calls >= 2 * n_cases, (n_cases, n_calls)  # (+ the candidate generator on every other call)
