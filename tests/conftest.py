"""pytest configuration.

markers
  gpu   needs a real MI355X (run with ``-m gpu`` on the GPU box); everything else runs on CPU.
"""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real AMD GPU (MI355X); deselected on CPU with -m "not gpu"')


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    g = {k: z[k] for k in z.files}
    M, dsub, Ks, N, B, seed, K = (int(v) for v in g['meta'])
    g.update(M=M, dsub=dsub, Ks=Ks, N=N, B=B, seed=seed, K=K, D=M * dsub)
    return g


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure): oracle/pq_oracle.py over oracle/libpq_oracle.so."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import pq_oracle

    pq_oracle.build()
    return pq_oracle


@pytest.fixture(params=golden_names())
def golden(request):
    return load_golden(request.param)


def reload_knobs():
    """The library reads its ANNLITE_* switches once, at load (common.h: Knobs); a test that changes them says so."""
    from annlite_amd import _capi

    _capi.knobs_reload()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with the library told whenever an ANNLITE_* variable changes (setenv / delenv) and once more when
    the test's changes are undone -- the tests' A/B switches keep working although nothing on the search path reads the
    environment any more."""
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def _setenv(name, value, prepend=None):
        setenv(name, value, prepend)
        if name.startswith('ANNLITE_'):
            reload_knobs()

    def _delenv(name, raising=True):
        delenv(name, raising)
        if name.startswith('ANNLITE_'):
            reload_knobs()

    monkeypatch.setenv, monkeypatch.delenv = _setenv, _delenv
    yield monkeypatch
    monkeypatch.undo()
    reload_knobs()


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
