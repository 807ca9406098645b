"""CPU: the C oracle under AddressSanitizer + UBSan (the reference has no sanitizer
coverage and runs its Cython loops with bounds checks off)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which('gcc') is None, reason='gcc not available')
def test_oracle_under_asan_ubsan():
    d = os.path.join(ROOT, 'oracle')
    r = subprocess.run(['make', '-C', d, '-B', 'pq_oracle_asan'], capture_output=True, text=True)
    if r.returncode != 0 and ('asan' in r.stderr.lower() or 'sanitize' in r.stderr.lower()):
        pytest.skip('sanitizer runtime not installed: ' + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1', OMP_NUM_THREADS='2')
    r = subprocess.run([os.path.join(d, 'pq_oracle_asan')], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert 'sanitize OK' in r.stdout
