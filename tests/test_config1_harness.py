"""BASELINE config 1 (examples/pq_benchmark.py plumbing at 1k docs, m=8, cosine): the harness runs, and on a GPU the
product path returns exactly what the CPU restatement of the reference path returns."""
import importlib.util
import os
import sys

import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _harness():
    spec = importlib.util.spec_from_file_location('bench_config1', os.path.join(ROOT, 'scripts', 'bench_config1.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_config1_cpu_path(monkeypatch, capsys):
    """no GPU: seed 123 / make_blobs / test_size=20 data, sklearn-trained codebooks (pq.py:89-115), CPU path metrics"""
    mod = _harness()
    monkeypatch.setattr(sys, 'argv', ['bench_config1.py', '--cpu-only', '--repeat', '1'])
    rec = mod.main()
    assert 'gpu' not in rec and rec['cpu']['cores'] == 1
    assert 0.0 <= rec['cpu']['recall'] <= 1.0 and rec['cpu']['recall'] == rec['cpu']['precision']  # 10 of 10 returned
    # the metric helpers are the reference's (examples/utils.py:40-71): precision divides by len(predicted)
    assert mod._precision([1, 2, 3, 4], [1, 2], 2) == 0.5 and mod._recall([1, 2, 3, 4], [1, 2], 2) == 1.0


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_config1_gpu_equals_cpu(monkeypatch):
    mod = _harness()
    monkeypatch.setattr(sys, 'argv', ['bench_config1.py', '--repeat', '2'])
    rec = mod.main()
    eq = rec['gpu_equals_cpu']
    assert eq['ids_identical'] and eq['distances_identical'] and eq['recall_identical'] and eq['precision_identical'], rec
