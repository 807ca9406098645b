"""bench.py with TWO ranks, before an 8-GPU SCALE run is their debut: `python -m torch.distributed.run --nproc-per-node 2
bench.py --backend gloo ...` with both ranks on cuda:0 (RCCL refuses two ranks on one device; under gloo the script's collectives
and sharded.py's packed exchange go through host memory).  Everything the N > 1 branches do runs: shard ranges, broadcast of the
codebooks, per-rank scan of its own rows, the packed [B, k, 2] exchange + merge_lists_kernel, max-over-ranks timing, the per-rank
records, the merged brute-force truth for recall, the re-rank through gather_and_merge.

What must hold: the merged result IS the single-GPU result -- `result_sha256` equals the N = 1 digest of the
same workload --, the ranks' additive shard checksums add up to the N = 1 checksum, every rank reports the same digest head.
The fan-out / merge the reference itself tests is Jina `shards=3`, tests/executor/test_executor.py:326-350."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKLOAD = ['--rows', '300000', '--steps', '6', '--warmup', '2', '--prewarm-steps', '4', '--recall-queries', '32',
            '--cpu-queries', '0', '--legs', 'none']


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{')]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


@pytest.fixture(scope='module')
def single(tmp_path_factory):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + WORKLOAD, capture_output=True, text=True,
                       env=env, timeout=900, cwd=str(tmp_path_factory.mktemp('n1')))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return _line(r.stdout)


@pytest.mark.parametrize('world', [2, 3])
def test_bench_two_ranks_equal_the_single_gpu_line(tmp_path, single, world):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(29541 + world), os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--backend', 'gloo'] + WORKLOAD
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    rec = _line(r.stdout)
    assert rec['n_gpus'] == world and rec['config']['n_ranks'] == world and rec['config']['backend'].startswith('gloo')
    assert rec['scaling'] == 'strong' and rec['value'] > 0 and rec['ms_per_step'] > 0
    # the merged result is the single-GPU result
    assert rec['result_sha256'] == single['result_sha256'], (rec['result_sha256_per_batch'], single['result_sha256_per_batch'])
    # ... on every rank
    heads = [p['result_sha256_head'] for p in rec['per_rank']]
    assert len(heads) == world and set(heads) == {single['result_sha256'][:8]}
    # the shards partition the table: rows add up, additive checksums add up to the N = 1 figure
    assert sum(p['rows'] for p in rec['per_rank']) == 300000 and all(p['rows'] > 0 for p in rec['per_rank'])
    assert rec['shard_codes_checksum_sum'] == single['shard_codes_checksum_sum']
    total = sum(int(p['shard_codes_checksum'], 16) for p in rec['per_rank']) & 0xFFFFFFFFFFFFFFFF
    assert '%016x' % total == single['shard_codes_checksum_sum']
    # recall of the merged result against the merged brute-force truth == the single-GPU figure (same ids)
    assert rec['recall_at_10'] == pytest.approx(single['recall_at_10'], abs=1e-12)
    # the re-rank leg ran through the general exchange (gather_and_merge) and reaches the north-star recall
    assert rec['rerank'] is not None and rec['rerank']['recall_at_10'] >= 0.9
    # the tail of the line carries the evidence
    s = rec['summary']
    assert list(rec.keys())[-1] == 'summary' and len(json.dumps(s)) <= 1536
    assert s['main']['n'] == world and s['main']['sha'] == single['result_sha256'][:8]
    assert s['ranks']['sha'] == heads and s['ranks']['checksum_sum'] == single['shard_codes_checksum_sum']


def test_single_gpu_line_ends_with_the_summary(single):
    s = single['summary']
    assert list(single.keys())[-1] == 'summary' and len(json.dumps(s)) <= 1536
    assert s['main']['n'] == 1 and s['main']['sha'] == single['result_sha256'][:8] and s['main']['qps'] > 0


def test_bench_graph_leg(tmp_path):
    """The `graph` leg (round 6): HNSW-over-PQ over the bench's own rows -- level 0 built on the GPU, pair walk, fused exact re-rank --
    next to the exhaustive scan's line: present, timed, recall measured against the same brute-force truth, and in the summary."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    args = ['--rows', '300000', '--steps', '6', '--warmup', '2', '--prewarm-steps', '4', '--recall-queries', '64', '--cpu-queries', '0',
            '--legs', 'graph']
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + args, capture_output=True, text=True, env=env,
                       timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    rec = _line(r.stdout)
    g = rec['graph']
    assert g and 'error' not in g, g
    assert g['rows'] == 300000 and g['value'] > 0 and g['build_s'] > 0
    assert g['recall_at_10'] >= 0.9 and g['ef_search_160']['recall_at_10'] >= g['recall_at_10'] - 0.01  # (300k rows: the 128 best by PQ distance hold the neighbours)
    assert g['recall_at_10'] > rec['recall_at_10']  # (the plain ADC top-10 of the main line)
    s = rec['summary']
    assert list(rec.keys())[-1] == 'summary' and s['graph']['qps'] > 0 and len(json.dumps(s)) <= 1536
