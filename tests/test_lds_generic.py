"""Round 6: shapes without a filter kernel scan with the query's fp32 table in LDS (``adc_scan_lds_kernel``, scan.hip) where it fits
(<= 144 KB) -- the reference example's ``n_subvectors = 128`` with 1-float sub-vectors (examples/pq_benchmark.py:44), odd M, uint16
codes at M = 16 -- and a lane's code row comes in 16- / 4-byte loads instead of one byte load per (row, sub-space); larger tables
keep the generic kernel (table through L2).  Exact ascending-m fp32 sums either way (pq_bindings.pyx:30-47), the oracle's top-k
under the fixed tie-break (math.py:94-120): bit-exact on ragged sizes, every load width, validity bitmaps, heavy ties, k up to 64."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


def _scan(ops, codes, lut, k, valid=None, row_base=0):
    import torch
    from annlite_amd._capi import scan_plan

    B, M, Ks = lut.shape
    plan = scan_plan(codes.shape[0], M, Ks, codes.dtype.itemsize, B, k)
    assert plan.fast == 0 and plan.qt == 1
    assert plan.waves == (16 if M * Ks * 4 <= 144 * 1024 else 4)  # (table in LDS / through L2)
    vb = None
    if valid is not None:
        bits = np.zeros(((len(valid) + 31) // 32 + 2) * 32, dtype=bool)
        bits[:len(valid)] = valid
        vb = ops.to_dev(np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
    d, i = ops.adc_scan_topk(ops.to_dev(codes), ops.to_dev(lut), B, k, M, Ks, valid_bits=vb, row_base=row_base)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy()


# M, Ks, code dtype: 16-byte row loads (M * size % 16 == 0), 4-byte, element-wise; a table too large for the LDS
SHAPES = [(128, 256, np.uint8), (48, 256, np.uint8), (12, 100, np.uint8), (5, 33, np.uint8), (16, 768, np.uint16), (24, 300, np.uint16),
          (3, 300, np.uint16), (6, 70_000, np.uint32), (128, 512, np.uint16)]


@pytest.mark.parametrize('N,B,k', [(1, 3, 1), (63, 5, 10), (4097, 9, 64), (50_000, 17, 10), (130_001, 6, 33)])
@pytest.mark.parametrize('M,Ks,dt', SHAPES)
def test_odd_shapes_equal_the_oracle(ops, oracle, M, Ks, dt, N, B, k):
    if Ks > 60_000 and N > 5000:
        pytest.skip('one huge-Ks case is enough')
    rs = np.random.RandomState(M * 7 + Ks + N + k)
    if Ks > 60_000:
        B = 2
    lut = rs.rand(B, M, Ks).astype(np.float32)
    lut[B // 2] -= 0.5
    if N > 1000 and (M + N) % 2:  # few distinct rows: exact ties everywhere
        base = rs.randint(0, Ks, size=(19, M)).astype(dt)
        codes = base[rs.randint(0, 19, size=N)]
    else:
        codes = rs.randint(0, Ks, size=(N, M)).astype(dt)
    valid = rs.rand(N) < 0.8 if N % 3 == 0 else None
    d, i = _scan(ops, codes, lut, k, valid=valid, row_base=500)
    if valid is None:
        rd, ri = oracle.adc_search_c(lut, codes, k, id_base=500)
    else:
        idx = np.where(valid)[0]
        if len(idx):
            rd, ri = oracle.adc_search_c(lut, codes[idx], k)
            ri = np.where(ri >= 0, idx[np.clip(ri, 0, len(idx) - 1)] + 500, -1)
        else:
            rd, ri = np.full((B, k), np.inf, np.float32), np.full((B, k), -1, np.int64)
    assert np.array_equal(d, rd), (M, Ks, N, B, k)
    assert np.array_equal(i, ri), (M, Ks, N, B, k)


def test_example_shape_m128_through_the_index(ops, oracle):
    """queries in -> neighbours out at the reference example's extreme shape (128 sub-spaces of one float each), all three metrics,
    a batch large enough that several row slices and workgroup passes take part."""
    import torch
    from annlite_amd import Metric, PQCodec
    from annlite_amd.core.index.pq_flat_gpu import PQFlatGpuIndex

    rs = np.random.RandomState(9)
    N, D, M, B, k = 60_000, 128, 128, 300, 10
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 16).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    for metric, omet in ((Metric.EUCLIDEAN, oracle.EUCLIDEAN), (Metric.INNER_PRODUCT, oracle.INNER_PRODUCT), (Metric.COSINE, oracle.COSINE)):
        codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=metric, n_init=1)
        codec.seed = 2
        codec.fit(x[:8192], iter=4)
        index = PQFlatGpuIndex(dim=D, metric=metric, pq_codec=codec, initial_size=N)
        index.add_with_ids(x, np.arange(N))
        d, i = index.search_batch(q, limit=k)
        codes = ops.codes_to_numpy(index._plain_codes(N))
        rd, ri = oracle.index_search(q, codec.codebooks, codes, omet, k, threads=oracle.max_threads())
        assert np.array_equal(i, ri), metric
        assert np.array_equal(d, rd), metric
