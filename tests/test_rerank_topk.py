"""Round 6: annlite_rerank_topk -- the exact re-rank of candidate lists (GPU analogue of FlatIndex.search,
annlite/core/index/flat_index.py:15-39; hnswlib space_l2.h / space_ip.h) fused with the top-k -- against the steps it replaces:
annlite_exact_gather_dist -> masking -> annlite_topk_rows -> gather of the ids -> sqrt.  Bit for bit."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs a GPU')]


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


@pytest.mark.parametrize('metric', [1, 2, 3])
@pytest.mark.parametrize('D,R,k', [(128, 128, 10), (64, 128, 10), (96, 50, 64), (768, 200, 16), (20, 7, 10), (128, 64, 1), (130, 129, 33)])
def test_fused_rerank_equals_gather_topk_gather(ops, metric, D, R, k):
    import torch

    rs = np.random.RandomState(D + R + k + metric)
    N, B = 5000, 37
    x = rs.randn(N, D).astype(np.float32)
    x[100:140] = x[100]  # ties: the position in the list decides
    q = rs.randn(B, D).astype(np.float32)
    cand = rs.randint(0, N, size=(B, R)).astype(np.int64)
    cand[:, : min(R, 30)] = rs.randint(100, 140, size=(B, min(R, 30)))
    cand[rs.rand(B, R) < 0.1] = -1
    cand[0, :] = -1            # a query without candidates
    cand[1, 1:] = -1           # ... with one
    cand[2, rs.randint(0, R)] = N + 3  # beyond the table
    valid = rs.rand(N) < 0.85
    bits = np.zeros(((N + 31) // 32 + 2) * 32, bool)
    bits[:N] = valid
    vb = ops.to_dev(np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
    xd, qd, cd = ops.to_dev(x), ops.to_dev(q), ops.to_dev(cand)
    valid_d = ops.to_dev(valid)
    for vbits in (None, vb):
        for sqrt in (False, True):
            fd, fi = ops.rerank_topk(metric, qd, xd, cd, k, valid_bits=vbits, sqrt=sqrt)
            c2 = cd.clone()
            c2[c2 >= N] = -1
            if vbits is not None:
                ok = (c2 >= 0) & valid_d[c2.clamp(min=0)]
                c2 = torch.where(ok, c2, torch.full_like(c2, -1))
            exact = ops.exact_gather_dist(metric, qd, xd, c2)
            kk = min(k, R)
            d, pos = ops.topk_rows(exact, kk)
            i = torch.gather(c2, 1, pos.clamp(min=0))
            i = torch.where((pos < 0) | torch.isinf(d), torch.full_like(i, -1), i)
            if sqrt:
                d = torch.sqrt(d)
            if kk < k:
                d = torch.cat([d, torch.full((B, k - kk), float('inf'), device=d.device)], dim=1)
                i = torch.cat([i, torch.full((B, k - kk), -1, dtype=torch.int64, device=i.device)], dim=1)
            torch.cuda.synchronize()
            assert np.array_equal(fi.cpu().numpy(), i.cpu().numpy()), (metric, vbits is not None, sqrt)
            assert np.array_equal(fd.cpu().numpy().view(np.uint32), d.cpu().numpy().view(np.uint32))
    # and against numpy, loosely (the sums are fp32 in another order)
    fd, fi = ops.rerank_topk(metric, qd, xd, cd, k)
    fd, fi = fd.cpu().numpy(), fi.cpu().numpy()
    for b in (3, 17):
        ok = fi[b] >= 0
        ref = ((x[fi[b][ok]] - q[b]) ** 2).sum(1) if metric == 1 else 1.0 - x[fi[b][ok]] @ q[b]
        assert np.allclose(fd[b][ok], ref, rtol=1e-4, atol=1e-4)
