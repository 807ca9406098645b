"""The integer filter bound of the tile-mode scan (scan_qfilter.hip `TileState`), restated in numpy
and checked as a property: with tables quantised the way `lut_quantise_fused_kernel` / `lut_l2_build_quantise_kernel`
quantise them, NO row of the exact top-k (fp32 ascending-m sums, ties by id) has an integer sum above
``Sk + margin`` -- where Sk is the k-th smallest integer sum of the cell, i.e. the tightest bound the kernel can reach.
(The GPU tests check the kernel's results; this checks the inequality the kernel relies on, far beyond their sizes.)"""
import numpy as np
import pytest


def _quantise(lut):
    """lut f32 [M][Ks] of one query -> (Q int [M][Ks], step f32, margin int) as the kernels compute them"""
    M = lut.shape[0]
    qmax = 32767 // M
    lo = lut.min(axis=1).astype(np.float32)
    hi = lut.max(axis=1).astype(np.float32)
    rng = np.float32(0)
    smax = np.float32(0)
    for m in range(M):  # (fp32 accumulation like the kernel)
        rng = np.maximum(rng, np.float32(hi[m] - lo[m]))
        smax = np.float32(smax + np.maximum(np.abs(lo[m]), np.abs(hi[m])))
    step = np.float32(rng / np.float32(qmax))
    if not step > 0:
        step = np.float32(1)
    t = np.floor(((lut - lo[:, None]).astype(np.float32) / step).astype(np.float32))
    Q = np.clip(np.nan_to_num(t, nan=0.0), 0, qmax).astype(np.int64)
    slack = float(smax) * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0))
    margin = int(min(32767.0, np.floor(1.004 * M + 2.0 * slack / float(step)) + 1.0))
    return Q, step, margin


def _exact_sums(lut, codes):
    acc = np.zeros(codes.shape[0], dtype=np.float32)
    for m in range(lut.shape[0]):  # ascending m, fp32: the reference's order
        acc = (acc + lut[m, codes[:, m]]).astype(np.float32)
    return acc


@pytest.mark.parametrize('M', [8, 16, 32, 64])
@pytest.mark.parametrize('shape', ['gauss', 'one_huge_subspace', 'tiny_ranges', 'constant_subspace', 'heavy_ties', 'negative'])
def test_no_top_k_row_exceeds_the_integer_bound(M, shape):
    rs = np.random.RandomState(hash((M, shape)) % (2 ** 31))
    Ks = 256
    for trial in range(6):
        N = int(rs.choice([200, 3000, 20000]))
        k = int(rs.choice([1, 10, 64]))
        lut = (rs.rand(M, Ks).astype(np.float32) ** 2) * np.float32(rs.choice([1e-3, 1.0, 1e4]))
        if shape == 'one_huge_subspace':
            lut[rs.randint(M)] *= np.float32(1e4)
        elif shape == 'tiny_ranges':
            lut = (np.float32(1000.0) + lut * np.float32(1e-3)).astype(np.float32)
        elif shape == 'constant_subspace':
            lut[rs.randint(M)] = np.float32(0.25)
        elif shape == 'negative':  # inner-product style tables: 1/Ks - dot
            lut = (np.float32(1.0 / Ks) - (rs.randn(M, Ks) * 0.3).astype(np.float32)).astype(np.float32)
        codes = rs.randint(0, Ks, size=(N, M))
        if shape == 'heavy_ties':
            codes = codes[rs.randint(0, max(N // 50, 1), size=N)]  # every row repeated ~50 times
        Q, step, margin = _quantise(lut)
        S = Q[np.arange(M)[None, :], codes].sum(axis=1)
        assert S.max() <= 32767
        d = _exact_sums(lut, codes)
        kk = min(k, N)
        top = np.lexsort((np.arange(N), d))[:kk]
        Sk = np.sort(S)[kk - 1]
        assert S[top].max() <= Sk + margin, (M, shape, trial, int(S[top].max()), int(Sk), margin)


@pytest.mark.parametrize('M', [8, 16, 32, 64])
def test_streaming_filter_bound(M):
    """The exhaustive kernels' bound (scan_common.h `qbound_from_key`): a row whose exact fp32 distance is <= thr has
    S <= floor((thr + slack32 - L) / step) + 1 with L = sum_m lo_m in double -- no row at or below the k-th distance is
    filtered out, whatever the table's scale."""
    rs = np.random.RandomState(1000 + M)
    Ks = 256
    for trial in range(12):
        N = int(rs.choice([500, 5000, 30000]))
        k = int(rs.choice([1, 10, 64]))
        lut = (rs.rand(M, Ks).astype(np.float32) ** 2) * np.float32(rs.choice([1e-3, 1.0, 1e4]))
        if trial % 3 == 1:
            lut[rs.randint(M)] *= np.float32(1e4)
        if trial % 3 == 2:
            lut = (np.float32(1.0 / Ks) - (rs.randn(M, Ks) * 0.3).astype(np.float32)).astype(np.float32)
        codes = rs.randint(0, Ks, size=(N, M))
        Q, step, _ = _quantise(lut)
        S = Q[np.arange(M)[None, :], codes].sum(axis=1)
        d = _exact_sums(lut, codes)
        thr = np.sort(d)[min(k, N) - 1]
        lo = lut.min(axis=1).astype(np.float32)
        hi = lut.max(axis=1).astype(np.float32)
        smax = np.float32(0)
        for m in range(M):
            smax = np.float32(smax + np.maximum(np.abs(lo[m]), np.abs(hi[m])))
        slack = float(smax) * (2.0 * M * 5.9604644775390625e-08 * (1.0 + 1.0 / 1024.0))
        L = float(np.sum(lo.astype(np.float64)))
        qthr = np.floor((float(thr) + slack - L) / float(step)) + 1.0
        qthr = min(max(qthr, 0.0), 32767.0)
        assert S[d <= thr].max() <= qthr, (M, trial)
