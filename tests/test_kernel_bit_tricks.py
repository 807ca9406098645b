"""Integer identities the HIP kernels rely on, restated in numpy (CPU, no GPU): a wrong constant here is a silently wrong
filter there.  Each case cites the kernel line it mirrors."""
import itertools

import numpy as np


def test_row_queue_nibble_gather():
    """scan_q8.hip q8_row_pass_mask: the four per-byte pass flags of a dword (bit 7 of every byte, shifted down to bits 0, 8,
    16, 24) are gathered into a nibble by ONE multiplication: (hb * 0x00204081) >> 21 & 0xf -- bit j of the nibble = byte j."""
    for flags in itertools.product((0, 1), repeat=4):
        hb = sum(f << (8 * j) for j, f in enumerate(flags))
        nib = ((hb * 0x00204081) & 0xFFFFFFFF) >> 21 & 0xF
        assert nib == sum(f << j for j, f in enumerate(flags)), (flags, nib)


def test_byte_filter_word():
    """scan_q8.hip (the scanning waves' filter and q8_row_pass_mask): per byte, S <= T  <=>  bit 7 of
    ((0x80 | T) - (S & 0x7f)) & ~S, for every T <= 127 and every S <= 255 -- and no borrow crosses a byte."""
    rs = np.random.RandomState(0)
    T = rs.randint(0, 128, size=(20000, 4)).astype(np.uint32)
    S = rs.randint(0, 256, size=(20000, 4)).astype(np.uint32)
    S[:2000] = np.minimum(S[:2000], T[:2000] + rs.randint(-2, 3, size=(2000, 4)).clip(-200, 200) % 256)  # near the bound
    th = sum(((0x80 | T[:, j]) << (8 * j)) for j in range(4)).astype(np.uint32)
    sm = sum((S[:, j] << (8 * j)) for j in range(4)).astype(np.uint32)
    w = ((th - (sm & np.uint32(0x7F7F7F7F))) & ~sm & np.uint32(0x80808080)).astype(np.uint32)
    for j in range(4):
        assert np.array_equal(((w >> (8 * j + 7)) & 1).astype(bool), S[:, j] <= T[:, j])


def test_wide_filter_word_and_widening():
    """scan_q8.hip (M = 64): byte sums of 16 look-ups (<= 240) are widened into u16 sums -- bytes 0, 2 by a mask, bytes 1, 3 by
    a byte permute -- and the half-word filter (0x8000 | T) - S keeps bit 15 iff S <= T (S <= 960, T <= 32767)."""
    rs = np.random.RandomState(1)
    b = rs.randint(0, 241, size=(5000, 4)).astype(np.uint32)
    dword = sum(b[:, j] << (8 * j) for j in range(4)).astype(np.uint32)
    even = dword & np.uint32(0x00FF00FF)
    odd = ((dword >> 8) & np.uint32(0x00FF00FF))  # what v_perm_b32(0, x, 0x0c030c01) produces
    assert np.array_equal(even & 0xFFFF, b[:, 0]) and np.array_equal(even >> 16, b[:, 2])
    assert np.array_equal(odd & 0xFFFF, b[:, 1]) and np.array_equal(odd >> 16, b[:, 3])
    T = rs.randint(0, 961, size=(5000, 2)).astype(np.uint32)
    S = rs.randint(0, 961, size=(5000, 2)).astype(np.uint32)
    th = ((0x8000 | T[:, 0]) | ((0x8000 | T[:, 1]) << 16)).astype(np.uint32)
    sm = (S[:, 0] | (S[:, 1] << 16)).astype(np.uint32)
    w = ((th - sm) & np.uint32(0x80008000)).astype(np.uint32)
    assert np.array_equal(((w >> 15) & 1).astype(bool), S[:, 0] <= T[:, 0])
    assert np.array_equal(((w >> 31) & 1).astype(bool), S[:, 1] <= T[:, 1])


def test_merge_first_cut_rule():
    """scan_q8.hip q8_merge_tile: n_full ascending lists lie wholly in the first 64 keys; with j = ceil(k / n_full) the largest
    of their j-th keys is >= the k-th smallest of the union of ALL lists -- so keys above it can be dropped before ranking."""
    rs = np.random.RandomState(2)
    for k, n_slices in [(10, 8), (16, 8), (1, 8), (10, 4), (7, 16), (10, 1), (16, 32)]:
        for _ in range(200):
            lists = np.sort(rs.randint(0, 1000, size=(n_slices, k)), axis=1)
            if rs.rand() < 0.3:
                lists[rs.randint(n_slices), rs.randint(k):] = 10 ** 9  # a short list: padded with +inf keys
            total = n_slices * k
            n_full = min(total, 64) // k
            if n_full == 0:
                continue
            j = -(-k // n_full)
            assert j <= k and n_full * j >= k
            cut = lists[:n_full, j - 1].max()
            kth = np.sort(lists.reshape(-1))[k - 1]
            assert kth <= cut
