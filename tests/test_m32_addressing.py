"""The arithmetic of the M = 32 byte-table shape (scan_q8.hip, launch id 3250; scan_common.h ``Q8Cfg<32>``) restated in numpy
and checked as properties -- no GPU: the look-up address built by ONE ``v_perm_b32`` from a code dword and a lane constant lands on
the entry the table build stored for (code, sub-space); a ``ds_read_b128`` lane group never hits an LDS bank slot twice; byte sums
of 32 clipped entries never carry; the byte filter keeps exactly the sums at or below the bound."""
import numpy as np

M, QMAX, QOPEN, TMAX = 32, 240 // 32, 112 // 32, 127


def perm_b32(s0, s1, sel):
    """v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {s1 = bytes 0..3, s0 = bytes 4..7}; selector 0x0c = 0x00"""
    pool = [(s1 >> (8 * b)) & 0xff for b in range(4)] + [(s0 >> (8 * b)) & 0xff for b in range(4)]
    out = 0
    for i in range(4):
        c = (sel >> (8 * i)) & 0xff
        out |= (0 if c == 0x0c else pool[c]) << (8 * i)
    return out


def lane_constants(lane):
    """k32[j], j < 16: bytes (0, 1) = (column, half) of look-up t = 2 j, bytes (2, 3) of t = 2 j + 1, for the lane's skew s = lane % 32
    (stored byte t of row n is the code of sub-space (t + n) mod 32: the SKEWED layout; PLAIN rows are rotated into it)"""
    s = lane % 32
    k = []
    for j in range(16):
        s0, s1 = (s + 2 * j) % 32, (s + 2 * j + 1) % 32
        k.append(((s0 & 15) << 4) | ((s0 >> 4) << 8) | ((s1 & 15) << 20) | ((s1 >> 4) << 24))
    return k


def build_address(code, m):
    """where the table build stores the 16-byte entry (16 queries x 1 byte) of (code, sub-space m): two half tables of 16 sub-spaces,
    [256][16][16 B] each, 64 KB apart"""
    return ((m >> 4) << 16) + (code << 8) + (m & 15) * 16


def lookup_address(row_dwords, lane, t):
    b = t % 2
    sel = 0x0c000000 | ((2 * b + 1) << 16) | ((4 + t % 4) << 8) | (2 * b)
    return perm_b32(row_dwords[t // 4], lane_constants(lane)[t // 2], sel)


def test_one_permute_builds_the_address_of_the_right_entry():
    rs = np.random.RandomState(0)
    for lane in range(64):
        codes = rs.randint(0, 256, size=M)                      # codes of sub-spaces 0 .. 31 of the lane's row
        stored = np.array([codes[(t + lane) % M] for t in range(M)], dtype=np.uint32)   # SKEWED: byte t = sub-space (t + n) mod M, n = lane (mod 32)
        dwords = [int(stored[4 * w] | (stored[4 * w + 1] << 8) | (stored[4 * w + 2] << 16) | (stored[4 * w + 3] << 24)) for w in range(M // 4)]
        seen = set()
        for t in range(M):
            m = (t + lane) % M
            ad = lookup_address(dwords, lane, t)
            assert ad == build_address(int(codes[m]), m), (lane, t)
            assert ad % 16 == 0 and ad + 16 <= 131072
            seen.add(m)
        assert seen == set(range(M))                            # every sub-space exactly once per row


def test_a_lane_group_never_hits_a_bank_slot_twice():
    """ds_read_b128 serves 16 lanes at a time, each 16 bytes = 4 of the 64 banks: conflict-free iff the 16 lanes' 16-byte slots
    (address / 16 mod 16) differ -- whatever the codes are (code << 8 and half << 16 are multiples of the 256-byte bank line)"""
    rs = np.random.RandomState(1)
    for t in range(M):
        for g0 in range(0, 64, 16):
            slots = set()
            for lane in range(g0, g0 + 16):
                m = (t + lane) % M
                slots.add((build_address(int(rs.randint(0, 256)), m) >> 4) & 15)
            assert len(slots) == 16, (t, g0)


def test_byte_sums_never_carry_and_open_tables_pass_everything():
    assert M * QMAX <= 255 and M * QMAX == 224      # a byte sum of 32 clipped entries stays inside its byte
    assert M * QOPEN <= TMAX                        # a slot without a bound yet clips at QOPEN: T = 127 passes every row
    rs = np.random.RandomState(2)
    e = rs.randint(0, QMAX + 1, size=(1000, M, 4)).astype(np.uint32)            # 4 queries of a dword
    packed = (e[..., 0] | (e[..., 1] << 8) | (e[..., 2] << 16) | (e[..., 3] << 24)).sum(axis=1, dtype=np.uint64)
    for q in range(4):
        assert np.array_equal((packed >> (8 * q)) & 0xff, e[..., q].sum(axis=1))


def test_the_byte_filter_keeps_exactly_the_sums_at_or_below_the_bound():
    s = np.arange(0, 256, dtype=np.uint32)[:, None]
    t = np.arange(0, 128, dtype=np.uint32)[None, :]
    hit = (((0x80 | t) - (s & 0x7f)) & ~s & 0x80) != 0
    assert np.array_equal(hit, s <= t)
    # a pad slot (TMAX without the flag bit) and a packed dword: no borrow crosses a byte
    assert not ((((TMAX - (s & 0x7f)) & ~s & 0x80) != 0).any())
    rs = np.random.RandomState(3)
    S = rs.randint(0, 225, size=(5000, 4)).astype(np.uint32)
    T = rs.randint(0, 128, size=(5000, 4)).astype(np.uint32)
    sw = S[:, 0] | (S[:, 1] << 8) | (S[:, 2] << 16) | (S[:, 3] << 24)
    tw = (0x80808080 | T[:, 0] | (T[:, 1] << 8) | (T[:, 2] << 16) | (T[:, 3] << 24)).astype(np.uint32)
    bits = ((tw - (sw & 0x7f7f7f7f)) & ~sw & 0x80808080).astype(np.uint32)
    for q in range(4):
        assert np.array_equal(((bits >> (8 * q + 7)) & 1).astype(bool), S[:, q] <= T[:, q])
