"""The lane-level numpy model of the HIP FFT kernels (oracle/proto_stockham.py) against numpy.fft."""
import numpy as np
import pytest

import proto_stockham as ps


@pytest.mark.parametrize("n_fft", [64, 128, 256, 512, 1024, 2048, 4096])
def test_rfft_lane_model(n_fft):
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal(n_fft)
    np.testing.assert_allclose(ps.rfft_lanes(x), np.fft.rfft(x), atol=1e-11)


@pytest.mark.parametrize("n_fft", [256, 512, 1024, 2048])
def test_irfft_lane_model(n_fft):
    rng = np.random.default_rng(n_fft + 1)
    X = rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)
    np.testing.assert_allclose(ps.irfft_lanes(X), np.fft.irfft(X, n=n_fft), atol=1e-12)


def test_swizzle_is_a_bijection_and_linear():
    for nc in (128, 256, 512, 1024):
        e = np.arange(nc)
        assert sorted(ps.swz(e)) == list(range(nc))
    a, c = 37, 64 * 5          # bit-disjoint parts: swz(a + c) == swz(a) ^ swz(c)
    assert ps.swz(a + c) == ps.swz(a) ^ ps.swz(c)


@pytest.mark.parametrize("nc", [1024, 512])
def test_additive_skew_of_the_2048_kernels(nc):
    """kpr_fft.h SwzSkew (k_mel_ws, k_stft<1024>): the skewed index must (1) split additively into a
    per-lane part and a compile-time part -- that is what turns every exchange address into base
    register + immediate offset --, (2) be injective and fit the row, (3) be bank-conflict free for
    ds_write_b32 / ds_read_b32 (32 banks, lanes serviced in the groups 0-31 and 32-63)."""
    L = nc // 16                                                  # lanes per frame (64 / 32)
    for x, d in ps.skew_exchange_indices(nc).items():
        seen = set()
        for lane_part, consts in zip(d["write_lane"], d["write_const"]):
            for c in consts:
                full = ps.skew(lane_part + c, x)
                assert (full == ps.skew(lane_part, x) + ps.skew(c, x)).all()          # (1)
                for grp in (full[:32], full[32:]) if L == 64 else (full,):
                    assert len(set(grp % 32)) == 32                                    # (3) writes
                seen.update(full.tolist())
        assert len(seen) == nc and max(seen) < nc + nc // 32 + 24                      # (2)
        for c in d["read_const"]:
            full = ps.skew(d["read_lane"] + c, x)
            assert (full == ps.skew(d["read_lane"], x) + ps.skew(c, x)).all()
            for grp in (full[:32], full[32:]) if L == 64 else (full,):
                assert len(set(grp % 32)) == 32                                        # (3) reads
        # what is read back is exactly what was written
        reads = set()
        for c in d["read_const"]:
            reads.update(ps.skew(d["read_lane"] + c, x).tolist())
        assert reads == seen


@pytest.mark.parametrize("n_fft", [6, 12, 100, 300, 400, 480, 600, 1000])
def test_bluestein_real_fft_model(n_fft):
    """oracle/proto_bluestein.py (the step-by-step model of k_stft_bs) against numpy's rfft"""
    import proto_bluestein as pb

    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal(n_fft)
    m, wt, bt, t = pb.tables(n_fft)
    assert m >= 2 * (n_fft // 2) - 1 and m & (m - 1) == 0
    np.testing.assert_allclose(pb.rfft_bluestein(x), np.fft.rfft(x), atol=1e-10 * n_fft)


@pytest.mark.parametrize("n_fft", [6, 12, 100, 300, 400, 1000])
def test_bluestein_inverse_real_fft_model(n_fft):
    import proto_bluestein as pb

    rng = np.random.default_rng(n_fft + 7)
    X = rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)
    np.testing.assert_allclose(pb.irfft_bluestein(X), np.fft.irfft(X, n=n_fft), atol=1e-10)


@pytest.mark.parametrize("n_fft", [160, 200, 320, 400, 640, 800, 1000])
def test_mixed_radix_fft_model(n_fft):
    """oracle/proto_mixed_radix.py (lane / register / exchange-index model of kpr_fft_mr.h: 20 points
    per lane, passes of radix 20 | R2 | R3) against numpy's fft and rfft"""
    import proto_mixed_radix as pm

    p, r2, r3 = pm.PLANS[n_fft]
    assert p == 20 and p * r2 * r3 == n_fft // 2 and p % r2 == 0 and p % r3 == 0
    rng = np.random.default_rng(n_fft)
    z = rng.standard_normal(n_fft // 2) + 1j * rng.standard_normal(n_fft // 2)
    np.testing.assert_allclose(pm.mr_fft(z, p, r2, r3), np.fft.fft(z), atol=1e-11)
    x = rng.standard_normal(n_fft)
    np.testing.assert_allclose(pm.rfft_mr(x), np.fft.rfft(x), atol=1e-11)


@pytest.mark.parametrize("n_fft", [96, 120, 192, 240, 360, 384, 480, 600, 720, 768, 960])
def test_two_pass_fft_model(n_fft):
    """the two-pass plans of kpr_fft_mr.h (TwoPassFft<N1, N2>, sizes with a factor 3): index maps of the
    exchange with its odd row stride"""
    import proto_mixed_radix as pm

    n1, n2 = pm.PLANS_2P[n_fft]
    assert n1 * n2 == n_fft // 2 and max(n1, n2) <= 32
    rng = np.random.default_rng(n_fft)
    z = rng.standard_normal(n_fft // 2) + 1j * rng.standard_normal(n_fft // 2)
    np.testing.assert_allclose(pm.fft_2p(z, n1, n2), np.fft.fft(z), atol=1e-11)


# ------------------------------------------------------------------ wide (128-bit) exchange layout, NC = 1024
def _bank_conflicts(addrs, groups, width, nbanks):
    """worst number of DISTINCT addresses per bank within one lane group (1 = conflict free)."""
    worst = 0
    for g in groups:
        banks = {}
        for lane in g:
            for d in range(width):
                a = int(addrs[lane]) + d
                banks.setdefault(a % nbanks, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


B128_READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
                    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def test_wide_exchange_layout_is_a_bijection_and_static():
    lanes = np.arange(64)
    seen = np.zeros(1024, int)
    for m in range(16):
        seen[ps.wide_read_addr(lanes, m)] += 1
    assert (seen == 1).all()                                   # every dword of the row is exactly one (lane, slot)
    for x in (1, 2):
        hit = np.zeros(1024, int)
        for r in range(16):
            hit[ps.wide_write_addr(x, lanes, r)] += 1
        assert (hit == 1).all()
    # reads: slot 4j + t sits at chunk base + t, chunk bases = per-lane base[j & 1] + 2048 bytes * (j >> 1)
    for j in range(4):
        a = ps.wide_read_addr(lanes, 4 * j)
        assert (a % 4 == 0).all()
        for t in range(4):
            assert (ps.wide_read_addr(lanes, 4 * j + t) == a + t).all()
        assert (a == ps.wide_read_addr(lanes, 4 * (j & 1)) + 512 * (j >> 1)).all()
    # exchange 2: outputs r, r+4, r+8, r+12 fill ONE chunk in order (one ds_write_b128); base depends on r only
    # through (r >> 1) & 1 (a per-lane base) and an immediate
    for c in range(4):
        a = ps.wide_write_addr(2, lanes, c)
        assert (a % 4 == 0).all()
        for t in range(4):
            assert (ps.wide_write_addr(2, lanes, c + 4 * t) == a + t).all()
        base = ps.wide_write_addr(2, lanes, 2 * ((c >> 1) & 1))
        assert (a - base == 64 * (c & 1)).all()
    # exchange 1: address = per-lane base[(r >> 1) & 1][(r >> 2) & 1] + 4 * (r & 9) dwords
    for r in range(16):
        base = ps.wide_write_addr(1, lanes, r & 6)
        assert (ps.wide_write_addr(1, lanes, r) - base == 4 * (r & 9)).all()
    # ... and the device derives every base from ONE register per exchange (kpr_fft.h wide_write / wide_read)
    a_w1, a_w2, a_rd = ps.wide_write_addr(1, lanes, 0), ps.wide_write_addr(2, lanes, 0), ps.wide_read_addr(lanes, 0)
    for r in range(16):
        assert (ps.wide_write_addr(1, lanes, r) == (a_w1 ^ (4 * (r & 6))) + 4 * (r & 9)).all()
    for c in range(4):
        want = ((a_w2 ^ 8) + 128 if c & 2 else a_w2) + 64 * (c & 1)
        assert (ps.wide_write_addr(2, lanes, c) == want).all()
    for j in range(4):
        want = ((a_rd ^ 16) + 256 if j & 1 else a_rd) + 512 * (j >> 1)
        assert (ps.wide_read_addr(lanes, 4 * j) == want).all()


def test_wide_exchange_bank_conflicts():
    lanes = np.arange(64)
    # ds_read_b128: 4 x 16 lane groups, 64 banks
    for j in range(4):
        assert _bank_conflicts(ps.wide_read_addr(lanes, 4 * j), B128_READ_GROUPS, 4, 64) == 1
    # exchange 2 writes, ds_write_b128: 8 x 8 contiguous lanes, 32 banks
    groups8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
    for c in range(4):
        assert _bank_conflicts(ps.wide_write_addr(2, lanes, c), groups8, 4, 32) == 1
    # exchange 1 writes, ds_write_b32 (pairs merged to ds_write2_b32): 2 x 32 lanes, 32 banks; 2-way is free
    groups32 = [list(range(32)), list(range(32, 64))]
    for r in range(16):
        assert _bank_conflicts(ps.wide_write_addr(1, lanes, r), groups32, 1, 32) <= 2


def test_wide_exchange_fft_matches_numpy():
    rng = np.random.default_rng(7)
    z = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    lanes = np.arange(64)
    regs = z[lanes[:, None] + 64 * np.arange(16)[None, :]]
    out = ps.complex_fft_lanes_wide(regs, -1)
    ref = np.fft.fft(z)
    np.testing.assert_allclose(out, ref[lanes[:, None] + 64 * np.arange(16)[None, :]], rtol=1e-10, atol=1e-9)


# ------------------------------------------------------------------ 32 points per lane, radices (32, 32)
def test_p32_transpose_exchange_is_conflict_free_and_complete():
    lanes = np.arange(32)
    seen = np.zeros(32 * 33, int)
    for r in range(32):
        a = r * 33 + lanes                                            # writes of one instruction (output r)
        assert len(set(a % 32)) == 32
        seen[a] += 1
    for m in range(32):
        a = lanes * 33 + m                                            # reads of one instruction (slot m)
        assert len(set(a % 32)) == 32
        assert (seen[a] == 1).all()
    assert seen.sum() == 1024 and 32 * 33 <= 1058                     # fits the magnitude row of k_mel_ws


def test_p32_fft_and_rfft_match_numpy():
    rng = np.random.default_rng(32)
    z = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    lanes = np.arange(32)
    regs = z[lanes[:, None] + 32 * np.arange(32)[None, :]]
    ref = np.fft.fft(z)
    np.testing.assert_allclose(ps.p32_fft_lanes(regs, -1), ref[lanes[:, None] + 32 * np.arange(32)[None, :]],
                               rtol=1e-10, atol=1e-9)
    x = rng.standard_normal(2048)
    np.testing.assert_allclose(ps.p32_rfft_lanes(x), np.fft.rfft(x), rtol=1e-10, atol=1e-9)
