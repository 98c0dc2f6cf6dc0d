"""The band plan kpr_filterbank_pack appends for k_mel_pw (kapre_amd/csrc/kpr_mel_pw_kernels.h), checked on the CPU: the
section is parsed from the packed blob and EXECUTED here lane by lane exactly as the kernel does it -- 16 contiguous
bins per lane from the skewed magnitude row, one (S0, S1) running pair, partial sums appended to the list wherever the
lane's bit is set in the step's mask, then the per-filter sums through the offset table -- and compared with the
dense product mag @ fb.  Host-only: kpr_filterbank_pack runs on host memory, no GPU involved."""
import numpy as np
import pytest

from kapre_amd import _ffi, backend

HDR = 64


def pw_exch_words(nc):
    return nc + 4 if nc == 1024 else (nc + nc // 32 + 24 if nc == 512 else nc)


def pw_mag_word(k):
    return k + 4 * (k >> 6)


def pw_zero_word(nc):
    return (max(pw_exch_words(nc), pw_mag_word(nc) + 1) + 3) & ~3


def parse(blob, k_bins, n_filt):
    h = blob[:HDR].view(np.uint32)
    assert h[0] == 0x4B504642 and h[1] == k_bins and h[2] == n_filt
    if h[6] == 0:
        return None
    off, L, NR, CMQ, nlist, words = (int(v) for v in h[6:12])
    assert off == HDR + int(h[4]) * 512
    sec = blob[off:off + words].view(np.uint32)
    em = sec[:32].view(np.uint64)
    tab = sec[32:]
    assert words == 32 + 32 * L + L + NR * L + 4 * NR * CMQ * L
    t1 = tab[:32 * L].view(np.float32).reshape(8, L, 4)
    p = tab[32 * L:33 * L]
    wn = tab[33 * L:(33 + NR) * L].view(np.float32).reshape(NR, L)
    t2 = tab[(33 + NR) * L:].reshape(NR, CMQ, L, 4)
    return dict(L=L, NR=NR, CMQ=CMQ, nlist=nlist, em=em, t1=t1, p=p, wn=wn, t2=t2)


def run_plan(plan, mag, n_filt):
    """Emulates stage 1 + stage 2 of k_mel_pw for ONE frame (float32 arithmetic, the kernel's order of operations).  A plan laid
    out for more bins than the row has (round 6: n_freq - 1 not 16 L, the stand-alone kernel k_fb_pw only): the bins beyond the row
    count as zeros, the Nyquist bin is the row's last."""
    L, NR, CMQ = plan["L"], plan["NR"], plan["CMQ"]
    nc = 16 * L
    nb = len(mag) - 1
    mag = np.concatenate([mag[:nb], np.zeros(nc - nb, np.float32), mag[nb:]]).astype(np.float32)
    zero_w = pw_zero_word(nc)
    row = np.full(zero_w + 4, np.float32(np.nan), np.float32)          # anything not written must not be read
    row[zero_w:zero_w + 4] = 0.0
    for k in range(nc + 1):
        row[pw_mag_word(k)] = mag[k]
    mags = np.stack([row[16 * fl + 4 * (fl >> 2):16 * fl + 4 * (fl >> 2) + 16].copy() for fl in range(L)])
    magn = row[pw_mag_word(nc)]
    rowb = row.view(np.uint8)                                          # byte-addressed view for the list
    ptr = plan["p"].astype(np.int64).copy()
    acc = np.zeros((L, 2), np.float32)
    for i in range(16):
        j, e = i // 2, i % 2
        w = plan["t1"][j, :, 2 * e:2 * e + 2]                          # (w0, w1) of bin i, per lane
        acc = (mags[:, i:i + 1] * w + acc).astype(np.float32)          # v_pk_fma_f32 (fused: float64 product is exact enough here)
        mask = int(plan["em"][i])
        for fl in range(L):
            if (mask >> fl) & 1:
                assert ptr[fl] + 8 <= 4 * zero_w, "the list must stop short of the zero words"
                rowb[ptr[fl]:ptr[fl] + 8] = acc[fl].view(np.uint8)
                acc[fl] = 0.0
                ptr[fl] += 8
        if L < 64:                                                      # masks are replicated per lane group
            for gq in range(1, 64 // L):
                assert ((mask >> (L * gq)) & ((1 << L) - 1)) == (mask & ((1 << L) - 1))
    out = np.zeros(NR * L, np.float32)
    for r in range(NR):
        for fl in range(L):
            u = np.float32(0.0)
            d = np.float32(0.0)
            for q in range(CMQ):
                for e in range(4):
                    o = int(plan["t2"][r, q, fl, e])
                    u = np.float32(u + rowb[(o & 0xffff):(o & 0xffff) + 4].view(np.float32)[0])
                    d = np.float32(d + rowb[(o >> 16):(o >> 16) + 4].view(np.float32)[0])
            out[fl + L * r] = np.float32(plan["wn"][r, fl] * magn + np.float32(u + d))
    assert not np.isnan(out[:n_filt]).any()
    return out[:n_filt]


def check_bank(fb, seed=0, expect_plan=True):
    fb = np.ascontiguousarray(fb, np.float32)
    k_bins, n_filt = fb.shape
    blob = _ffi.filterbank_pack(fb, _ffi.filterbank_kranges(fb))
    plan = parse(blob, k_bins, n_filt)
    if not expect_plan:
        assert plan is None
        return None
    assert plan is not None, "no band plan for a bank that has one"
    rng = np.random.default_rng(seed)
    for trial in range(3):
        mag = rng.uniform(0, 1, k_bins).astype(np.float32)
        if trial == 1:
            mag *= np.logspace(0, -6, k_bins).astype(np.float32)        # steep spectral decay
        if trial == 2:
            mag[:] = 0
            mag[rng.integers(0, k_bins, 5)] = 1e3                        # a few isolated peaks
        got = run_plan(plan, mag, n_filt)
        want = mag.astype(np.float64) @ fb.astype(np.float64)
        err = np.abs(got - want)
        assert (err <= 2e-6 * np.abs(want) + 1e-6 * np.abs(want).max() * 1e-3 + 1e-30).all(), float(err.max())
    return plan


@pytest.mark.parametrize("sr, n_fft, n_mels, kw", [
    (44100, 2048, 128, {}),                      # the north-star bank
    (44100, 2048, 130, {}),
    (44100, 2048, 40, {}),
    (44100, 2048, 96, dict(htk=True)),
    (44100, 2048, 128, dict(norm=None)),
    (44100, 2048, 64, dict(f_min=300.0, f_max=8000.0)),     # empty bins above and below
    (16000, 1024, 80, {}),                       # cfg5
    (22050, 1024, 96, {}),
    (22050, 512, 40, dict(f_max=8000.0)),        # the reference's own test shape
    (22050, 512, 128, {}),
    (22050, 256, 40, {}),
    (8000, 256, 40, {}),
    (8000, 256, 64, {}),                         # more filters than make sense (empty ones, one-bin ones): eight per lane
])
def test_band_plan_reproduces_the_dense_product(sr, n_fft, n_mels, kw):
    fb = backend.filterbank_mel(sr, n_fft // 2 + 1, n_mels, **kw)
    plan = check_bank(np.asarray(fb, np.float32), seed=n_fft + n_mels)
    assert plan["L"] == n_fft // 32 and plan["NR"] == -(-n_mels // plan["L"])


@pytest.mark.parametrize("sr, n_fft, n_mels", [(16000, 400, 80), (16000, 400, 40), (16000, 320, 64), (16000, 160, 40), (8000, 200, 40),
                                               (48000, 960, 128), (44100, 1000, 128), (22050, 2000, 128), (16000, 480, 80), (16000, 24, 8)])
def test_band_plan_for_rows_that_are_not_sixteen_bins_per_lane(sr, n_fft, n_mels):
    """round 6: the stand-alone kernel takes any n_freq - 1 that is a multiple of four (every n_fft that is a multiple of eight): the
    plan is laid out for the next 16 x {8, 16, 32, 64} bins, the bins the rows do not have carry zero weights"""
    k_bins = n_fft // 2 + 1
    fb = np.asarray(backend.filterbank_mel(sr, k_bins, n_mels), np.float32)
    plan = check_bank(fb, seed=n_fft)
    L = 8
    while 16 * L < k_bins - 1:
        L *= 2
    assert plan["L"] == L and plan["NR"] == -(-n_mels // L)


def test_band_plan_random_two_band_matrices():
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(30):
        nc = int(rng.choice([128, 256, 512, 1024]))
        L = nc // 16
        n_filt = int(rng.integers(1, min(8 * L, 200) + 1))
        # random non-decreasing a(k) with jumps of 0 / 1 / 2, segments at most ~4 lanes long
        a = np.zeros(nc, np.int64)
        cur, run = 0, 0
        for k in range(nc):
            if cur < n_filt - 1 and (run > 50 or rng.random() < n_filt / nc * 1.2):
                cur = min(n_filt - 1, cur + int(rng.integers(1, 3)))
                run = 0
            a[k] = cur
            run += 1
        fb = np.zeros((nc + 1, n_filt), np.float32)
        run = 0
        for k in range(nc):
            run = run + 1 if k and a[k] == a[k - 1] else 1
            kind = rng.integers(0, 4) if run <= 200 else 0             # (a segment may span at most 16 lanes = 256 bins)
            if kind != 0:
                fb[k, a[k]] = rng.uniform(0.1, 1)
            if kind >= 2 and a[k] + 1 < n_filt:
                fb[k, a[k] + 1] = rng.uniform(0.1, 1)
        fb[nc, rng.integers(0, n_filt, 3)] = rng.uniform(0.1, 1, 3).astype(np.float32)      # any Nyquist row
        # pieces the plan needs at most (a lane's 16 bins cut wherever a(k) changes); the row holds zero_word / 2 of them
        pieces = sum(1 + int(np.count_nonzero(np.diff(a[16 * fl:16 * fl + 16]))) for fl in range(L))
        if pieces > pw_zero_word(nc) // 2:
            continue
        check_bank(fb, seed=trial)
        checked += 1
    assert checked >= 8


def test_matrices_without_band_structure_get_no_plan():
    k_bins = 1025
    rng = np.random.default_rng(2)
    dense = rng.uniform(0, 1, (k_bins, 32)).astype(np.float32)
    check_bank(dense, expect_plan=False)
    logfb = np.asarray(backend.filterbank_log(22050, 1025, n_bins=84, bins_per_octave=12), np.float32)
    check_bank(logfb, expect_plan=False)
    fb = np.asarray(backend.filterbank_mel(44100, 1025, 128), np.float32).copy()
    fb[500, 3] = 0.5                                                    # a stray third non-zero in one bin
    check_bank(fb, expect_plan=False)
    fb = np.asarray(backend.filterbank_mel(44100, 1025, 128), np.float32)[:, ::-1].copy()   # filters in descending order
    check_bank(fb, expect_plan=False)
    fb3 = np.asarray(backend.filterbank_mel(16000, 202, 80), np.float32)                    # 201 bins below Nyquist: not a multiple of four
    check_bank(fb3, expect_plan=False)
    fb4 = np.asarray(backend.filterbank_mel(16000, 81, 80), np.float32)                     # n_fft 160: eight lanes, ten filters per lane
    check_bank(fb4, expect_plan=False)
    wide = np.zeros((1025, 600), np.float32)                            # more than 8 filters per lane
    wide[np.arange(1024), np.arange(1024) * 600 // 1024] = 1.0
    check_bank(wide, expect_plan=False)
