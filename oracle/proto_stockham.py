"""Lane-level numpy model of the HIP real-FFT kernels (TEST INFRASTRUCTURE ONLY).

It mirrors, index for index, what kapre_amd/csrc/kapre_hip.hip does per frame:

  * a real n_fft-point transform is computed as an NC = n_fft/2 point complex Stockham FFT on
    z[n] = x[2n] + i x[2n+1];
  * a frame is owned by L = NC/16 lanes, each lane holds P = 16 complex points, always in the
    "lane + L*m" layout (m = register slot);
  * passes have radices R1*R2*R3 = NC (each <= 16); between passes values are exchanged
    through an LDS buffer of NC words (re and im in two rounds) with the XOR swizzle
    e ^ ((e >> 4) & 31);
  * real-FFT post-processing pairs bin k with NC-k, which lives in lane (L - lane) % L slot
    15-m (lane 0: its own slot (16-m)%16) -> one cross-lane shuffle per value;
  * the inverse applies the conjugate procedure (pre-process, inverse passes, un-interleave).

tests/test_proto_stockham.py checks this model against numpy.fft, so that the device code is a
transcription of verified index arithmetic rather than a fresh derivation.
"""
import numpy as np

P = 16


def radices_for(nc: int):
    table = {32: (16, 2), 64: (16, 4), 128: (16, 8), 256: (16, 16), 512: (16, 16, 2),
             1024: (16, 16, 4), 2048: (16, 16, 8)}
    return table[nc]


def swz(e):
    return e ^ ((e >> 4) & 31)


def skew(e, exchange):
    """Additive LDS skew used by the n_fft = 2048 kernels (kpr_fft.h: SwzSkew): exchange 1 (after
    pass 1) e + (e >> 5), exchange 2 (after pass 2) additionally 8 * (e >> 8)."""
    e = np.asarray(e)
    return e + (e >> 5) + (8 * (e >> 8) if exchange == 2 else 0)


def skew_exchange_indices(nc=1024):
    """For each exchange of the NC = 1024 transform: (lane part, const part) of every index a lane
    writes and reads, as the device computes them (fft_pass / exchange_read)."""
    L = nc // P
    lanes = np.arange(L)
    out = {}
    ns = 1
    for x, R in enumerate(radices_for(nc)[:-1], start=1):
        q_per = P // R
        wl, wc = [], []
        for q in range(q_per):
            t = lanes + L * q
            wl.append((lanes // ns) * (ns * R) + lanes % ns)          # FftTw::lane_base(fl, NS, R)
            wc.append([L * R * q + ns * r for r in range(R)])
        out[x] = dict(write_lane=wl, write_const=wc, read_lane=lanes, read_const=[L * m for m in range(P)])
        ns *= R
    return out


def _dft_small(v, sign):
    """(R,) complex -> DFT along axis 0 (explicit; the device uses hard-coded butterflies)."""
    r = v.shape[0]
    k = np.arange(r)
    w = np.exp(sign * 2j * np.pi * np.outer(k, k) / r)
    return w @ v


def complex_fft_lanes(regs, nc, sign=-1):
    """regs: (L, 16) complex, regs[lane, m] = z[lane + L*m].  Returns the same layout holding
    Z[lane + L*m] (sign=-1 forward, +1 unnormalised inverse)."""
    L = nc // P
    radices = radices_for(nc)
    ns = 1
    lanes = np.arange(L)
    for pi, R in enumerate(radices):
        q_per = P // R                   # butterflies per lane
        new = np.empty_like(regs)
        lds = np.zeros(nc, dtype=complex)
        for q in range(q_per):
            t = lanes + L * q            # butterfly ("thread") index, < nc/R
            # inputs in[t + (nc/R)*r] = slot m = q + (16/R)*r
            v = np.stack([regs[:, q + q_per * r] for r in range(R)])        # (R, L)
            kk = t % ns
            tw = np.exp(sign * 2j * np.pi * np.outer(np.arange(R), kk) / (ns * R))
            v = _dft_small(v * tw, sign)
            base = (t // ns) * ns * R + kk
            last = (ns * R == nc)
            for r in range(R):
                if last:
                    new[:, q + q_per * r] = v[r]     # base == t: in place, no exchange
                else:
                    lds[swz(base + ns * r)] = v[r]
        if ns * R != nc:
            for m in range(P):
                new[:, m] = lds[swz(lanes + L * m)]
        regs = new
        ns *= R
    return regs


def load_frame(x_frame, nc):
    """x_frame: (2*nc,) real -> regs[lane, m] = x[2n] + i x[2n+1], n = lane + L*m."""
    L = nc // P
    n = np.arange(L)[:, None] + L * np.arange(P)[None, :]
    return x_frame[2 * n] + 1j * x_frame[2 * n + 1]


def rfft_lanes(x_frame):
    """Full forward model.  Returns X[0..nc] (nc+1 bins)."""
    nfft = x_frame.shape[0]
    nc = nfft // 2
    L = nc // P
    z = complex_fft_lanes(load_frame(x_frame, nc), nc, -1)       # z[lane, m] = Z[lane + L*m]
    out = np.zeros(nc + 1, dtype=complex)
    lanes = np.arange(L)
    src_lane = (L - lanes) % L
    for m in range(P):
        k = lanes + L * m
        # partner value Z[nc - k]: shuffle slot 15-m from src_lane; lane 0 uses own slot (16-m)%16
        shuffled = z[src_lane, 15 - m]
        own = z[:, (16 - m) % 16]
        zp = np.where(lanes == 0, own, shuffled)
        zk = z[:, m]
        e = 0.5 * (zk + np.conj(zp))
        o = -0.5j * (zk - np.conj(zp))
        w = np.exp(-2j * np.pi * k / nfft)
        out[k] = e + w * o
    # Nyquist bin from Z[0] (lane 0, slot 0)
    out[nc] = z[0, 0].real - z[0, 0].imag
    return out


def irfft_lanes(X):
    """Inverse model: X[0..nc] -> x (2*nc,) real, scaled by 1/n_fft (tf.signal.irfft).
    Imaginary parts of X[0] and X[nc] are ignored, as irfft does."""
    nc = X.shape[0] - 1
    nfft = 2 * nc
    L = nc // P
    lanes = np.arange(L)
    regs = np.empty((L, P), dtype=complex)
    for m in range(P):
        k = lanes + L * m
        xk = X[k].copy()
        xp = X[nc - k].copy()          # device: global load of bin nc-k (no shuffle needed)
        # k == 0 pairs DC with Nyquist: use real parts only
        is0 = (k == 0)
        xk = np.where(is0, X[0].real, xk)
        xp = np.where(is0, X[nc].real, xp)
        e = xk + np.conj(xp)
        o = (xk - np.conj(xp)) * np.exp(2j * np.pi * k / nfft)
        regs[:, m] = e + 1j * o        # = 2 * Z[k]
    z = complex_fft_lanes(regs, nc, +1)          # z[lane, m] = 2*nc * zt[lane + L*m]
    x = np.empty(nfft)
    n = lanes[:, None] + L * np.arange(P)[None, :]
    x[2 * n] = z.real / nfft
    x[2 * n + 1] = z.imag / nfft
    return x


# ---------------------------------------------------------------------------------------------------
# "Wide" exchange layout for NC = 1024 (kpr_fft.h: SwzWide) -- 128-bit LDS accesses.
#
# A wave may keep only 15 LDS instructions in flight (lgkmcnt), so an exchange of 64 narrow accesses is issued in
# more than four latency windows; the cost of an exchange is its INSTRUCTION count.  Here the row is organised in
# 16-byte chunks: chunk p (0..255) holds the four consecutive register slots 4j..4j+3 of one reader lane g,
#     p = p_low + 16 * p_high,   p_low = (g & 15) ^ (2 (g >> 5) + 4 (j & 1)),
#                                p_high = ((g >> 4) & 1) + 2 (g >> 5) + 4 (j & 1) + 8 (j >> 1)
# so every lane reads its 16 slots with FOUR ds_read_b128 (the same addresses for both exchanges: two per-lane
# bases + immediates).  Writers: exchange 1 sends a lane's 16 outputs to 16 different readers (ds_write_b32, merged
# in pairs: chunks of neighbouring readers are adjacent), exchange 2 sends outputs r, r+4, r+8, r+12 to the four
# slots of ONE chunk (one ds_write_b128).  Per round (re, im): 8 + 4 and 4 + 4 instructions instead of 32 and 32.
# ---------------------------------------------------------------------------------------------------
def wide_chunk(g, j):
    g = np.asarray(g)
    p_low = (g & 15) ^ (2 * (g >> 5) + 4 * (j & 1))
    p_high = ((g >> 4) & 1) + 2 * (g >> 5) + 4 * (j & 1) + 8 * (j >> 1)
    return p_low + 16 * p_high


def wide_read_addr(lane, m):
    """dword address at which reader `lane` finds slot m (m = 4 j + t)."""
    return 4 * wide_chunk(lane, m >> 2) + (m & 3)


def wide_write_addr(exchange, lane, r):
    """dword address to which writer `lane` sends its pass output r (0..15) in exchange 1 / 2 of NC = 1024."""
    lane = np.asarray(lane)
    if exchange == 1:       # output index 16 lane + r -> reader (16 lane + r) % 64, slot (16 lane + r) // 64
        e = 16 * lane + r
    else:                   # output index 256 (lane >> 4) + (lane & 15) + 16 r
        e = 256 * (lane >> 4) + (lane & 15) + 16 * r
    return wide_read_addr(e % 64, e // 64)


def complex_fft_lanes_wide(regs, sign=-1):
    """complex_fft_lanes for NC = 1024 with the wide exchange layout (the data flow the device uses)."""
    nc, L = 1024, 64
    radices = radices_for(nc)
    ns = 1
    lanes = np.arange(L)
    for pi, R in enumerate(radices):
        new = np.empty_like(regs)
        lds = np.full(nc, np.nan + 0j)
        t = lanes
        v = np.stack([regs[:, r] for r in range(R)]) if R == 16 else None
        if R == 16:
            kk = t % ns
            tw = np.exp(sign * 2j * np.pi * np.outer(np.arange(R), kk) / (ns * R))
            v = _dft_small(v * tw, sign)
            for r in range(R):
                lds[wide_write_addr(pi + 1, lanes, r)] = v[r]
            assert not np.isnan(lds).any()
            for m in range(P):
                new[:, m] = lds[wide_read_addr(lanes, m)]
        else:                            # last pass: radix 4 on 4 groups per lane, in place
            q_per = P // R
            for q in range(q_per):
                tq = lanes + L * q
                vv = np.stack([regs[:, q + q_per * r] for r in range(R)])
                kk = tq % ns
                tw = np.exp(sign * 2j * np.pi * np.outer(np.arange(R), kk) / (ns * R))
                vv = _dft_small(vv * tw, sign)
                for r in range(R):
                    new[:, q + q_per * r] = vv[r]
        regs = new
        ns *= R
    return regs


# ---------------------------------------------------------------------------------------------------
# 32 points per lane (experiment "P32" of k_mel_ws, n_fft = 2048): a frame is owned by 32 lanes x 32 slots, radices
# (32, 32) -- ONE LDS exchange instead of two -- which is a plain 32 x 32 transpose (lane <-> slot): writer lane fl,
# output r -> reader lane r, slot fl.  Row layout: element (reader g, slot m) at g * 33 + m (odd stride: the dword
# writes r * 33 + fl of consecutive lanes and the reads g * 33 + m of consecutive lanes are both conflict free).
# ---------------------------------------------------------------------------------------------------
def p32_fft_lanes(regs, sign=-1):
    """regs[fl, m] = z[fl + 32 m] (32 x 32) -> the same layout of the 1024-point FFT."""
    nc, L, R = 1024, 32, 32
    lanes = np.arange(L)
    # pass 1: radix 32 over the slots (stride-32 inputs), no twiddle, output index 32 fl + r
    v = _dft_small(regs.T.copy(), sign)                                   # v[r, fl]
    row = np.full(32 * 33, np.nan + 0j)
    for r in range(R):
        row[r * 33 + lanes] = v[r]                                        # writer fl, output r -> (reader r, slot fl)
    z = np.stack([row[lanes * 33 + m] for m in range(32)], axis=1)        # z[g, m] = index g + 32 m
    assert not np.isnan(z).any()
    # pass 2: NS = 32, kk = fl: twiddle w_1024^{r fl}, radix 32, output index fl + 32 r (natural layout)
    r_ = np.arange(R)
    tw = np.exp(sign * 2j * np.pi * np.outer(lanes, r_) / nc)             # [fl, r]
    return _dft_small((z * tw).T.copy(), sign).T


def p32_rfft_lanes(x_frame):
    """2048 real samples -> 1025 bins with the P32 data flow (window-less); mirrors rfft_lanes()."""
    nc, L = 1024, 32
    lanes = np.arange(L)
    z = (x_frame[0::2] + 1j * x_frame[1::2])
    regs = z[lanes[:, None] + L * np.arange(32)[None, :]]
    Z = p32_fft_lanes(regs, -1)                                           # Z[fl, m] = bin fl + 32 m
    X = np.zeros(nc + 1, complex)
    for fl in range(L):
        for m in range(16):                                               # k = fl + 32 m < 512
            k = fl + L * m
            if fl == 0:
                zp = Z[0, (32 - m) % 32]                                  # own slot (32 - m) % 32
            else:
                zp = Z[(L - fl) % L, 31 - m]                              # partner lane, slot 31 - m
            e = Z[fl, m] + np.conj(zp)
            t = -1j * np.exp(-2j * np.pi * k / (2 * nc)) * (Z[fl, m] - np.conj(zp))
            X[k] = 0.5 * (e + t)
            X[nc - k] = 0.5 * np.conj(e - t)
    zz = Z[0, 16]                                                         # k = NC / 2, self-paired (lane 0, slot 16)
    e = zz + np.conj(zz)
    t = -1j * np.exp(-2j * np.pi * (nc // 2) / (2 * nc)) * (zz - np.conj(zz))
    X[nc // 2] = 0.5 * (e + t)
    return X
