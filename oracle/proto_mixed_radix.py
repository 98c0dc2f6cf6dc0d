"""Lane-level numpy model of the mixed-radix (2^a 5^b) FFT of kapre_amd/csrc/kpr_fft_mr.h (P = 20).
Test infrastructure only (tests/test_proto_stockham.py checks it against numpy.fft).

N = P * R2 * R3 complex points, L = R2 * R3 lanes per frame, P points per lane.
  pass 1: lane l holds x[l + L m], m < P: DFT-P over m (in registers), times W_N^{l k1}
  exchange 1: item (a, b, k1) -- writer lane l = a + R3 b -- at row index l + L k1
  pass 2: lane (a, b') takes k1 = b' + R2 j: DFT-R2 over b, times W_L^{a kb}
  exchange 2 (R3 > 1): item (a, c = k1 + P kb) at row index c + P R2 a
  pass 3: lane l3 takes c = l3 + L j3: DFT-R3 over a
  result: lane l3, register (j3, ka) holds X[l3 + L j3 + P R2 ka]
"""
import numpy as np

# n_fft -> (P, R2, R3): the plans of mixed_radix_plan() in kapre_hip.hip
PLANS = {160: (20, 4, 1), 200: (20, 5, 1), 320: (20, 4, 2), 400: (20, 10, 1), 640: (20, 4, 4),
         800: (20, 20, 1), 1000: (20, 5, 5)}


def dft(v):
    n = len(v)
    k = np.arange(n)
    return np.exp(-2j * np.pi * np.outer(k, k) / n) @ v


def dft_comp(v, a_, b_=5):
    """DFT-P, P = A*5, the way the registers do it: m = u + 5 t, k = kt + A ku."""
    p_ = a_ * b_
    y = np.zeros((b_, a_), complex)
    for u in range(b_):
        y[u] = dft(v[u::b_])                                     # DFT-A over t
        y[u] *= np.exp(-2j * np.pi * u * np.arange(a_) / p_)     # W_P^{u kt}
    out = np.zeros(p_, complex)
    for kt in range(a_):
        out[kt::a_] = dft(y[:, kt])                              # DFT-5 over u -> k = kt + A ku
    return out


def mr_fft(x, p_, r2, r3):
    n = p_ * r2 * r3
    l_ = r2 * r3
    assert len(x) == n
    w = lambda num, den: np.exp(-2j * np.pi * num / den)
    # pass 1
    regs = np.zeros((l_, p_), complex)
    for l in range(l_):
        regs[l] = dft_comp(x[l::l_], p_ // 5) * w(l * np.arange(p_), n)
    row = np.zeros(n, complex)
    for l in range(l_):
        for k1 in range(p_):
            row[l + l_ * k1] = regs[l, k1]
    # pass 2
    q = p_ // r2
    regs2 = np.zeros((l_, q, r2), complex)                       # [lane][j][kb]
    for a in range(r3):
        for bp in range(r2):
            lane = a + r3 * bp
            for j in range(q):
                k1 = bp + r2 * j
                v = np.array([row[(a + r3 * b) + l_ * k1] for b in range(r2)])
                regs2[lane, j] = dft(v) * w(a * np.arange(r2), l_)
    out = np.zeros(n, complex)
    if r3 == 1:
        for bp in range(r2):
            for j in range(q):
                for kb in range(r2):
                    out[bp + r2 * j + p_ * kb] = regs2[bp, j, kb]
        return out
    # exchange 2
    row2 = np.zeros(n, complex)
    for a in range(r3):
        for bp in range(r2):
            for j in range(q):
                for kb in range(r2):
                    c = (bp + r2 * j) + p_ * kb
                    row2[c + p_ * r2 * a] = regs2[a + r3 * bp, j, kb]
    q3 = p_ // r3
    for l3 in range(l_):
        for j3 in range(q3):
            c = l3 + l_ * j3
            v = np.array([row2[c + p_ * r2 * a] for a in range(r3)])
            res = dft(v)
            for ka in range(r3):
                out[c + p_ * r2 * ka] = res[ka]
    return out


def rfft_mr(x):
    """Real FFT of even length n_fft in PLANS via the half-length complex FFT + pairing."""
    n_fft = len(x)
    p_, r2, r3 = PLANS[n_fft]
    n = n_fft // 2
    z = mr_fft(x[0::2] + 1j * x[1::2], p_, r2, r3)
    k = np.arange(n + 1)
    zk = z[k % n]
    zp = np.conj(z[(n - k) % n])
    t = np.exp(-2j * np.pi * k / n_fft)
    return (zk + zp) / 2 - 1j * t * (zk - zp) / 2


# ---- two-pass ("four-step") plans for the sizes with a factor 3: n_fft -> (N1, N2), N = N1 * N2
#   pass 1: lane l < N2 holds x[l + N2 m], m < N1: DFT-N1 over m, times W_N^{l k1}
#   exchange: item (l, k1) at row index l + (N2 | 1) k1
#   pass 2: lane l1 < N1 reads the N2 items of k1 = l1: DFT-N2 -> register k2 holds X[l1 + N1 k2]
PLANS_2P = {96: (8, 6), 120: (4, 15), 192: (8, 12), 240: (8, 15), 360: (12, 15), 384: (16, 12),
            480: (16, 15), 600: (20, 15), 720: (15, 24), 768: (16, 24), 960: (20, 24)}


def fft_2p(x, n1, n2):
    n = n1 * n2
    assert len(x) == n
    n2p = n2 | 1
    row = np.zeros(n2p * n1, complex)
    for l in range(n2):
        y = dft(x[l::n2]) * np.exp(-2j * np.pi * l * np.arange(n1) / n)
        for k1 in range(n1):
            row[l + n2p * k1] = y[k1]
    out = np.zeros(n, complex)
    for l1 in range(n1):
        out[l1::n1] = dft(np.array([row[l2 + n2p * l1] for l2 in range(n2)]))
    return out
