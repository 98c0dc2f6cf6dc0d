"""numpy model of the size-generic FFT engine of kapre_amd/csrc/kpr_generic_kernels.h (k_stft_gen / k_irfft_gen): the
run-time radix plan, the Stockham pass formula with its twiddle indices, the real-FFT packing of even sizes and the
float-reciprocal index arithmetic -- the same formulas, checked on the CPU against numpy.fft
(tests/test_proto_generic.py).  TEST INFRASTRUCTURE (lives under oracle/): nothing in kapre_amd/ imports it."""
import numpy as np


def gen_plan(n):
    """kapre_hip.hip: gen_plan -- 4s first, then a 2, then the odd primes up to 64; None when a larger prime remains."""
    radix, m = [], n
    while m % 4 == 0:
        radix.append(4)
        m //= 4
    if m % 2 == 0:
        radix.append(2)
        m //= 2
    f = 3
    while f <= 64 and m > 1:
        while m % f == 0:
            radix.append(f)
            m //= f
        f += 2
    return radix if (m == 1 and n >= 2 and len(radix) <= 16) else None


def gen_fft_len(n_fft):
    return n_fft // 2 if (n_fft % 2 == 0 and n_fft >= 4) else n_fft


def float_div(o, d):
    """(int)((o + 0.5f) * (1.0f / d)) in float32, the device's replacement for o / d."""
    inv = np.float32(1.0) / np.float32(d)
    return ((np.asarray(o, np.float32) + np.float32(0.5)) * inv).astype(np.int64)


def gen_fft(a, radix, tw, tws, sign):
    """gen_fft: a = N complex points, tw = exp(-2 pi i j / (tws N)) table; Stockham autosort, one pass per radix.
    Butterfly j < N / R with k = j mod Ns reads a[j + r N/R], multiplies by W^{r k N/(Ns R)} (table index
    r * k * step * tws, no reduction needed) and writes the DFT-R outputs to (j - k) R + k + q Ns."""
    n = len(a)
    ns = 1
    a = np.asarray(a, np.complex128)
    for r_ in radix:
        nr, step = n // r_, n // (ns * r_)
        j = np.arange(nr)
        k = j - float_div(j, ns) * ns
        assert (k == j % ns).all()
        v = np.stack([a[j + r * nr] for r in range(r_)])                      # (R, nr)
        idx = np.outer(np.arange(r_), k * step)
        assert idx.max(initial=0) < n
        w = tw[idx * tws]
        v = v * (w if sign < 0 else np.conj(w))
        dft = np.exp(sign * 2j * np.pi * np.outer(np.arange(r_), np.arange(r_)) / r_)
        o = dft @ v                                                           # o[q] = sum_r v[r] e^{sign 2 pi i r q / R}
        b = np.empty(n, np.complex128)
        for q in range(r_):
            b[(j - k) * r_ + k + q * ns] = o[q]
        a = b
        ns *= r_
    return a


def rfft_generic(x, n_fft):
    """k_stft_gen: window-less real FFT of one frame of n_fft samples -> n_fft // 2 + 1 bins."""
    x = np.asarray(x, np.float64)
    tw = np.exp(-2j * np.pi * np.arange(n_fft) / n_fft)
    m = gen_fft_len(n_fft)
    plan = gen_plan(m) or [m]
    if m != n_fft:                                   # packed: z[n] = x[2n] + i x[2n+1]
        z = gen_fft(x[0::2] + 1j * x[1::2], plan, tw, 2, -1)
        k = np.arange(m + 1)
        zk, zm = z[k % m], np.conj(z[(m - k) % m])
        e, d = 0.5 * (zk + zm), 0.5 * (zk - zm)
        out = e + tw[k] * (-1j * d)                  # (k = M reads tw[M] = -1)
        out[0] = out[0].real
        out[-1] = out[-1].real
        return out
    z = gen_fft(x.astype(np.complex128), plan, tw, 1, -1)
    out = z[: n_fft // 2 + 1].copy()
    out[0] = out[0].real
    return out


def irfft_generic(spec, n_fft):
    """k_irfft_gen: n_fft // 2 + 1 bins -> n_fft real samples (numpy.fft.irfft semantics: the imaginary parts of the
    DC and Nyquist bins are ignored)."""
    spec = np.asarray(spec, np.complex128)
    tw = np.exp(-2j * np.pi * np.arange(n_fft) / n_fft)
    m = gen_fft_len(n_fft)
    plan = gen_plan(m) or [m]
    if m != n_fft:
        k = np.arange(m)
        xk, xm = spec[k].copy(), spec[m - k].copy()
        xk[0] = xk[0].real
        xm[0] = xm[0].real
        e, d = 0.5 * (xk + np.conj(xm)), 0.5 * (xk - np.conj(xm))
        z = e + 1j * (np.conj(tw[k]) * d)
        y = gen_fft(z, plan, tw, 2, +1) / m
        out = np.empty(n_fft)
        out[0::2], out[1::2] = y.real, y.imag
        return out
    kfull = np.arange(n_fft)
    kk = np.where(kfull < n_fft // 2 + 1, kfull, n_fft - kfull)
    a = np.where(kfull < n_fft // 2 + 1, spec[kk], np.conj(spec[kk]))
    a[kk == 0] = a[kk == 0].real
    return gen_fft(a, plan, tw, 1, +1).real / n_fft
