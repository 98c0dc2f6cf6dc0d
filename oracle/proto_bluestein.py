"""numpy model of the Bluestein (chirp-z) real FFT used by the HIP kernels for even transform sizes
that are not powers of two (k_stft_bs in kapre_amd/csrc/kapre_hip.hip).  TEST INFRASTRUCTURE ONLY.

A real n_fft-point transform (n_fft even) is an NCr = n_fft/2 point complex DFT of
z[n] = x[2n] + i x[2n+1] plus the usual pairing pass.  The NCr-point DFT is evaluated as a
convolution with a chirp, i.e. two power-of-two FFTs of size M >= 2 NCr - 1:

    w[n]  = exp(-i pi n^2 / NCr)                     (n^2 reduced mod 2 NCr: exact angles)
    a[n]  = z[n] w[n]                (n < NCr, zero padded to M)
    b[n]  = conj(w[|n|])             (n = -(NCr-1) .. NCr-1, wrapped mod M)
    Z[k]  = w[k] * IFFT_M( FFT_M(a) * FFT_M(b) )[k]         k = 0 .. NCr-1
    X[k]  = 1/2 [ (Z[k] + conj Z[NCr-k]) - i exp(-2 pi i k / n_fft) (Z[k] - conj Z[NCr-k]) ]

The device keeps Bt = FFT_M(b) / (2 M) (the 1/M of the inverse FFT and the 1/2 of the pairing) and
evaluates the inverse transform as conj(FFT(conj(.))).
"""
import numpy as np


def next_pow2(n):
    m = 1
    while m < n:
        m *= 2
    return m


def tables(n_fft):
    """(M, w[M] (zero beyond NCr), Bt[M], t[NCr+1]) -- what the host uploads per n_fft."""
    assert n_fft % 2 == 0
    ncr = n_fft // 2
    m = max(next_pow2(2 * ncr - 1), 128)
    n = np.arange(ncr)
    w = np.exp(-1j * np.pi * ((n * n) % (2 * ncr)) / ncr)
    b = np.zeros(m, dtype=complex)
    b[:ncr] = np.conj(w)
    b[m - ncr + 1:] = np.conj(w[1:][::-1])
    wt = np.zeros(m, dtype=complex)
    wt[:ncr] = w
    bt = np.fft.fft(b) / (2.0 * m)
    t = np.exp(-2j * np.pi * np.arange(ncr + 1) / n_fft)
    return m, wt, bt, t


def rfft_bluestein(x):
    """x: (n_fft,) real (already windowed) -> (n_fft/2 + 1,) complex, following the device steps."""
    n_fft = x.shape[0]
    ncr = n_fft // 2
    m, wt, bt, t = tables(n_fft)
    z = np.zeros(m, dtype=complex)
    z[:ncr] = x[0::2] + 1j * x[1::2]
    a = z * wt
    big = np.fft.fft(a) * bt
    c = np.conj(np.fft.fft(np.conj(big)))                 # unnormalised inverse (1/M is in bt)
    zz = c[:ncr] * wt[:ncr]                               # = Z / 2
    k = np.arange(ncr + 1)
    zk = zz[k % ncr]
    zp = np.conj(zz[(ncr - k) % ncr])
    return (zk + zp) - 1j * t * (zk - zp)


def irfft_bluestein(X):
    """X: (n_fft/2 + 1,) complex -> (n_fft,) real = numpy.fft.irfft(X, n_fft), following the device
    steps of k_irfft_bs: inverse pairing -> Z, then the NCr-point inverse DFT as
    conj(DFT(conj Z)) / NCr with the same chirp machinery as the forward transform."""
    ncr = X.shape[0] - 1
    n_fft = 2 * ncr
    m, wt, bt, t = tables(n_fft)
    k = np.arange(ncr)
    xk = X[:ncr].copy()
    xp = np.conj(X[ncr - k])                              # conj X[NCr - k]
    xk[0] = X[0].real + 0j                                # irfft ignores Im of DC / Nyquist
    xp[0] = X[ncr].real + 0j
    e = 0.5 * (xk + xp)
    od = 0.5 * (xk - xp) * np.conj(t[:ncr])
    zk = e + 1j * od                                      # Z[k], k < NCr
    a = np.zeros(m, dtype=complex)
    a[:ncr] = np.conj(zk) * wt[:ncr]
    big = np.fft.fft(a) * bt
    c = np.conj(np.fft.fft(np.conj(big)))
    y = c[:ncr] * wt[:ncr]                                # = DFT(conj Z) / 2
    z = np.conj(y) * (2.0 / ncr)
    out = np.empty(n_fft)
    out[0::2] = z.real
    out[1::2] = z.imag
    return out
