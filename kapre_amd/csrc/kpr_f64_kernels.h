// kpr_f64_kernels.h -- the float64 / complex128 variants of the layer chain (STFT, InverseSTFT, Magnitude, Phase,
// ApplyFilterbank, MagnitudeToDecibel).  Kapre computes in whatever dtype the Keras layer was built with
// (/root/reference/kapre/time_frequency.py:155: "complex128 if x is float64"); float64 is the rare case, so these are
// plain, size-generic kernels -- one workgroup per frame, the whole frame in LDS -- not the tuned fp32 family.
// Part of the single translation unit kapre_hip.hip (included there; not stand-alone).
#pragma once

namespace kpr {

constexpr int kF64Threads = 256;

// in-LDS complex FFT of N points (N a power of two): radix-2 Stockham, ping-pong between a and b; sign = -1 forward,
// +1 inverse (unscaled).  tw[j] = exp(-2 pi i j / N).  Returns the buffer that holds the result.
KPR_DEV double2* f64_fft_pow2(double2* a, double2* b, int N, const double2* __restrict__ tw, int sign) {
    for (int ns = 1; ns < N; ns <<= 1) {
        const int step = N / (2 * ns);
        for (int j = threadIdx.x; j < N / 2; j += kF64Threads) {
            const int k = j & (ns - 1);
            const double2 w0 = tw[k * step];
            const double wr = w0.x, wi = (sign < 0) ? w0.y : -w0.y;
            const double2 u = a[j], v = a[j + N / 2];
            const double tr = v.x * wr - v.y * wi, ti = v.x * wi + v.y * wr;
            const int o = ((j - k) << 1) + k;
            b[o] = double2{u.x + tr, u.y + ti};
            b[o + ns] = double2{u.x - tr, u.y - ti};
        }
        __syncthreads();
        double2* t = a; a = b; b = t;
    }
    return a;
}

// direct DFT of N points, bins [0, nb): X[k] = sum_n a[n] exp(sign 2 pi i n k / N) (exact twiddle index n k mod N)
KPR_DEV void f64_dft(const double2* a, double2* b, int N, int nb, const double2* __restrict__ tw, int sign) {
    for (int k = threadIdx.x; k < nb; k += kF64Threads) {
        double sr = 0.0, si = 0.0;
        int idx = 0;
        for (int n = 0; n < N; ++n) {
            const double2 w0 = tw[idx];
            const double wr = w0.x, wi = (sign < 0) ? w0.y : -w0.y;
            const double2 v = a[n];
            sr += v.x * wr - v.y * wi;
            si += v.x * wi + v.y * wr;
            idx += k;
            if (idx >= N) idx -= N;
        }
        b[k] = double2{sr, si};
    }
    __syncthreads();
}

// STFT, float64: one workgroup per frame (grid-stride).  LDS: 2 x n_fft double2.
// tf.signal.stft: frame of win samples x window, zero-padded at the END to n_fft, rfft (time_frequency.py:173-181).
__global__ __launch_bounds__(kF64Threads) void k_stft_f64(const double* __restrict__ x, Geom g,
                                                          const double* __restrict__ window,
                                                          const double2* __restrict__ tw, int pow2, int mode,
                                                          void* __restrict__ outv) {
    extern __shared__ __attribute__((aligned(16))) double2 smem64[];
    const int N = g.n_fft, K = g.K;
    double2* a = smem64;
    double2* b = smem64 + N;
    for (long long gf = blockIdx.x; gf < g.total_frames; gf += gridDim.x) {
        const FramePos p = frame_pos(g, gf);
        const double* sig = x + p.sig_off;
        for (int n = threadIdx.x; n < N; n += kF64Threads) {
            const long long t = p.s0 + n;
            double v = 0.0;
            if (n < g.win && t >= 0 && t < g.T) v = sig[t * p.es] * window[n];
            a[n] = double2{v, 0.0};
        }
        __syncthreads();
        const double2* r;
        if (pow2) r = f64_fft_pow2(a, b, N, tw, -1);
        else { f64_dft(a, b, N, K, tw, -1); r = b; }
        const long long base = spec_base(g, p, gf, K);
        const int st = spec_stride(g);
        for (int k = threadIdx.x; k < K; k += kF64Threads) {
            double2 v = r[k];
            if (k == 0 || 2 * k == N) v.y = 0.0;                 // real input: DC and Nyquist bins are real
            if (mode == KPR_OUT_COMPLEX) reinterpret_cast<double2*>(outv)[base + (long long)k * st] = v;
            else if (mode == KPR_OUT_MAGNITUDE) reinterpret_cast<double*>(outv)[base + (long long)k * st] = hypot(v.x, v.y);
            else reinterpret_cast<double*>(outv)[base + (long long)k * st] = atan2(v.y, v.x);
        }
        __syncthreads();
    }
}

// inverse real FFT of one frame x synthesis window -> frames[gf][win] (tf.signal.inverse_stft: irfft, first
// win samples, window; time_frequency.py:307-314).  The overlap-add is k_ola<double>.
__global__ __launch_bounds__(kF64Threads) void k_irfft_f64(const double2* __restrict__ spec, Geom g,
                                                           const double* __restrict__ synth_window,
                                                           const double2* __restrict__ tw, int pow2,
                                                           double* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) double2 smem64[];
    const int N = g.n_fft, K = g.K;
    double2* a = smem64;
    double2* b = smem64 + N;
    const double inv_n = 1.0 / (double)N;
    for (long long gf = blockIdx.x; gf < g.total_frames; gf += gridDim.x) {
        const FramePos p = frame_pos(g, gf);
        const long long base = spec_base(g, p, gf, K);
        const int st = spec_stride(g);
        for (int k = threadIdx.x; k < N; k += kF64Threads) {
            // Hermitian extension; irfft ignores the imaginary parts of the DC and Nyquist bins
            const int kk = (k < K) ? k : N - k;
            double2 v = spec[base + (long long)kk * st];
            if (k >= K) v.y = -v.y;
            if (kk == 0 || 2 * kk == N) v.y = 0.0;
            a[k] = v;
        }
        __syncthreads();
        const double2* r;
        if (pow2) r = f64_fft_pow2(a, b, N, tw, +1);
        else { f64_dft(a, b, N, N, tw, +1); r = b; }
        double* dst = frames + gf * (long long)g.win;
        for (int n = threadIdx.x; n < g.win; n += kF64Threads)
            dst[n] = (n < N) ? r[n].x * inv_n * synth_window[n] : 0.0;
        __syncthreads();
    }
}

// Magnitude / Phase on complex128 (tf.abs / tf.math.angle, time_frequency.py:359, :402)
__global__ void k_cplx_to_real_f64(const double2* __restrict__ x, long long n, int phase, double* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double2 v = x[i];
        out[i] = phase ? atan2(v.y, v.x) : hypot(v.x, v.y);
    }
}

// ApplyFilterbank, float64 (tf.tensordot over the frequency axis, time_frequency.py:535-548): rows = (batch, ch, frame)
// in either layout, es = element stride of the frequency axis (ch for channels_last, 1 for channels_first)
__global__ void k_filterbank_f64(const double* __restrict__ x, long long batch, int C, long long F, int n_freq,
                                 int layout_last, const double* __restrict__ fb, int n_filt, double* __restrict__ out) {
    const long long total = batch * C * F * n_filt;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i % n_filt);
        const long long row = i / n_filt;            // row = (b * C + c) * F + f
        const long long f = row % F, bc = row / F;
        const long long b = bc / C;
        const int c = (int)(bc - b * C);
        long long xin, xo;
        int es;
        if (layout_last) { xin = ((b * F + f) * n_freq) * C + c; xo = ((b * F + f) * n_filt + m) * C + c; es = C; }
        else { xin = row * n_freq; xo = row * n_filt + m; es = 1; }
        double acc = 0.0;
        for (int k = 0; k < n_freq; ++k) acc += x[xin + (long long)k * es] * fb[(long long)k * n_filt + m];
        out[xo] = acc;
    }
}

// MagnitudeToDecibel, float64 (backend.py:178-192): one workgroup per batch item -- log pass with the item maximum,
// then the dynamic-range clamp by the same workgroup
__global__ __launch_bounds__(1024) void k_db_f64(const double* __restrict__ x, long long item_size, double amin,
                                                 double ref_term, double dyn, double* __restrict__ out) {
    __shared__ double red[1024];
    const long long base = (long long)blockIdx.x * item_size;
    double mx = -INFINITY;
    for (long long i = threadIdx.x; i < item_size; i += blockDim.x) {
        const double d = 10.0 * log10(fmax(x[base + i], amin)) - ref_term;
        out[base + i] = d;
        mx = fmax(mx, d);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const double floor_db = red[0] - dyn;
    for (long long i = threadIdx.x; i < item_size; i += blockDim.x)     // each thread re-reads its own stores
        out[base + i] = fmax(out[base + i], floor_db);
}

}  // namespace kpr
