// kpr_f64_kernels.h -- size-generic kernels: (a) the float64 / complex128 variants of the layer chain (STFT, InverseSTFT,
// Magnitude, Phase, ApplyFilterbank, MagnitudeToDecibel) and (b) the float32 STFT / inverse FFT for transform sizes
// that have no tuned plan (n_fft = 1001, 1200, 1536, 2000, 3000 ...: any size whose prime factors are <= 64).
// Kapre computes in whatever dtype the Keras layer was built with (/root/reference/kapre/time_frequency.py:155:
// "complex128 if x is float64") and accepts every n_fft tf.signal.stft does; both are the rare case, so this is one
// plain engine -- one workgroup per frame, the whole frame in LDS, a mixed-radix Stockham FFT with run-time radices --
// not the tuned float32 power-of-two family.  Before it, such sizes took the DFT-as-GEMM path (O(n_fft^2) per frame:
// 0.5 - 1.5 ms for 64 x 44100 samples where the FFT sizes take 20 - 30 us).
// Part of the single translation unit kapre_hip.hip (included there; not stand-alone).
#pragma once

namespace kpr {

constexpr int kF64Threads = 256;
constexpr int kGenMaxPasses = 16;

template <class T> struct Cplx;
template <> struct Cplx<float> { typedef float2 type; };
template <> struct Cplx<double> { typedef double2 type; };

// run-time FFT plan: N = product of radix[0 .. npass)
struct GenPlan {
    int n, npass;
    int radix[kGenMaxPasses];
};

// In-LDS complex FFT of p.n points, Stockham autosort with run-time radices, ping-pong between a and b; sign = -1
// forward, +1 inverse (unscaled).  tw[j] = exp(-2 pi i j / N) (any address space the pointer can reach).
// Pass with radix R after Ns = product of the earlier radices (j < N/R, k = j mod Ns):
//   out[(j - k) R + k + q Ns] = sum_r in[j + r N/R] * W_N^{ r (k + q Ns) N / (Ns R) },   q < R
// evaluated directly, one thread per output: R multiply-adds with the twiddle index stepped modulo N (exact
// table entries, no recurrence).  N * sum(R) operations per frame instead of N log N: the radices are small
// (4, 2, 3, 5, 7 ...), and the point of this engine is to be within a small factor of the FFT curve for EVERY size.
// one output of a pass: RC > 0 = compile-time radix (all 2 (R - 1) loads are issued before the arithmetic; with a
// run-time trip count hipcc waits for each pair, ~300 cycles per term), RC == 0 = run-time radix in groups of four
template <int RC, class T2>
KPR_DEV T2 gen_output(const T2* a, const T2* tw, int j, int nr, int idx0, int N, int R, int sign) {
    auto sr = a[j].x, si = a[j].y;                                // r = 0: twiddle 1
    if constexpr (RC > 0) {
        T2 v[RC - 1], w[RC - 1];
        int idx = 0;
#pragma unroll
        for (int r = 1; r < RC; ++r) {
            idx += idx0;
            if (idx >= N) idx -= N;
            w[r - 1] = tw[idx];
            v[r - 1] = a[j + r * nr];
        }
#pragma unroll
        for (int r = 1; r < RC; ++r) {
            const auto wr = w[r - 1].x, wi = (sign < 0) ? w[r - 1].y : -w[r - 1].y;
            sr += v[r - 1].x * wr - v[r - 1].y * wi;
            si += v[r - 1].x * wi + v[r - 1].y * wr;
        }
    } else {
        int idx = 0;
        for (int r0 = 1; r0 < R; r0 += 4) {
            T2 v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                idx += idx0;
                if (idx >= N) idx -= N;
                const int r = min(r0 + u, R - 1);                 // clamped: in-range loads, masked below
                w[u] = tw[idx];
                v[u] = a[j + r * nr];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r0 + u < R) {
                    const auto wr = w[u].x, wi = (sign < 0) ? w[u].y : -w[u].y;
                    sr += v[u].x * wr - v[u].y * wi;
                    si += v[u].x * wi + v[u].y * wr;
                }
            }
        }
    }
    T2 o;
    o.x = sr;
    o.y = si;
    return o;
}

// one pass: U outputs per thread and step (their loads are all in flight together); o >= N is clamped for the loads
// and masked at the store
template <int RC, int U, class T2>
KPR_DEV void gen_pass(const T2* a, T2* b, const T2* tw, int N, int R, int ns, int sign) {
    const int nr = N / R, step = N / (ns * R);
    // o / (ns R) and rem / ns by float reciprocal: exact for these sizes -- (o + 0.5) / d is at least 0.5 / d away
    // from an integer and the float error is below o * 2^-22 / d (o < 2^14) -- and ~20x cheaper than integer division
    const float inv_blk = 1.0f / (float)(ns * R), inv_ns = 1.0f / (float)ns;
    for (int o0 = threadIdx.x; o0 < N; o0 += U * kF64Threads) {
        T2 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int o = min(o0 + u * kF64Threads, N - 1);
            const int blk = (int)(((float)o + 0.5f) * inv_blk), rem = o - blk * (ns * R);      // rem = q * ns + k
            const int k = rem - (int)(((float)rem + 0.5f) * inv_ns) * ns;
            r[u] = gen_output<RC>(a, tw, blk * ns + k, nr, rem * step, N, R, sign);             // (k + q ns) step < N
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (o0 + u * kF64Threads < N) b[o0 + u * kF64Threads] = r[u];
    }
}

template <class T2>
KPR_DEV T2* gen_fft(T2* a, T2* b, const GenPlan& p, const T2* tw, int sign) {
    const int N = p.n;
    int ns = 1;
    for (int ps = 0; ps < p.npass; ++ps) {
        const int R = p.radix[ps];
        switch (R) {                                                  // workgroup-uniform
            case 2: gen_pass<2, 4>(a, b, tw, N, R, ns, sign); break;
            case 3: gen_pass<3, 4>(a, b, tw, N, R, ns, sign); break;
            case 4: gen_pass<4, 4>(a, b, tw, N, R, ns, sign); break;
            case 5: gen_pass<5, 2>(a, b, tw, N, R, ns, sign); break;
            case 7: gen_pass<7, 2>(a, b, tw, N, R, ns, sign); break;
            default: gen_pass<0, 1>(a, b, tw, N, R, ns, sign); break;
        }
        __syncthreads();
        T2* t = a; a = b; b = t;
        ns *= R;
    }
    return a;
}

// STFT: one workgroup per frame (grid-stride).  LDS: a | b (N complex each) [| twiddle table when tw_lds].
// tf.signal.stft: frame of win samples x window, zero-padded at the END to n_fft, rfft (time_frequency.py:173-181).
// TWL: the twiddle table is copied into LDS (compile-time, so that every table read is a ds_read: one pointer that may
// be LDS or global makes hipcc emit flat loads with full counter drains -- measured 8x slower)
template <class T, bool TWL>
__global__ __launch_bounds__(kF64Threads) void k_stft_gen(const T* __restrict__ x, Geom g, const T* __restrict__ window,
                                                          const typename Cplx<T>::type* __restrict__ twg, GenPlan plan,
                                                          int mode, void* __restrict__ outv) {
    typedef typename Cplx<T>::type T2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gen[];
    const int N = g.n_fft, K = g.K;
    T2* a = reinterpret_cast<T2*>(smem_gen);
    T2* b = a + N;
    if constexpr (TWL) {
        T2* twl = b + N;
        for (int n = threadIdx.x; n < N; n += kF64Threads) twl[n] = twg[n];
    }
    for (long long gf = blockIdx.x; gf < g.total_frames; gf += gridDim.x) {
        const FramePos p = frame_pos(g, gf);
        const T* sig = x + p.sig_off;
        // eight samples per thread in flight (clamped addresses, masked values: a load under a condition is waited
        // for on the spot, one memory round trip per sample)
        for (int n0 = threadIdx.x; n0 < N; n0 += 8 * kF64Threads) {
            T sv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0 + u * kF64Threads;
                const long long t = min(max(p.s0 + n, 0LL), (long long)g.T - 1);
                sv[u] = sig[t * p.es];
                wv[u] = window[min(n, g.win - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0 + u * kF64Threads;
                const long long t = p.s0 + n;
                if (n < N) {
                    a[n].x = (n < g.win && t >= 0 && t < g.T) ? sv[u] * wv[u] : (T)0;
                    a[n].y = 0;
                }
            }
        }
        __syncthreads();
        const T2* r = TWL ? gen_fft<T2>(a, b, plan, b + N, -1) : gen_fft<T2>(a, b, plan, twg, -1);
        const long long base = spec_base(g, p, gf, K);
        const int st = spec_stride(g);
        for (int k = threadIdx.x; k < K; k += kF64Threads) {
            T2 v = r[k];
            if (k == 0 || 2 * k == N) v.y = 0;                   // real input: DC and Nyquist bins are real
            if (mode == KPR_OUT_COMPLEX) reinterpret_cast<T2*>(outv)[base + (long long)k * st] = v;
            else if (mode == KPR_OUT_MAGNITUDE) reinterpret_cast<T*>(outv)[base + (long long)k * st] = (T)hypot(v.x, v.y);
            else reinterpret_cast<T*>(outv)[base + (long long)k * st] = (T)atan2(v.y, v.x);
        }
        __syncthreads();
    }
}

// inverse real FFT of one frame x synthesis window -> frames[gf][win] (tf.signal.inverse_stft: irfft, first
// win samples, window; time_frequency.py:307-314).  The overlap-add is k_ola<T>.
template <class T, bool TWL>
__global__ __launch_bounds__(kF64Threads) void k_irfft_gen(const typename Cplx<T>::type* __restrict__ spec, Geom g,
                                                           const T* __restrict__ synth_window,
                                                           const typename Cplx<T>::type* __restrict__ twg, GenPlan plan,
                                                           T* __restrict__ frames) {
    typedef typename Cplx<T>::type T2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gen[];
    const int N = g.n_fft, K = g.K;
    T2* a = reinterpret_cast<T2*>(smem_gen);
    T2* b = a + N;
    if constexpr (TWL) {
        T2* twl = b + N;
        for (int n = threadIdx.x; n < N; n += kF64Threads) twl[n] = twg[n];
    }
    const T inv_n = (T)(1.0 / (double)N);
    for (long long gf = blockIdx.x; gf < g.total_frames; gf += gridDim.x) {
        const FramePos p = frame_pos(g, gf);
        const long long base = spec_base(g, p, gf, K);
        const int st = spec_stride(g);
        for (int k0 = threadIdx.x; k0 < N; k0 += 4 * kF64Threads) {
            // Hermitian extension; irfft ignores the imaginary parts of the DC and Nyquist bins
            T2 sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(k0 + u * kF64Threads, N - 1);
                sv[u] = spec[base + (long long)((k < K) ? k : N - k) * st];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * kF64Threads;
                if (k < N) {
                    const int kk = (k < K) ? k : N - k;
                    T2 v = sv[u];
                    if (k >= K) v.y = -v.y;
                    if (kk == 0 || 2 * kk == N) v.y = 0;
                    a[k] = v;
                }
            }
        }
        __syncthreads();
        const T2* r = TWL ? gen_fft<T2>(a, b, plan, b + N, +1) : gen_fft<T2>(a, b, plan, twg, +1);
        T* dst = frames + gf * (long long)g.win;
        for (int n = threadIdx.x; n < g.win; n += kF64Threads)
            dst[n] = (n < N) ? r[n].x * inv_n * synth_window[n] : (T)0;
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
