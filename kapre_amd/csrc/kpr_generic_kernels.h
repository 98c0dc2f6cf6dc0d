// kpr_generic_kernels.h -- size-generic kernels: (a) the float64 / complex128 variants of the layer chain (STFT, InverseSTFT,
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
KPR_DEV T2 gen_output(const T2* a, const T2* tw, int tws, int j, int nr, int idx0, int N, int R, int sign) {
    auto sr = a[j].x, si = a[j].y;                                // r = 0: twiddle 1
    if constexpr (RC > 0) {
        T2 v[RC - 1], w[RC - 1];
        int idx = 0;
#pragma unroll
        for (int r = 1; r < RC; ++r) {
            idx += idx0;
            if (idx >= N) idx -= N;
            w[r - 1] = tw[idx * tws];
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
                w[u] = tw[idx * tws];
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

// one pass, one thread per OUTPUT (any radix): U outputs per thread and step (their loads are all in flight together);
// o >= N is clamped for the loads and masked at the store
template <int RC, int U, class T2>
KPR_DEV void gen_pass(const T2* a, T2* b, const T2* tw, int tws, int N, int R, int ns, int sign) {
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
            r[u] = gen_output<RC>(a, tw, tws, blk * ns + k, nr, rem * step, N, R, sign);        // (k + q ns) step < N
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (o0 + u * kF64Threads < N) b[o0 + u * kF64Threads] = r[u];
    }
}

// complex helpers on the (x, y) structs of either precision
template <class T2> KPR_DEV T2 gc_add(T2 a, T2 b) { T2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
template <class T2> KPR_DEV T2 gc_sub(T2 a, T2 b) { T2 r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
template <class T2> KPR_DEV T2 gc_mul(T2 a, T2 w) { T2 r; r.x = a.x * w.x - a.y * w.y; r.y = a.x * w.y + a.y * w.x; return r; }
// a * (s i), s = +1 / -1
template <class T2> KPR_DEV T2 gc_muli(T2 a, int s) { T2 r; if (s > 0) { r.x = -a.y; r.y = a.x; } else { r.x = a.y; r.y = -a.x; } return r; }

// DFT-R of v[0 .. R) in registers, e^{sign 2 pi i r q / R}; R = 2, 3, 4, 5
template <int R, class T2>
KPR_DEV void gen_dft(T2 (&v)[R], int sign) {
    typedef decltype(v[0].x) T;
    if constexpr (R == 2) {
        const T2 a = v[0];
        v[0] = gc_add(a, v[1]);
        v[1] = gc_sub(a, v[1]);
    } else if constexpr (R == 4) {
        const T2 t0 = gc_add(v[0], v[2]), t1 = gc_sub(v[0], v[2]), t2 = gc_add(v[1], v[3]);
        const T2 t3 = gc_muli(gc_sub(v[1], v[3]), sign);                 // forward: -i (v1 - v3)
        v[0] = gc_add(t0, t2);
        v[2] = gc_sub(t0, t2);
        v[1] = gc_add(t1, t3);
        v[3] = gc_sub(t1, t3);
    } else if constexpr (R == 3) {
        const T h = (T)0.86602540378443864676;                           // sin(2 pi / 3)
        const T2 s = gc_add(v[1], v[2]);
        T2 d = gc_sub(v[1], v[2]);
        d.x *= h; d.y *= h;
        d = gc_muli(d, sign);
        T2 m;
        m.x = v[0].x - (T)0.5 * s.x;
        m.y = v[0].y - (T)0.5 * s.y;
        v[0] = gc_add(v[0], s);
        v[1] = gc_add(m, d);
        v[2] = gc_sub(m, d);
    } else {
        static_assert(R == 5, "register butterflies exist for radix 2, 3, 4, 5");
        const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;      // cos(2 pi / 5), cos(4 pi / 5)
        const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;       // sin(2 pi / 5), sin(4 pi / 5)
        const T2 t1 = gc_add(v[1], v[4]), t2 = gc_add(v[2], v[3]), t3 = gc_sub(v[1], v[4]), t4 = gc_sub(v[2], v[3]);
        T2 a1, a2, b1, b2;
        a1.x = v[0].x + c1 * t1.x + c2 * t2.x; a1.y = v[0].y + c1 * t1.y + c2 * t2.y;
        a2.x = v[0].x + c2 * t1.x + c1 * t2.x; a2.y = v[0].y + c2 * t1.y + c1 * t2.y;
        b1.x = s1 * t3.x + s2 * t4.x; b1.y = s1 * t3.y + s2 * t4.y;
        b2.x = s2 * t3.x - s1 * t4.x; b2.y = s2 * t3.y - s1 * t4.y;
        b1 = gc_muli(b1, sign);                                          // forward: o1 = a1 - i b1
        b2 = gc_muli(b2, sign);
        v[0] = gc_add(v[0], gc_add(t1, t2));
        v[1] = gc_add(a1, b1);
        v[4] = gc_sub(a1, b1);
        v[2] = gc_add(a2, b2);
        v[3] = gc_sub(a2, b2);
    }
}

// one pass of radix R = 2, 3, 4, 5 as register butterflies: one thread per butterfly j (N / R of them), R loads,
// R - 1 twiddles W_N^{r k N / (Ns R)} (index < N: no reduction needed), DFT-R, R stores.  U butterflies per thread
// and step in flight.
template <int R, int U, class T2>
KPR_DEV void gen_pass_bf(const T2* a, T2* b, const T2* tw, int tws, int N, int ns, int sign) {
    const int nr = N / R, step = N / (ns * R);
    const float inv_ns = 1.0f / (float)ns;
    for (int j0 = threadIdx.x; j0 < nr; j0 += U * kF64Threads) {
        T2 v[U][R], w[U][R - 1];
        int ob[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(j0 + u * kF64Threads, nr - 1);
            const int k = j - (int)(((float)j + 0.5f) * inv_ns) * ns;
            ob[u] = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) v[u][r] = a[j + r * nr];
#pragma unroll
            for (int r = 1; r < R; ++r) w[u][r - 1] = tw[(r * k * step) * tws];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                T2 ww = w[u][r - 1];
                if (sign > 0) ww.y = -ww.y;
                v[u][r] = gc_mul(v[u][r], ww);
            }
            gen_dft<R>(v[u], sign);
            if (j0 + u * kF64Threads < nr) {
#pragma unroll
                for (int q = 0; q < R; ++q) b[ob[u] + q * ns] = v[u][q];
            }
        }
    }
}

// tw = table of exp(-2 pi i j / NT) with NT = tws * p.n (tws = 2 when the transform is the half-length FFT of a packed
// real frame and the table is the full-length one)
template <class T2>
KPR_DEV T2* gen_fft(T2* a, T2* b, const GenPlan& p, const T2* tw, int tws, int sign) {
    const int N = p.n;
    int ns = 1;
    for (int ps = 0; ps < p.npass; ++ps) {
        const int R = p.radix[ps];
        switch (R) {                                                  // workgroup-uniform
            case 2: gen_pass_bf<2, 2>(a, b, tw, tws, N, ns, sign); break;
            case 3: gen_pass_bf<3, 2>(a, b, tw, tws, N, ns, sign); break;
            case 4: gen_pass_bf<4, 2>(a, b, tw, tws, N, ns, sign); break;
            case 5: gen_pass_bf<5, 1>(a, b, tw, tws, N, ns, sign); break;
            case 7: gen_pass<7, 2>(a, b, tw, tws, N, R, ns, sign); break;
            default: gen_pass<0, 1>(a, b, tw, tws, N, R, ns, sign); break;
        }
        __syncthreads();
        T2* t = a; a = b; b = t;
        ns *= R;
    }
    return a;
}

// STFT: one workgroup per frame (grid-stride).  Even n_fft: the frame is packed as z[n] = x[2n] + i x[2n+1], transformed
// with an M = n_fft / 2 point FFT (plan.n == M; the twiddle table stays the n_fft-point one, read with stride 2) and
// unpacked by the usual real-FFT pairing; odd n_fft: n_fft complex points with a zero imaginary part (plan.n == n_fft).
// LDS: a | b (plan.n complex each) [| twiddle table, n_fft entries, when TWL].
// tf.signal.stft: frame of win samples x window, zero-padded at the END to n_fft, rfft (time_frequency.py:173-181).
// TWL: the twiddle table is copied into LDS (compile-time, so that every table read is a ds_read: one pointer that may
// be LDS or global makes hipcc emit flat loads with full counter drains -- measured 8x slower)
template <class T, bool TWL>
__global__ __launch_bounds__(kF64Threads) void k_stft_gen(const T* __restrict__ x, Geom g, const T* __restrict__ window,
                                                          const typename Cplx<T>::type* __restrict__ twg, GenPlan plan,
                                                          int mode, void* __restrict__ outv) {
    typedef typename Cplx<T>::type T2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gen[];
    const int N = g.n_fft, K = g.K, M = plan.n;
    const bool packed = M != N;
    const int tws = packed ? 2 : 1;
    T2* a = reinterpret_cast<T2*>(smem_gen);
    T2* b = a + M;
    T2* twl = b + M;
    if constexpr (TWL) {
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
                    const T v = (n < g.win && t >= 0 && t < g.T) ? sv[u] * wv[u] : (T)0;
                    T* dst = reinterpret_cast<T*>(a);
                    if (packed) dst[n] = v;                  // (x[2n], x[2n+1]) = the interleaved (re, im) of z[n]
                    else { a[n].x = v; a[n].y = 0; }
                }
            }
        }
        __syncthreads();
        const T2* r = TWL ? gen_fft<T2>(a, b, plan, twl, tws, -1) : gen_fft<T2>(a, b, plan, twg, tws, -1);
        const long long base = spec_base(g, p, gf, K);
        const int st = spec_stride(g);
        for (int k = threadIdx.x; k < K; k += kF64Threads) {
            T2 v;
            if (packed) {
                // X[k] = E + W_N^k (-i D),  E = (Z[k] + conj Z[M-k]) / 2,  D = (Z[k] - conj Z[M-k]) / 2   (Z[M] = Z[0])
                const T2 zk = r[k == M ? 0 : k], zm = r[k == 0 ? 0 : M - k];
                const T2 w = TWL ? twl[k] : twg[k];
                T2 e, d;
                e.x = (T)0.5 * (zk.x + zm.x); e.y = (T)0.5 * (zk.y - zm.y);
                d.x = (T)0.5 * (zk.x - zm.x); d.y = (T)0.5 * (zk.y + zm.y);
                T2 o;                                        // -i D
                o.x = d.y; o.y = -d.x;
                v = gc_add(e, gc_mul(o, w));
            } else {
                v = r[k];
            }
            if (k == 0 || 2 * k == N) v.y = 0;                   // real input: DC and Nyquist bins are real
            if (mode == KPR_OUT_COMPLEX) reinterpret_cast<T2*>(outv)[base + (long long)k * st] = v;
            else if (mode == KPR_OUT_MAGNITUDE) reinterpret_cast<T*>(outv)[base + (long long)k * st] = (T)hypot(v.x, v.y);
            else reinterpret_cast<T*>(outv)[base + (long long)k * st] = (T)atan2(v.y, v.x);
        }
        __syncthreads();
    }
}

// inverse real FFT of one frame x synthesis window -> frames[gf][win] (tf.signal.inverse_stft: irfft, first
// win samples, window; time_frequency.py:307-314).  The overlap-add is k_ola<T>.  Even n_fft: inverse pairing into
// Z[k] (M = n_fft / 2 points), inverse FFT, z[n] = (x[2n], x[2n+1]); odd n_fft: Hermitian extension to n_fft points.
template <class T, bool TWL>
__global__ __launch_bounds__(kF64Threads) void k_irfft_gen(const typename Cplx<T>::type* __restrict__ spec, Geom g,
                                                           const T* __restrict__ synth_window,
                                                           const typename Cplx<T>::type* __restrict__ twg, GenPlan plan,
                                                           T* __restrict__ frames) {
    typedef typename Cplx<T>::type T2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gen[];
    const int N = g.n_fft, K = g.K, M = plan.n;
    const bool packed = M != N;
    const int tws = packed ? 2 : 1;
    T2* a = reinterpret_cast<T2*>(smem_gen);
    T2* b = a + M;
    T2* twl = b + M;
    if constexpr (TWL) {
        for (int n = threadIdx.x; n < N; n += kF64Threads) twl[n] = twg[n];
    }
    const T inv_m = (T)(1.0 / (double)M);
    for (long long gf = blockIdx.x; gf < g.total_frames; gf += gridDim.x) {
        const FramePos p = frame_pos(g, gf);
        const long long base = spec_base(g, p, gf, K);
        const int st = spec_stride(g);
        for (int k0 = threadIdx.x; k0 < M; k0 += 4 * kF64Threads) {
            // irfft ignores the imaginary parts of the DC and Nyquist bins
            T2 sv[4], sm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(k0 + u * kF64Threads, M - 1);
                if (packed) {
                    sv[u] = spec[base + (long long)k * st];
                    sm[u] = spec[base + (long long)(M - k) * st];
                } else {
                    sv[u] = spec[base + (long long)((k < K) ? k : N - k) * st];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * kF64Threads;
                if (k >= M) continue;
                if (packed) {
                    // Z[k] = Xe + i Xo,  Xe = (X[k] + conj X[M-k]) / 2,  Xo = W_N^{-k} (X[k] - conj X[M-k]) / 2
                    T2 xk = sv[u], xm = sm[u];
                    if (k == 0) { xk.y = 0; xm.y = 0; }      // X[0] and X[M]
                    const T2 w = TWL ? twl[k] : twg[k];
                    T2 e, d, wc;
                    e.x = (T)0.5 * (xk.x + xm.x); e.y = (T)0.5 * (xk.y - xm.y);
                    d.x = (T)0.5 * (xk.x - xm.x); d.y = (T)0.5 * (xk.y + xm.y);
                    wc.x = w.x; wc.y = -w.y;
                    const T2 xo = gc_mul(d, wc);
                    T2 z;
                    z.x = e.x - xo.y; z.y = e.y + xo.x;      // e + i xo
                    a[k] = z;
                } else {
                    const int kk = (k < K) ? k : N - k;
                    T2 v = sv[u];
                    if (k >= K) v.y = -v.y;
                    if (kk == 0 || 2 * kk == N) v.y = 0;
                    a[k] = v;
                }
            }
        }
        __syncthreads();
        const T2* r = TWL ? gen_fft<T2>(a, b, plan, twl, tws, +1) : gen_fft<T2>(a, b, plan, twg, tws, +1);
        T* dst = frames + gf * (long long)g.win;
        const T* rr = reinterpret_cast<const T*>(r);
        for (int n = threadIdx.x; n < g.win; n += kF64Threads) {
            T v = 0;
            if (n < N) v = (packed ? rr[n] : r[n].x) * inv_m * synth_window[n];    // packed: z[n/2] = (x[2m], x[2m+1])
            dst[n] = v;
        }
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
