// kpr_grad_kernels.h -- vector-Jacobian products (the "backward" of tf.GradientTape) of the elementwise layers of
// the path.  The reference's layers are differentiable TensorFlow graphs (a Kapre front end sits inside model.fit,
// time_frequency.py:146-187, :351-359, :535-548, backend.py:186-192); the linear layers reuse the forward kernels for
// their adjoints -- STFT^T is an inverse-STFT launch with window N w and the interior bins halved, InverseSTFT^T is an
// STFT launch with window 2 w_s / N and the edge bins halved, ApplyFilterbank^T is the same GEMM with the transposed
// matrix -- and this header holds what is left: the gradients of tf.abs / tf.math.angle on complex data, the bin
// scaling between the two conventions, and the decibel map (with the gradient that reaches an item's maximum through
// the dynamic-range floor).  Not a hot path: plain grid-stride kernels, T = float | double.
#pragma once

namespace kpr {

template <typename T> struct GCplx { T x, y; };

// tf.abs on complex x (math_grad.py _ComplexAbsGrad): grad * sign(x), sign(x) = x / |x| (0 at 0)
// tf.math.angle (math_grad.py _AngleGrad): -grad / (im + i re) = grad * (-im + i re) / |x|^2
template <typename T>
__global__ void k_cplx_to_real_bwd(const GCplx<T>* __restrict__ x, const T* __restrict__ g, long long n, int phase,
                                   GCplx<T>* __restrict__ gx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const GCplx<T> v = x[i];
        const T gi = g[i];
        const T r2 = v.x * v.x + v.y * v.y;
        GCplx<T> o;
        if (phase) {
            const T s = r2 > (T)0 ? gi / r2 : (T)0;
            o.x = -v.y * s;
            o.y = v.x * s;
        } else {
            const T r = sqrt(r2);
            const T s = r > (T)0 ? gi / r : (T)0;
            o.x = v.x * s;
            o.y = v.y * s;
        }
        gx[i] = o;
    }
}

// out = in * (s_edge on the bins a real transform keeps once -- DC, and Nyquist for even n_fft -- else s_mid).
// in / out: complex spectrogram with K bins, `inner` elements per bin step (C for channels_last, 1 otherwise);
// nyq = n_fft / 2 for even n_fft, -1 for odd.  In place (out == in) is allowed.
template <typename T>
__global__ void k_spec_edge_scale(const GCplx<T>* in, long long n, int K, int inner, int nyq, T s_edge, T s_mid,
                                  GCplx<T>* out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)((i / inner) % K);
        const T s = (k == 0 || k == nyq) ? s_edge : s_mid;
        GCplx<T> v = in[i];
        v.x *= s;
        v.y *= s;
        out[i] = v;
    }
}

template <typename T> KPR_DEV T db_value(T v, T amin, T ref_term);
template <> KPR_DEV float db_value<float>(float v, float amin, float ref_term) {
    // EXACTLY the forward's arithmetic (to_db, kpr_common.h: v_log_f32 and one fused multiply-add; amin already raised to the
    // smallest normal float by the caller): the masks [l >= max - dyn] and [l == max] must agree with what the forward
    // clamped, also for elements within an ulp of the floor and for exact ties (VERDICT r03 / ADVICE r03)
    return fmaf(3.01029995663981195f, __builtin_amdgcn_logf(fmaxf(v, amin)), -ref_term);
}
template <> KPR_DEV double db_value<double>(double v, double amin, double ref_term) {
    return 10.0 * log10(fmax(v, amin)) - ref_term;                                    // = k_db_f64
}

template <typename T, typename Op>
KPR_DEV T block_reduce_1024(T v, T* red, Op op) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = op(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const T r = red[0];
    __syncthreads();
    return r;
}

// backend.magnitude_to_decibel (backend.py:186-192) differentiated the way TensorFlow does it:
//   a = max(x, amin); l = 10 log10 a - ref; m = reduce_max(l) per item; y = max(l, m - dyn)
//   dL/dl = g [l >= m - dyn]  +  [l == m] / #{l == m} * sum(g [l < m - dyn])     (the floor moves with the maximum)
//   dL/dx = dL/dl * 10 / (ln 10 * a) * [x >= amin]
// One 1024-thread workgroup per item, three passes over the item (maximum; floor sum and tie count; gradient).
template <typename T>
__global__ __launch_bounds__(1024) void k_db_bwd(const T* __restrict__ x, const T* __restrict__ g, long long item_size,
                                                 T amin, T ref_term, T dyn, T* __restrict__ gx) {
    __shared__ T red[1024];
    const long long base = (long long)blockIdx.x * item_size;
    T mx = -INFINITY;
    for (long long i = threadIdx.x; i < item_size; i += blockDim.x) mx = fmax(mx, db_value<T>(x[base + i], amin, ref_term));
    mx = block_reduce_1024(mx, red, [](T a, T b) { return fmax(a, b); });
    const T thr = mx - dyn;
    T fsum = 0, ties = 0;
    for (long long i = threadIdx.x; i < item_size; i += blockDim.x) {
        const T l = db_value<T>(x[base + i], amin, ref_term);
        if (l < thr) fsum += g[base + i];
        if (l == mx) ties += (T)1;
    }
    fsum = block_reduce_1024(fsum, red, [](T a, T b) { return a + b; });
    ties = block_reduce_1024(ties, red, [](T a, T b) { return a + b; });
    const T to_max = ties > (T)0 ? fsum / ties : (T)0;
    const T c = (T)4.3429448190325182765;                                             // 10 / ln 10
    for (long long i = threadIdx.x; i < item_size; i += blockDim.x) {
        const T v = x[base + i];
        const T l = db_value<T>(v, amin, ref_term);
        const T gl = (l >= thr ? g[base + i] : (T)0) + (l == mx ? to_max : (T)0);
        gx[base + i] = v >= amin ? gl * c / v : (T)0;
    }
}

// ------------------------------------------------------------------------------------------
// Frame / Energy / Delta (kapre/signal.py:22-240, time_frequency.py:563-644): gather-form adjoints -- every thread owns
// one element of the input cotangent and sums what reaches it in a fixed order (no atomics, deterministic).
// ------------------------------------------------------------------------------------------
// frames f covering sample t: f hop <= t < f hop + L, f < F
KPR_DEV void frames_covering(long long t, const FrameArgs& a, int* f_lo, int* f_hi) {
    const long long hi = t / a.hop;
    *f_hi = (int)min(hi, (long long)a.F - 1);
    const long long lo = t - a.L + 1;
    *f_lo = lo <= 0 ? 0 : (int)((lo + a.hop - 1) / a.hop);
}

// tf.signal.frame^T: gx[t] = sum_f g[f][t - f hop] (samples of the padding receive nothing).  a.cl as in FrameArgs.
__global__ __launch_bounds__(256) void k_frame_bwd(const float* __restrict__ g, FrameArgs a, float* __restrict__ gx) {
    const long long total = a.n_sig * a.T;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        long long b, t;
        int c;
        if (a.cl) { b = e / (a.T * a.C); const long long r = e - b * a.T * a.C; t = r / a.C; c = (int)(r - t * a.C); }
        else { b = e / a.T; t = e - b * a.T; c = 0; }                  // b = signal index b * C + c
        int f_lo, f_hi;
        frames_covering(t, a, &f_lo, &f_hi);
        float acc = 0.0f;
        for (int f = f_lo; f <= f_hi; ++f) {
            const long long i = t - (long long)f * a.hop;
            acc += a.cl ? g[((b * a.F + f) * a.L + i) * a.C + c] : g[(b * a.F + f) * a.L + i];
        }
        gx[e] = acc;
    }
}

// Energy^T: E[f] = scale sum_{i < L} x[f hop + i]^2  ->  gx[t] = 2 scale x[t] sum_{f covering t} g[f]
__global__ __launch_bounds__(256) void k_energy_bwd(const float* __restrict__ x, const float* __restrict__ g, FrameArgs a,
                                                    float scale, float* __restrict__ gx) {
    const long long total = a.n_sig * a.T;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        long long b, t;
        int c;
        if (a.cl) { b = e / (a.T * a.C); const long long r = e - b * a.T * a.C; t = r / a.C; c = (int)(r - t * a.C); }
        else { b = e / a.T; t = e - b * a.T; c = 0; }
        int f_lo, f_hi;
        frames_covering(t, a, &f_lo, &f_hi);
        float acc = 0.0f;
        for (int f = f_lo; f <= f_hi; ++f) acc += a.cl ? g[(b * a.F + f) * a.C + c] : g[b * a.F + f];
        gx[e] = 2.0f * scale * x[e] * acc;
    }
}

// Delta^T.  y[t] = inv sum_j j (x[src(t + j)] - x[src(t - j)]) with src = delta_src_index (the padding mode), so
// gx[s] = inv sum_{u : src(u) == s} sum_j j (g[u - j] - g[u + j]), g = 0 outside [0, T): u = s itself plus its mirror
// images among the 2n padded positions.  x viewed as (outer, T, inner) like k_delta.
KPR_DEV float delta_bwd_term(const float* __restrict__ gcol, long long u, long long T, long long inner, int n) {
    float acc = 0.0f;
    for (int j = 1; j <= n; ++j) {
        const long long tm = u - j, tp = u + j;
        const float gm = (tm >= 0 && tm < T) ? gcol[tm * inner] : 0.0f;
        const float gp = (tp >= 0 && tp < T) ? gcol[tp * inner] : 0.0f;
        acc += (float)j * (gm - gp);
    }
    return acc;
}
__global__ __launch_bounds__(256) void k_delta_bwd(const float* __restrict__ g, long long outer, long long T, long long inner,
                                                   int n, int mode, float inv_denom, float* __restrict__ gx) {
    const long long total = outer * T * inner;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long q = e / inner, i = e - q * inner;
        const long long o = q / T, s = q - o * T;
        const float* gcol = g + o * T * inner + i;
        float acc = delta_bwd_term(gcol, s, T, inner, n);
        if (mode != KPR_PAD_CONSTANT) {
            if (s <= n)
                for (long long u = -n; u < 0; ++u)
                    if (delta_src_index(u, T, mode) == s) acc += delta_bwd_term(gcol, u, T, inner, n);
            if (s >= T - 1 - n)
                for (long long u = T; u < T + n; ++u)
                    if (delta_src_index(u, T, mode) == s) acc += delta_bwd_term(gcol, u, T, inner, n);
        }
        gx[e] = acc * inv_denom;
    }
}

}  // namespace kpr
