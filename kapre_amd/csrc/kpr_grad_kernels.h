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
    return 10.0f * (logf(fmaxf(v, amin)) * 0.43429448190325182765f) - ref_term;      // = to_db()
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

}  // namespace kpr
