// Mixed-radix (2^a 5^b) complex FFT for transform sizes that are not powers of two:
// N = 20 * R2 * R3 points, 20 per lane, L = R2 * R3 lanes per frame.  Same lane / register model as
// the power-of-two Stockham FFT of kpr_fft.h (in-register small DFTs, LDS exchange between passes),
// with the pass structure of a recursive four-step FFT:
//   pass 1: lane l holds x[l + L m], m < 20: DFT-20 over m in registers, times W_N^{l k1}
//   exchange 1: item (a, b, k1) -- writer lane l = a + R3 b -- at row index l + L k1
//   pass 2: lane (a, b') takes k1 = b' + R2 j, j < 20/R2: DFT-R2 over b, times W_L^{a kb}
//   exchange 2 (R3 > 1): item (a, c = k1 + 20 kb) at row index c + 20 R2 a
//   pass 3: lane l3 takes c = l3 + L j3, j3 < 20/R3: DFT-R3 over a
//   result: lane l, register j*R + r holds X[l + L j + (N/R) r]   (R = R3 if R3 > 1 else R2)
// oracle/proto_mixed_radix.py is the step-by-step numpy model (tests/test_proto_stockham.py).
// Replaces, for n_fft in {160, 200, 320, 400, 640, 800, 1000, 2000}, the Bluestein evaluation of
// tf.signal.stft's rfft (kapre/time_frequency.py:174-182): one N-point FFT instead of two
// M >= 2N point FFTs.
#pragma once
#include "kpr_fft.h"

namespace kpr {

constexpr int kMrP = 20;   // complex points per lane

// exp(-2 pi i j / 40), exact at the quarter turns
__host__ __device__ constexpr float mr_cos40(int j) {
    constexpr float t[40] = {1.0f, 0.987688341f, 0.951056516f, 0.891006524f, 0.809016994f, 0.707106781f, 0.587785252f, 0.4539905f, 0.309016994f, 0.156434465f, 0.0f, -0.156434465f, -0.309016994f, -0.4539905f, -0.587785252f, -0.707106781f, -0.809016994f, -0.891006524f, -0.951056516f, -0.987688341f, -1.0f, -0.987688341f, -0.951056516f, -0.891006524f, -0.809016994f, -0.707106781f, -0.587785252f, -0.4539905f, -0.309016994f, -0.156434465f, 0.0f, 0.156434465f, 0.309016994f, 0.4539905f, 0.587785252f, 0.707106781f, 0.809016994f, 0.891006524f, 0.951056516f, 0.987688341f};
    return t[j % 40];
}
__host__ __device__ constexpr float mr_sin40(int j) {
    constexpr float t[40] = {0.0f, 0.156434465f, 0.309016994f, 0.4539905f, 0.587785252f, 0.707106781f, 0.809016994f, 0.891006524f, 0.951056516f, 0.987688341f, 1.0f, 0.987688341f, 0.951056516f, 0.891006524f, 0.809016994f, 0.707106781f, 0.587785252f, 0.4539905f, 0.309016994f, 0.156434465f, 0.0f, -0.156434465f, -0.309016994f, -0.4539905f, -0.587785252f, -0.707106781f, -0.809016994f, -0.891006524f, -0.951056516f, -0.987688341f, -1.0f, -0.987688341f, -0.951056516f, -0.891006524f, -0.809016994f, -0.707106781f, -0.587785252f, -0.4539905f, -0.309016994f, -0.156434465f};
    return t[j % 40];
}
// x * W_40^j, j a compile-time constant after unrolling
KPR_DEV f2 cmul_w40(f2 x, int j) {
    j %= 40;
    if (j == 0) return x;
    if (j == 10) return f2{x.y, -x.x};      // -i
    if (j == 20) return f2{-x.x, -x.y};
    if (j == 30) return f2{-x.y, x.x};      // +i
    return cmul_s(x, f2{mr_cos40(j), -mr_sin40(j)});
}

template <> struct Dft<5> {
    // t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;  c_k = cos(2 pi k/5), s_k = sin(2 pi k/5)
    //   X0 = x0 + t1 + t2
    //   X1,4 = (x0 + c1 t1 + c2 t2) -/+ i (s1 t3 + s2 t4)
    //   X2,3 = (x0 + c2 t1 + c1 t2) -/+ i (s2 t3 - s1 t4)
    static KPR_DEV void run(f2 (&v)[5]) {
        constexpr float c1 = 0.309016994374947424f, c2 = -0.809016994374947424f;
        constexpr float s1 = 0.951056516295153572f, s2 = 0.587785252292473129f;
        const f2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
        const f2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
        const f2 a1 = v[0] + c1 * t1 + c2 * t2, a2 = v[0] + c2 * t1 + c1 * t2;
        const f2 b1 = s1 * t3 + s2 * t4, b2 = s2 * t3 - s1 * t4;
        v[0] = cadd(v[0], cadd(t1, t2));
        v[1] = cadd_mi(a1, b1);
        v[4] = cadd_pi(a1, b1);
        v[2] = cadd_mi(a2, b2);
        v[3] = cadd_pi(a2, b2);
    }
};

// DFT-5A, A in {2, 4}: m = u + 5 t, k = kt + A ku: DFT-A over t, times W_5A^{u kt}, DFT-5 over u
template <int A>
KPR_DEV void dft5a(f2 (&v)[5 * A]) {
    f2 y[5][A];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        f2 t[A];
#pragma unroll
        for (int s = 0; s < A; ++s) t[s] = v[u + 5 * s];
        Dft<A>::run(t);
#pragma unroll
        for (int kt = 0; kt < A; ++kt) y[u][kt] = cmul_w40(t[kt], (40 / (5 * A)) * u * kt);
    }
#pragma unroll
    for (int kt = 0; kt < A; ++kt) {
        f2 t[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) t[u] = y[u][kt];
        Dft<5>::run(t);
#pragma unroll
        for (int ku = 0; ku < 5; ++ku) v[kt + A * ku] = t[ku];
    }
}
template <> struct Dft<10> { static KPR_DEV void run(f2 (&v)[10]) { dft5a<2>(v); } };
template <> struct Dft<20> { static KPR_DEV void run(f2 (&v)[20]) { dft5a<4>(v); } };
template <> struct Dft<1> { static KPR_DEV void run(f2 (&)[1]) {} };

// `tab[j]` = exp(-2 pi i j / (2N)), j < 2N, in LDS (the table the real-FFT pairing needs anyway);
// `row` = this frame's exchange row, N complex words.  `active` = false for the lanes of a wave
// that do not belong to a whole frame: they run along but never write to LDS.
template <int R2, int R3>
struct MrFft {
    static constexpr int P = kMrP, L = R2 * R3, N = P * L, Q2 = P / R2, Q3 = P / R3;
    static constexpr int RL = (R3 > 1) ? R3 : R2;      // radix of the last pass
    static_assert(P % R2 == 0 && P % R3 == 0, "every pass works on whole groups of a lane's 20 points");

    // bin held by register `reg` of lane l after run()
    static KPR_DEV int bin(int l, int reg) { return l + L * (reg / RL) + (N / RL) * (reg % RL); }

    static KPR_DEV void run(f2 (&z)[P], int l, bool active, f2* row, const f2* tab) {
        Dft<P>::run(z);
#pragma unroll
        for (int k1 = 1; k1 < P; ++k1) z[k1] = cmul(z[k1], tab[2 * l * k1]);       // W_N^{l k1}
        if (active) {
#pragma unroll
            for (int k1 = 0; k1 < P; ++k1) row[l + L * k1] = z[k1];
        }
        const int a = l % R3, bp = l / R3;
#pragma unroll
        for (int j = 0; j < Q2; ++j) {
            f2 t[R2];
#pragma unroll
            for (int b = 0; b < R2; ++b) t[b] = row[(a + R3 * b) + L * (bp + R2 * j)];
            Dft<R2>::run(t);
#pragma unroll
            for (int kb = 0; kb < R2; ++kb)
                z[j * R2 + kb] = (R3 > 1 && kb > 0) ? cmul(t[kb], tab[2 * P * a * kb]) : t[kb];   // W_L^{a kb}
        }
        if constexpr (R3 > 1) {
            if (active) {
#pragma unroll
                for (int j = 0; j < Q2; ++j)
#pragma unroll
                    for (int kb = 0; kb < R2; ++kb) row[(bp + R2 * j + P * kb) + P * R2 * a] = z[j * R2 + kb];
            }
#pragma unroll
            for (int j3 = 0; j3 < Q3; ++j3) {
                f2 t[R3];
#pragma unroll
                for (int a2 = 0; a2 < R3; ++a2) t[a2] = row[(l + L * j3) + P * R2 * a2];
                Dft<R3>::run(t);
#pragma unroll
                for (int ka = 0; ka < R3; ++ka) z[j3 * R3 + ka] = t[ka];
            }
        }
    }
};

}  // namespace kpr
