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
// (TwoPassFft below covers the sizes with a factor 3.)
// Replaces, for n_fft in {160, 200, 320, 400, 640, 800, 1000} and {96, 120, 192, 240, 360, 384, 480, 600, 720, 768, 960}, the Bluestein evaluation of
// tf.signal.stft's rfft (kapre/time_frequency.py:174-182): one N-point FFT instead of two
// M >= 2N point FFTs.
#pragma once
#include "kpr_fft.h"

namespace kpr {

constexpr int kMrP = 20;   // complex points per lane of MrFft

// exp(-2 pi i j / 120): the inner twiddles of every composite in-register DFT used here
// (W_6, W_10, W_12, W_15, W_20, W_24 are powers of it); exact at the quarter turns
__host__ __device__ constexpr float mr_cos120(int j) {
    constexpr float t[120] = {1.0f, 0.998629535f, 0.994521895f, 0.987688341f, 0.978147601f, 0.965925826f, 0.951056516f, 0.933580426f, 0.913545458f, 0.891006524f, 0.866025404f, 0.838670568f, 0.809016994f, 0.777145961f, 0.743144825f, 0.707106781f, 0.669130606f, 0.629320391f, 0.587785252f, 0.544639035f, 0.5f, 0.4539905f, 0.406736643f, 0.35836795f, 0.309016994f, 0.258819045f, 0.207911691f, 0.156434465f, 0.104528463f, 0.0523359562f, 0.0f, -0.0523359562f, -0.104528463f, -0.156434465f, -0.207911691f, -0.258819045f, -0.309016994f, -0.35836795f, -0.406736643f, -0.4539905f, -0.5f, -0.544639035f, -0.587785252f, -0.629320391f, -0.669130606f, -0.707106781f, -0.743144825f, -0.777145961f, -0.809016994f, -0.838670568f, -0.866025404f, -0.891006524f, -0.913545458f, -0.933580426f, -0.951056516f, -0.965925826f, -0.978147601f, -0.987688341f, -0.994521895f, -0.998629535f, -1.0f, -0.998629535f, -0.994521895f, -0.987688341f, -0.978147601f, -0.965925826f, -0.951056516f, -0.933580426f, -0.913545458f, -0.891006524f, -0.866025404f, -0.838670568f, -0.809016994f, -0.777145961f, -0.743144825f, -0.707106781f, -0.669130606f, -0.629320391f, -0.587785252f, -0.544639035f, -0.5f, -0.4539905f, -0.406736643f, -0.35836795f, -0.309016994f, -0.258819045f, -0.207911691f, -0.156434465f, -0.104528463f, -0.0523359562f, 0.0f, 0.0523359562f, 0.104528463f, 0.156434465f, 0.207911691f, 0.258819045f, 0.309016994f, 0.35836795f, 0.406736643f, 0.4539905f, 0.5f, 0.544639035f, 0.587785252f, 0.629320391f, 0.669130606f, 0.707106781f, 0.743144825f, 0.777145961f, 0.809016994f, 0.838670568f, 0.866025404f, 0.891006524f, 0.913545458f, 0.933580426f, 0.951056516f, 0.965925826f, 0.978147601f, 0.987688341f, 0.994521895f, 0.998629535f};
    return t[j % 120];
}
__host__ __device__ constexpr float mr_sin120(int j) {
    constexpr float t[120] = {0.0f, 0.0523359562f, 0.104528463f, 0.156434465f, 0.207911691f, 0.258819045f, 0.309016994f, 0.35836795f, 0.406736643f, 0.4539905f, 0.5f, 0.544639035f, 0.587785252f, 0.629320391f, 0.669130606f, 0.707106781f, 0.743144825f, 0.777145961f, 0.809016994f, 0.838670568f, 0.866025404f, 0.891006524f, 0.913545458f, 0.933580426f, 0.951056516f, 0.965925826f, 0.978147601f, 0.987688341f, 0.994521895f, 0.998629535f, 1.0f, 0.998629535f, 0.994521895f, 0.987688341f, 0.978147601f, 0.965925826f, 0.951056516f, 0.933580426f, 0.913545458f, 0.891006524f, 0.866025404f, 0.838670568f, 0.809016994f, 0.777145961f, 0.743144825f, 0.707106781f, 0.669130606f, 0.629320391f, 0.587785252f, 0.544639035f, 0.5f, 0.4539905f, 0.406736643f, 0.35836795f, 0.309016994f, 0.258819045f, 0.207911691f, 0.156434465f, 0.104528463f, 0.0523359562f, 0.0f, -0.0523359562f, -0.104528463f, -0.156434465f, -0.207911691f, -0.258819045f, -0.309016994f, -0.35836795f, -0.406736643f, -0.4539905f, -0.5f, -0.544639035f, -0.587785252f, -0.629320391f, -0.669130606f, -0.707106781f, -0.743144825f, -0.777145961f, -0.809016994f, -0.838670568f, -0.866025404f, -0.891006524f, -0.913545458f, -0.933580426f, -0.951056516f, -0.965925826f, -0.978147601f, -0.987688341f, -0.994521895f, -0.998629535f, -1.0f, -0.998629535f, -0.994521895f, -0.987688341f, -0.978147601f, -0.965925826f, -0.951056516f, -0.933580426f, -0.913545458f, -0.891006524f, -0.866025404f, -0.838670568f, -0.809016994f, -0.777145961f, -0.743144825f, -0.707106781f, -0.669130606f, -0.629320391f, -0.587785252f, -0.544639035f, -0.5f, -0.4539905f, -0.406736643f, -0.35836795f, -0.309016994f, -0.258819045f, -0.207911691f, -0.156434465f, -0.104528463f, -0.0523359562f};
    return t[j % 120];
}
// x * W_120^j, j a compile-time constant after unrolling
KPR_DEV f2 cmul_w120(f2 x, int j) {
    j %= 120;
    if (j == 0) return x;
    if (j == 30) return f2{x.y, -x.x};      // -i
    if (j == 60) return f2{-x.x, -x.y};
    if (j == 90) return f2{-x.y, x.x};      // +i
    return cmul_s(x, f2{mr_cos120(j), -mr_sin120(j)});
}

template <> struct Dft<3> {
    // t1 = x1 + x2, t2 = x1 - x2:  X0 = x0 + t1,  X1,2 = (x0 - t1/2) -/+ i sin(2 pi/3) t2
    static KPR_DEV void run(f2 (&v)[3]) {
        constexpr float s = 0.866025403784438647f;
        const f2 t1 = cadd(v[1], v[2]), t2 = csub(v[1], v[2]);
        const f2 a = v[0] - 0.5f * t1, b = s * t2;
        v[0] = cadd(v[0], t1);
        v[1] = cadd_mi(a, b);
        v[2] = cadd_pi(a, b);
    }
};

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

// DFT-(A B) in registers from DFT-A and DFT-B: m = u + B t, k = kt + A ku:
// DFT-A over t, times W_AB^{u kt}, DFT-B over u
template <int A, int B>
KPR_DEV void dft_ab(f2 (&v)[A * B]) {
    static_assert(120 % (A * B) == 0, "inner twiddles come from the W_120 table");
    f2 y[B][A];
#pragma unroll
    for (int u = 0; u < B; ++u) {
        f2 t[A];
#pragma unroll
        for (int s = 0; s < A; ++s) t[s] = v[u + B * s];
        Dft<A>::run(t);
#pragma unroll
        for (int kt = 0; kt < A; ++kt) y[u][kt] = cmul_w120(t[kt], (120 / (A * B)) * u * kt);
    }
#pragma unroll
    for (int kt = 0; kt < A; ++kt) {
        f2 t[B];
#pragma unroll
        for (int u = 0; u < B; ++u) t[u] = y[u][kt];
        Dft<B>::run(t);
#pragma unroll
        for (int ku = 0; ku < B; ++ku) v[kt + A * ku] = t[ku];
    }
}
template <> struct Dft<1> { static KPR_DEV void run(f2 (&)[1]) {} };
template <> struct Dft<6> { static KPR_DEV void run(f2 (&v)[6]) { dft_ab<2, 3>(v); } };
template <> struct Dft<10> { static KPR_DEV void run(f2 (&v)[10]) { dft_ab<2, 5>(v); } };
template <> struct Dft<12> { static KPR_DEV void run(f2 (&v)[12]) { dft_ab<4, 3>(v); } };
template <> struct Dft<15> { static KPR_DEV void run(f2 (&v)[15]) { dft_ab<3, 5>(v); } };
template <> struct Dft<20> { static KPR_DEV void run(f2 (&v)[20]) { dft_ab<4, 5>(v); } };
template <> struct Dft<24> { static KPR_DEV void run(f2 (&v)[24]) { dft_ab<8, 3>(v); } };

// `tab[j]` = exp(-2 pi i j / (2N)), j < 2N, in LDS (the table the real-FFT pairing needs anyway);
// `row` = this frame's exchange row, N complex words.  `active` = false for the lanes of a wave
// that do not belong to a whole frame: they run along but never write to LDS.
template <int R2, int R3>
struct MrFft {
    static constexpr int P = kMrP, L = R2 * R3, N = P * L, Q2 = P / R2, Q3 = P / R3;
    static constexpr int RL = (R3 > 1) ? R3 : R2;      // radix of the last pass
    static_assert(P % R2 == 0 && P % R3 == 0, "every pass works on whole groups of a lane's 20 points");
    // what the kernels need to know (shared with TwoPassFft):
    //   input: lane l < LIN holds x[l + LIN m] in register m < PIN; output: bin(l, r) where holds(l, r);
    //   ROW = complex words of the exchange row
    static constexpr int PIN = P, LIN = L, ROW = N;
    static KPR_DEV bool holds(int, int) { return true; }

    // bin held by register `reg` of lane l after run()
    static KPR_DEV int bin(int l, int reg) { return l + L * (reg / RL) + (N / RL) * (reg % RL); }

    // the pass-1 twiddles W_N^{l k1} of lane l (k1 = 1 .. P-1 in tw1[k1]; tw1[0] unused): a persistent kernel keeps
    // them in registers instead of re-reading the LDS table for every frame
    static KPR_DEV void load_tw1(f2 (&tw1)[P], int l, const f2* tab) {
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) tw1[k1] = tab[2 * l * k1];
    }
    static KPR_DEV void run(f2 (&z)[P], int l, bool active, f2* row, const f2* tab) {
        f2 tw1[P];
        load_tw1(tw1, l, tab);
        run(z, l, active, row, tab, tw1);
    }
    static KPR_DEV void run(f2 (&z)[P], int l, bool active, f2* row, const f2* tab, const f2 (&tw1)[P]) {
        Dft<P>::run(z);
#pragma unroll
        for (int k1 = 1; k1 < P; ++k1) z[k1] = cmul(z[k1], tw1[k1]);               // W_N^{l k1}
        KPR_LDS_FENCE_W();                   // (kpr_fft.h: no access of the compiler crosses a fence of a wave-private hand-over)
        if (active) {
#pragma unroll
            for (int k1 = 0; k1 < P; ++k1) row[l + L * k1] = z[k1];
        }
        KPR_LDS_FENCE_R();
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
            KPR_LDS_FENCE_W();
            if (active) {
#pragma unroll
                for (int j = 0; j < Q2; ++j)
#pragma unroll
                    for (int kb = 0; kb < R2; ++kb) row[(bp + R2 * j + P * kb) + P * R2 * a] = z[j * R2 + kb];
            }
            KPR_LDS_FENCE_R();
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
        KPR_LDS_FENCE_X();
    }
};

// Two-pass ("four-step") FFT for N = N1 * N2 with DFT-N1 and DFT-N2 both small enough for registers:
//   pass 1: lane l < N2 holds x[l + N2 m], m < N1: DFT-N1 over m, times W_N^{l k1}
//   exchange: item (l, k1) at row index l + N2P k1 (N2P = N2 | 1: odd stride, conflict-free reads)
//   pass 2: lane l1 < N1 reads the N2 items of k1 = l1: DFT-N2 over l -> register k2 holds X[l1 + N1 k2]
// The lane count changes between the passes (N2, then N1): L = max(N1, N2) lanes per frame, the
// surplus lanes of a pass idle.  Used for the transform sizes with a factor 3 (n_fft = 96 ... 960),
// where no single register count divides into every pass radix.
template <int N1, int N2>
struct TwoPassFft {
    static constexpr int N = N1 * N2, P = (N1 > N2) ? N1 : N2, L = P;
    static constexpr int N2P = N2 | 1;
    static constexpr int PIN = N1, LIN = N2, ROW = N2P * N1;
    static KPR_DEV int bin(int l, int reg) { return l + N1 * reg; }
    static KPR_DEV bool holds(int l, int reg) { return l < N1 && reg < N2; }

    static KPR_DEV void run(f2 (&z)[P], int l, bool active, f2* row, const f2* tab) {
        {
            f2 t[N1];
#pragma unroll
            for (int m = 0; m < N1; ++m) t[m] = z[m];
            Dft<N1>::run(t);
#pragma unroll
            for (int k1 = 1; k1 < N1; ++k1) t[k1] = cmul(t[k1], tab[2 * min(l, N2 - 1) * k1]);   // W_N^{l k1}
            KPR_LDS_FENCE_W();
            if (active && l < N2) {
#pragma unroll
                for (int k1 = 0; k1 < N1; ++k1) row[l + N2P * k1] = t[k1];
            }
            KPR_LDS_FENCE_R();
        }
        {
            f2 t[N2];
            const int l1 = min(l, N1 - 1);
#pragma unroll
            for (int l2 = 0; l2 < N2; ++l2) t[l2] = row[l2 + N2P * l1];
            KPR_LDS_FENCE_X();
            Dft<N2>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) z[k2] = t[k2];
        }
    }
};

}  // namespace kpr
