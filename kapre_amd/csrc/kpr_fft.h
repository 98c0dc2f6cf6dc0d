// kpr_fft.h -- wave64 register/LDS Stockham FFT building blocks for gfx950 (CDNA4).
//
// A real n_fft-point transform is an NC = n_fft/2 point complex FFT of z[n] = x[2n] + i x[2n+1]
// plus a pairing pass.  One frame is owned by L = NC/16 lanes of a wave (G = 64/L frames per
// wave), each lane holds 16 complex points in registers, always in the "lane + L*m" layout
// (m = register slot).  Passes use radices <= 16 computed entirely in registers; between passes
// values cross lanes through an NC-word LDS row (re and im in two rounds so that the row that
// later receives the frame's magnitudes is big enough) with the bank swizzle
// swz(e) = e ^ ((e >> 4) & 31) -- conflict free for n_fft = 1024 / 2048.  swz is GF(2)-linear, so
// swz(lane_part + const_part) = swz(lane_part) ^ swz(const_part) whenever the two parts occupy
// disjoint bits: every LDS address is ONE v_xor of a per-lane register with a literal.
// (oracle/proto_stockham.py is the index-for-index numpy model; tests/test_proto_stockham.py.)
//
// Register budget: twiddles are kept factored (per-lane base values x compile-time roots of
// unity) so that window + twiddles + data stay well under the 256-VGPR budget of 2 waves/SIMD.
//
// Arithmetic this replaces: the rfft / irfft inside tf.signal.stft / tf.signal.inverse_stft as
// called from /root/reference/kapre/time_frequency.py:174-182 and :307-314.
#pragma once
#include <hip/hip_runtime.h>

#define KPR_DEV __device__ __forceinline__

namespace kpr {

constexpr int kPts = 16;  // complex points per lane

template <int NC> struct Radix;  // pass radices, product == NC
template <> struct Radix<128>  { static constexpr int r1 = 16, r2 = 8,  r3 = 1; };
template <> struct Radix<256>  { static constexpr int r1 = 16, r2 = 16, r3 = 1; };
template <> struct Radix<512>  { static constexpr int r1 = 16, r2 = 16, r3 = 2; };
template <> struct Radix<1024> { static constexpr int r1 = 16, r2 = 16, r3 = 4; };

__host__ __device__ constexpr int swz(int e) { return e ^ ((e >> 4) & 31); }

// cos / sin of 2*pi*m/32, m = 0..8 (first quadrant); everything else by symmetry
__host__ __device__ constexpr float q32(int m) {
    constexpr float t[9] = {1.0f,
                            0.98078528040323044913f,
                            0.92387953251128675613f,
                            0.83146961230254523708f,
                            0.70710678118654752440f,
                            0.55557023301960222474f,
                            0.38268343236508977173f,
                            0.19509032201612826785f,
                            0.0f};
    return t[m];
}
__host__ __device__ constexpr float cos32(int m) {   // cos(2 pi m / 32), any m >= 0
    m &= 31;
    if (m > 16) m = 32 - m;
    return (m <= 8) ? q32(m) : -q32(16 - m);
}
__host__ __device__ constexpr float sin32(int m) {   // sin(2 pi m / 32)
    m &= 31;
    return (m <= 16) ? ((m <= 8) ? q32(8 - m) : q32(m - 8)) : -((32 - m <= 8) ? q32(8 - (32 - m)) : q32((32 - m) - 8));
}

KPR_DEV void cmul(float& xr, float& xi, float wr, float wi) {
    float tr = xr * wr - xi * wi;
    xi = xr * wi + xi * wr;
    xr = tr;
}

// multiply by the compile-time root of unity w32^m = exp(-2 pi i m / 32); after unrolling m is a
// constant and the trivial cases fold away
KPR_DEV void cmul_w32(float& xr, float& xi, int m) {
    m &= 31;
    if (m == 0) return;
    if (m == 8)  { float t = xr; xr = xi;  xi = -t; return; }   // -i
    if (m == 16) { xr = -xr; xi = -xi; return; }
    if (m == 24) { float t = xr; xr = -xi; xi = t;  return; }   // +i
    cmul(xr, xi, cos32(m), -sin32(m));
}

// ---- small forward DFTs (e^{-2 pi i rs/R}), natural order, in registers --------------------
template <int R> struct Dft;

template <> struct Dft<2> {
    static KPR_DEV void run(float (&re)[2], float (&im)[2]) {
        float ar = re[0], ai = im[0];
        re[0] = ar + re[1]; im[0] = ai + im[1];
        re[1] = ar - re[1]; im[1] = ai - im[1];
    }
};

KPR_DEV void dft4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3,
                  float& i3) {
    float t0r = r0 + r2, t0i = i0 + i2;
    float t1r = r0 - r2, t1i = i0 - i2;
    float t2r = r1 + r3, t2i = i1 + i3;
    float dr = r1 - r3, di = i1 - i3;   // t3 = -i * d = (di, -dr)
    r0 = t0r + t2r; i0 = t0i + t2i;
    r1 = t1r + di;  i1 = t1i - dr;
    r2 = t0r - t2r; i2 = t0i - t2i;
    r3 = t1r - di;  i3 = t1i + dr;
}

template <> struct Dft<4> {
    static KPR_DEV void run(float (&re)[4], float (&im)[4]) {
        dft4(re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3]);
    }
};

template <> struct Dft<8> {
    // s = 4*n1 + n2 (n1 in {0,1}), r = k1 + 2*k2
    static KPR_DEV void run(float (&re)[8], float (&im)[8]) {
        float yr[4][2], yi[4][2];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            yr[n2][0] = re[n2] + re[n2 + 4]; yi[n2][0] = im[n2] + im[n2 + 4];
            yr[n2][1] = re[n2] - re[n2 + 4]; yi[n2][1] = im[n2] - im[n2 + 4];
            cmul_w32(yr[n2][1], yi[n2][1], 4 * n2);            // w8^{n2}
        }
#pragma unroll
        for (int k1 = 0; k1 < 2; ++k1) {
            dft4(yr[0][k1], yi[0][k1], yr[1][k1], yi[1][k1], yr[2][k1], yi[2][k1], yr[3][k1],
                 yi[3][k1]);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) { re[k1 + 2 * k2] = yr[k2][k1]; im[k1 + 2 * k2] = yi[k2][k1]; }
        }
    }
};

template <> struct Dft<16> {
    // s = 4*n1 + n2, r = k1 + 4*k2: DFT4 over n1, twiddle w16^{n2 k1}, DFT4 over n2
    static KPR_DEV void run(float (&re)[16], float (&im)[16]) {
        float yr[4][4], yi[4][4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            float a0r = re[n2], a0i = im[n2], a1r = re[4 + n2], a1i = im[4 + n2];
            float a2r = re[8 + n2], a2i = im[8 + n2], a3r = re[12 + n2], a3i = im[12 + n2];
            dft4(a0r, a0i, a1r, a1i, a2r, a2i, a3r, a3i);
            yr[n2][0] = a0r; yi[n2][0] = a0i; yr[n2][1] = a1r; yi[n2][1] = a1i;
            yr[n2][2] = a2r; yi[n2][2] = a2i; yr[n2][3] = a3r; yi[n2][3] = a3i;
#pragma unroll
            for (int k1 = 1; k1 < 4; ++k1) cmul_w32(yr[n2][k1], yi[n2][k1], 2 * n2 * k1);  // w16^{n2 k1}
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            dft4(yr[0][k1], yi[0][k1], yr[1][k1], yi[1][k1], yr[2][k1], yi[2][k1], yr[3][k1],
                 yi[3][k1]);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) { re[k1 + 4 * k2] = yr[k2][k1]; im[k1 + 4 * k2] = yi[k2][k1]; }
        }
    }
};

// ---- per-lane state: factored twiddles + swizzled LDS address bases --------------------------
// table[j] = exp(-2 pi i j / n_fft), j in [0, n_fft)
template <int NC>
struct FftTw {
    static constexpr int L = NC / kPts;
    static constexpr int NFFT = 2 * NC;
    static constexpr int R1 = Radix<NC>::r1, R2 = Radix<NC>::r2, R3 = Radix<NC>::r3;
    static constexpr int Q2 = kPts / R2;
    static constexpr int H2 = R2 / 4 - 1;            // "high digit" factors of pass 2 (r = 4, 8, 12)
    // pass 2 (NS = R1): w_{R1 R2}^{r kk}, kk = (fl + L q) mod R1;  r = 4a + b -> hi[a-1] * lo[b-1]
    float p2lo_r[Q2][3], p2lo_i[Q2][3];
    float p2hi_r[Q2][H2 > 0 ? H2 : 1], p2hi_i[Q2][H2 > 0 ? H2 : 1];
    // pass 3 (NS = R1 R2): w_NC^{r (fl + L q)} = w_NC^{r fl} * w16^{r q}
    float p3_r[R3 > 1 ? R3 - 1 : 1], p3_i[R3 > 1 ? R3 - 1 : 1];
    // pairing: w_NFFT^{fl + L m} = w_NFFT^{fl} * w32^{m}
    float pp_r, pp_i;
    // LDS address bases (word units, already swizzled)
    int a_rd;        // swz(fl)
    int a_w1;        // swz(lane part of pass-1 output index)
    int a_w2;        // swz(lane part of pass-2 output index)

    static KPR_DEV int lane_base(int fl, int NS, int R) {
        // expand(t) = (t / NS) * NS * R + t % NS with t = fl (the q part is a compile-time term)
        return (fl / NS) * (NS * R) + (fl % NS);
    }

    KPR_DEV void load(const float2* __restrict__ table, int fl) {
#pragma unroll
        for (int q = 0; q < Q2; ++q) {
            const int kk = (fl + L * q) & (R1 - 1);
            constexpr int step = NFFT / (R1 * R2);
#pragma unroll
            for (int b = 1; b <= 3; ++b) {
                float2 w = table[(b * kk * step) & (NFFT - 1)];
                p2lo_r[q][b - 1] = w.x; p2lo_i[q][b - 1] = w.y;
            }
#pragma unroll
            for (int a = 1; a <= H2; ++a) {
                float2 w = table[(4 * a * kk * step) & (NFFT - 1)];
                p2hi_r[q][a - 1] = w.x; p2hi_i[q][a - 1] = w.y;
            }
        }
        if constexpr (R3 > 1) {
#pragma unroll
            for (int r = 1; r < R3; ++r) {
                float2 w = table[(r * fl * 2) & (NFFT - 1)];
                p3_r[r - 1] = w.x; p3_i[r - 1] = w.y;
            }
        }
        {
            float2 w = table[fl];
            pp_r = w.x; pp_i = w.y;
        }
        a_rd = swz(fl);
        a_w1 = swz(lane_base(fl, 1, R1));
        a_w2 = swz(lane_base(fl, R1, R2));
    }
};

// One Stockham pass: radix R, NS = product of earlier radices, PASS = 1, 2 or 3.
// `row` is this lane's frame's NC-word LDS exchange row.  Mirrors complex_fft_lanes() in
// oracle/proto_stockham.py.
template <int NC, int PASS, int R, int NS>
KPR_DEV void fft_pass(float (&re)[kPts], float (&im)[kPts], const FftTw<NC>& tw, float* row) {
    constexpr int L = NC / kPts;
    constexpr int Q = kPts / R;
    constexpr bool LAST = (NS * R == NC);
    static_assert(L >= NS || LAST, "lane/const bit split needs L >= NS");
    float outr[kPts], outi[kPts];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        float vr[R], vi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { vr[r] = re[q + Q * r]; vi[r] = im[q + Q * r]; }
        if constexpr (PASS == 2) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const int a = r >> 2, b = r & 3;
                if (b) cmul(vr[r], vi[r], tw.p2lo_r[q][b - 1], tw.p2lo_i[q][b - 1]);
                if (a) cmul(vr[r], vi[r], tw.p2hi_r[q][a - 1], tw.p2hi_i[q][a - 1]);
            }
        } else if constexpr (PASS == 3) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                cmul(vr[r], vi[r], tw.p3_r[r - 1], tw.p3_i[r - 1]);
                cmul_w32(vr[r], vi[r], 2 * r * q);               // w16^{r q}
            }
        }
        Dft<R>::run(vr, vi);
#pragma unroll
        for (int r = 0; r < R; ++r) { outr[q + Q * r] = vr[r]; outi[q + Q * r] = vi[r]; }
    }
    if constexpr (LAST) {
#pragma unroll
        for (int m = 0; m < kPts; ++m) { re[m] = outr[m]; im[m] = outi[m]; }
    } else {
        const int aw = (PASS == 1) ? tw.a_w1 : tw.a_w2;
        // output index = expand(fl + L q) + NS r = lane_base(fl) + [L R q + NS r]
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) row[aw ^ swz(L * R * q + NS * r)] = outr[q + Q * r];
#pragma unroll
        for (int m = 0; m < kPts; ++m) re[m] = row[tw.a_rd ^ swz(L * m)];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) row[aw ^ swz(L * R * q + NS * r)] = outi[q + Q * r];
#pragma unroll
        for (int m = 0; m < kPts; ++m) im[m] = row[tw.a_rd ^ swz(L * m)];
    }
}

// Forward NC-point complex FFT, in / out layout "fl + L*m".
template <int NC>
KPR_DEV void cfft_forward(float (&re)[kPts], float (&im)[kPts], const FftTw<NC>& tw, float* row) {
    using Rx = Radix<NC>;
    fft_pass<NC, 1, Rx::r1, 1>(re, im, tw, row);
    fft_pass<NC, 2, Rx::r2, Rx::r1>(re, im, tw, row);
    if constexpr (Rx::r3 > 1) fft_pass<NC, 3, Rx::r3, Rx::r1 * Rx::r2>(re, im, tw, row);
}

// Pairing pass of the real FFT: Z (complex FFT of the packed frame) -> X[k], k = fl + L*m.
// Partner bin NC-k lives in lane (L-fl)%L slot 15-m; lane 0 pairs with its own slot (16-m)%16.
// nyq receives X[NC] (real) and is valid on lanes with fl == 0 only.
template <int NC>
KPR_DEV void rfft_pair(float (&re)[kPts], float (&im)[kPts], const FftTw<NC>& tw, int fl,
                       int lane, float& nyq) {
    constexpr int L = NC / kPts;
    const int src = (lane - fl) + ((L - fl) & (L - 1));
    float xr[kPts], xi[kPts];
#pragma unroll
    for (int m = 0; m < kPts; ++m) {
        float zpr = __shfl(re[kPts - 1 - m], src, 64);
        float zpi = __shfl(im[kPts - 1 - m], src, 64);
        if (fl == 0) { zpr = re[(kPts - m) & (kPts - 1)]; zpi = im[(kPts - m) & (kPts - 1)]; }
        float zr = re[m], zi = im[m];
        float er = 0.5f * (zr + zpr), ei = 0.5f * (zi - zpi);
        float orr = 0.5f * (zi + zpi), oi = -0.5f * (zr - zpr);
        // w = w_NFFT^{fl} * w32^{m}
        cmul_w32(orr, oi, m);
        cmul(orr, oi, tw.pp_r, tw.pp_i);
        xr[m] = er + orr;
        xi[m] = ei + oi;
    }
    nyq = re[0] - im[0];
#pragma unroll
    for (int m = 0; m < kPts; ++m) { re[m] = xr[m]; im[m] = xi[m]; }
}

// Inverse pairing: X[k] and X[NC-k] (k = fl + L m) -> conj(2 Z[k]), ready for cfft_forward;
// the caller conjugates again after the FFT (IFFT(z) = conj(FFT(conj z))).
template <int NC>
KPR_DEV void irfft_pair_one(float xkr, float xki, float xpr, float xpi, const FftTw<NC>& tw,
                            int m, float& zr, float& zi) {
    float er = xkr + xpr, ei = xki - xpi;
    float dr = xkr - xpr, di = xki + xpi;
    // o = d * conj(w), w = w_NFFT^{fl} * w32^{m}
    di = -di;                       // conj(d)
    cmul_w32(dr, di, m);
    cmul(dr, di, tw.pp_r, tw.pp_i); // conj(d) * w = conj(d * conj(w))
    float orr = dr, oi = -di;       // o
    zr = er - oi;
    zi = -(ei + orr);
}

}  // namespace kpr
