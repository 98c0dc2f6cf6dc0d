// kpr_fft32.h -- 1024-point complex FFT with 32 points per lane (k_mel_ws<1024, ..., P32>): a frame is owned by 32 lanes
// x 32 register slots, radices (32, 32), ONE LDS exchange per frame (a 32 x 32 transpose, lane <-> slot) instead of the
// two exchanges of the 16-points-per-lane form; a wave transforms two frames at once (lane groups 0-31 and 32-63).
// oracle/proto_stockham.py::p32_* is the numpy model (tests/test_proto_stockham.py).
// Part of the single translation unit kapre_hip.hip (included there, after kpr_fft.h / kpr_common.h).
#pragma once

namespace kpr {

constexpr int kPts32 = 32;

// cos / sin of 2 pi m / 64, m = 0 .. 16 (first quadrant)
__host__ __device__ constexpr float q64(int m) {
    constexpr float t[17] = {1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f,
                             0.95694033573220882438f, 0.92387953251128673848f, 0.88192126434835504956f,
                             0.83146961230254523567f, 0.77301045336273699338f, 0.70710678118654757274f,
                             0.63439328416364548779f, 0.55557023301960228867f, 0.47139673682599780857f,
                             0.38268343236508983729f, 0.29028467725446233105f, 0.19509032201612833135f,
                             0.09801714032956077016f, 0.0f};
    return t[m];
}
// x * exp(-2 pi i m / 64), 0 <= m <= 16 (compile-time m after unrolling)
KPR_DEV f2 cmul_w64(f2 x, int m) {
    if (m == 0) return x;
    if (m == 16) return f2{x.y, -x.x};      // -i
    return cmul_s(x, f2{q64(m), -q64(16 - m)});
}

// DFT-32, natural order in / out: decimation in time over two DFT-16
template <> struct Dft<32> {
    static KPR_DEV void run(f2 (&v)[32]) {
        f2 e[16], o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
        Dft<16>::run(e);
        Dft<16>::run(o);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f2 t = cmul_w32(o[k], k);
            v[k] = cadd(e[k], t);
            v[k + 16] = csub(e[k], t);
        }
    }
};

// per-lane twiddle factors of the second pass, w_1024^{r fl} = hi[r >> 3] * lo[r & 7], and of the pairing
struct Tw32 {
    f2 lo[7];      // w_1024^{b fl}, b = 1 .. 7
    f2 hi[3];      // w_1024^{8 a fl}, a = 1 .. 3
    f2 pp;         // w_2048^{fl}
    KPR_DEV void load(const float2* __restrict__ table, int fl) {      // table[j] = exp(-2 pi i j / 2048)
#pragma unroll
        for (int b = 1; b <= 7; ++b) { const float2 w = table[(2 * b * fl) & 2047]; lo[b - 1] = f2{w.x, w.y}; }
#pragma unroll
        for (int a = 1; a <= 3; ++a) { const float2 w = table[(16 * a * fl) & 2047]; hi[a - 1] = f2{w.x, w.y}; }
        const float2 w = table[fl];
        pp = f2{w.x, w.y};
    }
};

// raw samples of one frame, 32 points per lane: z[m] = (x[2n], x[2n+1]), n = fl + 32 m.  Returns the validity masks
// (bit 2m / 2m+1 of vm[m >> 4] for slot m & 15); see fetch_frame() for the clamped-address scheme.
KPR_DEV void fetch_frame32(const float* __restrict__ x, const Geom& g, const FramePos& p, bool valid, int fl,
                           f2 (&z)[kPts32], unsigned (&vm)[2]) {
    constexpr int L = 32, NC = 1024;
    const float* sig = x + p.sig_off;
    const bool interior = valid && p.s0 >= 0 && (p.s0 + 2 * NC) <= g.T && g.win >= 2 * NC;
    if (interior && p.es == 1) {
        const float* fp = sig + p.s0;
        if ((((unsigned long long)fp) & 7ull) == 0) {
            const float2* fp2 = reinterpret_cast<const float2*>(fp) + fl;
#pragma unroll
            for (int m = 0; m < kPts32; ++m) {
                const float2 v = fp2[L * m];
                z[m] = f2{v.x, v.y};
            }
        } else {
#pragma unroll
            for (int m = 0; m < kPts32; ++m) {
                const int n = 2 * (fl + L * m);
                z[m] = f2{fp[n], fp[n + 1]};
            }
        }
        vm[0] = vm[1] = 0xffffffffu;
        return;
    }
    vm[0] = vm[1] = 0;
    const int es = p.es, omax = (int)(g.T - 1) * es;
    const int o_base = ((int)p.s0 + 2 * fl) * es;
#pragma unroll
    for (int m = 0; m < kPts32; ++m) {
        const int n = 2 * (fl + L * m);
        const int o0 = o_base + m * (2 * L) * es, o1 = o0 + es;
        z[m] = f2{sig[min(max(o0, 0), omax)], sig[min(max(o1, 0), omax)]};
        vm[m >> 4] |= (valid && n < g.win && (unsigned)o0 <= (unsigned)omax) ? (1u << (2 * (m & 15))) : 0u;
        vm[m >> 4] |= (valid && n + 1 < g.win && (unsigned)o1 <= (unsigned)omax) ? (2u << (2 * (m & 15))) : 0u;
        if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

KPR_DEV void mask_frame32(f2 (&z)[kPts32], const unsigned (&vm)[2]) {
    if (__all(vm[0] == 0xffffffffu && vm[1] == 0xffffffffu)) return;      // wave-uniform: interior frames
#pragma unroll
    for (int m = 0; m < kPts32; ++m) {
        const unsigned w = vm[m >> 4];
        const unsigned kx = (unsigned)(-(int)((w >> (2 * (m & 15))) & 1u));
        const unsigned ky = (unsigned)(-(int)((w >> (2 * (m & 15) + 1)) & 1u));
        z[m] = f2{__uint_as_float(__float_as_uint(z[m].x) & kx), __uint_as_float(__float_as_uint(z[m].y) & ky)};
    }
}

// the exchange between the two passes: writer lane fl holds outputs r = 0 .. 31 (index 32 fl + r), reader lane g slot m
// wants index g + 32 m, i.e. (writer m, output g).  Row layout: element (writer w, output r) at w * 33 + r -- a lane
// stores its 32 values contiguously (ds_write2_b32 pairs off ONE base register; consecutive lanes 33 words apart:
// conflict free), and the reads of slot m by consecutive lanes are consecutive words.  re and im in two rounds (the row
// is the frame's magnitude row: 1056 of its 1058 words).
template <int I>
KPR_DEV void x32_store(const f2 (&z)[kPts32], unsigned wa, int c) {
    if constexpr (I < 16) {
        if (c == 0)
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4"
                         :: "v"(wa), "v"(z[2 * I].x), "v"(z[2 * I + 1].x), "n"(2 * I), "n"(2 * I + 1) : "memory");
        else
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4"
                         :: "v"(wa), "v"(z[2 * I].y), "v"(z[2 * I + 1].y), "n"(2 * I), "n"(2 * I + 1) : "memory");
        x32_store<I + 1>(z, wa, c);
    }
}

KPR_DEV void exchange32(f2 (&z)[kPts32], float* row, int fl) {
    const unsigned wa = (unsigned)(size_t)(row + 33 * fl);
    const float* rd = row + fl;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // (by hand: hipcc does not merge dword stores whose data are halves of 64-bit registers)
        x32_store<0>(z, wa, c);
        if (c == 0) {
            // the x halves of z are dead once their stores are issued (LDS executes a wave's operations in order)
#pragma unroll
            for (int m = 0; m < kPts32; ++m) z[m].x = rd[33 * m];
        } else {
#pragma unroll
            for (int m = 0; m < kPts32; ++m) z[m].y = rd[33 * m];
        }
    }
}

// forward 1024-point FFT of the frame in z (layout fl + 32 m in and out)
KPR_DEV void cfft32_forward(f2 (&z)[kPts32], const Tw32& tw, float* row, int fl) {
    Dft<32>::run(z);                                   // pass 1: output index 32 fl + r
    exchange32(z, row, fl);
#pragma unroll
    for (int r = 1; r < 32; ++r) {                     // pass 2: NS = 32, kk = fl: w_1024^{r fl}
        const int a = r >> 3, b = r & 7;
        if (b) z[r] = cmul(z[r], tw.lo[b - 1]);
        if (a) z[r] = cmul(z[r], tw.hi[a - 1]);
    }
    Dft<32>::run(z);                                   // output index fl + 32 r: natural layout
}

// real-FFT pairing, 32 points per lane (see rfft_pair): lane fl evaluates slots m = 0 .. 15 (k = fl + 32 m < 512) and
// emits X[k] and X[1024 - k]; the partner Z[1024 - k] is slot 31 - m of lane (32 - fl) % 32 of the same lane group
// (lane 0: its own slot (32 - m) % 32); k = 512 is slot 16 of lane 0, self-paired.
template <class Emit>
KPR_DEV void rfft_pair32(const f2 (&z)[kPts32], const Tw32& tw, int fl, int lane, Emit&& emit) {
    constexpr int L = 32, NC = 1024;
    const int src = (lane - fl) + ((L - fl) & (L - 1));
    const f2 ppmi = f2{tw.pp.y, -tw.pp.x};            // -i * w_2048^{fl}
    f2 zq[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        zq[m].x = __shfl(z[31 - m].x, src, 64);
        zq[m].y = __shfl(z[31 - m].y, src, 64);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        f2 zp = zq[m];
        if (fl == 0) zp = z[(32 - m) & 31];
        const f2 e = cadd_conj(z[m], zp);
        const f2 t = cmul(cmul_w64(csub_conj(z[m], zp), m), ppmi);
        const f2 xk = cadd(e, t);
        const f2 xm = csub(e, t);
        const int k = fl + L * m;
        emit(k, xk, NC - k, f2{xm.x, -xm.y});
    }
    if (fl == 0) {
        const f2 zz = z[16];
        const f2 e = cadd_conj(zz, zz);
        const f2 t = cmul(cmul_w64(csub_conj(zz, zz), 16), ppmi);
        emit(NC / 2, cadd(e, t), -1, f2{0.0f, 0.0f});
    }
}

}  // namespace kpr
