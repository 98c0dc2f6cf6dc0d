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
// Arithmetic is PACKED: a complex value is one 64-bit VGPR pair and every complex add / sub /
// (x +- i*y) is ONE v_pk_add_f32 (op_sel / neg modifiers do the swap and the signs), a complex
// multiply is v_pk_mul_f32 + v_pk_fma_f32.  Measured on MI355X a wave issues one VALU instruction
// per ~4 cycles, so per-frame time is proportional to the instruction count (rocprofv3:
// SQ_ACTIVE_INST_VALU ~= SQ_INSTS_VALU quad-cycles); packing halves it.
//
// Register budget: twiddles are kept factored (per-lane base values x compile-time roots of
// unity held in SGPR pairs) so that window + twiddles + data stay under 256 VGPRs (2 waves/SIMD).
//
// Arithmetic this replaces: the rfft / irfft inside tf.signal.stft / tf.signal.inverse_stft as
// called from /root/reference/kapre/time_frequency.py:174-182 and :307-314.
#pragma once
#include <hip/hip_runtime.h>

#define KPR_DEV __device__ __forceinline__

namespace kpr {

constexpr int kPts = 16;  // complex points per lane

typedef float f2 __attribute__((ext_vector_type(2)));   // (re, im) in one VGPR pair
typedef float f4 __attribute__((ext_vector_type(4)));   // a register quad (128-bit LDS accesses)

// ---- lds_wave_fence: ordering of a wave-private LDS hand-over ------------------------------------------------------
// Lanes of ONE wave hand words to each other through an LDS row without s_barrier and without s_waitcnt: the LDS executes
// a wave's instructions in issue order, so a group of stores followed by loads of OTHER lanes' words is correct in
// hardware.  The compiler, however, only honours the per-thread memory model: it may move a store above a load (or a load
// above a store) of the SAME lane whenever it proves the two addresses of that lane different.  That is what made
// k_istft_pw<512, 2> wrong in round 4: hipcc hoisted the first ds_write2_b32 of the exchange's second component above the
// last ds_read_b32 of the first -- disjoint for every single lane, but lanes 30 / 31 overwrote the words lanes 0, 1 / 16, 17
// still had to read (profiles/r05_hazard_rootcause.md: reproduced, bisected and repaired at the ISA level).
// The fence is an empty asm statement with a "memory" clobber: no instruction, vector-ALU work still moves across it, but no
// memory access of the compiler crosses it.  Rule: one fence between every group of stores and the loads that read other
// lanes' words, and one between those loads and the next stores to the same row.  The marker comment lands in the ISA:
// tests/test_asm_audit.py checks for every kernel that no LDS store sits in a region opened by an "R" fence and no LDS load
// in a region opened by a "W" fence ("X" closes a region without opening one).
#define KPR_LDS_FENCE_W() asm volatile("; kpr_lds_fence W" ::: "memory")   /* what follows: stores of this lane's words   */
#define KPR_LDS_FENCE_R() asm volatile("; kpr_lds_fence R" ::: "memory")   /* what follows: loads of other lanes' words   */
#define KPR_LDS_FENCE_X() asm volatile("; kpr_lds_fence X" ::: "memory")   /* end of the hand-over                        */

template <int NC> struct Radix;  // pass radices, product == NC
template <> struct Radix<128>  { static constexpr int r1 = 16, r2 = 8,  r3 = 1; };
template <> struct Radix<256>  { static constexpr int r1 = 16, r2 = 16, r3 = 1; };
template <> struct Radix<512>  { static constexpr int r1 = 16, r2 = 16, r3 = 2; };
template <> struct Radix<1024> { static constexpr int r1 = 16, r2 = 16, r3 = 4; };

__host__ __device__ constexpr int swz(int e) { return e ^ ((e >> 4) & 31); }

// LDS index policies for the exchange rows.  An exchange index is always lane_part + const_part
// (disjoint bit fields); a policy maps it to a bank-conflict-free word offset in the row.
//   SwzXor  : swz(lane) ^ swz(const).  Row = NC words.  Every address costs a v_xor + v_lshl_add.
//   SwzSkew : additive skew, sk(lane) + sk(const) -- the constant part becomes the IMMEDIATE
//             offset of the DS instruction (no VALU work per address, neighbouring accesses merge
//             into ds_read2/ds_write2).  Exchange 1 uses e + (e>>5), exchange 2 adds 8*(e>>8);
//             both are additive and conflict free for NC = 1024 (16,16,4) and NC = 512 (16,16,2)
//             (tests/test_proto_stockham.py checks both properties for every lane).
//             Row = NC + NC/32 + 24 words.
struct SwzXor {
    template <int PASS> static constexpr int lane(int e) { return swz(e); }
    template <int PASS> static KPR_DEV int at(int lane_part, int c) { return lane_part ^ swz(c); }
    static constexpr bool kXor = true;
};
struct SwzSkew {
    template <int PASS> static constexpr int sk(int e) {
        return PASS == 1 ? e + (e >> 5) : e + (e >> 5) + 8 * (e >> 8);
    }
    template <int PASS> static constexpr int lane(int e) { return sk<PASS>(e); }
    template <int PASS> static KPR_DEV int at(int lane_part, int c) { return lane_part + sk<PASS>(c); }
    static constexpr bool kXor = false;
    static constexpr int row_words(int NC) { return NC + NC / 32 + 24; }
};
//   SwzWide : NC = 1024 only, 128-bit accesses (wide_chunk() below; oracle/proto_stockham.py wide_*).  A wave can
//             keep 15 LDS instructions in flight (lgkmcnt), so the cost of an exchange is its instruction count:
//             40 per frame here against 128 with one dword per access.  The row must start on a 16-byte boundary.
struct SwzWide {
    static constexpr bool kXor = false;
    static constexpr int row_words(int NC) { return NC + 4; }       // + slack to align the row start
};
template <class SW> struct IsWide { static constexpr bool value = false; };
template <> struct IsWide<SwzWide> { static constexpr bool value = true; };

// 16-byte chunk p (0..255) of the exchange row holds register slots 4j .. 4j+3 of reader lane g
__host__ __device__ constexpr int wide_chunk(int g, int j) {
    return ((g & 15) ^ (2 * (g >> 5) + 4 * (j & 1))) + 16 * (((g >> 4) & 1) + 2 * (g >> 5) + 4 * (j & 1) + 8 * (j >> 1));
}
__host__ __device__ constexpr int wide_addr(int e) {          // word offset of exchange index e = g + 64 m
    return 4 * wide_chunk(e & 63, e >> 8) + ((e >> 6) & 3);
}
typedef f4 f4a __attribute__((may_alias, aligned(16)));
typedef f2 f2a __attribute__((may_alias, aligned(8)));

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

#ifdef KPR_FFT_PLAIN   /* development (tools/build_variant.py x -DKPR_FFT_PLAIN): the same primitives in plain C++ -- hipcc picks the
                          instructions: headline 48.6-49.1 vs 45.2 us (cold clocks), cfg5 262-265 vs 214-216, cfg4 STFT 99 vs 71
                          (profiles/r04_probes/plain_cpp_primitives_and_nofence.log) */
KPR_DEV f2 cadd(f2 a, f2 b) { return a + b; }
KPR_DEV f2 csub(f2 a, f2 b) { return a - b; }
KPR_DEV f2 cadd_mi(f2 a, f2 b) { return f2{a.x + b.y, a.y - b.x}; }
KPR_DEV f2 cadd_pi(f2 a, f2 b) { return f2{a.x - b.y, a.y + b.x}; }
KPR_DEV f2 cadd_conj(f2 a, f2 b) { return f2{a.x + b.x, a.y - b.y}; }
KPR_DEV f2 csub_conj(f2 a, f2 b) { return f2{a.x - b.x, a.y + b.y}; }
KPR_DEV f2 cmul(f2 a, f2 w) { return f2{fmaf(-a.y, w.y, a.x * w.x), fmaf(a.x, w.y, a.y * w.x)}; }        // (same rounding order as the asm)
KPR_DEV f2 cmul_conj(f2 a, f2 w) { return f2{fmaf(a.y, w.y, a.x * w.x), fmaf(-a.x, w.y, a.y * w.x)}; }
KPR_DEV f2 cmul_s(f2 a, f2 w) { return cmul(a, w); }
KPR_DEV f2 pmul(f2 a, f2 w) { return a * w; }
#else
// ---- packed complex primitives (VOP3P; op_sel[i] / op_sel_hi[i] pick the half of source i that
// feeds the low / high result, neg_lo / neg_hi negate it) ------------------------------------
KPR_DEV f2 cadd(f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
KPR_DEV f2 csub(f2 a, f2 b) {
    f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// a + (-i) b = (a.x + b.y, a.y - b.x)
KPR_DEV f2 cadd_mi(f2 a, f2 b) {
    f2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
KPR_DEV f2 cadd_pi(f2 a, f2 b) {
    f2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// a + conj(b), a - conj(b)
KPR_DEV f2 cadd_conj(f2 a, f2 b) {
    f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
KPR_DEV f2 csub_conj(f2 a, f2 b) {
    f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// a * w (complex), w in VGPRs
KPR_DEV f2 cmul(f2 a, f2 w) {
    f2 r;     // one block = one hipcc boundary pad instead of two (no VALU->VALU hazard inside)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=&v"(r) : "v"(a), "v"(w));
    return r;
}
// a * conj(w)
KPR_DEV f2 cmul_conj(f2 a, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * w with w a compile-time constant kept in an SGPR pair
KPR_DEV f2 cmul_s(f2 a, f2 w) {
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=&v"(r) : "v"(a), "s"(w));
    return r;
}
// a * (wr, wi) elementwise (window)
KPR_DEV f2 pmul(f2 a, f2 w) { f2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(w)); return r; }

#endif

// "Planar" operands (the 128-bit exchange of the 1024-point FFT hands over four re parts in one register quad and the
// four im parts in another): element S of the pair xp is re, element S of the pair yp is im.
//   cmul_planar<S>: (re + i im) * w, result interleaved -- the twiddle multiply that follows an exchange does the
//                   repacking for free (op_sel picks the halves);
//   pk_join<S>    : (re, im) for the one slot per pass that has no twiddle.
template <int S>
KPR_DEV f2 cmul_planar(f2 xp, f2 yp, f2 w) {
    f2 r;
    if constexpr (S == 0)
        asm("v_pk_mul_f32 %0, %1, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
            "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]"
            : "=&v"(r) : "v"(xp), "v"(yp), "v"(w));
    else
        asm("v_pk_mul_f32 %0, %1, %3 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
            "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
            : "=&v"(r) : "v"(xp), "v"(yp), "v"(w));
    return r;
}
template <int S>
KPR_DEV f2 pk_join(f2 xp, f2 yp) {
    f2 r;
    if constexpr (S == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(r) : "v"(xp), "v"(yp));
    else                  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(xp), "v"(yp));
    return r;
}
template <int E> KPR_DEV f2 quad_pair(f4 q) {          // the 64-bit half of the quad that holds element E
    if constexpr (E < 2) return __builtin_shufflevector(q, q, 0, 1);
    else return __builtin_shufflevector(q, q, 2, 3);
}

// multiply by the compile-time root of unity w32^m = exp(-2 pi i m / 32); after unrolling m is a
// constant: quarter turns are register renames + sign flips, the rest one packed complex multiply
KPR_DEV f2 cmul_w32(f2 x, int m) {
    m &= 31;
    if (m == 0) return x;
    if (m == 8)  return f2{x.y, -x.x};      // -i
    if (m == 16) return f2{-x.x, -x.y};
    if (m == 24) return f2{-x.y, x.x};      // +i
    return cmul_s(x, f2{cos32(m), -sin32(m)});
}

// ---- small forward DFTs (e^{-2 pi i rs/R}), natural order, in registers --------------------
KPR_DEV void dft4(f2& a0, f2& a1, f2& a2, f2& a3) {
    // 8 packed adds in ONE asm block (registers r0..r3 = a0..a3, t = scratch):
    //   t  = r0 + r2 (t0)   r2 = r0 - r2 (t1)   r0 = r1 + r3 (t2)   r3 = r1 - r3 (d)
    //   r1 = t - r0  (o2)   r0 = t + r0  (o0)   t  = r2 - i r3 (o1) r3 = r2 + i r3 (o3)
    f2 t;
    asm("v_pk_add_f32 %4, %0, %2\n\t"
        "v_pk_add_f32 %2, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %1, %3\n\t"
        "v_pk_add_f32 %3, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %4, %0 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %4, %0\n\t"
        "v_pk_add_f32 %4, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t));
    // results: o0 in a0, o1 in t, o2 in a1, o3 in a3
    a2 = a1;
    a1 = t;
}

// dft4 with PLANAR outputs: xq = (o0.x, o2.x, o1.x, o3.x), yq = (o0.y, o2.y, o1.y, o3.y) -- the last four packed adds of
// dft4 with other operand selections; the quads are what one 128-bit store of the exchange takes (their element
// order 0, 2, 1, 3 is undone by the reader's compile-time slot map)
KPR_DEV void dft4_planar(f2 a0, f2 a1, f2 a2, f2 a3, f4& xq, f4& yq) {
    f2 t, A, B, C, D;
    asm("v_pk_add_f32 %8, %0, %2\n\t"                                               // t0
        "v_pk_add_f32 %2, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"                     // t1
        "v_pk_add_f32 %0, %1, %3\n\t"                                               // t2
        "v_pk_add_f32 %3, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"                     // d
        "v_pk_add_f32 %4, %8, %0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"     // (o0.x, o2.x) = t0.x +- t2.x
        "v_pk_add_f32 %6, %8, %0 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"     // (o0.y, o2.y) = t0.y +- t2.y
        "v_pk_add_f32 %5, %2, %3 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"     // (o1.x, o3.x) = t1.x +- d.y
        "v_pk_add_f32 %7, %2, %3 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]"           // (o1.y, o3.y) = t1.y -+ d.x
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(A), "=&v"(B), "=&v"(C), "=&v"(D), "=&v"(t));
    xq = f4{A.x, A.y, B.x, B.y};
    yq = f4{C.x, C.y, D.x, D.y};
}

template <int R> struct Dft;

template <> struct Dft<2> {
    static KPR_DEV void run(f2 (&v)[2]) {
        f2 a = v[0];
        v[0] = cadd(a, v[1]);
        v[1] = csub(a, v[1]);
    }
};

template <> struct Dft<4> {
    static KPR_DEV void run(f2 (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }
};

template <> struct Dft<8> {
    // s = 4*n1 + n2 (n1 in {0,1}), r = k1 + 2*k2
    static KPR_DEV void run(f2 (&v)[8]) {
        f2 y[4][2];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            y[n2][0] = cadd(v[n2], v[n2 + 4]);
            y[n2][1] = cmul_w32(csub(v[n2], v[n2 + 4]), 4 * n2);            // w8^{n2}
        }
#pragma unroll
        for (int k1 = 0; k1 < 2; ++k1) {
            dft4(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) v[k1 + 2 * k2] = y[k2][k1];
        }
    }
};

template <> struct Dft<16> {
    // s = 4*n1 + n2, r = k1 + 4*k2: DFT4 over n1, twiddle w16^{n2 k1}, DFT4 over n2
    static KPR_DEV void run(f2 (&v)[16]) {
        f2 y[4][4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            f2 a0 = v[n2], a1 = v[4 + n2], a2 = v[8 + n2], a3 = v[12 + n2];
            dft4(a0, a1, a2, a3);
            y[n2][0] = a0;
            y[n2][1] = cmul_w32(a1, 2 * n2 * 1);   // w16^{n2 k1}
            y[n2][2] = cmul_w32(a2, 2 * n2 * 2);
            y[n2][3] = cmul_w32(a3, 2 * n2 * 3);
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            dft4(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = y[k2][k1];
        }
    }
    // the same transform with planar outputs: quad k1 holds outputs k1 + 4 k2 in the element order k2 = 0, 2, 1, 3
    static KPR_DEV void run_planar(const f2 (&v)[16], f4 (&xq)[4], f4 (&yq)[4]) {
        f2 y[4][4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            f2 a0 = v[n2], a1 = v[4 + n2], a2 = v[8 + n2], a3 = v[12 + n2];
            dft4(a0, a1, a2, a3);
            y[n2][0] = a0;
            y[n2][1] = cmul_w32(a1, 2 * n2 * 1);
            y[n2][2] = cmul_w32(a2, 2 * n2 * 2);
            y[n2][3] = cmul_w32(a3, 2 * n2 * 3);
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4_planar(y[0][k1], y[1][k1], y[2][k1], y[3][k1], xq[k1], yq[k1]);
    }
};

// ---- per-lane state: factored twiddles + swizzled LDS address bases --------------------------
// table[j] = exp(-2 pi i j / n_fft), j in [0, n_fft)
template <int NC, class SW = SwzXor>
struct FftTw {
    static constexpr int L = NC / kPts;
    static constexpr int NFFT = 2 * NC;
    static constexpr int R1 = Radix<NC>::r1, R2 = Radix<NC>::r2, R3 = Radix<NC>::r3;
    static constexpr int Q2 = kPts / R2;
    static constexpr int H2 = R2 / 4 - 1;            // "high digit" factors of pass 2 (r = 4, 8, 12)
    // pass 2 (NS = R1): w_{R1 R2}^{r kk}, kk = (fl + L q) mod R1;  r = 4a + b -> hi[a-1] * lo[b-1]
    f2 p2lo[Q2][3];
    f2 p2hi[Q2][H2 > 0 ? H2 : 1];
    // pass 3 (NS = R1 R2): w_NC^{r (fl + L q)} = w_NC^{r fl} * w16^{r q}
    f2 p3[R3 > 1 ? R3 - 1 : 1];
    // pairing: w_NFFT^{fl + L m} = w_NFFT^{fl} * w32^{m}
    f2 pp;
    // LDS address bases (word units, already swizzled)
    int a_rd;        // policy(fl)                               (identical for both exchanges)
    int a_w1;        // policy(lane part of pass-1 output index)
    int a_w2;        // policy(lane part of pass-2 output index)

    // Call once per frame: makes the three address bases opaque so that the ~48 swizzled LDS
    // addresses derived from them (one v_xor each) are recomputed per frame instead of being
    // hoisted out of the frame loop into 48 permanently live VGPRs (which costs a wave per SIMD).
    KPR_DEV void refresh() { asm volatile("" : "+v"(a_rd), "+v"(a_w1), "+v"(a_w2)); }

    // the twiddle registers in a fixed order (a kernel that stages one lane-set through LDS for all its waves: k_mel_pw)
    static constexpr int kNumTw = Q2 * 3 + Q2 * (H2 > 0 ? H2 : 1) + (R3 > 1 ? R3 - 1 : 1) + 1;
    template <class F> KPR_DEV void for_each_tw(F&& f) {
        int i = 0;
#pragma unroll
        for (int q = 0; q < Q2; ++q)
#pragma unroll
            for (int b = 0; b < 3; ++b) f(p2lo[q][b], i++);
#pragma unroll
        for (int q = 0; q < Q2; ++q)
#pragma unroll
            for (int a = 0; a < (H2 > 0 ? H2 : 1); ++a) f(p2hi[q][a], i++);
#pragma unroll
        for (int r = 0; r < (R3 > 1 ? R3 - 1 : 1); ++r) f(p3[r], i++);
        f(pp, i++);
    }

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
                p2lo[q][b - 1] = f2{w.x, w.y};
            }
#pragma unroll
            for (int a = 1; a <= H2; ++a) {
                float2 w = table[(4 * a * kk * step) & (NFFT - 1)];
                p2hi[q][a - 1] = f2{w.x, w.y};
            }
        }
        if constexpr (R3 > 1) {
#pragma unroll
            for (int r = 1; r < R3; ++r) {
                float2 w = table[(r * fl * 2) & (NFFT - 1)];
                p3[r - 1] = f2{w.x, w.y};
            }
        }
        {
            float2 w = table[fl];
            pp = f2{w.x, w.y};
        }
        set_addresses(fl);
    }
    KPR_DEV void set_addresses(int fl) {
        if constexpr (IsWide<SW>::value) {
            static_assert(NC == 1024, "the wide layout is derived for 64 lanes x 16 slots");
            a_rd = wide_addr(fl);                                   // slots 0..3; see wide_read for the others
            a_w1 = wide_addr(lane_base(fl, 1, R1));                 // pass-1 output r = 0
            a_w2 = wide_addr(lane_base(fl, R1, R2));                // pass-2 output r = 0
        } else {
            a_rd = SW::template lane<1>(fl);
            a_w1 = SW::template lane<1>(lane_base(fl, 1, R1));
            a_w2 = SW::template lane<2>(lane_base(fl, R1, R2));
        }
    }
};

// Read back z[m] = row[policy(fl + L m)] (component C).  XOR policy: slots m and m+8 share the
// swizzle XOR when L = 64 and sit exactly 8L words apart, so they are written as adjacent accesses
// off ONE address register: hipcc merges them into ds_read2st64_b32 (half the LDS instructions).
// Skew policy: every slot is base register + immediate.
template <int L, int C, int PASS, class SW>
KPR_DEV void exchange_read(f2 (&z)[kPts], int a_rd, const float* row) {
    if constexpr (!SW::kXor) {
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            const float v = row[SW::template at<PASS>(a_rd, L * m)];
            if (C == 0) z[m].x = v; else z[m].y = v;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (j is a constant after unrolling; the condition folds at compile time)
            if ((swz(L * (j + 8)) ^ swz(L * j)) == 8 * L && (swz(L * j) & (8 * L)) == 0) {
                const float* q = row + (a_rd ^ swz(L * j));
                if (C == 0) { z[j].x = q[0]; z[j + 8].x = q[8 * L]; }
                else        { z[j].y = q[0]; z[j + 8].y = q[8 * L]; }
            } else {
                if (C == 0) { z[j].x = row[a_rd ^ swz(L * j)]; z[j + 8].x = row[a_rd ^ swz(L * (j + 8))]; }
                else        { z[j].y = row[a_rd ^ swz(L * j)]; z[j + 8].y = row[a_rd ^ swz(L * (j + 8))]; }
            }
        }
    }
}

// One Stockham pass: radix R, NS = product of earlier radices, PASS = 1, 2 or 3.
// `row` is this lane's frame's NC-word LDS exchange row.  Mirrors complex_fft_lanes() in
// oracle/proto_stockham.py.  The pass comes in two halves so that a caller can put other work between the
// butterflies and the LDS traffic (k_mel_ws interleaves two frames per wave):
//   pass_compute   : twiddles + radix-R butterflies, z -> out (registers only)
//   exchange_issue : out -> row (re), row -> z.x, out -> row (im), row -> z.y -- LDS executes a wave's
//                    operations in order, so the four groups need no wait between them; the values arrive in z
//                    whenever the caller first touches z
template <int NC, int PASS, int R, int NS, class SW = SwzXor>
KPR_DEV void pass_compute(const f2 (&z)[kPts], const FftTw<NC, SW>& tw, f2 (&out)[kPts]) {
    constexpr int Q = kPts / R;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        f2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = z[q + Q * r];
        if constexpr (PASS == 2) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const int a = r >> 2, b = r & 3;
                if (b) v[r] = cmul(v[r], tw.p2lo[q][b - 1]);
                if (a) v[r] = cmul(v[r], tw.p2hi[q][a - 1]);
            }
        } else if constexpr (PASS == 3) {
#pragma unroll
            for (int r = 1; r < R; ++r)
                v[r] = cmul_w32(cmul(v[r], tw.p3[r - 1]), 2 * r * q);       // w16^{r q}
        }
        Dft<R>::run(v);
#pragma unroll
        for (int r = 0; r < R; ++r) out[q + Q * r] = v[r];
    }
}

// Wide layout (SwzWide), one component C of all 16 outputs / slots.  Address algebra (checked for every lane in
// tests/test_proto_stockham.py):
//   exchange 1: addr(r) = (a_w1 ^ 4 (r & 6)) + 4 (r & 9)            -> 4 bases, ds_write2_b32 (offsets 0/4, 32/36)
//   exchange 2: outputs c, c+4, c+8, c+12 fill one chunk at (c & 2 ? (a_w2 ^ 8) + 128 : a_w2) + 64 (c & 1)
//   reads     : slots 4j .. 4j+3 at (j & 1 ? (a_rd ^ 16) + 256 : a_rd) + 512 (j >> 1)
template <int PASS, int C>
KPR_DEV void wide_write(const f2 (&out)[kPts], int aw, float* xr) {
    if constexpr (PASS == 1) {
        // ds_write2_b32 by hand: hipcc does not merge dword stores whose data are halves of 64-bit registers.
        // (LDS executes a wave's operations in order and the "memory" clobber keeps the compiler's own accesses
        // to the row on their side; its lgkmcnt bookkeeping merely becomes conservative.)
        const unsigned xa = (unsigned)(size_t)xr;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned p = xa + 4u * (unsigned)(aw ^ (8 * b));
            asm volatile("ds_write2_b32 %0, %1, %2 offset1:4"
                         :: "v"(p), "v"(C == 0 ? out[2 * b].x : out[2 * b].y),
                            "v"(C == 0 ? out[2 * b + 1].x : out[2 * b + 1].y) : "memory");
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:32 offset1:36"
                         :: "v"(p), "v"(C == 0 ? out[2 * b + 8].x : out[2 * b + 8].y),
                            "v"(C == 0 ? out[2 * b + 9].x : out[2 * b + 9].y) : "memory");
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* p = xr + (((c & 2) ? ((aw ^ 8) + 128) : aw) + 64 * (c & 1));
            f4 v;
            v.x = C == 0 ? out[c].x      : out[c].y;
            v.y = C == 0 ? out[c + 4].x  : out[c + 4].y;
            v.z = C == 0 ? out[c + 8].x  : out[c + 8].y;
            v.w = C == 0 ? out[c + 12].x : out[c + 12].y;
            *reinterpret_cast<f4a*>(p) = v;
        }
    }
}
template <int C>
KPR_DEV void wide_read(f2 (&z)[kPts], int a_rd, const float* xr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* p = xr + (((j & 1) ? ((a_rd ^ 16) + 256) : a_rd) + 512 * (j >> 1));
        const f4 v = *reinterpret_cast<const f4a*>(p);
        if (C == 0) { z[4 * j].x = v.x; z[4 * j + 1].x = v.y; z[4 * j + 2].x = v.z; z[4 * j + 3].x = v.w; }
        else        { z[4 * j].y = v.x; z[4 * j + 1].y = v.y; z[4 * j + 2].y = v.z; z[4 * j + 3].y = v.w; }
    }
}

template <int NC, int PASS, int R, int NS, class SW = SwzXor>
KPR_DEV void exchange_issue(const f2 (&out)[kPts], f2 (&z)[kPts], const FftTw<NC, SW>& tw, float* row) {
    constexpr int L = NC / kPts;
    constexpr int Q = kPts / R;
    static_assert(L >= NS, "lane/const bit split needs L >= NS");
    const int aw = (PASS == 1) ? tw.a_w1 : tw.a_w2;
    if constexpr (IsWide<SW>::value) {
        static_assert(R == 16 && Q == 1 && PASS <= 2, "wide exchange: after the two radix-16 passes of NC = 1024");
        float* xr = static_cast<float*>(__builtin_assume_aligned(row, 16));
        KPR_LDS_FENCE_W();
        wide_write<PASS, 0>(out, aw, xr);
        KPR_LDS_FENCE_R();
        wide_read<0>(z, tw.a_rd, xr);
        KPR_LDS_FENCE_W();
        wide_write<PASS, 1>(out, aw, xr);
        KPR_LDS_FENCE_R();
        wide_read<1>(z, tw.a_rd, xr);
        KPR_LDS_FENCE_X();
    } else {
        // output index = expand(fl + L q) + NS r = lane_base(fl) + [L R q + NS r]
        KPR_LDS_FENCE_W();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) row[SW::template at<PASS>(aw, L * R * q + NS * r)] = out[q + Q * r].x;
        KPR_LDS_FENCE_R();
        exchange_read<L, 0, PASS, SW>(z, tw.a_rd, row);
        KPR_LDS_FENCE_W();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) row[SW::template at<PASS>(aw, L * R * q + NS * r)] = out[q + Q * r].y;
        KPR_LDS_FENCE_R();
        exchange_read<L, 1, PASS, SW>(z, tw.a_rd, row);
        KPR_LDS_FENCE_X();
    }
}

// ---- the 1024-point forward FFT on the wide (128-bit) exchange with PLANAR hand-over: the exchange reads deliver quads
// of re parts and quads of im parts, which the twiddle multiplies of the next pass take directly (cmul_planar), and
// pass 2 emits quads (Dft<16>::run_planar) that are stored with one ds_write_b128 each -- no repack moves (the
// interleaved form needed 86 v_mov per frame).  Same arithmetic, same order of operations as fft_pass x 3.
KPR_DEV f2 planar_cmul_at(const f4& xq, const f4& yq, int e, f2 w) {        // e is a constant after unrolling
    switch (e) {
        case 0: return cmul_planar<0>(quad_pair<0>(xq), quad_pair<0>(yq), w);
        case 1: return cmul_planar<1>(quad_pair<1>(xq), quad_pair<1>(yq), w);
        case 2: return cmul_planar<0>(quad_pair<2>(xq), quad_pair<2>(yq), w);
        default: return cmul_planar<1>(quad_pair<3>(xq), quad_pair<3>(yq), w);
    }
}
KPR_DEV f2 planar_join_at(const f4& xq, const f4& yq, int e) {
    switch (e) {
        case 0: return pk_join<0>(quad_pair<0>(xq), quad_pair<0>(yq));
        case 1: return pk_join<1>(quad_pair<1>(xq), quad_pair<1>(yq));
        case 2: return pk_join<0>(quad_pair<2>(xq), quad_pair<2>(yq));
        default: return pk_join<1>(quad_pair<3>(xq), quad_pair<3>(yq));
    }
}
KPR_DEV void wide_read_quads(f4 (&q)[4], int a_rd, const float* xr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* p = xr + (((j & 1) ? ((a_rd ^ 16) + 256) : a_rd) + 512 * (j >> 1));
        q[j] = *reinterpret_cast<const f4a*>(p);
    }
}
KPR_DEV void wide_write_quads(const f4 (&q)[4], int aw, float* xr) {       // exchange 2: quad c = outputs c, c+4, c+8, c+12
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float* p = xr + (((c & 2) ? ((aw ^ 8) + 128) : aw) + 64 * (c & 1));
        *reinterpret_cast<f4a*>(p) = q[c];
    }
}
KPR_DEV void cfft_forward_wide_planar(f2 (&z)[kPts], const FftTw<1024, SwzWide>& tw, float* row) {
    float* xr = static_cast<float*>(__builtin_assume_aligned(row, 16));
    f4 X1[4], Y1[4];
    {
        f2 out[kPts];
        pass_compute<1024, 1, 16, 1, SwzWide>(z, tw, out);                  // pass 1: no twiddles
        KPR_LDS_FENCE_W();
        wide_write<1, 0>(out, tw.a_w1, xr);
        KPR_LDS_FENCE_R();
        wide_read_quads(X1, tw.a_rd, xr);
        KPR_LDS_FENCE_W();
        wide_write<1, 1>(out, tw.a_w1, xr);
        KPR_LDS_FENCE_R();
        wide_read_quads(Y1, tw.a_rd, xr);
        KPR_LDS_FENCE_X();
    }
    f4 X2[4], Y2[4];
    {
        // pass 2 (NS = 16): slot r = 4 j + e sits at element e of quad j; w = p2hi[a - 1] * p2lo[b - 1], r = 4 a + b
        f2 v[kPts];
#pragma unroll
        for (int r = 0; r < kPts; ++r) {
            const int j = r >> 2, e = r & 3, a = r >> 2, b = r & 3;
            if (r == 0) v[r] = planar_join_at(X1[j], Y1[j], e);
            else if (b) {
                v[r] = planar_cmul_at(X1[j], Y1[j], e, tw.p2lo[0][b - 1]);
                if (a) v[r] = cmul(v[r], tw.p2hi[0][a - 1]);
            } else v[r] = planar_cmul_at(X1[j], Y1[j], e, tw.p2hi[0][a - 1]);
        }
        f4 XO[4], YO[4];
        Dft<16>::run_planar(v, XO, YO);
        KPR_LDS_FENCE_W();
        wide_write_quads(XO, tw.a_w2, xr);
        KPR_LDS_FENCE_R();
        wide_read_quads(X2, tw.a_rd, xr);
        KPR_LDS_FENCE_W();
        wide_write_quads(YO, tw.a_w2, xr);
        KPR_LDS_FENCE_R();
        wide_read_quads(Y2, tw.a_rd, xr);
        KPR_LDS_FENCE_X();
    }
    // pass 3 (radix 4, NS = 256, last): v[r] = slot q + 4 r = quad r, stored element order 0, 2, 1, 3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = (q == 1) ? 2 : (q == 2) ? 1 : q;
        f2 v[4];
        v[0] = planar_join_at(X2[0], Y2[0], e);
#pragma unroll
        for (int r = 1; r < 4; ++r) v[r] = cmul_w32(planar_cmul_at(X2[r], Y2[r], e, tw.p3[r - 1]), 2 * r * q);
        Dft<4>::run(v);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[q + 4 * r] = v[r];
    }
}

template <int NC, int PASS, int R, int NS, class SW = SwzXor>
KPR_DEV void fft_pass(f2 (&z)[kPts], const FftTw<NC, SW>& tw, float* row) {
    constexpr bool LAST = (NS * R == NC);
    f2 out[kPts];
    pass_compute<NC, PASS, R, NS, SW>(z, tw, out);
    if constexpr (LAST) {
#pragma unroll
        for (int m = 0; m < kPts; ++m) z[m] = out[m];
    } else {
        exchange_issue<NC, PASS, R, NS, SW>(out, z, tw, row);
    }
}

// Forward NC-point complex FFT, in / out layout "fl + L*m".
template <int NC, class SW = SwzXor>
KPR_DEV void cfft_forward(f2 (&z)[kPts], const FftTw<NC, SW>& tw, float* row) {
    using Rx = Radix<NC>;
    fft_pass<NC, 1, Rx::r1, 1, SW>(z, tw, row);
    fft_pass<NC, 2, Rx::r2, Rx::r1, SW>(z, tw, row);
    if constexpr (Rx::r3 > 1) fft_pass<NC, 3, Rx::r3, Rx::r1 * Rx::r2, SW>(z, tw, row);
}

// exchange-row policy per transform size: the additive skew is derived (and tested) for NC = 1024
// and 512, whose two exchanges follow radix-16 passes; everything else keeps the XOR swizzle
template <int NC> struct SwzFor { typedef SwzXor type; };
template <> struct SwzFor<1024> { typedef SwzSkew type; };
template <> struct SwzFor<512> { typedef SwzSkew type; };     // radices (16,16,2): same two exchanges

// Pairing pass of the real FFT.  With Z the complex FFT of the packed frame,
//   2 X[k]    =       (Z[k] + conj Z[NC-k]) - i w_k (Z[k] - conj Z[NC-k])   =      e + t
//   2 X[NC-k] = conj( (Z[k] + conj Z[NC-k]) + i w_k (Z[k] - conj Z[NC-k]) ) = conj(e - t)
// (w_k = w_NFFT^{fl} * w32^{m}; the factor 2 is left to the caller: fold 0.5 into the window),
// so ONE evaluation yields both bins of a pair.  Lane fl evaluates its slots m = 0..7
// (k = fl + L m < NC/2) and emits X[k] together with X[NC-k]; the results go to LDS anyway, so no
// lane has to produce "its own" upper bins.  Partner Z[NC-k] lives in lane (L-fl)%L slot 15-m
// (lane 0: its own slot (16-m)%16) -> one __shfl per component.  k = NC/2 pairs with itself
// and is emitted by lane 0 (slot 8); k = 0 pairs with the Nyquist bin NC.
//   emit(k, Xk, kp, Xkp) is called with kp == NC - k, or kp < 0 when there is no second bin.
template <int NC, class SW, class Emit>
KPR_DEV void rfft_pair(const f2 (&z)[kPts], const FftTw<NC, SW>& tw, int fl, int lane, Emit&& emit) {
    constexpr int L = NC / kPts;
    const int src = (lane - fl) + ((L - fl) & (L - 1));
    const f2 ppmi = f2{tw.pp.y, -tw.pp.x};            // -i * w_NFFT^{fl}
    // all 16 cross-lane reads first, then the arithmetic: interleaved with the emits (LDS stores)
    // hipcc waits for every ds_bpermute pair separately -- 8 exposed LDS round trips per frame
    f2 zq[kPts / 2];
#pragma unroll
    for (int m = 0; m < kPts / 2; ++m) {
        zq[m].x = __shfl(z[kPts - 1 - m].x, src, 64);
        zq[m].y = __shfl(z[kPts - 1 - m].y, src, 64);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < kPts / 2; ++m) {
        f2 zp = zq[m];
        if (fl == 0) zp = z[(kPts - m) & (kPts - 1)];
        const f2 e = cadd_conj(z[m], zp);
        const f2 t = cmul(cmul_w32(csub_conj(z[m], zp), m), ppmi);
        const f2 xk = cadd(e, t);
        const f2 xm = csub(e, t);
        const int k = fl + L * m;
        emit(k, xk, NC - k, f2{xm.x, -xm.y});
    }
    if (fl == 0) {                                      // k = NC/2 (slot 8 of lane 0), self-paired
        const f2 zz = z[kPts / 2];
        const f2 e = cadd_conj(zz, zz);
        const f2 t = cmul(cmul_w32(csub_conj(zz, zz), kPts / 2), ppmi);
        emit(NC / 2, cadd(e, t), -1, f2{0.0f, 0.0f});
    }
}

// The same pairing in two halves (k_mel_ws interleaves two frames per wave and puts the other frame's butterflies
// between them).  rfft_pair_issue starts the 16 cross-lane reads and lets the partner values land IN PLACE of
// z[8..15] (a lane's own upper slots are only needed by its partner -- except on lane fl = 0, which pairs with its
// own slots (16 - m) & 15: it ships those through the same permute, and keeps z[8] for the self-paired bin NC/2);
// rfft_pair_finish does the arithmetic: afterwards z[15 - m] holds Z[NC - k] for slot m.
template <int NC>
KPR_DEV void rfft_pair_issue(f2 (&z)[kPts], f2& z8, int fl, int lane) {
    constexpr int L = NC / kPts;
    const int src = (lane - fl) + ((L - fl) & (L - 1));
    z8 = z[kPts / 2];
    f2 t[kPts / 2];
#pragma unroll
    for (int m = 0; m < kPts / 2; ++m) {       // what the partner (or, on lane 0, the lane itself) needs from this lane
        const f2 own = z[(kPts - m) & (kPts - 1)], up = z[kPts - 1 - m];
        t[m] = f2{fl == 0 ? own.x : up.x, fl == 0 ? own.y : up.y};
    }
#pragma unroll
    for (int m = 0; m < kPts / 2; ++m) {
        z[kPts - 1 - m].x = __shfl(t[m].x, src, 64);
        z[kPts - 1 - m].y = __shfl(t[m].y, src, 64);
    }
}

template <int NC, class SW, class Emit>
KPR_DEV void rfft_pair_finish(const f2 (&z)[kPts], f2 z8, const FftTw<NC, SW>& tw, int fl, Emit&& emit) {
    constexpr int L = NC / kPts;
    const f2 ppmi = f2{tw.pp.y, -tw.pp.x};            // -i * w_NFFT^{fl}
#pragma unroll
    for (int m = 0; m < kPts / 2; ++m) {
        const f2 zp = z[kPts - 1 - m];
        const f2 e = cadd_conj(z[m], zp);
        const f2 t = cmul(cmul_w32(csub_conj(z[m], zp), m), ppmi);
        const f2 xk = cadd(e, t);
        const f2 xm = csub(e, t);
        const int k = fl + L * m;
        emit(k, xk, NC - k, f2{xm.x, -xm.y});
    }
    if (fl == 0) {                                      // k = NC/2 (slot 8 of lane 0), self-paired
        const f2 e = cadd_conj(z8, z8);
        const f2 t = cmul(cmul_w32(csub_conj(z8, z8), kPts / 2), ppmi);
        emit(NC / 2, cadd(e, t), -1, f2{0.0f, 0.0f});
    }
}

// Inverse pairing: X[k] and X[NC-k] (k = fl + L m) -> conj(2 Z[k]), ready for cfft_forward;
// the caller conjugates again after the FFT (IFFT(z) = conj(FFT(conj z))).
//   2 Z[k] = (X[k] + conj X[NC-k]) + i conj(w_k) (X[k] - conj X[NC-k])
template <int NC, class SW>
KPR_DEV f2 irfft_pair_one(f2 xk, f2 xp, const FftTw<NC, SW>& tw, int m) {
    const f2 e = cadd_conj(xk, xp);
    f2 d = csub_conj(xk, xp);
    // conj(d) * w = conj(d * conj(w)); o = d * conj(w)
    f2 t = cmul(cmul_w32(f2{d.x, -d.y}, m), tw.pp);
    const f2 o = f2{t.x, -t.y};
    // 2Z = e + i o = (e.x - o.y, e.y + o.x); return its conjugate
    return f2{e.x - o.y, -(e.y + o.x)};
}

}  // namespace kpr
