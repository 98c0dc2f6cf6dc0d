// kpr_common.h -- errors, frame geometry shared by host and device, sample fetch, small device helpers.
// Part of the single translation unit kapre_hip.hip (included there, in this order; not stand-alone).
#pragma once

namespace kpr {

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define KPR_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(KPR_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// ------------------------------------------------------------------------------------------
// device status word (round 5)
// ------------------------------------------------------------------------------------------
// Conditions only a kernel can see.  The ring / per-wave kernels wait for each other through LDS flags, and every such wait is
// BOUNDED (a protocol error must not hang the GPU).  A wait that runs out used to end as wrong samples only a parity test would notice; now the wave also ORs a bit
// into a word of mapped host memory, and the next API call -- or kpr_device_status() -- fails with KPR_E_DEVICE.
// g_status_word: device pointer of that word (0 until the first launcher of such a kernel installed it on this device).
enum : unsigned {
    kStMelWs = 1u << 0,        // k_mel_ws: producer / consumer hand-over
    kStIstftWsCons = 1u << 1,  // k_istft_ws / k_istft_ws_mr: the consumer waited for frames
    kStIstftWsProd = 1u << 2,  // ... a producer waited for ring rows
    kStIstftPw = 1u << 3,      // k_istft_pw: a run waited for its successor's partial blocks
    kStMelPwSlot = 1u << 5,    // k_mel_pw (PAIR, staged channels_last store): a wave waited for an output slot
    kStStalePlan = 1u << 4,    // k_mel_pw: the packed filterbank at this address is not the one whose band plan the host cached
    kStSelfTest = 1u << 31     // kpr_debug_spin_timeout
};
__device__ unsigned* g_status_word = nullptr;
__device__ __forceinline__ void status_raise(unsigned bits) {             // (cold: callers test their spin count first)
    unsigned* w = g_status_word;
    if (w) __hip_atomic_fetch_or(w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------
// geometry shared by host and device
// ------------------------------------------------------------------------------------------
struct Geom {
    long long total_frames;  // B * C * F
    long long T;
    int F, C;
    int n_fft, win, hop, pad_left;
    int K;
    int in_cl, out_cl;
    int cfast;   // frame numbering: 0 -> g = (b*C + c)*F + f,  1 -> g = (b*F + f)*C + c.
                 // Channel-fastest is used for channels_last waveforms with C > 1: the C frames
                 // that share the same interleaved cache lines then sit in the same tile.
    // division by F and by C as multiply + shift (geom_set_magic, always set by the host): a 32-bit division
    // is ~25 vector instructions even for a wave-uniform operand, and frame_pos runs once per frame in every kernel
    unsigned mF, mC;
    int sF, sC;
};

// u / d for u < 2^31 as (u * m) >> (31 + s), s = ceil(log2 d), m = floor(2^(31+s) / d) + 1: exact, since
// m d = 2^(31+s) + e with 0 < e <= d makes the error term u e / (d 2^(31+s)) < 2^-s <= 1/d
inline void magic_for(unsigned d, unsigned* m, int* s) {
    int sh = 0;
    while ((1ull << sh) < d) ++sh;
    *s = sh;
    *m = (unsigned)(((1ull << (31 + sh)) / d) + 1ull);
}
inline void geom_set_magic(Geom& g) {
    g.mF = g.mC = 0; g.sF = g.sC = 0;
    if (g.F >= 1 && g.C >= 1) { magic_for((unsigned)g.F, &g.mF, &g.sF); magic_for((unsigned)g.C, &g.mC, &g.sC); }
}
__host__ __device__ inline unsigned magic_div(unsigned u, unsigned m, int s) {
    return (unsigned)(((unsigned long long)u * m) >> (31 + s));
}

struct FramePos {
    long long sig_off;   // element offset of sample 0 of this (b, c) signal
    int es;              // element stride between consecutive samples
    long long s0;        // time index of frame sample 0 (may be negative with pad_begin)
    long long bc;        // b*C + c
    int b, c, f;
};

KPR_DEV FramePos frame_pos(const Geom& g, long long gf) {
    FramePos p;
    if (g.total_frames < 0x7fffffffLL) {   // multiply + shift (scalar ALU when gf is wave-uniform); every host-side
                                           // Geom goes through geom_set_magic
        const unsigned u = (unsigned)gf;
        if (g.cfast) {
            const unsigned q = (g.C == 1) ? u : magic_div(u, g.mC, g.sC);
            p.c = (int)(u - q * (unsigned)g.C);
            p.b = (int)magic_div(q, g.mF, g.sF);
            p.f = (int)(q - (unsigned)p.b * (unsigned)g.F);
        } else {
            const unsigned bc = magic_div(u, g.mF, g.sF);
            p.f = (int)(u - bc * (unsigned)g.F);
            p.b = (g.C == 1) ? (int)bc : (int)magic_div(bc, g.mC, g.sC);
            p.c = (int)(bc - (unsigned)p.b * (unsigned)g.C);
        }
    } else {   // >= 2^31 frames: 64-bit division.  Cold; the divisors are laundered so that the reciprocal set-up of
               // the division cannot be hoisted out of this branch into registers that the hot loops then pay for
        long long dC = g.C, dF = g.F;
        asm volatile("" : "+s"(dC), "+s"(dF));
        if (g.cfast) {
            const long long q = gf / dC;
            p.c = (int)(gf - q * dC);
            p.b = (int)(q / dF);
            p.f = (int)(q - (long long)p.b * dF);
        } else {
            const long long bc = gf / dF;
            p.f = (int)(gf - bc * dF);
            p.b = (int)(bc / dC);
            p.c = (int)(bc - (long long)p.b * dC);
        }
    }
    p.bc = (long long)p.b * g.C + p.c;
    if (g.in_cl) { p.sig_off = (long long)p.b * g.T * g.C + p.c; p.es = g.C; }
    else         { p.sig_off = p.bc * g.T;                        p.es = 1;   }
    p.s0 = (long long)p.f * g.hop - g.pad_left;
    return p;
}

// frame_pos for launches of fewer than 2^31 frames (the caller checked): the multiply + shift branch alone -- without the
// 64-bit fallback in the instruction stream (k_mel_pw runs this once or twice per frame: ~100 scalar instructions and a dozen
// reloads of spilled SGPRs less per frame)
KPR_DEV FramePos frame_pos32(const Geom& g, unsigned u) {
    FramePos p;
    if (g.cfast) {
        const unsigned q = (g.C == 1) ? u : magic_div(u, g.mC, g.sC);
        p.c = (int)(u - q * (unsigned)g.C);
        p.b = (int)magic_div(q, g.mF, g.sF);
        p.f = (int)(q - (unsigned)p.b * (unsigned)g.F);
    } else {
        const unsigned bc = magic_div(u, g.mF, g.sF);
        p.f = (int)(u - bc * (unsigned)g.F);
        p.b = (g.C == 1) ? (int)bc : (int)magic_div(bc, g.mC, g.sC);
        p.c = (int)(bc - (unsigned)p.b * (unsigned)g.C);
    }
    p.bc = (long long)p.b * g.C + p.c;
    if (g.in_cl) { p.sig_off = (long long)p.b * g.T * g.C + p.c; p.es = g.C; }
    else         { p.sig_off = p.bc * g.T;                        p.es = 1;   }
    p.s0 = (long long)p.f * g.hop - g.pad_left;
    return p;
}

// spectrogram addressing: element (frame, q) of an axis with Q entries lives at
// spec_base(...) + q * spec_stride(g)   (elements of the output dtype)
KPR_DEV long long spec_base(const Geom& g, const FramePos& p, long long gf, int Q) {
    (void)gf;
    if (g.out_cl) return (((long long)p.b * g.F + p.f) * Q) * g.C + p.c;
    return (p.bc * g.F + p.f) * Q;
}
KPR_DEV int spec_stride(const Geom& g) { return g.out_cl ? g.C : 1; }

// order preserving float <-> uint map for atomic max / min
KPR_DEV unsigned enc_f(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
KPR_DEV float dec_f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct DbDev {
    int enabled;
    float amin;
    float ref_term;   // 10*log10(max(amin, ref))
    float dyn;
    // statistics slots (fused mel kernels, small batches): with few items every workgroup's closing atomics land on the
    // same few words and serialise there (~100 ns each: 8 six-channel items, 256 workgroups x 4 waves: +13 us on a 12 us
    // kernel).  A workgroup uses slot blockIdx & slot_mask, slot s of item b sits at stats[s * slot_stride + 2 b]
    // (slot_stride = 2 * items); k_db_clamp reduces the slots.  slot_mask = 0: one slot, the layout of rounds 1-2.
    int slot_mask;
    int slot_stride;
};

KPR_DEV float to_db(float v, const DbDev& db) {
    // backend.py:186-188: 10*log10(max(x, amin)) - 10*log10(max(amin, ref)) = 10 log10(2) * log2(max(x, amin)) - ref_term.
    // The hardware logarithm (v_log_f32, 1 ulp) directly: libm's logf wraps the same instruction in a denormal rescue and
    // a split-constant multiply -- two v_cndmask on VCC (~20 cycles each on gfx950) and a dozen more instructions per
    // value, four values per lane and tile: 134.8 -> 153.2 us of kernel time on the speech shape with decibels on
    // (20 M outputs).  The argument is never denormal: the host raises amin to the smallest normal float (make_db).
    return fmaf(3.01029995663981195f, __builtin_amdgcn_logf(fmaxf(v, db.amin)), -db.ref_term);
}

// Per-item decibel statistics (backend.py:186-192 needs each item's maximum).  A global atomic on ONE address costs the
// L2 ~100 ns and same-address atomics serialise: the round-1/2 epilogues issued a pair per wave and tile plus one pair PER
// THREAD on every tile that straddles two items -- 170k atomics on 256 addresses for the reference's own test shape
// (n_fft 512, 2 channels), +67 us on a 70 us kernel.  Now every lane keeps a running (item, max, min) across tiles and
// the wave flushes it -- one atomic pair per distinct item among its lanes -- only when some lane moves on to another
// item, and once at the end.
struct DbRun {
    int b;            // item the running extrema belong to (-1: none yet)
    float mx, mn;
    KPR_DEV void reset() { b = -1; mx = -INFINITY; mn = INFINITY; }
};
KPR_DEV void db_flush_wave(DbRun& r, unsigned* __restrict__ item_stats, const DbDev& db) {
    item_stats += (long long)((int)blockIdx.x & db.slot_mask) * db.slot_stride;      // this workgroup's slot
    unsigned long long live = __ballot(r.b >= 0 && r.mx >= r.mn);
    while (live) {                                                     // one turn per distinct item (wave-uniform loop)
        const int b = __builtin_amdgcn_readlane(r.b, (int)__builtin_ctzll(live));
        const bool mine = r.b == b && r.mx >= r.mn;
        float mx = mine ? r.mx : -INFINITY, mn = mine ? r.mn : INFINITY;
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            mn = fminf(mn, __shfl_xor(mn, o, 64));
        }
        if ((int)(threadIdx.x & 63) == (int)__builtin_ctzll(live)) {
            atomicMax(&item_stats[2 * b], enc_f(mx));
            atomicMin(&item_stats[2 * b + 1], enc_f(mn));
        }
        live &= ~__ballot(r.b == b);
    }
    r.reset();
}
// add value v of item b to a lane's running statistics; flushes the whole wave first when any lane changes item
// (must be called by all lanes of the wave together; `have` = this lane has a value)
KPR_DEV void db_account(DbRun& r, bool have, int b, float vmax, float vmin, unsigned* __restrict__ item_stats,
                        const DbDev& db) {
    if (__any(have && r.b >= 0 && r.b != b)) db_flush_wave(r, item_stats, db);
    if (have) { r.b = b; r.mx = fmaxf(r.mx, vmax); r.mn = fminf(r.mn, vmin); }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence +
// s_barrier, and the fence drains vmcnt(0): global loads issued as a PREFETCH before the barrier
// (the next tile's samples, ~3 us from HBM when the tile is far away) would have to land before
// any wave may pass it.  Here only this wave's LDS operations are waited for.
KPR_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// frame load: z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1], n = fl + L*m
// ------------------------------------------------------------------------------------------
template <int NC>
struct WinRegs {
    f2 w[kPts];     // (scale * window[2n], scale * window[2n+1]), n = fl + L*m
    // scale = 0.5 for the forward transforms: rfft_pair() yields 2 X[k]
    KPR_DEV void load(const float* __restrict__ window, int win, int fl, float scale) {
        constexpr int L = NC / kPts;
        // unconditional loads (clamped index, masked scale): a per-element "load or zero" makes
        // hipcc branch around every load and drain vmcnt(0) 32 times (~700 cycles each)
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            const int n = 2 * (fl + L * m);
            const float a = window[min(n, win - 1)];
            const float b = window[min(n + 1, win - 1)];
            w[m].x = a * ((n < win) ? scale : 0.0f);
            w[m].y = b * ((n + 1 < win) ? scale : 0.0f);
        }
    }
};

// raw (un-windowed) samples of one frame: z[m] = (x[2n], x[2n+1]), n = fl + L*m.
// Returns the validity mask vm (bit 2m: z[m].x is a real sample, bit 2m+1: z[m].y); samples whose
// bit is 0 (zero padding, beyond a short window, frame beyond the end) were loaded from a clamped
// address and must be zeroed with mask_frame() WHEN THE FRAME IS CONSUMED.  Keeping the mask out of
// the load path matters twice: with a visible "ok ? x : 0" hipcc sinks each load under its
// condition (32 exec-masked branches, each draining vmcnt(0): one memory latency per sample pair),
// and a prefetched frame must not be touched before it is used.
// WIN_ZEROS: the caller's window table holds zeros beyond win_length, so a frame whose samples all exist takes the fast
// path for any window length (the samples beyond the window are multiplied by those zeros; see fetch_frame_z).
template <int NC, bool WIN_ZEROS = false>
KPR_DEV unsigned fetch_frame(const float* __restrict__ x, const Geom& g, const FramePos& p, bool valid,
                             int fl, f2 (&z)[kPts]) {
    constexpr int L = NC / kPts;
    const float* sig = x + p.sig_off;
    const bool interior = valid && p.s0 >= 0 && (p.s0 + 2 * NC) <= g.T && (WIN_ZEROS || g.win >= 2 * NC);
    if (interior && p.es == 1) {
        const float* fp = sig + p.s0;
        if ((((unsigned long long)fp) & 7ull) == 0) {       // 8-byte aligned: one dwordx2 per point
            const float2* fp2 = reinterpret_cast<const float2*>(fp) + fl;
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                float2 v = fp2[L * m];
                z[m] = f2{v.x, v.y};
            }
        } else {
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                int n = 2 * (fl + L * m);
                z[m] = f2{fp[n], fp[n + 1]};
            }
        }
        return 0xffffffffu;
    }
    // edge frames (zero padding), short windows, channels_last: unconditional loads from a clamped
    // index
    unsigned vm = 0;
    const long long tmax = g.T - 1;
    {
        // 32-bit ELEMENT offsets (check_geom rejects signals of 2^30 elements or more):
        // clamp(t, 0, T-1) * es == clamp(t * es, 0, (T-1) * es), and t * es is linear in m -- one
        // multiply per frame instead of one 64-bit multiply per sample
        const int es = p.es, omax = (int)tmax * es;
        const int o_base = ((int)p.s0 + 2 * fl) * es;
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            const int n = 2 * (fl + L * m);
            const int o0 = o_base + m * (2 * L) * es, o1 = o0 + es;
            z[m] = f2{sig[min(max(o0, 0), omax)], sig[min(max(o1, 0), omax)]};
            vm |= (valid && n < g.win && (unsigned)o0 <= (unsigned)omax) ? (1u << (2 * m)) : 0u;
            vm |= (valid && n + 1 < g.win && (unsigned)o1 <= (unsigned)omax) ? (2u << (2 * m)) : 0u;
            // issue in groups of four: without the fence hipcc computes all 32 64-bit addresses
            // first (64 live VGPRs -> spills in the 168-register kernels)
            if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    return vm;
}

// fetch_frame without a validity mask (round 3): z[m] = (x[2n], x[2n+1]), n = fl + L*m, samples that do not exist = 0.
//   * every sample of the frame exists (any element stride, any window length): plain loads, no clamps, no masks --
//     one dwordx2 per point for contiguous signals (the hardware takes 4-byte aligned 8-byte loads), two dword loads per
//     point for interleaved (channels_last, C > 1) signals.  A window shorter than n_fft needs nothing here: the samples
//     beyond it meet the zeros of the window (the reference, tf.signal.stft with frame_length < fft_length, never reads
//     them; for finite input 0 * x == 0 is the same number).
//   * otherwise (zero padding at either end of the signal, frame beyond the end): the registers are zeroed first and only
//     the existing samples are loaded, each load under an EXEC mask set by hand, and waited for before returning (no
//     prefetch for these few frames).  A visible "ok ? load : 0" makes hipcc
//     branch around every load and drain vmcnt there; the round-1/2 form (clamped address + 32-bit validity mask applied
//     when the frame is consumed) cost two clamps and four mask instructions per sample and ~30 VGPRs of masks.
//   * CHANNEL PAIRS, interleaved (channels_last, C even; round 3: stereo), frames numbered channel-fastest, 16 or 32 lanes per frame (`lane` >= 0
//     enables it): the two channel-frames of one (item, frame) sit in neighbouring lane groups of the wave, and the four
//     floats both need per point -- x[2n][0], x[2n][1], x[2n+1][0], x[2n+1][1] -- are 16 contiguous bytes.  The channel-0
//     lane loads the first 8, the channel-1 lane the second 8 (ONE dwordx2 each instead of two dword loads with a stride
//     of 8 bytes: half the vector-memory instructions for the same cache lines), and *pair_swap = true tells the caller
//     to run stereo_unswap() on the registers WHEN IT CONSUMES them: one v_permlane16/32_swap per point turns
//     (a, b | a', b') into (a, a' | b, b').  Wave-uniform decision; every other case returns *pair_swap = false.
template <int NC>
KPR_DEV void stereo_unswap(f2 (&z)[kPts]) {
    constexpr int L = NC / kPts;
    static_assert(L == 16 || L == 32, "neighbouring lane groups are rows of 16 or halves of 32");
#pragma unroll
    for (int m = 0; m < kPts; ++m) {
        // (the operands come from vector-memory loads; s_nop 1 covers the VALU-write -> swap hazard if the compiler
        //  moved them first)
        if constexpr (L == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(z[m].x), "+v"(z[m].y));
        else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(z[m].x), "+v"(z[m].y));
    }
}
template <int NC>
KPR_DEV void fetch_frame_z(const float* __restrict__ x, const Geom& g, const FramePos& p, bool valid, int fl, f2 (&z)[kPts],
                           int lane = -1, bool* pair_swap = nullptr) {
    constexpr int L = NC / kPts;
    const float* sig = x + p.sig_off;
    struct __attribute__((aligned(4))) float2u { float x, y; };      // 8-byte load from a 4-byte aligned address
    const bool inside = valid && p.s0 >= 0 && (p.s0 + 2 * NC) <= g.T;
    if constexpr (L == 16 || L == 32) {
        if (pair_swap) {
            *pair_swap = false;
            // (round 4: any even channel count -- channels c, c + 1 of neighbouring lane groups; es = C floats between samples)
            if (lane >= 0 && g.cfast && __all(inside && (p.es & 1) == 0 && (p.c & 1) == ((lane / L) & 1))) {
                const int q = p.c & 1, es = p.es;
                const float* fp = sig - q + (p.s0 + q + 2 * fl) * es;     // channel c: sample 2n of (c, c+1); c+1: sample 2n+1 of them
#pragma unroll
                for (int m = 0; m < kPts; ++m) {
                    float2u v = *reinterpret_cast<const float2u*>(fp + (2 * L * m) * es);
                    z[m] = f2{v.x, v.y};
                    if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (addresses four points at a time)
                }
                *pair_swap = true;
                return;
            }
        }
    }
    if (inside) {
        if (p.es == 1) {
            const float2u* fp2 = reinterpret_cast<const float2u*>(sig + p.s0) + fl;
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                float2u v = fp2[L * m];
                z[m] = f2{v.x, v.y};
            }
        } else {
            const float* fp = sig + (p.s0 + 2 * fl) * p.es;
            const int es = p.es;
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                z[m] = f2{fp[(2 * L * m) * es], fp[(2 * L * m + 1) * es]};
                if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (addresses four points at a time)
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = f2{0.0f, 0.0f};
    if (!valid) return;
    {
        const int es = p.es, T = (int)g.T;                 // (check_geom rejects signals of 2^30 elements or more)
        const int t_base = (int)p.s0 + 2 * fl;
        const float* fp = sig + (long long)t_base * es;
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            const int t0 = t_base + 2 * L * m;
            const unsigned long long k0 = __ballot((unsigned)t0 < (unsigned)T), k1 = __ballot((unsigned)(t0 + 1) < (unsigned)T);
            const float* a0 = fp + (long long)(2 * L * m) * es;
            const float* a1 = a0 + es;
            unsigned long long sv;
            asm volatile("s_and_saveexec_b64 %[sv], %[mk]\n\tglobal_load_dword %[d], %[a], off\n\ts_mov_b64 exec, %[sv]"
                         : [d] "+v"(z[m].x), [sv] "=&s"(sv) : [mk] "s"(k0), [a] "v"(a0) : "memory");
            asm volatile("s_and_saveexec_b64 %[sv], %[mk]\n\tglobal_load_dword %[d], %[a], off\n\ts_mov_b64 exec, %[sv]"
                         : [d] "+v"(z[m].y), [sv] "=&s"(sv) : [mk] "s"(k1), [a] "v"(a1) : "memory");
        }
        // The compiler does not know that these loads complete later: it may copy a destination register before the data
        // is there (seen as results that changed from call to call once the register allocation around the call site
        // changed).  Edge frames are a handful per signal, so their loads are simply waited for here; the registers are
        // "defined" by the asm statements below, after the wait.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < kPts; ++m) asm volatile("" : "+v"(z[m].x), "+v"(z[m].y));
    }
}

// zero the samples of a fetched frame whose validity bit is clear (see fetch_frame)
KPR_DEV void mask_frame(f2 (&z)[kPts], unsigned vm) {
    if (__all(vm == 0xffffffffu)) return;        // wave-uniform: interior frames pay one compare
#pragma unroll
    for (int m = 0; m < kPts; ++m) {
        const unsigned kx = (unsigned)(-(int)((vm >> (2 * m)) & 1u));
        const unsigned ky = (unsigned)(-(int)((vm >> (2 * m + 1)) & 1u));
        z[m] = f2{__uint_as_float(__float_as_uint(z[m].x) & kx), __uint_as_float(__float_as_uint(z[m].y) & ky)};
    }
}

template <int NC>
KPR_DEV void apply_window(const WinRegs<NC>& w, f2 (&z)[kPts]) {
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = pmul(z[m], w.w[m]);
}

}  // namespace kpr
