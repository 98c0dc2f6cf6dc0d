// kapre_hip.hip -- gfx950 kernels + C ABI (include/kapre_hip.h) for Kapre's time-frequency path.
//
// Kernels
//   k_mel_fused<NC>   waveform -> [frame+window+rFFT -> |X| -> (K x M) filterbank on fp32 MFMA
//                     -> optional 10 log10] ; the whole Sequential of composed.py:138-261 in
//                     one launch, nothing but the waveform read and the mel tile written.
//   k_stft<NC>        frame+window+rFFT with complex / magnitude / phase epilogue
//                     (time_frequency.py:164-185 [+ :359 / :402]).
//   k_irfft<NC>       pairing + inverse FFT + synthesis window -> windowed frames
//   k_ola             gather-style overlap-add (no atomics)      (time_frequency.py:304-317)
//   k_gemm<...>       generic fp32-MFMA GEMM with accessor/epilogue policies: stand-alone
//                     ApplyFilterbank (time_frequency.py:544) and the DFT-as-GEMM path for
//                     transform sizes the Stockham kernels do not cover (the idea of the
//                     reference's own kapre/tflite_compatible_stft.py:14-75).
//   k_db_*            magnitude_to_decibel (backend.py:126-194): log pass with per-item
//                     max/min statistics, then the dynamic-range clamp.
//
// gfx950 only: wave64, v_mfma_f32_16x16x4_f32, 160 KiB LDS.  No CUDA/compat paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kapre_hip.h"
#include "kpr_fft.h"
#include "kpr_fft_mr.h"

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
};

struct FramePos {
    long long sig_off;   // element offset of sample 0 of this (b, c) signal
    int es;              // element stride between consecutive samples
    long long s0;        // time index of frame sample 0 (may be negative with pad_begin)
    long long bc;        // b*C + c
    int b, c, f;
};

KPR_DEV FramePos frame_pos(const Geom& g, long long gf) {
    FramePos p;
    if (g.total_frames < 0x7fffffffLL) {          // 32-bit division is ~10x cheaper on the GPU
        const unsigned u = (unsigned)gf;
        if (g.cfast) {
            const unsigned q = u / (unsigned)g.C;
            p.c = (int)(u - q * (unsigned)g.C);
            p.b = (int)(q / (unsigned)g.F);
            p.f = (int)(q - (unsigned)p.b * (unsigned)g.F);
        } else {
            const unsigned bc = u / (unsigned)g.F;
            p.f = (int)(u - bc * (unsigned)g.F);
            p.b = (int)(bc / (unsigned)g.C);
            p.c = (int)(bc - (unsigned)p.b * (unsigned)g.C);
        }
    } else if (g.cfast) {
        const long long q = gf / g.C;
        p.c = (int)(gf - q * g.C);
        p.b = (int)(q / g.F);
        p.f = (int)(q - (long long)p.b * g.F);
    } else {
        const long long bc = gf / g.F;
        p.f = (int)(gf - bc * g.F);
        p.b = (int)(bc / g.C);
        p.c = (int)(bc - (long long)p.b * g.C);
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
};

KPR_DEV float to_db(float v, const DbDev& db) {
    // backend.py:186-188: 10*log10(max(x, amin)) - 10*log10(max(amin, ref)), log10 = ln/ln10
    return 10.0f * (logf(fmaxf(v, db.amin)) * 0.43429448190325182765f) - db.ref_term;
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
template <int NC>
KPR_DEV unsigned fetch_frame(const float* __restrict__ x, const Geom& g, const FramePos& p, bool valid,
                             int fl, f2 (&z)[kPts]) {
    constexpr int L = NC / kPts;
    const float* sig = x + p.sig_off;
    const bool interior = valid && p.s0 >= 0 && (p.s0 + 2 * NC) <= g.T && g.win >= 2 * NC;
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

// ------------------------------------------------------------------------------------------
// fused mel kernel
// ------------------------------------------------------------------------------------------
#ifndef KPR_RING_DEPTH
#define KPR_RING_DEPTH 3
#endif
constexpr int kMaxTiles = 64;   // up to 1024 filters
constexpr int kFT = 16;         // frames per workgroup == MFMA N

constexpr int kMaxSegs = kMaxTiles + 4;

struct MelSched {
    int M;                        // number of filters
    int ntiles;                   // ceil(M/16)
    int nseg;                     // segments = (filter tile x contiguous chunk run) pieces
    short klo[kMaxTiles], khi[kMaxTiles];   // padded to whole chunks (multiples of kChunkRows)
    unsigned short chunk0[kMaxTiles];       // first chunk of tile t in the packed filterbank
    // The chunk stream (tiles in natural order) is cut into 4 equal contiguous slices, one per
    // wave; a tile that straddles a cut becomes two segments whose partial results are added in
    // the epilogue (fixed order -> deterministic).
    int wave_seg0[5];                       // segments of wave w: [wave_seg0[w], wave_seg0[w+1])
    unsigned short wave_chunk0[4];          // first chunk of wave w's slice
    unsigned short wave_nchunks[4];         // chunks in wave w's slice
    unsigned char seg_tile[kMaxSegs];       // filter tile of segment i
    // 32-bit on purpose: the MFMA pipeline reads these with a wave-uniform index and they must be
    // SCALAR loads (s_load has no sub-dword form; a vector load inside the counted-vmcnt region
    // would make hipcc drain the whole pipeline -- tests/test_asm_audit.py checks the ISA)
    int seg_nch[kMaxSegs];                  // chunks in segment i
    int seg_k0[kMaxSegs];                   // first magnitude row (k) of segment i
    unsigned char t_s0[kMaxTiles], t_ns[kMaxTiles];   // segments of tile t: [t_s0, t_s0 + t_ns)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kChunkRows = 32;   // MFMA loop granularity: 8 k-steps of 4 rows

__host__ __device__ inline int mel_row_cap(int K) { return (K + kChunkRows - 1) / kChunkRows * kChunkRows; }
__host__ __device__ inline int mel_row_stride(int K) {
    // S >= roundup(K,32) (tile k-ranges are padded to whole chunks and must stay inside the
    // zero-padded row), S % 16 == 2 -> conflict-free MFMA operand reads (banks 2j+h / 18j+h)
    return mel_row_cap(K) + 2;
}

template <int NC>
__global__ __launch_bounds__(256, 2) void k_mel_fused(const float* __restrict__ x, Geom g,
                                                      const float* __restrict__ window,
                                                      const float2* __restrict__ twtab,
                                                      const float* __restrict__ fbp, MelSched sch,
                                                      DbDev db, unsigned* __restrict__ item_stats,
                                                      float* __restrict__ out, int ntiles,
                                                      long long* __restrict__ dbg) {
    constexpr int L = NC / kPts;       // lanes per frame
    constexpr int G = 64 / L;          // frames per wave per round
    constexpr int ROUNDS = kFT / (4 * G);
    constexpr int CH = 8;              // k-steps per software-pipelined MFMA chunk
    static_assert(ROUNDS >= 1, "tile too small for this NC");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = NC + 1;
    const int S = mel_row_stride(K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & (L - 1), grp = lane / L;
    const int jcol = lane & 15, kq = lane >> 4;

    int dbi = 0;
#define KPR_STAMP() do { if (dbg && blockIdx.x == 0 && (tid & 63) == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
    KPR_STAMP();
    FftTw<NC> tw;
    tw.load(twtab, fl);
    WinRegs<NC> wr;
    wr.load(window, g.win, fl, 0.5f);
    KPR_STAMP();

    f2 nz[kPts];
    unsigned nvm;
    {
        const long long gf = (long long)blockIdx.x * kFT + wave * G + grp;
        const bool valid = gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        nvm = fetch_frame<NC>(x, g, p, valid, fl, nz);
    }
    // persistent: a workgroup walks tiles blockIdx.x, +gridDim.x, ... (prologue paid once)
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long tile0 = (long long)tile * kFT;

        // ---- phase 1: FFT + magnitude of 16 frames into smem[j*S + k] ---------------------
        // (nz already holds this tile's first frame: fetched before the loop / during phase 2)
#pragma unroll 1
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int j = rd * (4 * G) + wave * G + grp;       // frame slot in the tile
            float* row = smem + j * S;
            f2 z[kPts];
#pragma unroll
            for (int m = 0; m < kPts; ++m) z[m] = nz[m];
            mask_frame(z, nvm);
            if (rd + 1 < ROUNDS) {                              // prefetch the next frame's samples
                const long long gfn = tile0 + j + 4 * G;
                const bool validn = gfn < g.total_frames;
                FramePos pn = frame_pos(g, validn ? gfn : 0);
                nvm = fetch_frame<NC>(x, g, pn, validn, fl, nz);
            }
#ifdef KPR_FINE_STAMPS
#define KPR_FS() do { if (rd == 1 && tile == (int)blockIdx.x) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); KPR_STAMP(); } } while (0)
#else
#define KPR_FS() do { } while (0)
#endif
            KPR_FS();
            apply_window<NC>(wr, z);
            KPR_FS();
            {
                using Rx = Radix<NC>;
                tw.refresh();
                fft_pass<NC, 1, Rx::r1, 1>(z, tw, row);
                KPR_FS();
                fft_pass<NC, 2, Rx::r2, Rx::r1>(z, tw, row);
                KPR_FS();
                if constexpr (Rx::r3 > 1) fft_pass<NC, 3, Rx::r3, Rx::r1 * Rx::r2>(z, tw, row);
                KPR_FS();
            }
            rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                row[k] = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y);
                if (kp >= 0) row[kp] = __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y);
            });
            KPR_FS();
            // zero pad columns K .. S-1 (read by the last k-step; must be finite)
            for (int k = K + fl; k < S; k += L) row[k] = 0.0f;
            KPR_FS();
#undef KPR_FS
            KPR_STAMP();
        }
        __syncthreads();
        KPR_STAMP();

        // ---- phase 2: D[filter][frame] = sum_k fb[k][filter] * mag[frame][k] on fp32 MFMA --
        // Each wave walks ONE stream of A chunks: the chunks of all its filter tiles back to back
        // (the packed filterbank is laid out in exactly this order), so the software pipeline is
        // filled and drained once per frame tile.  Tile results go to an LDS staging tile
        // dst[frame][filter]; no global store happens inside the pipeline (vmcnt also counts
        // stores and would make the counted waits wait for them).
        {
            float* dpart = smem + kFT * S;               // [nseg][frame 16][filter 16]
            const int total = __builtin_amdgcn_readfirstlane((int)sch.wave_nchunks[wave]);
            int si = __builtin_amdgcn_readfirstlane(sch.wave_seg0[wave]);
            const int si_end = __builtin_amdgcn_readfirstlane(sch.wave_seg0[wave + 1]);
            if (total > 0) {
                int rem = __builtin_amdgcn_readfirstlane(sch.seg_nch[si]);
                const float* brow = smem + jcol * S + kq;
                const float* bcur = brow + __builtin_amdgcn_readfirstlane(sch.seg_k0[si]);
                const float* fa = fbp + ((long long)sch.wave_chunk0[wave] * 2) * 256 + lane * 4;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                constexpr int D = KPR_RING_DEPTH;      // register sets in flight (D-1 chunks ahead)
                f32x4 ar[D][2];
#define KPR_ISSUE(set, chunk)                                                                  \
    do {                                                                                       \
        const float* p_ = fa + (long long)max(0, min((chunk), total - 1)) * 512;               \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(set[0]) : "v"(p_));             \
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(set[1]) : "v"(p_)); \
    } while (0)
    // operand-less wait + sched_barrier: a "+v" wait makes the register allocator copy the
    // in-flight registers BEFORE the wait (stale data); nothing may be scheduled across.
#define KPR_WAIT(n)                                                                            \
    do {                                                                                       \
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n) : "memory");                               \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#define KPR_MMA(set)                                                                           \
    do {                                                                                       \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                     \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(set[g_][0], bcur[16 * g_], acc0, 0, 0, 0);      \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(set[g_][1], bcur[16 * g_ + 4], acc1, 0, 0, 0);  \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(set[g_][2], bcur[16 * g_ + 8], acc0, 0, 0, 0);  \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(set[g_][3], bcur[16 * g_ + 12], acc1, 0, 0, 0); \
        }                                                                                      \
        bcur += kChunkRows;                                                                    \
        if (--rem == 0) {   /* segment done: lane holds D[filter 4kq+r][frame jcol] (partial) */ \
            *reinterpret_cast<f32x4*>(dpart + si * 256 + jcol * 16 + 4 * kq) = acc0 + acc1;    \
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};                                                  \
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};                                                  \
            ++si;                                                                              \
            if (si < si_end) {                                                                 \
                rem = __builtin_amdgcn_readfirstlane(sch.seg_nch[si]);                         \
                bcur = brow + __builtin_amdgcn_readfirstlane(sch.seg_k0[si]);                  \
            }                                                                                  \
        }                                                                                      \
    } while (0)
                // every set has ONE issue point (no PHI copies of in-flight registers): the loop
                // starts D chunks early and only issues during its first trip.  At the wait of
                // step u the D-1 younger sets (2 loads each) may stay in flight.
#pragma unroll 1
                for (int c = -D; c < total; c += D) {
#pragma unroll
                    for (int u = 0; u < D; ++u) {
                        KPR_ISSUE(ar[(u + D - 1) % D], c + u + D - 1);
                        KPR_WAIT(2 * (D - 1));
                        if (c + u >= 0 && c + u < total) KPR_MMA(ar[u]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // drain: no asm load may still be in flight into a register hipcc considers free
                KPR_WAIT(0);
#undef KPR_ISSUE
#undef KPR_WAIT
#undef KPR_MMA
            }
        }
        // per-frame output base / batch index, computed once per tile by 16 lanes (the epilogue's
        // 256 threads would otherwise each do two integer divisions per item)
        long long* fbase = reinterpret_cast<long long*>(smem + kFT * S + sch.nseg * 256);
        int* fitem = reinterpret_cast<int*>(fbase + kFT);
        if (tid < kFT) {
            const long long gfc = tile0 + tid;
            const bool ok = gfc < g.total_frames;
            FramePos pc = frame_pos(g, ok ? gfc : 0);
            fbase[tid] = ok ? spec_base(g, pc, gfc, sch.M) : -1;
            fitem[tid] = pc.b;
        }
        if (tile + (int)gridDim.x < ntiles) {          // next tile's first frame: fetch it now, the
            const long long gf = (long long)(tile + gridDim.x) * kFT + wave * G + grp;   // epilogue
            const bool valid = gf < g.total_frames;                                      // covers
            FramePos p = frame_pos(g, valid ? gf : 0);                                   // the HBM
            nvm = fetch_frame<NC>(x, g, p, valid, fl, nz);                               // latency
        }
        __syncthreads();
        KPR_STAMP();

        // ---- epilogue: dB + fully coalesced stores of the staged 16 x M tile -----------------
        {
            const float* dpart = smem + kFT * S;
            const int q4 = sch.ntiles * 4;                      // float4 groups per frame
            const int ostride = spec_stride(g);
            float wmax = -INFINITY, wmin = INFINITY;
            int my_b = -1;
            for (int it = tid; it < kFT * q4; it += 256) {
                const int j = it / q4, m4 = it - j * q4;
                const long long ob = fbase[j];
                if (ob < 0) continue;                           // frame beyond the end
                const int t = m4 >> 2, off = (m4 & 3) * 4;
                const int s0 = sch.t_s0[t], ns = sch.t_ns[t];
                f32x4 v = *reinterpret_cast<const f32x4*>(dpart + s0 * 256 + j * 16 + off);
                for (int u = 1; u < ns; ++u)                    // partials of a split tile, in order
                    v += *reinterpret_cast<const f32x4*>(dpart + (s0 + u) * 256 + j * 16 + off);
                const int mel = 4 * m4;
                if (db.enabled) {
                    const int b_here = fitem[j];
                    if (my_b >= 0 && my_b != b_here && wmax >= wmin) {   // rare: thread spans items
                        atomicMax(&item_stats[2 * my_b], enc_f(wmax));
                        atomicMin(&item_stats[2 * my_b + 1], enc_f(wmin));
                        wmax = -INFINITY; wmin = INFINITY;
                    }
                    my_b = b_here;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = to_db(v[r], db);
                        if (mel + r < sch.M) { wmax = fmaxf(wmax, v[r]); wmin = fminf(wmin, v[r]); }
                    }
                }
                float* outc = out + ob;
                if (!g.out_cl && (sch.M & 3) == 0 && mel + 3 < sch.M) {
                    *reinterpret_cast<float4*>(outc + mel) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (mel + r < sch.M) outc[(long long)(mel + r) * ostride] = v[r];
                }
            }
            if (db.enabled) {
                // one atomic pair per wave when the whole wave works on one batch item
                const int b0 = __builtin_amdgcn_readfirstlane(my_b);
                const bool uniform = __all(my_b == b0);
                if (uniform && b0 >= 0) {
                    for (int o = 32; o > 0; o >>= 1) {
                        wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
                        wmin = fminf(wmin, __shfl_xor(wmin, o, 64));
                    }
                    if (lane == 0 && wmax >= wmin) {
                        atomicMax(&item_stats[2 * b0], enc_f(wmax));
                        atomicMin(&item_stats[2 * b0 + 1], enc_f(wmin));
                    }
                } else if (my_b >= 0 && wmax >= wmin) {
                    atomicMax(&item_stats[2 * my_b], enc_f(wmax));
                    atomicMin(&item_stats[2 * my_b + 1], enc_f(wmin));
                }
            }
        }
        // no barrier here: the next tile's phase 1 only writes mag rows (every MFMA read of them
        // is behind the barrier above); dst is rewritten only after the next phase-1 barrier
        KPR_STAMP();
    }
#undef KPR_STAMP
}


// ------------------------------------------------------------------------------------------
// fused mel kernel, wave-specialised variant (the default whenever it fits in LDS):
// 768 threads = 12 waves, ONE workgroup per CU, persistent over tiles of 16 frames.
//   waves 0..7   producers: frame fetch + window + rFFT + |X| of tile i into mag[i & 1]
//                (VALU + LDS work; two of them per SIMD keep the vector ALU busy)
//   waves 8..11  consumers: banded MFMA GEMM + dB + coalesced stores of tile i-1 from
//                mag[(i-1) & 1] (matrix pipe + HBM work; one per SIMD)
// ONE __syncthreads per tile hands the buffers over, so the MFMA / epilogue phases of the ring
// kernel (a third of its time, during which the vector ALU idles) run UNDER the next tile's FFTs.
// The four consumer waves need one more sync between their GEMM slices and the epilogue (partial
// tiles are summed there); gfx950 has no named barriers, so that is an LDS counter they spin on
// (all four are resident by construction).  The window lives in LDS (ds_read_b64 at use) to keep
// the producers under the 168-VGPR budget of 3 waves/SIMD.
//   LDS = mag[2][16][S] | dpart[nseg][16x16] | fbase[16] fitem[16] sync | window[NC] (f2)
// ------------------------------------------------------------------------------------------
// one frame of k_mel_ws: mask + window the prefetched samples, prefetch this wave's next frame,
// FFT, pairing, |X| into `row` (G == 1: the whole wave owns the frame)
#ifdef KPR_WS_XOR
template <int NC> struct WsSwzFor { typedef SwzXor type; };
#else
template <int NC> struct WsSwzFor { typedef typename SwzFor<NC>::type type; };
#endif
// one ticket of k_mel_ws = G frames (one per lane group): gf_next is the first frame of the wave's
// next ticket (wave-uniform), lane group grp takes frame gf_next + grp
template <int NC>
KPR_DEV void ws_frame(const float* __restrict__ x, const Geom& g, FftTw<NC, typename WsSwzFor<NC>::type>& tw,
                      const f2* winl, float* row, int gf_next, int f_end, int fl, int grp, int lane, int K, int S,
                      f2 (&nz)[kPts], unsigned& nvm, long long* dbgw, int& dbi) {
    constexpr int L = NC / kPts;
    typedef typename WsSwzFor<NC>::type WsSwz;
#ifdef KPR_FINE_STAMPS
#define KPR_FS() do { if (dbgw && lane == 0 && dbi < 32) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dbgw[dbi++] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define KPR_FS() do { (void)dbgw; (void)dbi; } while (0)
#endif
    KPR_FS();
    f2 z[kPts];
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = nz[m];
    mask_frame(z, nvm);
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = pmul(z[m], winl[fl + L * m]);
    KPR_FS();
    if (gf_next < f_end) {                                  // wave-uniform
        const bool validn = gf_next + grp < f_end;
        FramePos pn = frame_pos(g, validn ? gf_next + grp : gf_next);
        nvm = fetch_frame<NC>(x, g, pn, validn, fl, nz);
    }
    {
        using Rx = Radix<NC>;
        tw.refresh();
        KPR_FS();
        fft_pass<NC, 1, Rx::r1, 1, WsSwz>(z, tw, row);
        KPR_FS();
        fft_pass<NC, 2, Rx::r2, Rx::r1, WsSwz>(z, tw, row);
        KPR_FS();
        if constexpr (Rx::r3 > 1) fft_pass<NC, 3, Rx::r3, Rx::r1 * Rx::r2, WsSwz>(z, tw, row);
        KPR_FS();
    }
    rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
        row[k] = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y);
        if (kp >= 0) row[kp] = __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y);
    });
    // zero pad columns K .. S-1 (read by the last k-step; must be finite)
    for (int k = K + fl; k < S; k += L) row[k] = 0.0f;
    KPR_FS();
#undef KPR_FS
}

// loader producers of k_mel_ws<NC, true> (see there): tickets of RPT rows, PER loads of 64 floats per
// row, two register sets (the next ticket's rows are in flight while the current ones are written)
template <int RPT, int PER>
KPR_DEV void ws_loader(const float* __restrict__ x, int K, int S, int f_begin, int n_total, float* smem,
                       int* sync, int lane) {
    static_assert(kFT % RPT == 0, "a ticket never straddles two tiles");
    const int kend = mel_row_cap(K) + 2;               // columns the consumers may read
    const int n_tickets = (n_total + RPT - 1) / RPT;
#define WL_TICKET(dst_)                                                                          \
    do {                                                                                         \
        int v_ = 0;                                                                              \
        if (lane == 0) v_ = __hip_atomic_fetch_add(&sync[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        dst_ = __builtin_amdgcn_readfirstlane(v_);                                               \
    } while (0)
#define WL_LOAD(set_, n_)                                                                        \
    do {                                                                                         \
        _Pragma("unroll") for (int r = 0; r < RPT; ++r) {                                        \
            const float* src_ = x + (long long)(f_begin + min(RPT * (n_) + r, n_total - 1)) * K; \
            _Pragma("unroll") for (int u = 0; u < PER; ++u)                                      \
                if (64 * u < K) set_[r][u] = src_[min(lane + 64 * u, K - 1)];  /* wave-uniform guard */ \
        }                                                                                        \
    } while (0)
#define WL_STORE(set_, n_)                                                                       \
    do {                                                                                         \
        const int q0_ = RPT * (n_), t_ = q0_ >> 4;                                               \
        /* buffer t & 1 is free once all four consumers have read tile t - 2 */                  \
        if (t_ >= 2)                                                                             \
            while (__hip_atomic_load(&sync[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (t_ - 1)) \
                __builtin_amdgcn_s_sleep(2);                                                     \
        _Pragma("unroll") for (int r = 0; r < RPT; ++r) {                                        \
            if (q0_ + r < n_total) {                                                             \
                float* row_ = smem + (t_ & 1) * (kFT * S) + ((q0_ & (kFT - 1)) + r) * S;         \
                _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                \
                    const int k_ = lane + 64 * u;                                                \
                    if (64 * u < kend && k_ < kend) row_[k_] = (k_ < K) ? set_[r][u] : 0.0f;     \
                }                                                                                \
                for (int k_ = lane + 64 * PER; k_ < kend; k_ += 64) row_[k_] = 0.0f;             \
            }                                                                                    \
        }                                                                                        \
        if (lane == 0)                                                                           \
            __hip_atomic_fetch_add(&sync[t_ & 1], min(RPT, n_total - q0_), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
    } while (0)
    float va[RPT][PER], vb[RPT][PER];
    int n;
    WL_TICKET(n);
    if (n < n_tickets) WL_LOAD(va, n);
#pragma unroll 1
    while (n < n_tickets) {
        int n2;
        WL_TICKET(n2);
        if (n2 < n_tickets) WL_LOAD(vb, n2);
        WL_STORE(va, n);
        n = n2;
        if (n >= n_tickets) break;
        WL_TICKET(n2);
        if (n2 < n_tickets) WL_LOAD(va, n2);
        WL_STORE(vb, n);
        n = n2;
    }
#undef WL_TICKET
#undef WL_LOAD
#undef WL_STORE
}

#ifndef KPR_WS_CONS_PRIO
#define KPR_WS_CONS_PRIO 3
#endif
constexpr int kWsProd = 8;
constexpr int kWsThreads = 768;

// magnitude row stride of k_mel_ws: the row doubles as the skewed FFT exchange row (WsSwz needs
// NC + NC/32 + 24 words) and must keep S % 16 == 2 for the MFMA operand reads
__host__ __device__ inline int mel_ws_row_stride(int K) {
    const int NC = K - 1;
    bool skew = NC == 1024 || NC == 512;
#ifdef KPR_WS_XOR
    skew = false;
#endif
    if (!skew) return mel_row_stride(K);
    const int need = std::max(mel_row_cap(K), SwzSkew::row_words(NC));
    return (need + 13) / 16 * 16 + 2;
}

__host__ __device__ inline size_t mel_ws_lds_bytes(int NC, int nseg) {
    const int S = mel_ws_row_stride(NC + 1);
    return sizeof(float) * ((size_t)2 * kFT * S + (size_t)nseg * 256) +
           kFT * (sizeof(long long) + sizeof(int)) + 8 * sizeof(int) + (size_t)NC * 2 * sizeof(float);
}

// FROM_MAG = true: the same kernel as a stand-alone ApplyFilterbank -- `x` holds magnitude rows
// (g.K floats per frame, contiguous) and the producers merely copy them into the tile; consumers,
// counters, tickets and the epilogue are shared.
template <int NC, bool FROM_MAG>
__global__ __launch_bounds__(kWsThreads) void k_mel_ws(const float* __restrict__ x, Geom g,
                                                       const float* __restrict__ window,
                                                       const float2* __restrict__ twtab,
                                                       const float* __restrict__ fbp, MelSched sch,
                                                       DbDev db, unsigned* __restrict__ item_stats,
                                                       float* __restrict__ out, int ntiles,
                                                       long long* __restrict__ dbg) {
    constexpr int L = NC / kPts;       // lanes per frame
    constexpr int G = 64 / L;          // frames per wave per round
    typedef typename WsSwzFor<NC>::type WsSwz;
    static_assert(!FROM_MAG || G == 1, "loader producers copy one row per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = FROM_MAG ? g.K : NC + 1;
    const int S = mel_ws_row_stride(NC + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    float* dpart = smem + 2 * kFT * S;                                   // [nseg][frame 16][filter 16]
    long long* fbase = reinterpret_cast<long long*>(dpart + sch.nseg * 256);
    int* fitem = reinterpret_cast<int*>(fbase + kFT);
    // monotonic LDS counters: sync[0], sync[1] rows written into mag buffer 0 / 1 (producers),
    // sync[2] consumer waves done reading a tile, sync[3] consumer-group barrier, sync[4] frame tickets
    int* sync = fitem + kFT;
    f2* winl = reinterpret_cast<f2*>(sync + 8);                          // (0.5 w[2n], 0.5 w[2n+1])
#define WS_SIGNAL_N(p_, n_) do { if (lane == 0) __hip_atomic_fetch_add((p_), (n_), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define WS_SIGNAL(p_) WS_SIGNAL_N(p_, 1)
#define WS_SPIN_UNTIL(p_, n_, nap_) do { while (__hip_atomic_load((p_), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (n_)) __builtin_amdgcn_s_sleep(nap_); } while (0)

    int dbi = 0;
    // development aid: dbg[12*32] selects the workgroup whose waves record cycle stamps
    const bool stamp_me = dbg && (long long)blockIdx.x == dbg[12 * 32];
#define KPR_STAMP() do { if (stamp_me && lane == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
    KPR_STAMP();
    if constexpr (!FROM_MAG) {
        for (int i = tid; i < NC; i += kWsThreads) {
            const int n = 2 * i;
            const float a = window[min(n, g.win - 1)], b = window[min(n + 1, g.win - 1)];
            winl[i] = f2{(n < g.win) ? 0.5f * a : 0.0f, (n + 1 < g.win) ? 0.5f * b : 0.0f};
        }
    }
    if (tid < 8) sync[tid] = 0;
    // A workgroup owns a CONTIGUOUS run of frames [f_begin, f_end), cut at ticket granularity (G
    // frames), so the runs differ by at most one ticket; it walks the run in tiles of 16 frames, the
    // last one possibly short.  Contiguous, not grid-strided: the next tile's samples overlap the
    // current one's and sit in the same pages.
    // (frame numbers fit in 32 bits here: the launcher falls back to k_mel_fused otherwise)
    const long long ngroups = (g.total_frames + G - 1) / G;
    const int f_begin = (int)(ngroups * blockIdx.x / gridDim.x * G);
    const int f_end = (int)min(g.total_frames, ngroups * (blockIdx.x + 1) / gridDim.x * G);
    const int my = (f_end - f_begin + kFT - 1) / kFT;             // my tiles
    (void)ntiles;
    __syncthreads();

#define KPR_PREFETCH(gf_)                                                                       \
    do {                                                                                        \
        const bool v_ = (gf_) + grp < f_end;                                                    \
        FramePos p_ = frame_pos(g, v_ ? (gf_) + grp : (gf_));                                   \
        nvm = fetch_frame<NC>(x, g, p_, v_, fl, nz);                                            \
    } while (0)
#ifdef KPR_FINE_STAMPS   /* stamps of workgroup 0 in tile 2 only (fits the 32-slot row) */
#define KPR_DO_FRAME(row_, gf_next_) ws_frame<NC>(x, g, tw, winl, (row_), (gf_next_), f_end, fl, grp, lane, K, S, nz, nvm, (stamp_me && t == 2) ? dbg + wave * 32 : nullptr, dbi)
#else
#define KPR_DO_FRAME(row_, gf_next_) ws_frame<NC>(x, g, tw, winl, (row_), (gf_next_), f_end, fl, grp, lane, K, S, nz, nvm, nullptr, dbi)
#endif

    if (wave < kWsProd) {
        // ================================ producers ==========================================
        const int n_total = f_end - f_begin;
#define WS_TICKET(dst_)                                                                          \
    do {                                                                                         \
        int v_ = 0;                                                                              \
        if (lane == 0) v_ = __hip_atomic_fetch_add(&sync[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        dst_ = __builtin_amdgcn_readfirstlane(v_);                                               \
    } while (0)
        if constexpr (FROM_MAG) {
            // loader producers: row n of the run -> row n & 15 of tile n >> 4 (coalesced dword loads:
            // a row of K floats starts at an arbitrary 4-byte boundary).  A ticket is RPT consecutive
            // rows, short rows travel four or two at a time, and the next ticket's loads are issued
            // before the current rows are written: with one 201-float row per ticket and nothing in
            // flight behind it (the first version) a wave moved one row per HBM round trip.
            if (K <= 256) ws_loader<4, 4>(x, K, S, f_begin, n_total, smem, sync, lane);
            else if (K <= 512) ws_loader<2, 8>(x, K, S, f_begin, n_total, smem, sync, lane);
            else ws_loader<1, (NC + 1 + 63) / 64>(x, K, S, f_begin, n_total, smem, sync, lane);
        } else {
        const int fl = lane & (L - 1), grp = lane / L;     // lane group grp owns frame G*ticket + grp
        FftTw<NC, WsSwz> tw;
        tw.load(twtab, fl);
        f2 nz[kPts];
        unsigned nvm = 0xffffffffu;
        // Frames are handed out DYNAMICALLY (an LDS ticket counter): ticket n = the G frames
        // G*n .. G*n + G-1 of the run, frame q going to row q & 15 of tile q >> 4.  With a static
        // assignment the four older producer waves, which win the SIMD's issue arbitration, finish
        // early and idle a quarter of every tile; now they simply take more tickets.  A wave holds
        // its next ticket while it works on the current one, so the sample prefetch still runs one
        // ticket ahead.
        // (Also tried: some frames done by the consumers after their GEMM + epilogue -- 13 %
        // slower, a third FFT wave per SIMD does not raise the VALU utilisation.)
        const int n_tickets = (n_total + G - 1) / G;
        int n;
        WS_TICKET(n);
        if (n < n_tickets) KPR_PREFETCH(f_begin + G * n);
        KPR_STAMP();
#pragma unroll 1
        while (n < n_tickets) {
            int n2;
            WS_TICKET(n2);
            const int q0 = G * n;                                     // first frame of the ticket
            const int t = q0 >> 4, j = (q0 & (kFT - 1)) + grp;
            // buffer t & 1 is free once all four consumers have read tile t - 2
            if (t >= 2) WS_SPIN_UNTIL(&sync[2], 4 * (t - 1), 2);
            KPR_DO_FRAME(smem + (t & 1) * (kFT * S) + j * S, (n2 < n_tickets) ? f_begin + G * n2 : f_end);
            WS_SIGNAL_N(&sync[t & 1], min(G, n_total - q0));          // rows written into this buffer
            KPR_STAMP();
            n = n2;
        }
        }
#undef WS_TICKET
    } else {
        // ================================ consumers ==========================================
        const int cw = wave - kWsProd, ctid = tid - kWsProd * 64;
        const int jcol = lane & 15, kq = lane >> 4;
        // The consumers issue few instructions (one MFMA per 32 matrix-pipe cycles) but each one
        // competes for the SIMD's VALU issue port with two producers that always have work ready;
        // at equal priority the port goes to the older (producer) waves and the GEMM runs 2.5x
        // slower than alone.  Raise the consumers' priority.
        __builtin_amdgcn_s_setprio(KPR_WS_CONS_PRIO);
        // this wave's slice of the chunk stream (at most 64 chunks: one lane of cinfo per chunk)
        const int total = __builtin_amdgcn_readfirstlane((int)sch.wave_nchunks[cw]);
        const float* fa = fbp + ((long long)sch.wave_chunk0[cw] * 2) * 256 + lane * 4;
        int cinfo = 0;
        {
            int cbase = 0;
            for (int sj = sch.wave_seg0[cw]; sj < sch.wave_seg0[cw + 1]; ++sj) {
                const int n = sch.seg_nch[sj], r = lane - cbase;
                if (r >= 0 && r < n)
                    cinfo = (4 * (sch.seg_k0[sj] + kChunkRows * r)) | ((r == n - 1) ? 0x10000 : 0) | (sj << 17);
                cbase += n;
            }
        }
#pragma unroll 1
        for (int it = 1; it <= my; ++it) {                  // it - 1 = tile index
            {
                const int tile0 = f_begin + (it - 1) * kFT;
                const float* mag = smem + ((it - 1) & 1) * (kFT * S);
                // all rows of the tile written?  (rows of this buffer so far: 16 per earlier tile)
                // (poll rarely and at low priority: the producers need the issue slots)
                __builtin_amdgcn_s_setprio(0);
                WS_SPIN_UNTIL(&sync[(it - 1) & 1], kFT * ((it - 1) >> 1) + min(kFT, f_end - tile0), 8);
                __builtin_amdgcn_s_setprio(KPR_WS_CONS_PRIO);
                KPR_STAMP();
                // per-frame output base / batch index, once per tile by 16 lanes
                if (ctid < kFT) {
                    const int gfc = tile0 + ctid;
                    const bool ok = gfc < f_end;
                    FramePos pc = frame_pos(g, ok ? gfc : 0);
                    fbase[ctid] = ok ? spec_base(g, pc, gfc, sch.M) : -1;
                    fitem[ctid] = pc.b;
                }
                // ---- D[filter][frame] = sum_k fb[k][filter] * mag[frame][k] on fp32 MFMA ------------
                // One software pipeline per wave over its slice of the chunk stream, BOTH operands
                // prefetched D-1 chunks ahead by inline-asm loads into static register sets: A (packed
                // filterbank, L2) with global_load_dwordx4 / vmcnt, B (magnitudes, LDS) with
                // ds_read2_b32 / lgkmcnt.  The producers keep the LDS pipeline busy, so an LDS read
                // issued at its use costs ~1k cycles here; LDS returns in order, and anything the
                // compiler adds to lgkmcnt (scalar loads, the dpart store) only makes the counted wait
                // more conservative.
                {
                    if (total > 0) {
                        const unsigned bbase = (unsigned)(uintptr_t)(mag + jcol * S + kq);   // LDS bytes
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        constexpr int D = KPR_RING_DEPTH;
                        f32x4 ar[D][2];
                        f2 br[D][4];
                        // chunk n of the slice: cinfo lane n = (k0 * 4 bytes) | last-of-segment << 16
                        // | segment id << 17; v_readlane with a wave-uniform index, no memory op
#define KPR_ISSUE(sa, sb, chunk)                                                               \
    do {                                                                                       \
        const int n_ = max(0, min((chunk), total - 1));                                        \
        const float* p_ = fa + (long long)n_ * 512;                                            \
        const unsigned b_ = bbase + (unsigned)(__builtin_amdgcn_readlane(cinfo, n_) & 0xffff); \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sa[0]) : "v"(p_));              \
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(sa[1]) : "v"(p_));  \
        asm volatile("ds_read2_b32 %0, %1 offset1:4" : "=v"(sb[0]) : "v"(b_));                \
        asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:12" : "=v"(sb[1]) : "v"(b_));     \
        asm volatile("ds_read2_b32 %0, %1 offset0:16 offset1:20" : "=v"(sb[2]) : "v"(b_));    \
        asm volatile("ds_read2_b32 %0, %1 offset0:24 offset1:28" : "=v"(sb[3]) : "v"(b_));    \
    } while (0)
#define KPR_WAIT(nv, nl)                                                                       \
    do {                                                                                       \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"i"(nv), "i"(nl) : "memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#define KPR_MMA(sa, sb, chunk)                                                                 \
    do {                                                                                       \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][0], sb[0].x, acc0, 0, 0, 0);         \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][1], sb[0].y, acc1, 0, 0, 0);         \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][2], sb[1].x, acc0, 0, 0, 0);         \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][3], sb[1].y, acc1, 0, 0, 0);         \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][0], sb[2].x, acc0, 0, 0, 0);         \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][1], sb[2].y, acc1, 0, 0, 0);         \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][2], sb[3].x, acc0, 0, 0, 0);         \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][3], sb[3].y, acc1, 0, 0, 0);         \
        const int i_ = __builtin_amdgcn_readlane(cinfo, (chunk));                              \
        if (i_ & 0x10000) { /* segment done: lane holds D[filter 4kq+r][frame jcol] (partial) */ \
            *reinterpret_cast<f32x4*>(dpart + (i_ >> 17) * 256 + jcol * 16 + 4 * kq) = acc0 + acc1; \
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};                                                  \
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};                                                  \
        }                                                                                      \
    } while (0)
                        // every set has ONE issue point; the loop starts D chunks early and only
                        // issues during its first trip.  At the wait of step u the D-1 younger sets
                        // (2 global + 4 LDS loads each) may stay in flight.
#pragma unroll 1
                        for (int c = -D; c < total; c += D) {
#pragma unroll
                            for (int u = 0; u < D; ++u) {
                                KPR_ISSUE(ar[(u + D - 1) % D], br[(u + D - 1) % D], c + u + D - 1);
                                KPR_WAIT(2 * (D - 1), 4 * (D - 1));
                                if (c + u >= 0 && c + u < total) KPR_MMA(ar[u], br[u], c + u);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        KPR_WAIT(0, 0);
#undef KPR_ISSUE
#undef KPR_WAIT
#undef KPR_MMA
                    }
                }
                KPR_STAMP();
                // ---- consumer-group barrier (4 waves): LDS counter, monotonically increasing ----
                WS_SIGNAL(&sync[2]);                         // this wave is done reading the mag buffer
                WS_SIGNAL(&sync[3]);
                WS_SPIN_UNTIL(&sync[3], 8 * it - 4, 1);      // all four GEMM slices are in dpart
                KPR_STAMP();
                // ---- epilogue: dB + fully coalesced stores of the staged 16 x M tile ----------
                {
                    const int q4 = sch.ntiles * 4;                      // float4 groups per frame
                    const int ostride = spec_stride(g);
                    float wmax = -INFINITY, wmin = INFINITY;
                    int my_b = -1;
                    for (int e = ctid; e < kFT * q4; e += 256) {
                        const int j = e / q4, m4 = e - j * q4;
                        const long long ob = fbase[j];
                        if (ob < 0) continue;                           // frame beyond the end
                        const int t = m4 >> 2, off = (m4 & 3) * 4;
                        const int s0 = sch.t_s0[t], ns = sch.t_ns[t];
                        f32x4 v = *reinterpret_cast<const f32x4*>(dpart + s0 * 256 + j * 16 + off);
                        for (int u = 1; u < ns; ++u)                    // partials of a split tile, in order
                            v += *reinterpret_cast<const f32x4*>(dpart + (s0 + u) * 256 + j * 16 + off);
                        const int mel = 4 * m4;
                        if (db.enabled) {
                            const int b_here = fitem[j];
                            if (my_b >= 0 && my_b != b_here && wmax >= wmin) {   // rare: thread spans items
                                atomicMax(&item_stats[2 * my_b], enc_f(wmax));
                                atomicMin(&item_stats[2 * my_b + 1], enc_f(wmin));
                                wmax = -INFINITY; wmin = INFINITY;
                            }
                            my_b = b_here;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = to_db(v[r], db);
                                if (mel + r < sch.M) { wmax = fmaxf(wmax, v[r]); wmin = fminf(wmin, v[r]); }
                            }
                        }
                        float* outc = out + ob;
                        if (!g.out_cl && (sch.M & 3) == 0 && mel + 3 < sch.M) {
                            *reinterpret_cast<float4*>(outc + mel) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (mel + r < sch.M) outc[(long long)(mel + r) * ostride] = v[r];
                        }
                    }
                    if (db.enabled) {
                        const int b0 = __builtin_amdgcn_readfirstlane(my_b);
                        const bool uniform = __all(my_b == b0);
                        if (uniform && b0 >= 0) {
                            for (int o = 32; o > 0; o >>= 1) {
                                wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
                                wmin = fminf(wmin, __shfl_xor(wmin, o, 64));
                            }
                            if (lane == 0 && wmax >= wmin) {
                                atomicMax(&item_stats[2 * b0], enc_f(wmax));
                                atomicMin(&item_stats[2 * b0 + 1], enc_f(wmin));
                            }
                        } else if (my_b >= 0 && wmax >= wmin) {
                            atomicMax(&item_stats[2 * my_b], enc_f(wmax));
                            atomicMin(&item_stats[2 * my_b + 1], enc_f(wmin));
                        }
                    }
                }
                // dpart / fbase are rewritten by the next tile: wait until all four waves are done
                WS_SIGNAL(&sync[3]);
                WS_SPIN_UNTIL(&sync[3], 8 * it, 1);
                KPR_STAMP();
            }
        }
    }
#undef KPR_STAMP
#undef WS_SIGNAL_N
#undef WS_SIGNAL
#undef WS_SPIN_UNTIL
#undef KPR_PREFETCH
#undef KPR_DO_FRAME
}


// ------------------------------------------------------------------------------------------
// Thin GEMM: out[rows][N] = A[rows][K] x B[K][N] for small K and N (LogmelToMFCC: 80 x 13,
// 128 x 20, ...; any narrow ApplyFilterbank matrix on contiguous rows).  HBM-bound: A is read once
// with 16-byte loads (lane (m, kq) takes A[row m][16j + 4kq .. +3]; those four values feed four
// MFMA k-steps, the B fragments in LDS are stored in the matching order), every wave owns 16 rows
// per step and keeps all N-tiles' accumulators in registers.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_thin_gemm(const float* __restrict__ a, long long rows, int K,
                                                   const float* __restrict__ bm, int N,
                                                   float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int J = (K + 15) / 16;
    f32x4* bfrag = reinterpret_cast<f32x4*>(smem);               // [NT][J][64]
    for (int idx = threadIdx.x; idx < NT * J * 64; idx += blockDim.x) {
        const int l = idx & 63, j = (idx >> 6) % J, nt = (idx >> 6) / J;
        const int n = nt * 16 + (l & 15), kq = l >> 4;
        f32x4 v;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = 16 * j + 4 * kq + s4;
            v[s4] = (k < K && n < N) ? bm[(long long)k * N + n] : 0.0f;
        }
        bfrag[idx] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const long long nblk = (rows + 15) / 16;
    for (long long rb = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); rb < nblk; rb += (long long)gridDim.x * 4) {
        const long long row = rb * 16 + m;
        const float* ar = a + min(row, rows - 1) * K;            // rows past the end: clamped, never stored
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < J; ++j) {
            const int k0 = 16 * j + 4 * kq;
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (k0 + 3 < K) av = *reinterpret_cast<const f32x4*>(ar + k0);      // K % 4 == 0: all or nothing
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 bv = bfrag[(nt * J + j) * 64 + lane];
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc[nt], 0, 0, 0);
            }
        }
        // lane holds D[row 4*kq + r][col m]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long orow = rb * 16 + 4 * kq + r;
                if (orow < rows && n < N) out[orow * N + n] = acc[nt][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Frame / Energy / Delta (kapre/signal.py:22-213, time_frequency.py:563-644): bandwidth kernels
// ------------------------------------------------------------------------------------------
struct FrameArgs {
    long long n_sig;        // batch * channels
    long long T;
    int C, F, L, hop;
    int cl;                 // waveform (b, t, c) and frames (b, f, l, c) if 1; (b, c, t) / (b, c, f, l) if 0
    float pad_value;
};

// Output-stationary copy: every thread produces VEC consecutive output floats (one 16-byte store when
// VEC = 4) of one row; a row = one frame of one batch item (all channels, channels_last: out[b][f]
// is L*C contiguous floats and so is its source) or of one signal (channels_first).  The source of
// a float4 is only 4-byte aligned in general (hop is arbitrary), so it is read as four dwords --
// still fully coalesced across the wave.
template <int VEC>
__global__ __launch_bounds__(256) void k_frame(const float* __restrict__ x, FrameArgs a,
                                               float* __restrict__ out, long long nrows) {
    const int rowlen = a.cl ? a.L * a.C : a.L;
    const int per_row = rowlen / VEC;                           // VEC == 4 only when rowlen % 4 == 0
    const long long total = nrows * per_row;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / per_row;
        const int i = (int)(e - row * per_row) * VEC;
        const long long bq = row / a.F;                          // batch item (cl) or signal b*C + c (cf)
        const int f = (int)(row - bq * a.F);
        const long long t0 = (long long)f * a.hop;
        const float* src = a.cl ? x + (bq * a.T + t0) * a.C : x + bq * a.T + t0;
        const long long avail = (a.T - t0) * (a.cl ? a.C : 1);   // valid elements from src on
        float v[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) v[u] = (i + u < avail) ? src[i + u] : a.pad_value;
        float* dst = out + row * rowlen + i;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else dst[0] = v[0];
    }
}

// Energy: every sample is read ONCE.  With L = q*hop + r a frame is q whole hop-blocks plus the first
// r samples of the next one, so a workgroup (4 waves) that owns kEnFrames consecutive frames of one
// signal first reduces each of its kEnFrames + q hop-blocks to two numbers in LDS -- the block's
// sum of squares and the sum of its first r squares -- and then adds q + 1 of them per output.
// Samples beyond the end of the signal count as pad_value (tf.signal.frame pad_end semantics).
constexpr int kEnFrames = 64;

__global__ __launch_bounds__(256) void k_energy(const float* __restrict__ x, FrameArgs a, float scale,
                                                float* __restrict__ out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = a.L / a.hop, r = a.L - q * a.hop;
    const int nblk = kEnFrames + q + (r ? 1 : 0);               // hop-blocks this workgroup needs
    float* full = smem;                                          // [nblk]
    float* pre = smem + nblk;                                    // [nblk]
    const long long total = a.n_sig * chunks;
    for (long long wg = blockIdx.x; wg < total; wg += gridDim.x) {
        const long long sig = wg / chunks;                       // b*C + c  (cf)  /  b, c from it (cl)
        const int f0 = (int)(wg - sig * chunks) * kEnFrames;
        const long long b = sig / a.C;
        const int c = (int)(sig - b * a.C);
        const float* src = a.cl ? x + b * a.T * a.C + c : x + sig * a.T;
        const int es = a.cl ? a.C : 1;
        for (int i = wave; i < nblk; i += 4) {
            const long long t0 = (long long)(f0 + i) * a.hop;
            float s_all = 0.0f, s_pre = 0.0f;
            for (int l = lane; l < a.hop; l += 64) {
                const long long t = t0 + l;
                const float v = src[min(t, a.T - 1) * es];       // unconditional load, then select
                const float w = (t < a.T) ? v : a.pad_value;
                const float w2 = w * w;
                s_all += w2;
                s_pre += (l < r) ? w2 : 0.0f;
            }
            for (int sft = 32; sft > 0; sft >>= 1) {
                s_all += __shfl_xor(s_all, sft, 64);
                s_pre += __shfl_xor(s_pre, sft, 64);
            }
            if (lane == 0) { full[i] = s_all; pre[i] = s_pre; }
        }
        __syncthreads();
        if (threadIdx.x < kEnFrames && f0 + (int)threadIdx.x < a.F) {
            const int f = threadIdx.x;
            float acc = 0.0f;
            for (int k = 0; k < q; ++k) acc += full[f + k];
            if (r) acc += pre[f + q];
            const long long fo = f0 + f;
            out[a.cl ? (b * a.F + fo) * a.C + c : sig * a.F + fo] = scale * acc;
        }
        __syncthreads();
    }
}

// x viewed as (outer, T, inner): channels_last (b, t, f, c): outer = b, inner = f*c;
// channels_first (b, c, t, f): outer = b*c, inner = f
__device__ __forceinline__ long long delta_src_index(long long t, long long T, int mode) {
    if (t >= 0 && t < T) return t;
    if (mode == KPR_PAD_CONSTANT) return -1;
    if (T == 1) return 0;
    if (mode == KPR_PAD_SYMMETRIC) {             // ... 1 0 | 0 1 2 ... T-1 | T-1 T-2 ...
        const long long p = 2 * T;
        long long m = t % p; if (m < 0) m += p;
        return m < T ? m : p - 1 - m;
    }
    const long long p = 2 * T - 2;               // reflect: ... 2 1 | 0 1 ... T-1 | T-2 ...
    long long m = t % p; if (m < 0) m += p;
    return m < T ? m : p - m;
}

// every thread produces VEC consecutive outputs along `inner` of one (o, t) row; the 2n neighbour
// rows are read with the same vector width (they are L1/L2 hits for all but the first reader)
template <int VEC>
__global__ __launch_bounds__(256) void k_delta(const float* __restrict__ x, long long outer, long long T,
                                               long long inner, int n, int mode, float inv_denom,
                                               float* __restrict__ out) {
    typedef float vf __attribute__((ext_vector_type(VEC)));
    const long long per_row = inner / VEC;
    const long long total = outer * T * per_row;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long q = e / per_row;
        const long long i = (e - q * per_row) * VEC;
        const long long o = q / T;
        const long long t = q - o * T;
        const float* base = x + o * T * inner + i;
        vf acc = {};
        const bool interior = t - n >= 0 && t + n < T;
        for (int j = 1; j <= n; ++j) {            // pairs (+j, -j): j * (x[t+j] - x[t-j])
            long long ip = t + j, im = t - j;
            if (!interior) { ip = delta_src_index(ip, T, mode); im = delta_src_index(im, T, mode); }
            vf vp = {}, vm = {};
            if (ip >= 0) vp = *reinterpret_cast<const vf*>(base + ip * inner);
            if (im >= 0) vm = *reinterpret_cast<const vf*>(base + im * inner);
            acc += (float)j * (vp - vm);
        }
        *reinterpret_cast<vf*>(out + q * inner + i) = acc * inv_denom;
    }
}

// ------------------------------------------------------------------------------------------
// stand-alone STFT kernel (complex / magnitude / phase epilogue)
// ------------------------------------------------------------------------------------------
#ifdef KPR_STFT_NT
#define KPR_STFT_STORE(p_, v_) __builtin_nontemporal_store((v_), (p_))
#else
#define KPR_STFT_STORE(p_, v_) (*(p_) = (v_))
#endif
#ifndef KPR_STFT_WAVES
#define KPR_STFT_WAVES 4
#endif
#ifndef KPR_STFT_OCC
#define KPR_STFT_OCC 2          /* workgroups (4 waves each) per CU the register budget is sized for */
#endif
// LDS words of one k_stft workgroup: 4*G spectrum/exchange rows + window + ticket counter
__host__ __device__ inline size_t stft_lds_bytes(int NC) {
    const int G = 64 / (NC / kPts);
    return sizeof(float) * ((size_t)KPR_STFT_WAVES * G * (2 * NC + 8) + 2 * (size_t)NC) + 4 * sizeof(int);
}

// MODE (KPR_OUT_*) and the output layout are compile-time: the complex / channels_first instance
// then fits the 168-VGPR budget of three workgroups per CU (the phase epilogue alone needs ~60 more)
template <int NC, int MODE, bool OUT_CL>
__global__ __launch_bounds__(64 * KPR_STFT_WAVES, (MODE == KPR_OUT_PHASE || OUT_CL) ? 2 : 3) void k_stft(const float* __restrict__ x, Geom g,
                                                 const float* __restrict__ window,
                                                 const float2* __restrict__ twtab,
                                                 void* __restrict__ outv, long long ngroups,
                                                 long long* __restrict__ dbg) {
    constexpr int L = NC / kPts;
    constexpr int G = 64 / L;
    typedef typename SwzFor<NC>::type SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & (L - 1), grp = lane / L;
    const int K = NC + 1;
    // one buffer per frame slot: exchange row of the FFT passes first, then the finished spectrum
    float* stage = smem + (wave * G + grp) * (2 * NC + 8);                 // 16B aligned
    float* row = stage;
    f2* winl = reinterpret_cast<f2*>(smem + KPR_STFT_WAVES * G * (2 * NC + 8));          // (0.5 w[2n], 0.5 w[2n+1])
    int* ticket = reinterpret_cast<int*>(winl + NC);
    int dbi = 0;
#define KPR_STAMP() do { if (dbg && blockIdx.x == 0 && (tid & 63) == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
    KPR_STAMP();
    // A workgroup owns a CONTIGUOUS run of frame groups (G frames = one wave-load) and its waves
    // draw groups from an LDS ticket counter: neighbouring frames (overlapping samples, same
    // pages) are in flight together, and waves that lose the issue arbitration take fewer groups.
    const long long g_begin = ngroups * blockIdx.x / gridDim.x;
    const int n_total = (int)(ngroups * (blockIdx.x + 1) / gridDim.x - g_begin);
    f2 nz[kPts];
    unsigned nvm = 0xffffffffu;
    int n = wave;                                           // first ticket is static: no sync needed
#define KPR_FETCH(n_)                                                                            \
    do {                                                                                         \
        const long long gf_ = (g_begin + (n_)) * G + grp;                                        \
        const bool valid_ = gf_ < g.total_frames;                                                \
        FramePos p_ = frame_pos(g, valid_ ? gf_ : 0);                                            \
        nvm = fetch_frame<NC>(x, g, p_, valid_, fl, nz);                                         \
    } while (0)
    if (n < n_total) KPR_FETCH(n);
    FftTw<NC, SW> tw;
    tw.load(twtab, fl);
    for (int i = tid; i < NC; i += 64 * KPR_STFT_WAVES) {
        const int m = 2 * i;
        const float a = window[min(m, g.win - 1)], b = window[min(m + 1, g.win - 1)];
        winl[i] = f2{(m < g.win) ? 0.5f * a : 0.0f, (m + 1 < g.win) ? 0.5f * b : 0.0f};
    }
    if (tid == 0) *ticket = KPR_STFT_WAVES;
    __syncthreads();
    const int ostride = spec_stride(g);
    KPR_STAMP();
#pragma unroll 1
    while (n < n_total) {
        const long long gf = (g_begin + n) * G + grp;
        const bool valid = gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        int n2 = 0;
        if (lane == 0) n2 = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        n2 = __builtin_amdgcn_readfirstlane(n2);
        f2 z[kPts];
#pragma unroll
        for (int m = 0; m < kPts; ++m) z[m] = nz[m];
        mask_frame(z, nvm);
#pragma unroll
        for (int m = 0; m < kPts; ++m) z[m] = pmul(z[m], winl[fl + L * m]);
#ifdef KPR_FINE_STAMPS
#define KPR_FS() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); KPR_STAMP(); } while (0)
#else
#define KPR_FS() do { } while (0)
#endif
        KPR_FS();
        if (n2 < n_total) KPR_FETCH(n2);                    // next group's samples, one ahead
        // pin the loads here: without the fence hipcc sinks them to the end of the loop body
        // (behind the spectrum stores), i.e. no prefetch at all -- 13k instead of 8k cycles/frame
        asm volatile("" ::: "memory");
        n = n2;
        KPR_FS();
        tw.refresh();
        cfft_forward<NC, SW>(z, tw, row);
        KPR_FS();
        KPR_STAMP();
        if constexpr (!OUT_CL) {
            // channels_first: the frame's K bins are contiguous in HBM.  16 narrow (4/8-byte)
            // stores per lane are store-ISSUE bound (cdna_hip_programming.md T21), so the frame is
            // transposed through LDS and written as 16-byte-per-lane, 1-KiB-per-instruction stores.
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            if constexpr (MODE == KPR_OUT_COMPLEX) {
                f2* st2 = reinterpret_cast<f2*>(stage);
                rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                    st2[k] = xk;
                    if (kp >= 0) st2[kp] = (kp == NC) ? f2{xp.x, 0.0f} : xp;
                });
                KPR_FS();
                if (valid) {
                    float* out = reinterpret_cast<float*>(outv) + 2 * spec_base(g, p, gf, K);
#pragma unroll
                    for (int q = 0; q < (2 * NC / 4) / L; ++q) {
                        const int i4 = fl + L * q;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * i4);
                        KPR_STFT_STORE(reinterpret_cast<f4u*>(out + 4 * i4), v);
                        // two at a time: all eight ds_read_b128 up front cost 32 live VGPRs
                        if (q & 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (fl == 0) { out[2 * NC] = stage[2 * NC]; out[2 * NC + 1] = 0.0f; }
                }
                KPR_STAMP();
            } else {
                rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                    stage[k] = (MODE == KPR_OUT_MAGNITUDE)
                                   ? __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y)
                                   : atan2f(k == 0 ? 0.0f : xk.y, xk.x);
                    if (kp >= 0)
                        stage[kp] = (MODE == KPR_OUT_MAGNITUDE)
                                        ? __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y)
                                        : atan2f(kp == NC ? 0.0f : xp.y, xp.x);
                });
                if (valid) {
                    float* out = reinterpret_cast<float*>(outv) + spec_base(g, p, gf, K);
#pragma unroll
                    for (int q = 0; q < (NC / 4) / L; ++q) {
                        const int i4 = fl + L * q;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * i4);
                        KPR_STFT_STORE(reinterpret_cast<f4u*>(out + 4 * i4), v);
                    }
                    if (fl == 0) out[NC] = stage[NC];
                }
            }
            continue;
        }
        // channels_last: bins of one frame are C elements apart -> narrow strided stores
        const long long ob = spec_base(g, p, gf, K);
        if constexpr (MODE == KPR_OUT_COMPLEX) {
            float2* out = reinterpret_cast<float2*>(outv) + ob;
            rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                if (valid) {
                    out[(long long)k * ostride] = make_float2(xk.x, k == 0 ? 0.0f : xk.y);
                    if (kp >= 0) out[(long long)kp * ostride] = make_float2(xp.x, kp == NC ? 0.0f : xp.y);
                }
            });
        } else {
            float* out = reinterpret_cast<float*>(outv) + ob;
            rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                if (valid) {
                    out[(long long)k * ostride] = (MODE == KPR_OUT_MAGNITUDE)
                                                      ? sqrtf(xk.x * xk.x + xk.y * xk.y)
                                                      : atan2f(k == 0 ? 0.0f : xk.y, xk.x);
                    if (kp >= 0)
                        out[(long long)kp * ostride] = (MODE == KPR_OUT_MAGNITUDE)
                                                           ? sqrtf(xp.x * xp.x + xp.y * xp.y)
                                                           : atan2f(kp == NC ? 0.0f : xp.y, xp.x);
                }
            });
        }
    }
#undef KPR_STAMP
#undef KPR_FETCH
}

// ------------------------------------------------------------------------------------------
// STFT for even transform sizes that are not powers of two (n_fft = 400, 480, 1000, ...; the
// reference's own tests use 1000): Bluestein / chirp-z on top of the power-of-two Stockham FFT.
// The NCr = n_fft/2 point complex DFT of z[n] = x[2n] + i x[2n+1] is a convolution with a chirp,
// evaluated with two M-point FFTs (M = power of two >= 2 NCr - 1), then the usual real-FFT pairing
// (oracle/proto_bluestein.py is the step-by-step numpy model, tests/test_proto_stockham.py):
//   a[n] = z[n] w[n],  Z[k]/2 = w[k] conj(FFT(conj(FFT(a) Bt)))[k],  Bt = FFT(chirp) / (2M)
//   X[k] = (Z[k] + conj Z[NCr-k])/2 - i t[k] (Z[k] - conj Z[NCr-k])/2,  t[k] = exp(-2 pi i k/n_fft)
// Tables (per n_fft, device cache): bs[0..M) = w (0 beyond NCr), bs[M..2M) = Bt, bs[2M..2M+NCr] = t.
// One LDS buffer per frame slot: exchange row of the FFTs | Z/2 (NCr complex) | finished spectrum.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int bs_slot_words(int M, int ncr) {
    return (M + M / 32 + 24 + 3) / 4 * 4 + 2 * ncr + 2 * (ncr + 1) + 2;
}

template <int M>
__global__ __launch_bounds__(256, 2) void k_stft_bs(const float* __restrict__ x, Geom g,
                                                    const float* __restrict__ window,
                                                    const float2* __restrict__ twtab,
                                                    const float2* __restrict__ bs, int mode,
                                                    void* __restrict__ outv, long long ngroups) {
    constexpr int L = M / kPts;
    constexpr int G = 64 / L;
    typedef typename SwzFor<M>::type SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & (L - 1), grp = lane / L;
    const int ncr = g.n_fft / 2, K = ncr + 1;
    const int slot = bs_slot_words(M, ncr);
    float* row = smem + (wave * G + grp) * slot;                           // FFT exchange row
    f2* zrow = reinterpret_cast<f2*>(row + (M + M / 32 + 24 + 3) / 4 * 4);  // Z/2, NCr complex
    float* stage = reinterpret_cast<float*>(zrow + ncr);                   // spectrum, 2K floats
    // window and the three tables live in LDS (ds_read_b64 at use): in registers they cost 128
    // VGPRs and the kernel spilled
    f2* winl = reinterpret_cast<f2*>(smem + 4 * G * slot);                 // (w[2n], w[2n+1])
    f2* cwl = winl + M;                                                    // chirp w (0 beyond NCr)
    f2* btl = cwl + M;                                                     // Bt
    f2* tkl = btl + M;                                                     // t[0 .. NCr]
    for (int i = tid; i < M; i += 256) {
        const int n = 2 * i;
        const float a = window[min(n, g.win - 1)], b = window[min(n + 1, g.win - 1)];
        winl[i] = f2{(n < g.win) ? a : 0.0f, (n + 1 < g.win) ? b : 0.0f};
        const float2 c = bs[i], d = bs[M + i];
        cwl[i] = f2{c.x, c.y};
        btl[i] = f2{d.x, d.y};
        if (i <= ncr) { const float2 e = bs[2 * M + i]; tkl[i] = f2{e.x, e.y}; }
    }
    FftTw<M, SW> tw;
    tw.load(twtab, fl);
    __syncthreads();
    const int ostride = spec_stride(g);
#pragma unroll 1
    for (long long grpi = (long long)blockIdx.x * 4 + wave; grpi < ngroups; grpi += (long long)gridDim.x * 4) {
        const long long gf = grpi * G + grp;
        const bool valid = gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        f2 z[kPts];
        const unsigned vm = fetch_frame<M>(x, g, p, valid, fl, z);         // n >= win: masked to zero
        mask_frame(z, vm);
#pragma unroll
        for (int m = 0; m < kPts; ++m) z[m] = cmul(pmul(z[m], winl[fl + L * m]), cwl[fl + L * m]);   // a = z w
        tw.refresh();
        cfft_forward<M, SW>(z, tw, row);
#pragma unroll
        for (int m = 0; m < kPts; ++m) { const f2 v = cmul(z[m], btl[fl + L * m]); z[m] = f2{v.x, -v.y}; }
        cfft_forward<M, SW>(z, tw, row);
#pragma unroll
        for (int m = 0; m < kPts; ++m) {                                   // Z/2 = w conj(.)
            z[m] = cmul(f2{z[m].x, -z[m].y}, cwl[fl + L * m]);
            const int j = fl + L * m;
            if (j < ncr) zrow[j] = z[m];
        }
        // the frame's lanes all sit in this wave: LDS is in order, no barrier needed.  The partner
        // reads Z[NCr - k] go through inline asm: with a compiler-visible data-dependent LDS load
        // hipcc kept a shadow copy of z[] in scratch memory (144 bytes per lane, ~100 scratch
        // instructions per frame)
        f2 zp[kPts], z0;
        {
            const unsigned zbase = (unsigned)(uintptr_t)zrow;
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                const int kc = min(fl + L * m, ncr);
                const int kpi = ncr - kc;                                  // k = 0 and NCr pair with Z[0]
                const unsigned addr = zbase + 8u * (unsigned)(kpi == ncr ? 0 : kpi);
                asm volatile("ds_read_b64 %0, %1" : "=v"(zp[m]) : "v"(addr) : "memory");
            }
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(z0) : "v"(zbase) : "memory");
        }
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            const int k = fl + L * m;
            const int kc = min(k, ncr);                                    // lanes past the end idle along
            const f2 zk = (k < ncr) ? z[m] : z0;
            const f2 e = cadd_conj(zk, zp[m]), d = csub_conj(zk, zp[m]);
            f2 X = cadd_mi(e, cmul(d, tkl[kc]));                           // e - i t d
            if (kc == 0 || kc == ncr) X.y = 0.0f;
            if (k <= ncr) {
                if (mode == KPR_OUT_COMPLEX) { stage[2 * k] = X.x; stage[2 * k + 1] = X.y; }
                else stage[k] = (mode == KPR_OUT_MAGNITUDE) ? __builtin_amdgcn_sqrtf(X.x * X.x + X.y * X.y)
                                                             : atan2f(X.y, X.x);
            }
        }
        if (valid) {
            const int nout = (mode == KPR_OUT_COMPLEX) ? 2 * K : K;
            if (ostride == 1) {
                float* out = reinterpret_cast<float*>(outv) + (mode == KPR_OUT_COMPLEX ? 2 : 1) * spec_base(g, p, gf, K);
                for (int i = fl; i < nout; i += L) out[i] = stage[i];
            } else if (mode == KPR_OUT_COMPLEX) {
                float2* out = reinterpret_cast<float2*>(outv) + spec_base(g, p, gf, K);
                for (int k = fl; k < K; k += L) out[(long long)k * ostride] = make_float2(stage[2 * k], stage[2 * k + 1]);
            } else {
                float* out = reinterpret_cast<float*>(outv) + spec_base(g, p, gf, K);
                for (int k = fl; k < K; k += L) out[(long long)k * ostride] = stage[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// STFT for n_fft = 2^a 5^b in {160, 200, 320, 400, 640, 800, 1000}: the N = n_fft/2 point complex
// FFT of z[n] = x[2n] + i x[2n+1] as a mixed-radix FFT (kpr_fft_mr.h: 20 points per lane,
// L = N/20 lanes per frame, G = 64 / L frames per wave), then the usual real-FFT pairing
//   X[k] = e - i t d,  X[N-k] = conj(e + i t d),  e = (Z[k] + conj Z[N-k])/2, d = (Z[k] - conj Z[N-k])/2,
//   t = exp(-2 pi i k / n_fft)
// done in place in the frame's LDS row, and a whole-wave copy of the finished spectra.
// One N-point FFT per frame instead of Bluestein's two M >= 2N point FFTs (k_stft_bs, kept for the
// remaining even sizes).  Replaces tf.signal.stft as called at kapre/time_frequency.py:174-182.
// ------------------------------------------------------------------------------------------
template <int R2, int R3>
__global__ __launch_bounds__(256, 3) void k_stft_mr(const float* __restrict__ x, Geom g,
                                                    const float* __restrict__ window,
                                                    const float2* __restrict__ twtab, int mode,
                                                    void* __restrict__ outv, long long ngroups) {
    typedef MrFft<R2, R3> F;
    constexpr int P = F::P, L = F::L, N = F::N, G = 64 / L, K = N + 1;
    constexpr int RSF = N + 1;                                    // row stride (complex words), odd
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool active = lane < G * L;                             // lanes beyond the last whole frame idle along
    const int grp = active ? lane / L : 0, l = active ? lane - grp * L : 0;
    f2* rows = reinterpret_cast<f2*>(smem);
    f2* row = rows + (wave * G + grp) * RSF;
    f2* winl = rows + 4 * G * RSF;                                // (w[2n], w[2n+1]) / 2
    f2* tab = winl + N;                                           // exp(-2 pi i j / n_fft), j < n_fft
    for (int i = tid; i < N; i += 256) {
        const int n = 2 * i;
        const float a = window[min(n, g.win - 1)], b = window[min(n + 1, g.win - 1)];
        winl[i] = f2{(n < g.win) ? 0.5f * a : 0.0f, (n + 1 < g.win) ? 0.5f * b : 0.0f};
    }
    for (int i = tid; i < 2 * N; i += 256) { const float2 t = twtab[i]; tab[i] = f2{t.x, t.y}; }
    __syncthreads();
    const int ostride = spec_stride(g);
#pragma unroll 1
    for (long long grpi = (long long)blockIdx.x * 4 + wave; grpi < ngroups; grpi += (long long)gridDim.x * 4) {
        const long long gf = grpi * G + grp;
        const bool valid = active && gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        // ---- samples (unconditional loads from clamped offsets, masked afterwards), window -------
        f2 z[P];
        {
            const float* sig = x + p.sig_off;
            const int es = p.es, omax = (int)(g.T - 1) * es;
            const int o_base = ((int)p.s0 + 2 * l) * es;
            unsigned long long vm = 0;
#pragma unroll
            for (int m = 0; m < P; ++m) {
                const int n = 2 * (l + L * m);
                const int o0 = o_base + m * (2 * L) * es, o1 = o0 + es;
                z[m] = f2{sig[min(max(o0, 0), omax)], sig[min(max(o1, 0), omax)]};
                vm |= (valid && n < g.win && (unsigned)o0 <= (unsigned)omax) ? (1ull << (2 * m)) : 0ull;
                vm |= (valid && n + 1 < g.win && (unsigned)o1 <= (unsigned)omax) ? (2ull << (2 * m)) : 0ull;
                if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < P; ++m) {
                const unsigned kx = (unsigned)(-(int)((vm >> (2 * m)) & 1ull));
                const unsigned ky = (unsigned)(-(int)((vm >> (2 * m + 1)) & 1ull));
                const f2 v = f2{__uint_as_float(__float_as_uint(z[m].x) & kx), __uint_as_float(__float_as_uint(z[m].y) & ky)};
                z[m] = pmul(v, winl[l + L * m]);
            }
        }
        // ---- Z/2 = FFT_N(z / 2), left in the row in natural order ---------------------------------
        F::run(z, l, active, row, tab);
        if (active) {
#pragma unroll
            for (int r = 0; r < P; ++r) row[F::bin(l, r)] = z[r];
        }
        // ---- pairing in place: the pair (k, N-k) -> X[k], X[N-k]; k = 0 -> X[0], X[N] ------------
        for (int k = l; 2 * k <= N; k += L) {
            const int kp = (k == 0) ? 0 : N - k;
            const f2 zk = row[k], zp = row[kp];
            const f2 e = cadd_conj(zk, zp), d = csub_conj(zk, zp);
            const f2 td = cmul(d, tab[k]);
            f2 xk = cadd_mi(e, td);                               // e - i t d
            f2 xq = cadd_pi(e, td);                               // e + i t d, conjugated below
            xq.y = -xq.y;
            if (k == 0) { xk.y = 0.0f; xq.y = 0.0f; }             // DC and Nyquist are real
            if (active) {
                row[k] = xk;
                if (2 * k != N) row[N - k] = xq;
            }
        }
        // ---- whole-wave copy of the G spectra ----------------------------------------------------
        const long long ob = valid ? spec_base(g, p, gf, K) : -1;
        const unsigned ob_lo = (unsigned)(unsigned long long)ob, ob_hi = (unsigned)((unsigned long long)ob >> 32);
#pragma unroll 1
        for (int gq = 0; gq < G; ++gq) {
            const long long o = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ob_hi, gq * L) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane((int)ob_lo, gq * L));
            if (o < 0) continue;                                  // wave-uniform
            const f2* src = rows + (wave * G + gq) * RSF;
            if (mode == KPR_OUT_COMPLEX) {
                float2* out = reinterpret_cast<float2*>(outv) + o;
                for (int k = lane; k < K; k += 64) { const f2 v = src[k]; out[(long long)k * ostride] = make_float2(v.x, v.y); }
            } else {
                float* out = reinterpret_cast<float*>(outv) + o;
                for (int k = lane; k < K; k += 64) {
                    const f2 v = src[k];
                    out[(long long)k * ostride] = (mode == KPR_OUT_MAGNITUDE) ? __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y)
                                                                              : atan2f(v.y, v.x);
                }
            }
        }
    }
}

// Inverse counterpart (InverseSTFT for the same transform sizes): inverse pairing X -> Z, the NCr-point
// inverse DFT as conj(DFT(conj Z)) / NCr through the same chirp machinery, synthesis window, and
// the windowed frame into the [total_frames][win] buffer that k_ola gathers from
// (oracle/proto_bluestein.py: irfft_bluestein).
template <int M>
__global__ __launch_bounds__(256, 2) void k_irfft_bs(const float2* __restrict__ spec, Geom g,
                                                     const float* __restrict__ synth,
                                                     const float2* __restrict__ twtab,
                                                     const float2* __restrict__ bs,
                                                     float* __restrict__ frames, long long ngroups) {
    constexpr int L = M / kPts;
    constexpr int G = 64 / L;
    typedef typename SwzFor<M>::type SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & (L - 1), grp = lane / L;
    const int ncr = g.n_fft / 2, K = ncr + 1;
    const int slot = (M + M / 32 + 24 + 3) / 4 * 4;
    float* row = smem + (wave * G + grp) * slot;                           // FFT exchange row
    f2* winl = reinterpret_cast<f2*>(smem + 4 * G * slot);                 // synthesis window * 2/NCr
    f2* cwl = winl + M;
    f2* btl = cwl + M;
    f2* tkl = btl + M;
    const float sc = 2.0f / (float)ncr;
    for (int i = tid; i < M; i += 256) {
        const int n = 2 * i;
        const float a = synth[min(n, g.win - 1)], b = synth[min(n + 1, g.win - 1)];
        winl[i] = f2{(n < g.win && n < g.n_fft) ? sc * a : 0.0f, (n + 1 < g.win && n + 1 < g.n_fft) ? sc * b : 0.0f};
        const float2 c = bs[i], d = bs[M + i];
        cwl[i] = f2{c.x, c.y};
        btl[i] = f2{d.x, d.y};
        if (i <= ncr) { const float2 e = bs[2 * M + i]; tkl[i] = f2{e.x, e.y}; }
    }
    FftTw<M, SW> tw;
    tw.load(twtab, fl);
    __syncthreads();
    const int ostride = spec_stride(g);
#pragma unroll 1
    for (long long grpi = (long long)blockIdx.x * 4 + wave; grpi < ngroups; grpi += (long long)gridDim.x * 4) {
        const long long gf = grpi * G + grp;
        const bool valid = gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        const float2* sp = spec + spec_base(g, p, gf, K);
        f2 z[kPts];
#pragma unroll
        for (int m = 0; m < kPts; ++m) {          // unconditional loads from clamped bins, masked below
            const int k = fl + L * m;
            const int kc = min(k, ncr - 1);
            float2 a = sp[(long long)kc * ostride], b = sp[(long long)(ncr - kc) * ostride];
            if (kc == 0) { a.y = 0.0f; b.y = 0.0f; }                        // irfft ignores Im of DC / Nyquist
            const f2 xk = f2{a.x, a.y}, xp = f2{b.x, -b.y};                 // X[k], conj X[NCr-k]
            const f2 e = cadd(xk, xp), d = csub(xk, xp);
            const f2 tc = tkl[kc];
            const f2 od = cmul(d, f2{tc.x, -tc.y});                        // (X - conj X') conj(t)
            f2 zk = f2{0.5f * (e.x - od.y), 0.5f * (e.y + od.x)};          // Z = E + i O
            if (!valid || k >= ncr) zk = f2{0.0f, 0.0f};
            z[m] = cmul(f2{zk.x, -zk.y}, cwl[fl + L * m]);                  // a = conj(Z) w
        }
        tw.refresh();
        cfft_forward<M, SW>(z, tw, row);
#pragma unroll
        for (int m = 0; m < kPts; ++m) { const f2 v = cmul(z[m], btl[fl + L * m]); z[m] = f2{v.x, -v.y}; }
        cfft_forward<M, SW>(z, tw, row);
        if (!valid) continue;
        float* fo = frames + gf * (long long)g.win;
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            const int n = fl + L * m;                                      // y[n] = DFT(conj Z)[n] / 2
            const f2 y = cmul(f2{z[m].x, -z[m].y}, cwl[n]);
            const f2 w = winl[n];                                          // (2/NCr) * synthesis window
            if (2 * n < g.win) fo[2 * n] = y.x * w.x;                      // z[n] = conj(y) * 2/NCr
            if (2 * n + 1 < g.win) fo[2 * n + 1] = -y.y * w.y;
        }
    }
}

// Inverse counterpart of k_stft_mr (InverseSTFT for the same transform sizes): inverse pairing
//   Z[k] = (E + i O)/2,  E = X[k] + conj X[N-k],  O = (X[k] - conj X[N-k]) conj(t[k]),
// the N-point inverse DFT as conj(FFT_N(conj Z)) / N, synthesis window, and the windowed frame into the
// [total_frames][win] buffer that k_ola gathers from (tf.signal.inverse_stft, kapre/time_frequency.py:307-314).
template <int R2, int R3>
__global__ __launch_bounds__(256, 2) void k_irfft_mr(const float2* __restrict__ spec, Geom g,
                                                     const float* __restrict__ synth,
                                                     const float2* __restrict__ twtab,
                                                     float* __restrict__ frames, long long ngroups) {
    typedef MrFft<R2, R3> F;
    constexpr int P = F::P, L = F::L, N = F::N, G = 64 / L, K = N + 1;
    constexpr int RSF = N + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool active = lane < G * L;
    const int grp = active ? lane / L : 0, l = active ? lane - grp * L : 0;
    f2* rows = reinterpret_cast<f2*>(smem);
    f2* row = rows + (wave * G + grp) * RSF;
    f2* winl = rows + 4 * G * RSF;                                // synthesis window / (2N), pairs
    f2* tab = winl + N;
    const float sc = 0.5f / (float)N;                             // 1/2 of the pairing, 1/N of the inverse DFT
    for (int i = tid; i < N; i += 256) {
        const int n = 2 * i;
        const float a = synth[min(n, g.win - 1)], b = synth[min(n + 1, g.win - 1)];
        winl[i] = f2{(n < g.win) ? sc * a : 0.0f, (n + 1 < g.win) ? sc * b : 0.0f};
    }
    for (int i = tid; i < 2 * N; i += 256) { const float2 t = twtab[i]; tab[i] = f2{t.x, t.y}; }
    __syncthreads();
    const int ostride = spec_stride(g);
#pragma unroll 1
    for (long long grpi = (long long)blockIdx.x * 4 + wave; grpi < ngroups; grpi += (long long)gridDim.x * 4) {
        const long long gf = grpi * G + grp;
        const bool valid = active && gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        const float2* sp = spec + spec_base(g, p, valid ? gf : 0, K);
        f2 z[P];
#pragma unroll
        for (int m = 0; m < P; ++m) {              // unconditional loads, masked below
            const int k = l + L * m;               // < N
            float2 a = sp[(long long)k * ostride], b = sp[(long long)(N - k) * ostride];
            if (k == 0) { a.y = 0.0f; b.y = 0.0f; }                        // irfft ignores Im of DC / Nyquist
            const f2 xk = f2{a.x, a.y}, xp = f2{b.x, -b.y};                 // X[k], conj X[N-k]
            const f2 e = cadd(xk, xp), d = csub(xk, xp);
            const f2 tc = tab[k];
            const f2 od = cmul(d, f2{tc.x, -tc.y});                        // (X - conj X') conj(t)
            f2 zc = f2{e.x - od.y, -(e.y + od.x)};                         // conj(2 Z) = conj(E + i O)
            if (!valid) zc = f2{0.0f, 0.0f};
            z[m] = zc;
        }
        F::run(z, l, active, row, tab);                                    // Y = FFT_N(conj 2Z)
        if (!valid) continue;
        float* fo = frames + gf * (long long)g.win;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const int n = F::bin(l, r);                                    // z[n] = conj(Y[n]) / (2N)
            const f2 w = winl[n];
            if (2 * n < g.win) fo[2 * n] = z[r].x * w.x;
            if (2 * n + 1 < g.win) fo[2 * n + 1] = -z[r].y * w.y;
        }
        // win_length > n_fft: the irfft output is right-padded with zeros
        for (int n = 2 * N + l; n < g.win; n += L) fo[n] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// inverse: spectrum -> windowed real frames (frames buffer is [total_frames][win])
// ------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256, 2) void k_irfft(const float2* __restrict__ spec, Geom g,
                                                  const float* __restrict__ synth,
                                                  const float2* __restrict__ twtab,
                                                  float* __restrict__ frames, long long nblocks) {
    constexpr int L = NC / kPts;
    constexpr int G = 64 / L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & (L - 1), grp = lane / L;
    const int K = NC + 1;
    float* row = smem + (wave * G + grp) * NC;
    float* stage = smem + 4 * G * NC + (wave * G + grp) * (2 * NC + 8);   // one spectrum, 16B aligned
    FftTw<NC> tw;
    tw.load(twtab, fl);
    WinRegs<NC> wr;
    wr.load(synth, g.win, fl, 1.0f / (float)(2 * NC));   // synthesis window with irfft's 1/n_fft
    const int ostride = spec_stride(g);
#pragma unroll 1
    for (long long fb = blockIdx.x; fb < nblocks; fb += gridDim.x) {
        const long long gf = fb * (4 * G) + wave * G + grp;
        const bool valid = gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        f2 z[kPts];
        // pairing: 2 Z[k] = (X[k] + conj X[NC-k]) + i (X[k] - conj X[NC-k]) e^{+2 pi i k/N}
        const float2* sp = spec + spec_base(g, p, gf, K);
        if (!g.out_cl) {
            // channels_first: stream the frame's K contiguous bins with 16-byte loads into LDS,
            // then pick X[k] and X[NC-k] from there (32 narrow global loads per lane otherwise)
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const float* spf = reinterpret_cast<const float*>(sp);
#pragma unroll
            for (int q = 0; q < (2 * NC / 4) / L; ++q) {
                const int i4 = fl + L * q;
                const f32x4 v = *reinterpret_cast<const f4u*>(spf + 4 * i4);
                *reinterpret_cast<f32x4*>(stage + 4 * i4) = v;
            }
            if (fl == 0) { stage[2 * NC] = spf[2 * NC]; stage[2 * NC + 1] = spf[2 * NC + 1]; }
            const float2* st2 = reinterpret_cast<const float2*>(stage);
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                const int k = fl + L * m;
                float2 a = st2[k], b = st2[NC - k];
                if (!valid) { a = make_float2(0.f, 0.f); b = a; }
                if (k == 0) { a.y = 0.0f; b.y = 0.0f; }   // irfft ignores Im of DC / Nyquist
                z[m] = irfft_pair_one<NC>(f2{a.x, a.y}, f2{b.x, b.y}, tw, m);
            }
        } else {
#pragma unroll
            for (int m = 0; m < kPts; ++m) {      // unconditional loads, masked below
                const int k = fl + L * m;
                float2 a = sp[(long long)k * ostride], b = sp[(long long)(NC - k) * ostride];
                if (!valid) { a = make_float2(0.f, 0.f); b = a; }
                if (k == 0) { a.y = 0.0f; b.y = 0.0f; }
                z[m] = irfft_pair_one<NC>(f2{a.x, a.y}, f2{b.x, b.y}, tw, m);
            }
        }
        tw.refresh();
        cfft_forward<NC>(z, tw, row);
        if (!valid) continue;
        float* fo = frames + gf * (long long)g.win;
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            int n = 2 * (fl + L * m);
            if (n < g.win) fo[n] = z[m].x * wr.w[m].x;
            if (n + 1 < g.win) fo[n + 1] = -z[m].y * wr.w[m].y;
        }
        // win_length > n_fft: irfft output is right-padded with zeros (tf.signal.inverse_stft)
        for (int n = 2 * NC + fl; n < g.win; n += L) fo[n] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// fused inverse: irFFT + synthesis window + overlap-add in ONE kernel, no frames workspace.
// A workgroup owns the output samples [c*FB*hop, (c+1)*FB*hop) of one signal.  It needs the frames
// fa .. fb that overlap them (FB frames plus a halo of R-1 = ceil(win/hop)-1 recomputed frames,
// NR = FB + R - 1 rows), transforms each into an LDS row (the row doubles as the FFT exchange
// buffer of its own frame), and then every output sample gathers its <= R contributions from
// LDS in ascending frame order (no atomics -> deterministic, same order as tf overlap_and_add).
// Replaces tf.signal.inverse_stft as called at kapre/time_frequency.py:307-314.
// ------------------------------------------------------------------------------------------
struct IstftPlan {
    long long n_sig;      // B * C
    long long t_out;      // (F-1)*hop + win
    int F, C, win, hop;
    int NR, FB, R;        // LDS rows, new frames per block, overlaps
    int RS;               // row stride (floats) >= max(win, NC)
    int chunks;           // blocks per signal = ceil(t_out / (FB*hop))
    int spec_cl, wave_cl; // layouts of the spectrogram / waveform
    int vec4;             // overlap-add in groups of four samples (hop, win % 4 == 0, contiguous out)
};

template <int NC, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_istft_fused(const float2* __restrict__ spec,
                                                            IstftPlan pl,
                                                            const float* __restrict__ synth,
                                                            const float2* __restrict__ twtab,
                                                            float* __restrict__ out,
                                                            long long nblocks) {
    constexpr int L = NC / kPts;
    constexpr int G = 64 / L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & (L - 1), grp = lane / L;
    const int K = NC + 1;
    FftTw<NC> tw;
    tw.load(twtab, fl);
    WinRegs<NC> wr;
    wr.load(synth, pl.win, fl, 1.0f / (float)(2 * NC));   // synthesis window with irfft's 1/n_fft
    // overlap-add walks (hop index q, 4-sample group o4) = divmod(tid + it * threads, hop / 4)
    const bool vec4 = pl.vec4 != 0;
    const int nq4 = vec4 ? pl.hop >> 2 : 1;
    const int q_first = tid / nq4, o4_first = tid - q_first * nq4;
    const int q_step = (NW * 64) / nq4, o4_step = (NW * 64) - q_step * nq4;
#pragma unroll 1
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long sig = blk / pl.chunks;
        const int c = (int)(blk - sig * pl.chunks);
        const long long b = sig / pl.C;
        const int ch = (int)(sig - b * pl.C);
        const long long t_lo = (long long)c * pl.FB * pl.hop;
        long long t_hi = t_lo + (long long)pl.FB * pl.hop;
        if (t_hi > pl.t_out) t_hi = pl.t_out;
        long long fa = (t_lo - pl.win + pl.hop) / pl.hop;             // ceil((t_lo - win + 1)/hop)
        if (t_lo - pl.win + 1 <= 0) fa = 0;
        long long fb = (t_hi - 1) / pl.hop;
        if (fb > pl.F - 1) fb = pl.F - 1;
        const int nrows = (int)(fb - fa + 1);                         // <= NR
        // spectrogram addressing of frame f: base + k * sstride (complex units)
        const long long sstride = pl.spec_cl ? pl.C : 1;

        // ---- phase A: irFFT of the rows ------------------------------------------------------
#pragma unroll 1
        for (int r0 = 0; r0 < pl.NR; r0 += NW * G) {
            const int r = r0 + wave * G + grp;
            const bool valid = r < nrows;
            const long long f = fa + (valid ? r : 0);
            const long long sbase = pl.spec_cl ? ((b * pl.F + f) * K) * pl.C + ch
                                               : ((b * pl.C + ch) * pl.F + f) * K;
            const float2* sp = spec + sbase;
            float* row = smem + (valid ? r : 0) * pl.RS;
            if (r0 + wave * G >= nrows) continue;                     // whole wave idle (uniform)
            f2 z[kPts];
#pragma unroll
            for (int m = 0; m < kPts; ++m) {          // unconditional loads, masked afterwards
                const int k = fl + L * m;
#ifdef KPR_ISTFT_NOLOAD
                float2 a = make_float2((float)k, 1.0f), bb = make_float2(1.0f, (float)m);
#else
                float2 a = sp[(long long)k * sstride], bb = sp[(long long)(NC - k) * sstride];
#endif
                if (!valid) { a = make_float2(0.f, 0.f); bb = a; }
                if (k == 0) { a.y = 0.0f; bb.y = 0.0f; }             // irfft ignores Im of DC / Nyquist
                z[m] = irfft_pair_one<NC>(f2{a.x, a.y}, f2{bb.x, bb.y}, tw, m);
            }
            tw.refresh();
            // idle frame slots (r >= nrows; never group 0 of an active wave) get a spare scratch row
            float* xrow = valid ? row : smem + (pl.NR + wave * (G > 1 ? G - 1 : 0) + (grp > 0 ? grp - 1 : 0)) * pl.RS;
#ifndef KPR_ISTFT_NOFFT
            cfft_forward<NC>(z, tw, xrow);
#endif
            if (valid) {
#pragma unroll
                for (int m = 0; m < kPts; ++m) {
                    const int n = 2 * (fl + L * m);
                    if (n < pl.win) row[n] = z[m].x * wr.w[m].x;
                    if (n + 1 < pl.win) row[n + 1] = -z[m].y * wr.w[m].y;
                }
                for (int n = 2 * NC + fl; n < pl.win; n += L) row[n] = 0.0f;   // win > n_fft: zeros
            }
        }
        __syncthreads();

        // ---- phase B: gather overlap-add from LDS ---------------------------------------------
        // 32-bit arithmetic relative to the chunk (t_lo is a multiple of hop): sample t = fh*hop +
        // off gets row f = fh - j at position j*hop + off, for the j with j*hop + off < win and
        // fa <= f <= fb.  Summed with f ASCENDING -- the order of the two-kernel path, bit for bit.
        const int n_here = (int)(t_hi - t_lo);
        const int fh0 = c * pl.FB;                                       // t_lo / hop
        const int ifa = (int)fa, ifb = (int)fb;
#ifdef KPR_ISTFT_NOB
        if (pl.F < 0)
#endif
        if (vec4) {
            // four consecutive samples per lane: hop, win, RS and t_lo are multiples of 4, so the four
            // share q, the row set and the bounds; one ds_read_b128 per contributing row and one
            // 16-byte store.  (q, o4) walk the chunk without a division; absent rows add nothing.
            const int n4 = n_here >> 2;                                  // t_out % 4 == 0
            int q = q_first, o4 = o4_first;
            float* const op = out + sig * pl.t_out + t_lo;
            for (int i = tid; i < n4; i += NW * 64) {
                const int fh = fh0 + q, off = 4 * o4;
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
                for (int j = pl.R - 1; j >= 0; --j) {
                    const int f = fh - j, pos = j * pl.hop + off;
                    if (pos < pl.win && f >= ifa && f <= ifb)
                        acc += *reinterpret_cast<const f32x4*>(smem + (f - ifa) * pl.RS + pos);
                }
                *reinterpret_cast<f32x4*>(op + 4 * i) = acc;
                o4 += o4_step; q += q_step;
                if (o4 >= nq4) { o4 -= nq4; ++q; }
            }
        } else {
            for (int tt = tid; tt < n_here; tt += NW * 64) {
                const int q = tt / pl.hop, off = tt - q * pl.hop;
                const int fh = fh0 + q;
                const int j_min = fh > ifb ? fh - ifb : 0;
                int j_max = (pl.win - 1 - off) / pl.hop;
                if (j_max > fh - ifa) j_max = fh - ifa;
                float acc = 0.0f;
                for (int j = j_max; j >= j_min; --j)
                    acc += smem[(fh - j - ifa) * pl.RS + j * pl.hop + off];
                const long long t = t_lo + tt;
                const long long o = pl.wave_cl ? (b * pl.t_out + t) * pl.C + ch : sig * pl.t_out + t;
                out[o] = acc;
            }
        }
        __syncthreads();       // rows are rewritten by the next block
    }
}

// ------------------------------------------------------------------------------------------
// k_istft_ws: the fused inverse, wave-specialised.  One workgroup per CU walks a SEGMENT of one
// signal (hop blocks q0 .. q1-1, i.e. output samples [q0*hop, q1*hop)) from left to right:
//   * 7 producer waves take tickets of G frames, load the spectrum rows one ticket ahead
//     (registers), run pairing + inverse FFT + synthesis window and leave the frame in slot
//     (frame - fa) & (NR-1) of an LDS ring of NR rows (the row is its own FFT exchange buffer);
//   * 1 consumer wave follows: when the frames of hop blocks [cq, cq+QB) are in the ring it sums,
//     for four samples per lane, the <= R rows that overlap them (ascending frame order, the order
//     of tf.signal.overlap_and_add) and stores 16 bytes.
// No workgroup barrier inside a segment: done[slot] = position + 1 (producer -> consumer, per
// frame) and sync[1] = hop blocks emitted (consumer -> producers: the frame NR positions back may be
// overwritten once block  f - NR + R - 1  is out).  Compared with k_istft_fused there is no halo
// of R-1 recomputed frames per chunk (only per segment), and spectrum loads, FFTs and the
// overlap-add of different frames overlap in time instead of alternating between two barriers.
// Replaces tf.signal.inverse_stft as called at kapre/time_frequency.py:307-314.
// ------------------------------------------------------------------------------------------
struct IstftWsPlan {
    long long t_out;      // (F-1)*hop + win
    int F, C, win, hop, R;
    int NR, RS;           // ring rows (power of two), row stride (floats)
    int Q;                // hop blocks per signal = F - 1 + R
    int segs, QS;         // segments per signal, hop blocks per segment
    int QB;               // hop blocks the consumer emits per batch
};
constexpr int kIwProd = 7;
constexpr int kIwThreads = 512;
// every wait is bounded (a few hundred ms): a protocol error must end as a wrong result that the
// parity tests catch, never as a hung device
constexpr int kIwSpinLimit = 1 << 22;
constexpr int kIwReads = 8;       // row reads (ds_read_b128) per consumer lane and pass

// One consumer pass of k_istft_ws = the 64 * IT four-sample groups of the hop blocks [cq, qe),
// RJ rows each (RJ >= R = ceil(win / hop)): IT * RJ = kIwReads independent ds_read_b128 plus the flag of
// one frame per lane, all issued together.  With the producers' FFT exchanges queued in the same
// LDS pipeline a read returns after ~1k cycles, so the consumer keeps TWO passes in flight: the
// reads of pass n+1 are issued before pass n is summed.  The flag is read FIRST and LDS executes a
// wave's reads in order: if every flag shows its frame, the rows read after it are complete; if
// not, the pass waits for the flags and reads its rows again.
template <int RJ>
struct IwPass {
    static constexpr int IT = kIwReads / RJ;
    f32x4 v[IT][RJ];
    int flag, want;       // done[] of the frame this lane checks, and the value that means "written"
    int cq, qe;
    bool full;            // every lane has IT groups and every group RJ rows (no predicates needed)
};
struct IwCtx {
    const float* smem;
    int* done;
    int fa, f_last, q0, R, hop, win, RS, rmask, t_out;
    bool regular;         // win == RJ * hop: every sample away from the signal's ends has RJ rows
};

template <int RJ>
KPR_DEV bool iw_use(const IwCtx& c, int fh, int off, int j, int& addr) {
    const int f = fh - j, pos = j * c.hop + off;
    const bool use = pos < c.win && f >= c.fa && f <= c.f_last;        // (j >= R: pos >= win)
    addr = use ? ((f - c.fa) & c.rmask) * c.RS + pos : 0;
    return use;
}

template <int RJ>
KPR_DEV void iw_issue(IwPass<RJ>& s, const IwCtx& c, int cq, int qe, int lane,
                      const int (&qk)[IwPass<RJ>::IT], const int (&o4k)[IwPass<RJ>::IT], bool with_flag) {
    constexpr int IT = IwPass<RJ>::IT;
    if (with_flag) {
        s.cq = cq; s.qe = qe;
        // frames max(fa, cq-R+1) .. min(qe-1, f_last), one lane per frame (host: at most 64)
        const int plo = max(c.fa, cq - c.R + 1) - c.fa, phi = min(qe - 1, c.f_last) - c.fa;
        const int pc = plo + lane;
        s.want = pc + 1;
        s.flag = 0x7fffffff;
        if (pc <= phi)
            s.flag = __hip_atomic_load(&c.done[pc & c.rmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");             // the rows are read after the flags
    }
    const int n4 = (min(qe * c.hop, c.t_out) - cq * c.hop) >> 2;
    if (with_flag)
        s.full = c.regular && n4 == 64 * IT && cq - (RJ - 1) >= c.fa && qe - 1 <= c.f_last;
    if (s.full) {                                                      // wave-uniform
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int base = cq + qk[u] - c.fa;
#pragma unroll
            for (int jj = RJ - 1; jj >= 0; --jj)
                s.v[u][jj] = *reinterpret_cast<const f32x4*>(
                    c.smem + ((base - jj) & c.rmask) * c.RS + jj * c.hop + 4 * o4k[u]);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int fh = (lane + 64 * u < n4) ? cq + qk[u] : -(1 << 20);   // beyond the batch: no row matches
#pragma unroll
        for (int jj = RJ - 1; jj >= 0; --jj) {
            int addr;
            (void)iw_use<RJ>(c, fh, 4 * o4k[u], jj, addr);
            s.v[u][jj] = *reinterpret_cast<const f32x4*>(c.smem + addr);
        }
    }
}

template <int RJ>
KPR_DEV void iw_consume(IwPass<RJ>& s, const IwCtx& c, float* __restrict__ osig, int* emitted, int lane,
                        const int (&qk)[IwPass<RJ>::IT], const int (&o4k)[IwPass<RJ>::IT]) {
    constexpr int IT = IwPass<RJ>::IT;
    if (!__all(s.flag >= s.want)) {
        // the producers are behind: wait for the frames, then read the rows again
        const int* flag = &c.done[(s.want - 1) & c.rmask];
        for (int spin = 0; spin < kIwSpinLimit; ++spin) {
            const bool ok = s.flag == 0x7fffffff ||
                __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= s.want;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(4);
        }
        iw_issue<RJ>(s, c, s.cq, s.qe, lane, qk, o4k, false);
    }
    const int n4 = (min(s.qe * c.hop, c.t_out) - s.cq * c.hop) >> 2;
    float* const op = osig + (long long)s.cq * c.hop;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    if (s.full) {
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            f32x4 acc = zero;
#pragma unroll
            for (int jj = RJ - 1; jj >= 0; --jj) acc += s.v[u][jj];   // descending j = ascending frame
            *reinterpret_cast<f32x4*>(op + 4 * (lane + 64 * u)) = acc;
        }
    } else
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const bool here = lane + 64 * u < n4;
        const int fh = here ? s.cq + qk[u] : -(1 << 20);
        f32x4 acc = zero;
#pragma unroll
        for (int jj = RJ - 1; jj >= 0; --jj) {          // descending j = ascending frame
            int addr;
            acc += iw_use<RJ>(c, fh, 4 * o4k[u], jj, addr) ? s.v[u][jj] : zero;
        }
        if (here) *reinterpret_cast<f32x4*>(op + 4 * (lane + 64 * u)) = acc;
    }
    // the rows of this pass have been read (their values are in `acc`): let the producers reuse them
    asm volatile("" ::: "memory");
    if (lane == 0)
        __hip_atomic_store(emitted, s.qe - c.q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NC, int RJ>
__global__ __launch_bounds__(kIwThreads) void k_istft_ws(const float2* __restrict__ spec,
                                                         IstftWsPlan pl,
                                                         const float* __restrict__ synth,
                                                         const float2* __restrict__ twtab,
                                                         float* __restrict__ out, int nitems,
                                                         long long* __restrict__ dbg) {
    constexpr int L = NC / kPts;
    constexpr int G = 64 / L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & (L - 1), grp = lane / L;
    const int K = NC + 1;
    const int rmask = pl.NR - 1;
    // development aid (tools/stamps_istft.py): cycle stamps of workgroup 0, 32 per wave
    int dbi = 0;
    const bool stamp_me = dbg && blockIdx.x == 0;
#define IW_STAMP() do { if (stamp_me && lane == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
#ifdef KPR_FINE_STAMPS
#define IW_FSTAMP() do { if (stamp_me && lane == 0 && dbi < 32) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define IW_FSTAMP() do { } while (0)
#endif
    IW_STAMP();
    float* spare = smem + pl.NR * pl.RS;                       // exchange rows of idle frame slots
    int* done = reinterpret_cast<int*>(spare + kIwProd * (G - 1) * pl.RS);   // [NR]
    int* sync = done + pl.NR;                                  // [0] tickets, [1] hop blocks emitted
    // (contiguous spectrogram rows only: the 32 loads of a frame are base + immediate offset)

    // one segment: hop blocks q0 .. q1-1 of signal `sig`, made from frames fa .. f_last
#define IW_ITEM_PARAMS()                                                                          \
        const int sig = item / pl.segs, seg = item - sig * pl.segs;                              \
        const int q0 = seg * pl.QS, q1 = min(pl.Q, q0 + pl.QS);                                  \
        const int fa = max(0, q0 - (pl.R - 1)), f_last = min(pl.F - 1, q1 - 1);                  \
        const int nframes = f_last - fa + 1 /* >= 1 */
    // flags and counters of the segment (the first kIwProd tickets are taken: ticket w = wave w)
#define IW_ITEM_SYNC()                                                                            \
        for (int i = tid; i < pl.NR; i += kIwThreads) done[i] = 0;                               \
        if (tid < 2) sync[tid] = tid == 0 ? kIwProd : 0;                                         \
        __syncthreads()

    // The two roles run the segment loop separately (the same two workgroup barriers per segment
    // in each): the twiddles / window of the producers and the two passes of the consumer are then
    // never live together and the allocator does not spill either.
    if (wave < kIwProd) {
        FftTw<NC> tw;
        WinRegs<NC> wr;
        float2 xa[kPts], xb[kPts];
#define IW_TICKET(dst_)                                                                          \
    do {                                                                                         \
        int v_ = 0;                                                                              \
        if (lane == 0) v_ = __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        dst_ = __builtin_amdgcn_readfirstlane(v_);                                               \
    } while (0)
        // unconditional loads from a clamped frame (idle slots are zeroed when consumed)
#define IW_LOAD(n_)                                                                              \
    do {                                                                                         \
        const int p_ = G * (n_) + grp;                                                           \
        const float2* sp_ = sp0 + (long long)(fa + (p_ < nframes ? p_ : 0)) * K + fl;            \
        _Pragma("unroll") for (int m = 0; m < kPts; ++m) {                                       \
            xa[m] = sp_[L * m];                                                                  \
            xb[m] = sp_[NC - 2 * fl - L * m];                                                    \
        }                                                                                        \
    } while (0)
        // The wave's first ticket of a segment is static (ticket = wave), so that its spectrum rows
        // can be requested before anything else: at kernel start they travel together with the
        // twiddle and window loads, and the three latencies are paid once, at the first barrier.
        {
            const int item = blockIdx.x;
            IW_ITEM_PARAMS();
            (void)q1;
            const float2* sp0 = spec + ((long long)sig * pl.F) * K;
            if (G * wave < nframes) IW_LOAD(wave);
        }
        tw.load(twtab, fl);
        wr.load(synth, pl.win, fl, 1.0f / (float)(2 * NC));   // synthesis window with irfft's 1/n_fft
        IW_FSTAMP();
#pragma unroll 1
        for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
            IW_ITEM_PARAMS();
            // ================================ producers ======================================
            const int n_tickets = (nframes + G - 1) / G;
            const float2* sp0 = spec + ((long long)sig * pl.F) * K;
            int n = wave;
            if (item != (int)blockIdx.x && n < n_tickets) IW_LOAD(n);
            IW_ITEM_SYNC();
            IW_FSTAMP();
#pragma unroll 1
            while (n < n_tickets) {
                int n2;
                IW_TICKET(n2);
                const int p = G * n + grp;
                const bool valid = p < nframes;
                f2 z[kPts];
#pragma unroll
                for (int m = 0; m < kPts; ++m) {
                    float2 a = xa[m], bb = xb[m];
                    if (!valid) { a = make_float2(0.f, 0.f); bb = a; }
                    if (fl + L * m == 0) { a.y = 0.0f; bb.y = 0.0f; }   // irfft ignores Im of DC / Nyquist
                    z[m] = irfft_pair_one<NC>(f2{a.x, a.y}, f2{bb.x, bb.y}, tw, m);
                }
                IW_FSTAMP();
                if (n2 < n_tickets) IW_LOAD(n2);                // next ticket's rows, in flight during the FFT
                tw.refresh();
                // the ring slots of this ticket are free once the consumer has emitted every block
                // that reads the frames NR positions back: blocks < f_hi - NR + R
                const int need = fa + min(G * n + G - 1, nframes - 1) - pl.NR + pl.R - q0;
                if (need > 0)
                    for (int spin = 0; spin < kIwSpinLimit &&
                         __hip_atomic_load(&sync[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spin)
                        __builtin_amdgcn_s_sleep(2);
                float* row = valid ? smem + (p & rmask) * pl.RS
                                   : spare + (wave * (G - 1) + (grp > 0 ? grp - 1 : 0)) * pl.RS;
                IW_FSTAMP();
#ifndef KPR_IW_NOFFT
                cfft_forward<NC>(z, tw, row);
#endif
                IW_FSTAMP();
                if (valid) {
#pragma unroll
                    for (int m = 0; m < kPts; ++m) {     // win is even here: samples t, t+1 share the test
                        const int t = 2 * (fl + L * m);
                        if (t < pl.win)
                            *reinterpret_cast<f2*>(row + t) = f2{z[m].x * wr.w[m].x, -z[m].y * wr.w[m].y};
                    }
                    for (int t = 2 * NC + fl; t < pl.win; t += L) row[t] = 0.0f;   // win > n_fft: zeros
                }
                // LDS executes a wave's instructions in order: the flag follows the row
                if (valid && fl == 0)
                    __hip_atomic_store(&done[p & rmask], p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                IW_STAMP();
                n = n2;
            }
#undef IW_TICKET
#undef IW_LOAD
            __syncthreads();       // ring, flags and counters are reused by the next segment
        }
    } else {
        // consumer: group lane + 64 u of a pass = 4-sample group o4k[u] of hop block qk[u] of the batch
        int qk[IwPass<RJ>::IT], o4k[IwPass<RJ>::IT];
        {
            const int nq4 = pl.hop >> 2;
#pragma unroll
            for (int u = 0; u < IwPass<RJ>::IT; ++u) {
                qk[u] = (lane + 64 * u) / nq4;
                o4k[u] = (lane + 64 * u) - qk[u] * nq4;
            }
        }
#pragma unroll 1
        for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
            IW_ITEM_PARAMS();
            (void)nframes;
            IW_FSTAMP();
            IW_ITEM_SYNC();
            IW_FSTAMP();
            // ================================ consumer =======================================
            float* const osig = out + (long long)sig * pl.t_out;
            __builtin_amdgcn_s_setprio(3);     // one wave against seven that always have work ready
            IwCtx c;
            c.smem = smem; c.done = done; c.fa = fa; c.f_last = f_last; c.q0 = q0; c.R = pl.R;
            c.hop = pl.hop; c.win = pl.win; c.RS = pl.RS; c.rmask = rmask; c.t_out = (int)pl.t_out;
            c.regular = pl.win == RJ * pl.hop;
            IwPass<RJ> pa, pb;
            iw_issue<RJ>(pa, c, q0, min(q0 + pl.QB, q1), lane, qk, o4k, true);
#pragma unroll 1
            for (;;) {
                const bool more_b = pa.qe < q1;
                if (more_b) iw_issue<RJ>(pb, c, pa.qe, min(pa.qe + pl.QB, q1), lane, qk, o4k, true);
                iw_consume<RJ>(pa, c, osig, &sync[1], lane, qk, o4k);
                IW_STAMP();
                if (!more_b) break;
                const bool more_a = pb.qe < q1;
                if (more_a) iw_issue<RJ>(pa, c, pb.qe, min(pb.qe + pl.QB, q1), lane, qk, o4k, true);
                iw_consume<RJ>(pb, c, osig, &sync[1], lane, qk, o4k);
                IW_STAMP();
                if (!more_a) break;
            }
            __syncthreads();
        }
    }
#undef IW_ITEM_PARAMS
#undef IW_ITEM_SYNC
#undef IW_STAMP
#undef IW_FSTAMP
}

// k_istft_ws for the mixed-radix transform sizes (n_fft = 2^a 5^b, kpr_fft_mr.h): the same ring of
// frames, flags, segments and consumer wave; the producers pair X[k], X[N-k] into conj(2 Z[k]) with
// the twiddle table in LDS, run MrFft (20 points per lane, G = 64 / L frames per ticket) with the
// frame's ring slot as exchange row, and leave conj(.) x synthesis window there.  Lane groups
// without a frame (beyond the segment's last one) and the lanes beyond the last whole group never
// write to LDS, so no spare rows are needed.
template <int R2, int R3, int RJ>
__global__ __launch_bounds__(kIwThreads) void k_istft_ws_mr(const float2* __restrict__ spec,
                                                            IstftWsPlan pl,
                                                            const float* __restrict__ synth,
                                                            const float2* __restrict__ twtab,
                                                            float* __restrict__ out, int nitems) {
    typedef MrFft<R2, R3> F;
    constexpr int P = F::P, L = F::L, N = F::N, G = 64 / L, K = N + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rmask = pl.NR - 1;
    int* done = reinterpret_cast<int*>(smem + pl.NR * pl.RS);  // [NR]
    int* sync = done + pl.NR;                                  // [0] tickets, [1] hop blocks emitted
    f2* winl = reinterpret_cast<f2*>(sync + 8);                // synthesis window / n_fft, pairs
    f2* tab = winl + N;                                        // exp(-2 pi i j / n_fft), j < n_fft
    {
        const float sc = 0.5f / (float)N;                      // 1/2 of the pairing, 1/N of the inverse DFT
        for (int i = tid; i < N; i += kIwThreads) {
            const int n = 2 * i;
            const float a = synth[min(n, pl.win - 1)], b = synth[min(n + 1, pl.win - 1)];
            winl[i] = f2{(n < pl.win) ? sc * a : 0.0f, (n + 1 < pl.win) ? sc * b : 0.0f};
        }
        for (int i = tid; i < 2 * N; i += kIwThreads) { const float2 t = twtab[i]; tab[i] = f2{t.x, t.y}; }
    }
#define IW_ITEM_PARAMS()                                                                          \
        const int sig = item / pl.segs, seg = item - sig * pl.segs;                              \
        const int q0 = seg * pl.QS, q1 = min(pl.Q, q0 + pl.QS);                                  \
        const int fa = max(0, q0 - (pl.R - 1)), f_last = min(pl.F - 1, q1 - 1);                  \
        const int nframes = f_last - fa + 1 /* >= 1 */
#define IW_ITEM_SYNC()                                                                            \
        for (int i = tid; i < pl.NR; i += kIwThreads) done[i] = 0;                               \
        if (tid < 2) sync[tid] = tid == 0 ? kIwProd : 0;                                         \
        __syncthreads()

    if (wave < kIwProd) {
        const bool active = lane < G * L;
        const int grp = active ? lane / L : 0, l = active ? lane - grp * L : 0;
        float2 xa[P], xb[P];
#define IW_TICKET(dst_)                                                                          \
    do {                                                                                         \
        int v_ = 0;                                                                              \
        if (lane == 0) v_ = __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        dst_ = __builtin_amdgcn_readfirstlane(v_);                                               \
    } while (0)
#define IW_LOAD(n_)                                                                              \
    do {                                                                                         \
        const int p_ = G * (n_) + grp;                                                           \
        const float2* sp_ = sp0 + (long long)(fa + (p_ < nframes ? p_ : 0)) * K + l;             \
        _Pragma("unroll") for (int m = 0; m < P; ++m) {                                          \
            xa[m] = sp_[L * m];                                                                  \
            xb[m] = sp_[N - 2 * l - L * m];                                                      \
        }                                                                                        \
    } while (0)
        {   // first ticket of the first segment: requested before the tables are built
            const int item = blockIdx.x;
            IW_ITEM_PARAMS();
            (void)q1;
            const float2* sp0 = spec + ((long long)sig * pl.F) * K;
            if (G * wave < nframes) IW_LOAD(wave);
        }
#pragma unroll 1
        for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
            IW_ITEM_PARAMS();
            const int n_tickets = (nframes + G - 1) / G;
            const float2* sp0 = spec + ((long long)sig * pl.F) * K;
            int n = wave;
            if (item != (int)blockIdx.x && n < n_tickets) IW_LOAD(n);
            IW_ITEM_SYNC();                                    // (first segment: also publishes winl / tab)
#pragma unroll 1
            while (n < n_tickets) {
                int n2;
                IW_TICKET(n2);
                const int p = G * n + grp;
                const bool valid = active && p < nframes;
                f2 z[P];
#pragma unroll
                for (int m = 0; m < P; ++m) {
                    const int k = l + L * m;                                   // < N
                    float2 a = xa[m], b = xb[m];
                    if (k == 0) { a.y = 0.0f; b.y = 0.0f; }                    // irfft ignores Im of DC / Nyquist
                    const f2 xk = f2{a.x, a.y}, xp = f2{b.x, -b.y};             // X[k], conj X[N-k]
                    const f2 e = cadd(xk, xp), d = csub(xk, xp);
                    const f2 tc = tab[k];
                    const f2 od = cmul(d, f2{tc.x, -tc.y});                    // (X - conj X') conj(t)
                    f2 zc = f2{e.x - od.y, -(e.y + od.x)};                     // conj(2 Z) = conj(E + i O)
                    if (!valid) zc = f2{0.0f, 0.0f};
                    z[m] = zc;
                }
                if (n2 < n_tickets) IW_LOAD(n2);                // next ticket's rows, in flight during the FFT
                const int need = fa + min(G * n + G - 1, nframes - 1) - pl.NR + pl.R - q0;
                if (need > 0)
                    for (int spin = 0; spin < kIwSpinLimit &&
                         __hip_atomic_load(&sync[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spin)
                        __builtin_amdgcn_s_sleep(2);
                float* row = smem + ((valid ? p : 0) & rmask) * pl.RS;
                F::run(z, l, valid, reinterpret_cast<f2*>(row), tab);           // Y = FFT_N(conj 2Z)
                if (valid) {
#pragma unroll
                    for (int r = 0; r < P; ++r) {               // win is even here: samples t, t+1 share the test
                        const int nn = F::bin(l, r), t = 2 * nn;
                        const f2 w = winl[nn];
                        if (t < pl.win) *reinterpret_cast<f2*>(row + t) = f2{z[r].x * w.x, -z[r].y * w.y};
                    }
                    for (int t = 2 * N + l; t < pl.win; t += L) row[t] = 0.0f;   // win > n_fft: zeros
                }
                if (valid && l == 0)
                    __hip_atomic_store(&done[p & rmask], p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                n = n2;
            }
#undef IW_TICKET
#undef IW_LOAD
            __syncthreads();
        }
    } else {
        int qk[IwPass<RJ>::IT], o4k[IwPass<RJ>::IT];
        {
            const int nq4 = pl.hop >> 2;
#pragma unroll
            for (int u = 0; u < IwPass<RJ>::IT; ++u) {
                qk[u] = (lane + 64 * u) / nq4;
                o4k[u] = (lane + 64 * u) - qk[u] * nq4;
            }
        }
#pragma unroll 1
        for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
            IW_ITEM_PARAMS();
            (void)nframes;
            IW_ITEM_SYNC();
            float* const osig = out + (long long)sig * pl.t_out;
            __builtin_amdgcn_s_setprio(3);
            IwCtx c;
            c.smem = smem; c.done = done; c.fa = fa; c.f_last = f_last; c.q0 = q0; c.R = pl.R;
            c.hop = pl.hop; c.win = pl.win; c.RS = pl.RS; c.rmask = rmask; c.t_out = (int)pl.t_out;
            c.regular = pl.win == RJ * pl.hop;
            IwPass<RJ> pa, pb;
            iw_issue<RJ>(pa, c, q0, min(q0 + pl.QB, q1), lane, qk, o4k, true);
#pragma unroll 1
            for (;;) {
                const bool more_b = pa.qe < q1;
                if (more_b) iw_issue<RJ>(pb, c, pa.qe, min(pa.qe + pl.QB, q1), lane, qk, o4k, true);
                iw_consume<RJ>(pa, c, osig, &sync[1], lane, qk, o4k);
                if (!more_b) break;
                const bool more_a = pb.qe < q1;
                if (more_a) iw_issue<RJ>(pa, c, pb.qe, min(pb.qe + pl.QB, q1), lane, qk, o4k, true);
                iw_consume<RJ>(pb, c, osig, &sync[1], lane, qk, o4k);
                if (!more_a) break;
            }
            __syncthreads();
        }
    }
#undef IW_ITEM_PARAMS
#undef IW_ITEM_SYNC
}

// overlap-add as a gather: out[t] = sum_{f : f*hop <= t < f*hop + win} frames[f][t - f*hop]
__global__ void k_ola(const float* __restrict__ frames, long long n_sig, int F, int C, int win,
                      int hop, long long t_out, int out_cl, float* __restrict__ out) {
    const long long total = n_sig * t_out;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long bc = i / t_out;
        const long long t = i - bc * t_out;
        long long f_hi = t / hop;
        if (f_hi > F - 1) f_hi = F - 1;
        long long f_lo = (t - win + hop) / hop;      // ceil((t - win + 1) / hop) for t-win+1 > 0
        if (t - win + 1 <= 0) f_lo = 0;
        float acc = 0.0f;
        for (long long f = f_lo; f <= f_hi; ++f)      // ascending frame order == tf overlap_and_add
            acc += frames[(bc * F + f) * win + (t - f * hop)];
        long long o;
        if (out_cl) { long long b = bc / C, c = bc - b * C; o = (b * t_out + t) * C + c; }
        else o = i;
        out[o] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// elementwise complex -> real
// ------------------------------------------------------------------------------------------
__global__ void k_cplx_to_real(const float2* __restrict__ x, long long n, int phase,
                               float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float2 v = x[i];
        out[i] = phase ? atan2f(v.y, v.x) : sqrtf(v.x * v.x + v.y * v.y);
    }
}

// ------------------------------------------------------------------------------------------
// decibel
// ------------------------------------------------------------------------------------------
__global__ void k_stats_init(unsigned* stats, long long n_items) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_items) { stats[2 * i] = 0u; stats[2 * i + 1] = 0xffffffffu; }
}

// log pass: out = 10 log10(max(x, amin)) - ref_term, per-item max/min into stats.
// VEC = 4: 16-byte loads / stores (item_size % 4 == 0 and 16-byte aligned bases; chunk bounds are
// then multiples of 4 as well)
template <int VEC>
__global__ __launch_bounds__(256) void k_db_log(const float* __restrict__ x, long long item_size, int chunks, DbDev db,
                         unsigned* __restrict__ stats, float* __restrict__ out) {
    typedef float vf __attribute__((ext_vector_type(VEC)));
    const long long item = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const long long nvec = item_size / VEC;
    const long long per = (nvec + chunks - 1) / chunks;
    const long long lo = chunk * per, hi = (lo + per < nvec) ? lo + per : nvec;
    const vf* xi = reinterpret_cast<const vf*>(x + item * item_size);
    vf* oi = reinterpret_cast<vf*>(out + item * item_size);
    float mx = -INFINITY, mn = INFINITY;
    long long i = lo + threadIdx.x;
    for (; i + 3 * (long long)blockDim.x < hi; i += 4 * (long long)blockDim.x) {   // four loads in flight
        vf v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = xi[i + q * (long long)blockDim.x];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                const float d = to_db(v[q][u], db);
                v[q][u] = d;
                mx = fmaxf(mx, d);
                mn = fminf(mn, d);
            }
            oi[i + q * (long long)blockDim.x] = v[q];
        }
    }
    for (; i < hi; i += blockDim.x) {
        vf v = xi[i];
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            const float d = to_db(v[u], db);
            v[u] = d;
            mx = fmaxf(mx, d);
            mn = fminf(mn, d);
        }
        oi[i] = v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        mn = fminf(mn, __shfl_xor(mn, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && mx >= mn) {
        atomicMax(&stats[2 * item], enc_f(mx));
        atomicMin(&stats[2 * item + 1], enc_f(mn));
    }
}

// clamp pass: out = max(out, item_max - dyn)  (backend.py:190-192); a whole item is skipped when
// its minimum is already above the threshold (nothing would change)
template <int VEC>
__global__ __launch_bounds__(256) void k_db_clamp(float* __restrict__ out, long long item_size, int chunks, float dyn,
                           const unsigned* __restrict__ stats) {
    typedef float vf __attribute__((ext_vector_type(VEC)));
    const long long item = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const float thr = dec_f(stats[2 * item]) - dyn;
    if (dec_f(stats[2 * item + 1]) >= thr) return;
    const long long nvec = item_size / VEC;
    const long long per = (nvec + chunks - 1) / chunks;
    const long long lo = chunk * per, hi = (lo + per < nvec) ? lo + per : nvec;
    vf* oi = reinterpret_cast<vf*>(out + item * item_size);
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        vf v = oi[i];
#pragma unroll
        for (int u = 0; u < VEC; ++u) v[u] = fmaxf(v[u], thr);
        oi[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// generic fp32-MFMA GEMM:  C[r][n] = sum_k A(r,k) * Bm[k][n]
// rows r are decomposed as r = (r2*D1 + r1)*D0 + r0 for input and output addressing
// ------------------------------------------------------------------------------------------
enum { A_PLAIN = 0, A_CABS = 1, A_FRAME = 2, A_CPLX = 3 };
enum { E_PLAIN = 0, E_CPLX = 1, E_WINDOW = 2, E_DB = 3 };

struct RowMap {
    long long rows;
    int D0, D1;
    long long s2, s1, s0;   // base = r2*s2 + r1*s1 + r0*s0
    long long es;           // element stride along k (input) / n (output)
    KPR_DEV long long base(long long r, long long* r2_out = nullptr) const {
        long long r0 = r % D0, q = r / D0;
        long long r1 = q % D1, r2 = q / D1;
        if (r2_out) *r2_out = r2;
        return r2 * s2 + r1 * s1 + r0 * s0;
    }
};

struct GemmArgs {
    RowMap in, out;
    int Kdim, N;            // reduction length, output columns
    int ldb;                // row stride of Bm
    // A_FRAME: time geometry
    long long T; int hop, pad_left; long long t_es;
    const float* window;    // A_FRAME analysis window / E_WINDOW synthesis window
    int win;
    DbDev db;
    unsigned* stats;
    int has_kr;             // per 64-column block k range
    short klo[kMaxTiles], khi[kMaxTiles];   // per 16-col tile (multiples of 4)
};

template <int AMODE>
KPR_DEV float gemm_load_a(const float* __restrict__ a, const GemmArgs& ga, long long r, int k) {
    if (r >= ga.in.rows || k >= ga.Kdim) return 0.0f;
    if constexpr (AMODE == A_PLAIN) {
        return a[ga.in.base(r) + (long long)k * ga.in.es];
    } else if constexpr (AMODE == A_CABS) {
        const float2 v = reinterpret_cast<const float2*>(a)[ga.in.base(r) + (long long)k * ga.in.es];
        return sqrtf(v.x * v.x + v.y * v.y);
    } else if constexpr (AMODE == A_CPLX) {
        // k indexes interleaved (re, im): complex element k>>1, part k&1
        return a[2 * (ga.in.base(r) + (long long)(k >> 1) * ga.in.es) + (k & 1)];
    } else {  // A_FRAME: rows are frames (r2 = b, r1 = c, r0 = f)
        long long r0 = r % ga.in.D0, q = r / ga.in.D0;
        long long r1 = q % ga.in.D1, r2 = q / ga.in.D1;
        long long t = r0 * ga.hop - ga.pad_left + k;
        if (t < 0 || t >= ga.T) return 0.0f;
        return a[r2 * ga.in.s2 + r1 * ga.in.s1 + t * ga.t_es] * ga.window[k];
    }
}

template <int AMODE, int EPI>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ a,
                                              const float* __restrict__ bm, GemmArgs ga,
                                              float* __restrict__ out) {
    constexpr int TM = 64, TN = 64, KC = 16, LDX = 18, LDB = 80;
    __shared__ float Xs[TM * LDX];
    __shared__ float Bs[KC * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = (long long)blockIdx.x * TM;
    const int col0 = blockIdx.y * TN;
    int klo = 0, khi = (ga.Kdim + 3) & ~3;
    if (ga.has_kr) {
        klo = 1 << 30; khi = 0;
        for (int t = col0 / 16; t < (col0 + TN) / 16 && t * 16 < ga.N; ++t) {
            klo = min(klo, (int)ga.klo[t]); khi = max(khi, (int)ga.khi[t]);
        }
        if (klo > khi) klo = khi;
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int jcol = lane & 15, kq = lane >> 4;
    for (int kc = klo; kc < khi; kc += KC) {
        {   // stage X tile: thread -> (row = tid>>2, 4 consecutive k)
            const int r = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                Xs[r * LDX + kk + i] = gemm_load_a<AMODE>(a, ga, row0 + r, kc + kk + i);
            // stage B tile: thread -> (k = tid>>4, 4 consecutive n)
            const int kb = tid >> 4, nn = (tid & 15) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int kg = kc + kb, ng = col0 + nn + i;
                Bs[kb * LDB + nn + i] =
                    (kg < ga.Kdim && ng < ga.N) ? bm[(long long)kg * ga.ldb + ng] : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            const float bfrag = Xs[(wave * 16 + jcol) * LDX + ks + kq];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float afrag = Bs[(ks + kq) * LDB + nt * 16 + jcol];
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag, bfrag, acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // lane holds C[row = row0 + wave*16 + jcol][col = col0 + nt*16 + 4*kq + r]
    const long long r = row0 + wave * 16 + jcol;
    if (r >= ga.out.rows) return;
    long long r2 = 0;
    const long long ob = ga.out.base(r, &r2);
    float mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = col0 + nt * 16 + 4 * kq + i;
            if (n >= ga.N) continue;
            float v = acc[nt][i];
            if constexpr (EPI == E_PLAIN) {
                out[ob + (long long)n * ga.out.es] = v;
            } else if constexpr (EPI == E_CPLX) {
                out[2 * (ob + (long long)(n >> 1) * ga.out.es) + (n & 1)] = v;
            } else if constexpr (EPI == E_WINDOW) {
                out[ob + (long long)n * ga.out.es] = (n < ga.win) ? v * ga.window[n] : 0.0f;
            } else {
                v = to_db(v, ga.db);
                mx = fmaxf(mx, v); mn = fminf(mn, v);
                out[ob + (long long)n * ga.out.es] = v;
            }
        }
    }
    if constexpr (EPI == E_DB) {
        if (mx >= mn) {
            atomicMax(&ga.stats[2 * r2], enc_f(mx));
            atomicMin(&ga.stats[2 * r2 + 1], enc_f(mn));
        }
    }
}

// development aid for PMC calibration: stream-read n float2 (8 B per lane, the access width of the
// frame loads) and write one float per workgroup
__global__ void k_calib_read8(const float2* __restrict__ x, long long n, float* __restrict__ out) {
    float acc = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float2 v = x[i];
        acc += v.x + v.y;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// zero-fill columns [n0, n1) of every output row (E_WINDOW with win_length > n_fft)
__global__ void k_fill_cols(float* out, long long rows, long long ld, int n0, int n1) {
    const long long total = rows * (n1 - n0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        out[(i / (n1 - n0)) * ld + n0 + (i % (n1 - n0))] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// host side: table caches
// ------------------------------------------------------------------------------------------
static std::mutex g_mu;
static std::map<std::pair<int, int>, float2*> g_tw;           // (device, n_fft) -> twiddles
static std::map<std::pair<int, int>, float*> g_dft_fwd;       // (device, n_fft) -> [n_fft][2K]
static std::map<std::pair<int, int>, float*> g_dft_inv;       // (device, n_fft) -> [2K][n_fft]

static int cur_device(int* dev) {
    KPR_HIP(hipGetDevice(dev));
    return 0;
}

static int get_twiddles(int n_fft, const float2** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tw.find({dev, n_fft});
    if (it == g_tw.end()) {
        std::vector<float2> h(n_fft);
        for (int j = 0; j < n_fft; ++j) {
            double a = -2.0 * M_PI * (double)j / (double)n_fft;
            h[j] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        KPR_HIP(hipMalloc(&d, sizeof(float2) * n_fft));
        KPR_HIP(hipMemcpy(d, h.data(), sizeof(float2) * n_fft, hipMemcpyHostToDevice));
        it = g_tw.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// Bluestein tables for an even n_fft that is not a power of two (k_stft_bs): M, then
// [w: M][Bt: M][t: NCr + 1] as float2; Bt = FFT_M(chirp) / (2M) computed in double precision
static int bluestein_m(int n_fft) {
    if (n_fft < 4 || (n_fft & 1)) return 0;
    const int ncr = n_fft / 2;
    int m = 128;
    while (m < 2 * ncr - 1) m *= 2;
    return m <= 1024 ? m : 0;
}

static std::map<std::pair<int, int>, float2*> g_bs;

static int get_bluestein(int n_fft, const float2** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_bs.find({dev, n_fft});
    if (it == g_bs.end()) {
        const int ncr = n_fft / 2, m = bluestein_m(n_fft);
        std::vector<double> wr(ncr), wi(ncr), br(m, 0.0), bi(m, 0.0);
        for (int n = 0; n < ncr; ++n) {
            const long long n2 = ((long long)n * n) % (2LL * ncr);           // exact angle reduction
            const double a = -M_PI * (double)n2 / (double)ncr;
            wr[n] = std::cos(a); wi[n] = std::sin(a);
        }
        for (int n = 0; n < ncr; ++n) { br[n] = wr[n]; bi[n] = -wi[n]; }
        for (int n = 1; n < ncr; ++n) { br[m - n] = wr[n]; bi[m - n] = -wi[n]; }
        // O(M^2) DFT of the chirp in double precision (once per n_fft and device; M <= 1024)
        std::vector<float2> h(2 * (size_t)m + ncr + 1);
        for (int n = 0; n < m; ++n) h[n] = n < ncr ? make_float2((float)wr[n], (float)wi[n]) : make_float2(0.f, 0.f);
        for (int k = 0; k < m; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < m; ++n) {
                if (br[n] == 0.0 && bi[n] == 0.0) continue;
                const double a = -2.0 * M_PI * (double)(((long long)k * n) % m) / (double)m;
                const double c = std::cos(a), sn = std::sin(a);
                sr += br[n] * c - bi[n] * sn;
                si += br[n] * sn + bi[n] * c;
            }
            h[m + k] = make_float2((float)(sr / (2.0 * m)), (float)(si / (2.0 * m)));
        }
        for (int k = 0; k <= ncr; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n_fft;
            h[2 * (size_t)m + k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        KPR_HIP(hipMalloc(&d, sizeof(float2) * h.size()));
        KPR_HIP(hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
        it = g_bs.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// forward DFT matrix [n_fft rows n][2K cols]: col 2k = cos(2 pi k n/N), col 2k+1 = -sin(...)
static int get_dft_fwd(int n_fft, const float** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dft_fwd.find({dev, n_fft});
    if (it == g_dft_fwd.end()) {
        const int K = n_fft / 2 + 1;
        std::vector<float> h((size_t)n_fft * 2 * K);
        for (int n = 0; n < n_fft; ++n)
            for (int k = 0; k < K; ++k) {
                long long kn = ((long long)k * n) % n_fft;     // exact angle reduction
                double a = 2.0 * M_PI * (double)kn / (double)n_fft;
                h[(size_t)n * 2 * K + 2 * k] = (float)std::cos(a);
                h[(size_t)n * 2 * K + 2 * k + 1] = (float)(-std::sin(a));
            }
        float* d = nullptr;
        KPR_HIP(hipMalloc(&d, h.size() * sizeof(float)));
        KPR_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        it = g_dft_fwd.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// inverse real DFT matrix [2K rows][n_fft cols]: row 2k = c_k cos(2 pi k n/N)/N,
// row 2k+1 = -c_k sin(2 pi k n/N)/N, c_k = 1 for DC (and Nyquist when N even) else 2
static int get_dft_inv(int n_fft, const float** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dft_inv.find({dev, n_fft});
    if (it == g_dft_inv.end()) {
        const int K = n_fft / 2 + 1;
        std::vector<float> h((size_t)2 * K * n_fft);
        for (int k = 0; k < K; ++k) {
            const bool edge = (k == 0) || ((n_fft % 2 == 0) && k == n_fft / 2);
            const double ck = (edge ? 1.0 : 2.0) / (double)n_fft;
            for (int n = 0; n < n_fft; ++n) {
                long long kn = ((long long)k * n) % n_fft;
                double a = 2.0 * M_PI * (double)kn / (double)n_fft;
                h[(size_t)(2 * k) * n_fft + n] = (float)(ck * std::cos(a));
                h[(size_t)(2 * k + 1) * n_fft + n] = edge ? 0.0f : (float)(-ck * std::sin(a));
            }
        }
        float* d = nullptr;
        KPR_HIP(hipMalloc(&d, h.size() * sizeof(float)));
        KPR_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        it = g_dft_inv.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// ------------------------------------------------------------------------------------------
// host side: geometry / validation
// ------------------------------------------------------------------------------------------
static bool fast_nfft(int n_fft) {
    return n_fft == 256 || n_fft == 512 || n_fft == 1024 || n_fft == 2048;
}

static long long frames_of(const kpr_stft_geom* s) {
    long long t = s->time + (s->pad_begin ? (s->n_fft - s->hop_length) : 0);
    if (s->pad_end) return (t + s->hop_length - 1) / s->hop_length;
    if (t < s->win_length) return 0;
    return 1 + (t - s->win_length) / s->hop_length;
}

static int check_geom(const kpr_stft_geom* s) {
    if (!s) return fail(KPR_E_BADARG, "geometry is NULL");
    if (s->batch < 0 || s->channels <= 0 || s->time < 0)
        return fail(KPR_E_BADARG, "bad batch/channels/time (%lld, %d, %lld)", (long long)s->batch,
                    s->channels, (long long)s->time);
    if (s->n_fft < 2 || s->win_length < 1 || s->hop_length < 1)
        return fail(KPR_E_BADARG, "bad n_fft/win_length/hop_length (%d, %d, %d)", s->n_fft,
                    s->win_length, s->hop_length);
    if ((unsigned)s->in_layout > 1u || (unsigned)s->out_layout > 1u)
        return fail(KPR_E_BADARG, "bad layout enum");
    if (s->pad_begin && s->n_fft < s->hop_length)
        return fail(KPR_E_BADARG, "pad_begin needs n_fft >= hop_length");
    // the kernels address one (batch item, channel) signal with 32-bit element offsets
    if (s->time * (long long)s->channels >= (1LL << 30))
        return fail(KPR_E_UNSUPPORTED, "time * channels = %lld elements per batch item: 2^30 or more is not supported",
                    (long long)(s->time * (long long)s->channels));
    return 0;
}

static Geom make_geom(const kpr_stft_geom* s, long long F) {
    Geom g;
    g.F = (int)F;
    g.C = s->channels;
    g.T = s->time;
    g.total_frames = s->batch * s->channels * F;
    g.n_fft = s->n_fft;
    g.win = s->win_length;
    g.hop = s->hop_length;
    g.pad_left = s->pad_begin ? (s->n_fft - s->hop_length) : 0;
    g.K = s->n_fft / 2 + 1;
    // with one channel the two layouts are the same memory image: take the contiguous paths
    // (Kapre's default is channels_last, so this is the common case)
    g.in_cl = s->in_layout == KPR_CHANNELS_LAST && s->channels > 1;
    g.out_cl = s->out_layout == KPR_CHANNELS_LAST && s->channels > 1;
    g.cfast = 0;
    return g;
}

static DbDev make_db(const kpr_db_params* db) {
    DbDev d{0, 1e-5f, 0.0f, 80.0f};
    if (db && db->enabled) {
        d.enabled = 1;
        d.amin = db->amin;
        d.ref_term = (float)(10.0 * std::log10(std::max((double)db->amin, (double)db->ref_value)));
        d.dyn = db->dynamic_range;
    }
    return d;
}

static int check_db(const kpr_db_params* db) {
    if (db && db->enabled) {
        // same checks (and order) as backend.py:168-173
        if (!(db->ref_value > 0)) return fail(KPR_E_BADARG, "ref_value must be positive");
        if (!(db->amin > 0)) return fail(KPR_E_BADARG, "amin must be positive");
        if (!(db->dynamic_range > 0)) return fail(KPR_E_BADARG, "dynamic_range must be positive");
    }
    return 0;
}

static int grid_1d(long long n, int block, int cap = 256 * 16) {
    long long b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

static int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(KPR_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// frame-row maps for GEMM paths: rows are global frames g = (b*C + c)*F + f
static RowMap frames_out_map(const Geom& g, long long Q) {
    RowMap m;
    m.rows = g.total_frames; m.D0 = g.F; m.D1 = g.C;
    if (g.out_cl) { m.s2 = (long long)g.F * Q * g.C; m.s1 = 1; m.s0 = Q * g.C; m.es = g.C; }
    else { m.s2 = (long long)g.C * g.F * Q; m.s1 = (long long)g.F * Q; m.s0 = Q; m.es = 1; }
    return m;
}
static RowMap frames_contig_map(const Geom& g, long long Q) {
    RowMap m;
    m.rows = g.total_frames; m.D0 = g.F; m.D1 = g.C;
    m.s2 = (long long)g.C * g.F * Q; m.s1 = (long long)g.F * Q; m.s0 = Q; m.es = 1;
    return m;
}

template <int AMODE, int EPI>
static int run_gemm(const float* a, const float* bm, const GemmArgs& ga, float* out,
                    hipStream_t st) {
    if (ga.in.rows <= 0 || ga.N <= 0) return 0;
    dim3 grid((unsigned)((ga.in.rows + 63) / 64), (unsigned)((ga.N + 63) / 64));
    hipLaunchKernelGGL((k_gemm<AMODE, EPI>), grid, dim3(256), 0, st, a, bm, ga, out);
    return launch_check("k_gemm");
}

// STFT of every frame into `out` (complex64), through the DFT-as-GEMM path
static int stft_gemm(const float* x, const kpr_stft_geom* s, const Geom& g, const float* window,
                     float* out_cplx, bool out_contig, hipStream_t st) {
    const float* dft = nullptr;
    if (int e = get_dft_fwd(g.n_fft, &dft)) return e;
    GemmArgs ga{};
    ga.in.rows = g.total_frames; ga.in.D0 = g.F; ga.in.D1 = g.C;
    if (g.in_cl) { ga.in.s2 = g.T * g.C; ga.in.s1 = 1; ga.t_es = g.C; }
    else { ga.in.s2 = (long long)g.C * g.T; ga.in.s1 = g.T; ga.t_es = 1; }
    ga.in.s0 = 0; ga.in.es = 0;
    ga.out = out_contig ? frames_contig_map(g, g.K) : frames_out_map(g, g.K);
    ga.Kdim = std::min(g.win, g.n_fft);
    ga.N = 2 * g.K;
    ga.ldb = 2 * g.K;
    ga.T = g.T; ga.hop = g.hop; ga.pad_left = g.pad_left;
    ga.window = window; ga.win = g.win;
    (void)s;
    return run_gemm<A_FRAME, E_CPLX>(x, dft, ga, out_cplx, st);
}

static int device_cus(int* cus);

// Kernels that use more than 64 KiB of dynamic LDS must opt in, once per (kernel, device).
// (benign race: the call is idempotent)
struct LdsOptIn { bool done[64] = {}; };
static int allow_big_lds(LdsOptIn& st, const void* fn) {
    int dev = 0;
    KPR_HIP(hipGetDevice(&dev));
    const bool slot = dev >= 0 && dev < 64;
    if (!slot || !st.done[dev]) {
        KPR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (slot) st.done[dev] = true;
    }
    return 0;
}
static long long* g_debug_stamps = nullptr;   // development aid: kpr_debug_stamps()

template <int NC, int NW>
static int launch_istft_fused(const float2* spec, const kpr_stft_geom* s, long long F,
                              const float* synth, const float2* tw, float* out, hipStream_t st,
                              bool* launched) {
    constexpr int L = NC / kPts, G = 64 / L;
    *launched = false;
    const int win = s->win_length, hop = s->hop_length;
    if (hop > win || F < 1) return 0;                       // gaps between frames: two-kernel path
    const int R = (win + hop - 1) / hop;
    const int RS = ((std::max(win, NC) + 3) & ~3) + 4;
    const int spare = (G > 1) ? NW * (G - 1) : 0;           // scratch rows for idle frame slots
    // rows: as many as fit next to a second workgroup on the CU (80 KiB each), at most 16; if that
    // leaves too few new frames per block, take the whole CU (160 KiB) instead
    // (an 8-wave workgroup at ~190 VGPRs fills the CU's register file on its own)
    int NR = (NW == 8) ? 0 : std::min(16, (int)(80 * 1024 / (sizeof(float) * RS)) - spare);
    if (NR < R + 3) NR = std::min(16, (int)(160 * 1024 / (sizeof(float) * RS)) - spare);
    if (NR < R + 1) return 0;
    IstftPlan pl;
    pl.n_sig = (long long)s->batch * s->channels;
    pl.t_out = (F - 1) * (long long)hop + win;
    pl.F = (int)F; pl.C = s->channels; pl.win = win; pl.hop = hop;
    pl.NR = NR; pl.R = R; pl.FB = NR - R + 1;
    pl.RS = RS;
    pl.chunks = (int)((pl.t_out + (long long)pl.FB * hop - 1) / ((long long)pl.FB * hop));
    pl.spec_cl = s->out_layout == KPR_CHANNELS_LAST;
    pl.wave_cl = s->in_layout == KPR_CHANNELS_LAST;
    pl.vec4 = hop % 4 == 0 && win % 4 == 0 && !(pl.wave_cl && s->channels > 1) &&
              (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const size_t lds = sizeof(float) * (size_t)(NR + spare) * pl.RS;
    if (lds > 160 * 1024) return 0;
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_istft_fused<NC, NW>))) return e;
    const long long nblocks = pl.n_sig * pl.chunks;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const int per_cu = std::max(1, (int)(160 * 1024 / lds));
    const unsigned grid = (unsigned)std::min<long long>(nblocks, (long long)std::min(per_cu, 2) * cus);
    hipLaunchKernelGGL((k_istft_fused<NC, NW>), dim3(grid), dim3(NW * 64), lds, st, spec, pl, synth,
                       tw, out, nblocks);
    *launched = true;
    return launch_check("k_istft_fused");
}


static int device_cus(int* cus) {
    int dev = 0;
    KPR_HIP(hipGetDevice(&dev));
    static int cached[64] = {0};
    int v = 256;
    if (dev >= 0 && dev < 64) {
        if (!cached[dev]) {
            int q = 0;
            KPR_HIP(hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev));
            cached[dev] = q > 0 ? q : 256;
        }
        v = cached[dev];
    }
    *cus = v;
    return 0;
}

// segments per signal for k_istft_ws: rounds of `cus` workgroups x (blocks + halo frames) per segment
static int istft_ws_segments(long long n_sig, int Q, int R, int cus) {
    int best = 1;
    double best_cost = 1e300;
    const int smax = std::max(1, std::min(64, Q / (4 * R)));
    for (int sg = 1; sg <= smax; ++sg) {
        const long long rounds = (n_sig * sg + cus - 1) / cus;
        const double cost = (double)rounds * ((Q + sg - 1) / sg + R - 1);
        if (cost < best_cost * 0.999) { best_cost = cost; best = sg; }
    }
    return best;
}

template <int NC, int RJ>
static int launch_istft_ws_inst(const float2* spec, const IstftWsPlan& pl, size_t lds, unsigned grid,
                                const float* synth, const float2* tw, float* out, int nitems, hipStream_t st) {
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_istft_ws<NC, RJ>))) return e;
    hipLaunchKernelGGL((k_istft_ws<NC, RJ>), dim3(grid), dim3(kIwThreads), lds, st, spec, pl, synth, tw, out,
                       nitems, g_debug_stamps);
    return launch_check("k_istft_ws");
}

// Plan of the ring kernels (k_istft_ws, k_istft_ws_mr); false when they do not apply.
//   row_min: floats a ring row needs as FFT exchange buffer, G: frames per producer ticket,
//   spare: extra rows (exchange rows of idle frame slots), extra: LDS bytes behind rows / flags / counters
static bool istft_ws_plan(const kpr_stft_geom* s, long long F, const float* out, int row_min, int G, int spare,
                          size_t extra, int cus, IstftWsPlan* plo, size_t* lds, int* rj, long long* nitems) {
    const int win = s->win_length, hop = s->hop_length;
    if (hop > win || F < 1 || getenv("KPR_ISTFT_NO_WS")) return false;
    // four samples per lane in the overlap-add: hop, win multiples of 4, contiguous waveform;
    // and contiguous spectrogram rows (channels_first, or one channel)
    if (hop % 4 || win % 4 || (s->in_layout == KPR_CHANNELS_LAST && s->channels > 1) ||
        (s->out_layout == KPR_CHANNELS_LAST && s->channels > 1) || (reinterpret_cast<uintptr_t>(out) & 15))
        return false;
    const long long n_sig = (long long)s->batch * s->channels;
    const long long t_out = (F - 1) * (long long)hop + win;
    if (n_sig * 64 >= (1LL << 31) || t_out + hop >= (1LL << 31)) return false;
    IstftWsPlan pl;
    pl.t_out = t_out;
    pl.F = (int)F; pl.C = s->channels; pl.win = win; pl.hop = hop;
    pl.R = (win + hop - 1) / hop;
    const int RJ = pl.R <= 2 ? 2 : pl.R <= 4 ? 4 : 8;         // rows read per sample group
    if (pl.R > 8) return false;
    const int per_pass = 64 * (kIwReads / RJ);                 // sample groups per consumer pass
    if (hop / 4 > per_pass) return false;                      // a hop block must fit one pass
    pl.RS = ((std::max(win, row_min) + 3) & ~3) + 4;
    pl.Q = (int)F - 1 + pl.R;
    pl.QB = std::min(16, per_pass / (hop / 4));
    auto bytes = [&](int nr) { return sizeof(float) * (size_t)(nr + spare) * pl.RS + sizeof(int) * (size_t)(nr + 8) + extra; };
    int NR = 128;
    while (NR > 1 && bytes(NR) > 160 * 1024) NR >>= 1;
    // room for the frames of the two passes in flight (R-1+2*QB), the producers' tickets and slack
    if (NR < pl.R - 1 + 2 * pl.QB + 2 * G + 1 || pl.R - 1 + pl.QB > 64) return false;
    pl.NR = NR;
    pl.segs = istft_ws_segments(n_sig, pl.Q, pl.R, cus);
    pl.QS = (pl.Q + pl.segs - 1) / pl.segs;
    pl.segs = (pl.Q + pl.QS - 1) / pl.QS;                       // no empty segment
    *plo = pl;
    *lds = bytes(NR);
    *rj = RJ;
    *nitems = n_sig * pl.segs;
    return true;
}

template <int NC>
static int launch_istft_ws(const float2* spec, const kpr_stft_geom* s, long long F, const float* synth,
                           const float2* tw, float* out, hipStream_t st, bool* launched) {
    constexpr int L = NC / kPts, G = 64 / L;
    *launched = false;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    IstftWsPlan pl;
    size_t lds;
    int RJ;
    long long nitems;
    if (!istft_ws_plan(s, F, out, NC, G, kIwProd * (G - 1), 0, cus, &pl, &lds, &RJ, &nitems)) return 0;
    const unsigned grid = (unsigned)std::min<long long>(nitems, cus);
    *launched = true;
    switch (RJ) {
        case 2:  return launch_istft_ws_inst<NC, 2>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
        case 4:  return launch_istft_ws_inst<NC, 4>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
        default: return launch_istft_ws_inst<NC, 8>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
    }
}

template <int NC, int MODE, bool OUT_CL>
static int launch_stft_inst(const float* x, const Geom& g, const float* window, const float2* tw,
                            void* out, hipStream_t st) {
    constexpr int L = NC / kPts, G = 64 / L;
    const long long ngroups = (g.total_frames + G - 1) / G;          // wave-loads of G frames
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = stft_lds_bytes(NC);
    // workgroups the hardware can keep resident per CU (registers + LDS), asked from the runtime
    static int resident_dev[64] = {0};
    int dev = 0;
    KPR_HIP(hipGetDevice(&dev));
    int& resident = resident_dev[(dev >= 0 && dev < 64) ? dev : 0];
    if (!resident) {
        KPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stft<NC, MODE, OUT_CL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int nb = 0;
        KPR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_stft<NC, MODE, OUT_CL>,
                                                             64 * KPR_STFT_WAVES, lds));
        resident = std::max(1, nb);
        if (getenv("KPR_VERBOSE"))
            fprintf(stderr, "[kapre_hip] k_stft<%d,%d,%d>: %d resident workgroups per CU (lds %zu B)\n", NC, MODE,
                    (int)OUT_CL, resident, lds);
    }
    // at least one group per wave when there is enough work
    const unsigned grid = (unsigned)std::max<long long>(
        1, std::min<long long>((ngroups + KPR_STFT_WAVES - 1) / KPR_STFT_WAVES, (long long)resident * cus));
    hipLaunchKernelGGL((k_stft<NC, MODE, OUT_CL>), dim3(grid), dim3(64 * KPR_STFT_WAVES), lds, st, x, g,
                       window, tw, out, ngroups, g_debug_stamps);
    return launch_check("k_stft");
}

template <int NC>
static int launch_stft_fast(const float* x, const Geom& g, const float* window, const float2* tw,
                            int mode, void* out, hipStream_t st) {
    const bool cl = g.out_cl != 0;
    switch (mode) {
        case KPR_OUT_COMPLEX:
            return cl ? launch_stft_inst<NC, KPR_OUT_COMPLEX, true>(x, g, window, tw, out, st)
                      : launch_stft_inst<NC, KPR_OUT_COMPLEX, false>(x, g, window, tw, out, st);
        case KPR_OUT_MAGNITUDE:
            return cl ? launch_stft_inst<NC, KPR_OUT_MAGNITUDE, true>(x, g, window, tw, out, st)
                      : launch_stft_inst<NC, KPR_OUT_MAGNITUDE, false>(x, g, window, tw, out, st);
        default:
            return cl ? launch_stft_inst<NC, KPR_OUT_PHASE, true>(x, g, window, tw, out, st)
                      : launch_stft_inst<NC, KPR_OUT_PHASE, false>(x, g, window, tw, out, st);
    }
}

// Bluestein STFT (even n_fft that is not a power of two, n_fft <= 1024, win_length <= n_fft)
static bool bluestein_ok(const kpr_stft_geom* s) {
    return !fast_nfft(s->n_fft) && bluestein_m(s->n_fft) > 0 && s->win_length <= s->n_fft;
}

template <int M>
static int launch_stft_bs_m(const float* x, const Geom& g, const float* window, const float2* tw,
                            const float2* bs, int mode, void* out, hipStream_t st) {
    constexpr int L = M / kPts, G = 64 / L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = sizeof(float) * ((size_t)4 * G * bs_slot_words(M, g.n_fft / 2) + 2 * (size_t)(3 * M + g.n_fft / 2 + 2));
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_stft_bs<M>))) return e;
    const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_stft_bs<M>), dim3(grid), dim3(256), lds, st, x, g, window, tw, bs, mode, out, ngroups);
    return launch_check("k_stft_bs");
}

// mixed-radix plans (kpr_fft_mr.h): n_fft -> (R2, R3), N = n_fft / 2 = 20 * R2 * R3
static bool mixed_radix_plan(int n_fft, int* r2, int* r3) {
    switch (n_fft) {
        case 160:  *r2 = 4;  *r3 = 1; return true;
        case 200:  *r2 = 5;  *r3 = 1; return true;
        case 320:  *r2 = 4;  *r3 = 2; return true;
        case 400:  *r2 = 10; *r3 = 1; return true;
        case 640:  *r2 = 4;  *r3 = 4; return true;
        case 800:  *r2 = 20; *r3 = 1; return true;
        case 1000: *r2 = 5;  *r3 = 5; return true;
        default:   return false;
    }
}

template <int R2, int R3>
static int launch_stft_mr_inst(const float* x, const Geom& g, const float* window, const float2* tw, int mode,
                               void* out, hipStream_t st) {
    typedef MrFft<R2, R3> F;
    constexpr int G = 64 / F::L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = sizeof(float) * 2 * ((size_t)4 * G * (F::N + 1) + 3 * (size_t)F::N);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_stft_mr<R2, R3>))) return e;
    const int per_cu = std::max(1, std::min(3, (int)(160 * 1024 / lds)));   // ~150 VGPRs: three workgroups per CU
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_stft_mr<R2, R3>), dim3(grid), dim3(256), lds, st, x, g, window, tw, mode, out, ngroups);
    return launch_check("k_stft_mr");
}

static int launch_stft_mr(const float* x, const Geom& g, const float* window, int mode, void* out, hipStream_t st) {
    const float2* tw = nullptr;
    if (int e = get_twiddles(g.n_fft, &tw)) return e;
    switch (g.n_fft) {
        case 160:  return launch_stft_mr_inst<4, 1>(x, g, window, tw, mode, out, st);
        case 200:  return launch_stft_mr_inst<5, 1>(x, g, window, tw, mode, out, st);
        case 320:  return launch_stft_mr_inst<4, 2>(x, g, window, tw, mode, out, st);
        case 400:  return launch_stft_mr_inst<10, 1>(x, g, window, tw, mode, out, st);
        case 640:  return launch_stft_mr_inst<4, 4>(x, g, window, tw, mode, out, st);
        case 800:  return launch_stft_mr_inst<20, 1>(x, g, window, tw, mode, out, st);
        default:   return launch_stft_mr_inst<5, 5>(x, g, window, tw, mode, out, st);
    }
}

static int launch_stft_bs(const float* x, const Geom& g, const float* window, int mode, void* out,
                          hipStream_t st) {
    {   // 2^a 5^b sizes: one mixed-radix FFT per frame instead of two chirp-z FFTs
        int r2, r3;
        if (mixed_radix_plan(g.n_fft, &r2, &r3) && !getenv("KPR_NO_MIXED_RADIX"))
            return launch_stft_mr(x, g, window, mode, out, st);
    }
    const int m = bluestein_m(g.n_fft);
    const float2 *tw = nullptr, *bs = nullptr;
    if (int e = get_twiddles(2 * m, &tw)) return e;
    if (int e = get_bluestein(g.n_fft, &bs)) return e;
    switch (m) {
        case 128:  return launch_stft_bs_m<128>(x, g, window, tw, bs, mode, out, st);
        case 256:  return launch_stft_bs_m<256>(x, g, window, tw, bs, mode, out, st);
        case 512:  return launch_stft_bs_m<512>(x, g, window, tw, bs, mode, out, st);
        default:   return launch_stft_bs_m<1024>(x, g, window, tw, bs, mode, out, st);
    }
}

template <int M>
static int launch_irfft_bs_m(const float2* spec, const Geom& g, const float* synth, const float2* tw,
                             const float2* bs, float* frames, hipStream_t st) {
    constexpr int L = M / kPts, G = 64 / L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = sizeof(float) * ((size_t)4 * G * ((M + M / 32 + 24 + 3) / 4 * 4) +
                                        2 * (size_t)(3 * M + g.n_fft / 2 + 2));
    const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_irfft_bs<M>), dim3(grid), dim3(256), lds, st, spec, g, synth, tw, bs, frames, ngroups);
    return launch_check("k_irfft_bs");
}

template <int R2, int R3>
static int launch_irfft_mr_inst(const float2* spec, const Geom& g, const float* synth, const float2* tw,
                                float* frames, hipStream_t st) {
    typedef MrFft<R2, R3> F;
    constexpr int G = 64 / F::L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = sizeof(float) * 2 * ((size_t)4 * G * (F::N + 1) + 3 * (size_t)F::N);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_irfft_mr<R2, R3>))) return e;
    const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_irfft_mr<R2, R3>), dim3(grid), dim3(256), lds, st, spec, g, synth, tw, frames, ngroups);
    return launch_check("k_irfft_mr");
}

static int launch_irfft_mr(const float2* spec, const Geom& g, const float* synth, float* frames, hipStream_t st) {
    const float2* tw = nullptr;
    if (int e = get_twiddles(g.n_fft, &tw)) return e;
    switch (g.n_fft) {
        case 160:  return launch_irfft_mr_inst<4, 1>(spec, g, synth, tw, frames, st);
        case 200:  return launch_irfft_mr_inst<5, 1>(spec, g, synth, tw, frames, st);
        case 320:  return launch_irfft_mr_inst<4, 2>(spec, g, synth, tw, frames, st);
        case 400:  return launch_irfft_mr_inst<10, 1>(spec, g, synth, tw, frames, st);
        case 640:  return launch_irfft_mr_inst<4, 4>(spec, g, synth, tw, frames, st);
        case 800:  return launch_irfft_mr_inst<20, 1>(spec, g, synth, tw, frames, st);
        default:   return launch_irfft_mr_inst<5, 5>(spec, g, synth, tw, frames, st);
    }
}

template <int R2, int R3, int RJ>
static int launch_istft_ws_mr_inst(const float2* spec, const IstftWsPlan& pl, size_t lds, unsigned grid,
                                   const float* synth, const float2* tw, float* out, int nitems, hipStream_t st) {
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_istft_ws_mr<R2, R3, RJ>))) return e;
    hipLaunchKernelGGL((k_istft_ws_mr<R2, R3, RJ>), dim3(grid), dim3(kIwThreads), lds, st, spec, pl, synth, tw,
                       out, nitems);
    return launch_check("k_istft_ws_mr");
}

template <int R2, int R3>
static int launch_istft_ws_mr_plan(const float2* spec, const kpr_stft_geom* s, long long F, const float* synth,
                                   const float2* tw, float* out, hipStream_t st, bool* launched) {
    typedef MrFft<R2, R3> FF;
    constexpr int G = 64 / FF::L;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    IstftWsPlan pl;
    size_t lds;
    int RJ;
    long long nitems;
    // rows double as exchange rows (N complex words); window pairs and the twiddle table behind the counters
    if (!istft_ws_plan(s, F, out, 2 * FF::N, G, 0, sizeof(float) * 2 * 3 * (size_t)FF::N, cus, &pl, &lds, &RJ, &nitems))
        return 0;
    if (RJ > 4) return 0;                                       // more than four overlapping frames: two-kernel path
    const unsigned grid = (unsigned)std::min<long long>(nitems, cus);
    *launched = true;
    if (RJ == 2) return launch_istft_ws_mr_inst<R2, R3, 2>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
    return launch_istft_ws_mr_inst<R2, R3, 4>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
}

// ring kernel for the mixed-radix transform sizes; *launched stays false when it does not apply
static int launch_istft_ws_mr(const float2* spec, const kpr_stft_geom* s, long long F, const float* synth,
                              float* out, hipStream_t st, bool* launched) {
    *launched = false;
    int r2, r3;
    if (!mixed_radix_plan(s->n_fft, &r2, &r3) || getenv("KPR_NO_MIXED_RADIX") || s->win_length > s->n_fft) return 0;
    const float2* tw = nullptr;
    if (int e = get_twiddles(s->n_fft, &tw)) return e;
    switch (s->n_fft) {
        case 160:  return launch_istft_ws_mr_plan<4, 1>(spec, s, F, synth, tw, out, st, launched);
        case 200:  return launch_istft_ws_mr_plan<5, 1>(spec, s, F, synth, tw, out, st, launched);
        case 320:  return launch_istft_ws_mr_plan<4, 2>(spec, s, F, synth, tw, out, st, launched);
        case 400:  return launch_istft_ws_mr_plan<10, 1>(spec, s, F, synth, tw, out, st, launched);
        case 640:  return launch_istft_ws_mr_plan<4, 4>(spec, s, F, synth, tw, out, st, launched);
        case 800:  return launch_istft_ws_mr_plan<20, 1>(spec, s, F, synth, tw, out, st, launched);
        default:   return launch_istft_ws_mr_plan<5, 5>(spec, s, F, synth, tw, out, st, launched);
    }
}

static int launch_irfft_bs(const float2* spec, const Geom& g, const float* synth, float* frames,
                           hipStream_t st) {
    {
        int r2, r3;
        if (mixed_radix_plan(g.n_fft, &r2, &r3) && !getenv("KPR_NO_MIXED_RADIX"))
            return launch_irfft_mr(spec, g, synth, frames, st);
    }
    const int m = bluestein_m(g.n_fft);
    const float2 *tw = nullptr, *bs = nullptr;
    if (int e = get_twiddles(2 * m, &tw)) return e;
    if (int e = get_bluestein(g.n_fft, &bs)) return e;
    switch (m) {
        case 128:  return launch_irfft_bs_m<128>(spec, g, synth, tw, bs, frames, st);
        case 256:  return launch_irfft_bs_m<256>(spec, g, synth, tw, bs, frames, st);
        case 512:  return launch_irfft_bs_m<512>(spec, g, synth, tw, bs, frames, st);
        default:   return launch_irfft_bs_m<1024>(spec, g, synth, tw, bs, frames, st);
    }
}

template <int NC>
static int launch_irfft_fast(const float2* spec, const Geom& g, const float* synth,
                             const float2* tw, float* frames, hipStream_t st) {
    constexpr int L = NC / kPts, G = 64 / L;
    const long long nblocks = (g.total_frames + 4 * G - 1) / (4 * G);
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const unsigned grid = (unsigned)std::min<long long>(nblocks, 2LL * cus);
    const size_t lds = sizeof(float) * (4 * G * NC + 4 * G * (2 * NC + 8));
    hipLaunchKernelGGL((k_irfft<NC>), dim3(grid), dim3(256), lds, st, spec, g, synth, tw, frames,
                       nblocks);
    return launch_check("k_irfft");
}


// Per 16-filter tile: the k range [lo, hi) the fused kernel walks, padded to whole chunks of
// kChunkRows rows inside [0, mel_row_cap(K)] (rows outside the caller's exact-zero range hold
// zeros for this tile, rows >= K do not exist and are packed as zeros).
static int tile_ranges(int K, int M, const int32_t* kr_host, int* lo_out, int* hi_out) {
    const int ntiles = (M + 15) / 16;
    if (ntiles > kMaxTiles)
        return fail(KPR_E_UNSUPPORTED, "n_filt=%d exceeds the %d-filter limit", M, kMaxTiles * 16);
    const int kp = (K + 3) & ~3;
    const int cap = mel_row_cap(K);
    for (int t = 0; t < ntiles; ++t) {
        int lo = 0, hi = kp;
        if (kr_host) {
            lo = kr_host[2 * t]; hi = kr_host[2 * t + 1];
            if (lo < 0 || hi > kp || lo > hi || (lo & 3) || (hi & 3))
                return fail(KPR_E_BADARG, "bad filterbank k-range for tile %d: [%d,%d)", t, lo, hi);
        }
        int need = std::max(kChunkRows, (hi - lo + kChunkRows - 1) / kChunkRows * kChunkRows);
        hi = std::min(cap, lo + need);
        lo = std::max(0, hi - need);
        if (hi - lo != need)
            return fail(KPR_E_UNSUPPORTED, "n_freq=%d too small for the fused kernel", K);
        lo_out[t] = lo; hi_out[t] = hi;
    }
    return 0;
}

static int build_sched(int K, int M, const int32_t* kr_host, MelSched* sch) {
    int lo[kMaxTiles], hi[kMaxTiles];
    if (int e = tile_ranges(K, M, kr_host, lo, hi)) return e;
    const int ntiles = (M + 15) / 16;
    sch->M = M;
    sch->ntiles = ntiles;
    int total = 0;
    for (int t = 0; t < ntiles; ++t) {
        sch->klo[t] = (short)lo[t]; sch->khi[t] = (short)hi[t];
        sch->chunk0[t] = (unsigned short)total;          // packed layout: tiles in natural order
        total += (hi[t] - lo[t]) / kChunkRows;
    }
    // 4 equal contiguous slices of the chunk stream; tiles straddling a cut are split
    int cut[5];
    for (int w = 0; w <= 4; ++w) cut[w] = (int)(((long long)total * w + 2) / 4);
    cut[0] = 0; cut[4] = total;
    int nseg = 0, w = 0;
    sch->wave_seg0[0] = 0;
    for (int t = 0; t < ntiles; ++t) {
        int c = sch->chunk0[t];
        const int cend = c + (hi[t] - lo[t]) / kChunkRows;
        sch->t_s0[t] = (unsigned char)nseg;
        sch->t_ns[t] = 0;
        while (c < cend) {
            while (w < 3 && c >= cut[w + 1]) { ++w; sch->wave_seg0[w] = nseg; }
            const int e = std::min(cend, cut[w + 1] > c ? cut[w + 1] : cend);
            if (nseg >= kMaxSegs) return fail(KPR_E_UNSUPPORTED, "too many filterbank segments");
            sch->seg_tile[nseg] = (unsigned char)t;
            sch->seg_nch[nseg] = e - c;
            sch->seg_k0[nseg] = lo[t] + (c - sch->chunk0[t]) * kChunkRows;
            ++sch->t_ns[t];
            ++nseg;
            c = e;
        }
    }
    while (w < 4) { ++w; sch->wave_seg0[w] = nseg; }
    sch->nseg = nseg;
    for (int i = 0; i < 4; ++i) {
        sch->wave_chunk0[i] = (unsigned short)cut[i];
        sch->wave_nchunks[i] = (unsigned short)(cut[i + 1] - cut[i]);
    }
    return 0;
}

template <int NC>
static int launch_mel_fast(const float* x, const Geom& g, const float* window, const float2* tw,
                           const float* fbp, const MelSched& sch, const DbDev& db, unsigned* stats,
                           float* out, hipStream_t st) {
    const int S = mel_row_stride(NC + 1);
    size_t lds = sizeof(float) * ((size_t)kFT * S + (size_t)sch.nseg * 256) +   // mag + partial tiles
                 kFT * (sizeof(long long) + sizeof(int));                        // + frame bases
    if (const char* pad = getenv("KPR_DEBUG_LDS_PAD")) lds += (size_t)atoi(pad);   // occupancy experiments
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_mel_fused<NC>))) return e;
    const long long ntiles = (g.total_frames + kFT - 1) / kFT;
    if (ntiles > 0x7fffffffLL) return fail(KPR_E_UNSUPPORTED, "too many frames");
    int dev = 0, cus = 256;
    KPR_HIP(hipGetDevice(&dev));
    static int cached_cus[64] = {0};
    if (dev >= 0 && dev < 64) {
        if (!cached_cus[dev]) {
            int v = 0;
            KPR_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
            cached_cus[dev] = v > 0 ? v : 256;
        }
        cus = cached_cus[dev];
    }
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 2LL * cus);   // 2 workgroups / CU
    hipLaunchKernelGGL((k_mel_fused<NC>), dim3(grid), dim3(256), lds, st, x, g, window, tw, fbp, sch,
                       db, stats, out, (int)ntiles, g_debug_stamps);
    return launch_check("k_mel_fused");
}


template <int NC, bool FROM_MAG = false>
static int launch_mel_ws(const float* x, const Geom& g, const float* window, const float2* tw,
                         const float* fbp, const MelSched& sch, const DbDev& db, unsigned* stats,
                         float* out, hipStream_t st) {
    const size_t lds = mel_ws_lds_bytes(NC, sch.nseg);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_mel_ws<NC, FROM_MAG>))) return e;
    const long long ntiles = (g.total_frames + kFT - 1) / kFT;
    if (ntiles > 0x7fffffffLL) return fail(KPR_E_UNSUPPORTED, "too many frames");
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    constexpr int RF = kWsProd * (64 / (NC / kPts));                           // frames per round
    const long long nrounds = (g.total_frames + RF - 1) / RF;
    const unsigned grid = (unsigned)std::min<long long>(nrounds, cus);         // 1 workgroup / CU
    hipLaunchKernelGGL((k_mel_ws<NC, FROM_MAG>), dim3(grid), dim3(kWsThreads), lds, st, x, g, window, tw, fbp,
                       sch, db, stats, out, (int)ntiles, g_debug_stamps);
    return launch_check("k_mel_ws");
}

static int db_clamp(float* out, long long n_items, long long item_size, float dyn,
                    const unsigned* stats, hipStream_t st) {
    if (n_items <= 0 || item_size <= 0) return 0;
    int chunks = (int)std::min<long long>(64, std::max<long long>(1, item_size / 4096));
    if ((item_size & 3) == 0 && (((uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(k_db_clamp<4>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, out,
                           item_size, chunks, dyn, stats);
    else
        hipLaunchKernelGGL(k_db_clamp<1>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, out,
                           item_size, chunks, dyn, stats);
    return launch_check("k_db_clamp");
}

}  // namespace kpr

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace kpr;

extern "C" {

int kpr_version(void) { return KPR_VERSION; }

/* development aid (not in the public header): device buffer of 12*32 + 1 int64: cycle stamps written by
 * the waves of one workgroup of k_mel_ws (the one whose index is stored in the last element; k_mel_fused
 * and k_stft: workgroup 0, 4*32 entries); NULL disables */
int kpr_debug_stamps(void* dev_buf) { g_debug_stamps = (long long*)dev_buf; return 0; }

/* development aid: known-traffic kernel for calibrating the FETCH_SIZE counter (reads n*8 bytes) */
int kpr_debug_calib_read8(const void* x, int64_t n_float2, float* out, kpr_stream_t stream) {
    hipLaunchKernelGGL(k_calib_read8, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (long long)n_float2, out);
    return launch_check("k_calib_read8");
}

const char* kpr_last_error(void) { return g_err.c_str(); }

int kpr_fft_fast_path(int n_fft) { return fast_nfft(n_fft) ? 1 : 0; }

int64_t kpr_num_frames(const kpr_stft_geom* s) {
    if (check_geom(s)) return -1;
    return frames_of(s);
}

int64_t kpr_stft_workspace_bytes(const kpr_stft_geom* s, int mode) {
    if (check_geom(s)) return -1;
    if (fast_nfft(s->n_fft) || bluestein_ok(s) || mode == KPR_OUT_COMPLEX) return 0;
    // DFT-GEMM path with a real-valued epilogue: complex spectrum staged in the workspace
    return (int64_t)sizeof(float) * 2 * s->batch * s->channels * frames_of(s) * (s->n_fft / 2 + 1);
}

int kpr_stft_f32(const float* x, const kpr_stft_geom* s, const float* window, void* out, int mode,
                 void* workspace, int64_t workspace_bytes, kpr_stream_t stream) {
    if (int e = check_geom(s)) return e;
    if (mode < 0 || mode > 2) return fail(KPR_E_BADARG, "bad output mode %d", mode);
    const long long F = frames_of(s);
    Geom g = make_geom(s, F);
    if (g.total_frames == 0) return 0;
    if (!x || !window || !out) return fail(KPR_E_BADARG, "x / window / out must not be NULL");
    hipStream_t st = (hipStream_t)stream;
    if (fast_nfft(s->n_fft)) {
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        g.cfast = (g.in_cl && g.C > 1) ? 1 : 0;
        switch (s->n_fft) {
            case 256:  return launch_stft_fast<128>(x, g, window, tw, mode, out, st);
            case 512:  return launch_stft_fast<256>(x, g, window, tw, mode, out, st);
            case 1024: return launch_stft_fast<512>(x, g, window, tw, mode, out, st);
            default:   return launch_stft_fast<1024>(x, g, window, tw, mode, out, st);
        }
    }
    if (bluestein_ok(s)) return launch_stft_bs(x, g, window, mode, out, st);
    if (mode == KPR_OUT_COMPLEX) return stft_gemm(x, s, g, window, (float*)out, false, st);
    const int64_t need = kpr_stft_workspace_bytes(s, mode);
    if (!workspace || workspace_bytes < need)
        return fail(KPR_E_WORKSPACE, "stft workspace: need %lld bytes", (long long)need);
    if (int e = stft_gemm(x, s, g, window, (float*)workspace, false, st)) return e;
    const long long n = g.total_frames * g.K;
    hipLaunchKernelGGL(k_cplx_to_real, dim3(grid_1d(n, 256)), dim3(256), 0, st,
                       (const float2*)workspace, n, mode == KPR_OUT_PHASE ? 1 : 0, (float*)out);
    return launch_check("k_cplx_to_real");
}

static bool fused_nfft(int n_fft) { return n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

static int64_t stats_region_bytes(int64_t batch) {
    int64_t b = 256 + (int64_t)sizeof(unsigned) * 2 * std::max<int64_t>(1, batch);
    return (b + 255) & ~(int64_t)255;
}

int64_t kpr_mel_workspace_bytes(const kpr_stft_geom* s, int n_filt, const kpr_db_params* db) {
    if (check_geom(s) || n_filt <= 0) return -1;
    (void)db;
    int64_t bytes = stats_region_bytes(s->batch);
    if (!fused_nfft(s->n_fft))   // two-kernel path stages the complex spectrum
        bytes += (int64_t)sizeof(float) * 2 * s->batch * s->channels * frames_of(s) *
                 (s->n_fft / 2 + 1);
    return bytes;
}

int64_t kpr_filterbank_pack_floats(int n_freq, int n_filt, const int32_t* fb_kranges_host) {
    if (n_freq <= 0 || n_filt <= 0) return -1;
    MelSched sch;
    if (build_sched(n_freq, n_filt, fb_kranges_host, &sch)) return -1;
    int64_t chunks = 0;
    for (int t = 0; t < sch.ntiles; ++t) chunks += (sch.khi[t] - sch.klo[t]) / kChunkRows;
    return chunks * 512;
}

int kpr_filterbank_pack(const float* fb_host, int n_freq, int n_filt, const int32_t* fb_kranges_host,
                        float* out_host) {
    if (!fb_host || !out_host || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad arguments to kpr_filterbank_pack");
    MelSched sch;
    if (int e = build_sched(n_freq, n_filt, fb_kranges_host, &sch)) return e;
    for (int t = 0; t < sch.ntiles; ++t) {
        size_t pos = (size_t)sch.chunk0[t] * 512;
        for (int c = 0; c < (sch.khi[t] - sch.klo[t]) / kChunkRows; ++c)
            for (int g = 0; g < 2; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int k = sch.klo[t] + kChunkRows * c + 16 * g + 4 * s4 + (l >> 4);
                        const int m = 16 * t + (l & 15);
                        out_host[pos++] = (k < n_freq && m < n_filt) ? fb_host[(size_t)k * n_filt + m] : 0.0f;
                    }
    }
    return 0;
}

int64_t kpr_mel_workspace_bytes_unpacked(const kpr_stft_geom* s, int n_filt) {
    if (check_geom(s) || n_filt <= 0) return -1;
    return stats_region_bytes(s->batch) +
           (int64_t)sizeof(float) * 2 * s->batch * s->channels * frames_of(s) * (s->n_fft / 2 + 1);
}

int kpr_mel_f32(const float* x, const kpr_stft_geom* s, const float* window, const float* fb,
                const float* fb_packed, int n_filt, const int32_t* fb_kranges_host,
                const kpr_db_params* db, float* out, void* workspace, int64_t workspace_bytes,
                kpr_stream_t stream) {
    if (int e = check_geom(s)) return e;
    if (int e = check_db(db)) return e;
    if (n_filt <= 0) return fail(KPR_E_BADARG, "n_filt must be positive");
    const long long F = frames_of(s);
    Geom g = make_geom(s, F);
    if (g.total_frames == 0) return 0;
    if (!x || !window || !fb || !out)
        return fail(KPR_E_BADARG, "x / window / fb / out must not be NULL");
    const int64_t need = kpr_mel_workspace_bytes(s, n_filt, db);
    if (!workspace || workspace_bytes < need)
        return fail(KPR_E_WORKSPACE, "mel workspace: need %lld bytes", (long long)need);
    hipStream_t st = (hipStream_t)stream;
    DbDev dbd = make_db(db);
    unsigned* stats = reinterpret_cast<unsigned*>(workspace);
    if (dbd.enabled) {
        hipLaunchKernelGGL(k_stats_init, dim3(grid_1d(s->batch, 256)), dim3(256), 0, st, stats,
                           (long long)s->batch);
        if (int e = launch_check("k_stats_init")) return e;
    }
    MelSched sch;
    if (int e = build_sched(g.K, n_filt, fb_kranges_host, &sch)) return e;
    const long long item_size = (long long)s->channels * F * n_filt;
    if (fused_nfft(s->n_fft) && fb_packed) {
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        int rc;
        g.cfast = (g.in_cl && g.C > 1) ? 1 : 0;
        const char* variant = getenv("KPR_MEL_VARIANT");
        // Default: the wave-specialised kernel (when its two magnitude buffers fit in LDS);
        // KPR_MEL_VARIANT=ring selects the 4-wave ring kernel for A/B runs.
        const bool want_ring = variant && std::strcmp(variant, "ring") == 0;
        int slice_max = 0;      // the consumers keep one lane of schedule per chunk of their slice
        for (int i = 0; i < 4; ++i) slice_max = std::max(slice_max, (int)sch.wave_nchunks[i]);
        // n_fft 2048 only: measured (profiles/) ws wins there by 14-40 %, while at n_fft 1024 (one
        // FFT round per tile, nothing for the consumers to hide behind) the ring kernel was 6 %
        // faster, so that size stays on it.
        if (!want_ring && (s->n_fft == 2048 || s->n_fft == 1024) && slice_max <= 64 &&
            g.total_frames < 0x7fffff00LL && mel_ws_lds_bytes(s->n_fft / 2, sch.nseg) <= 160 * 1024) {
            rc = (s->n_fft == 2048)
                     ? launch_mel_ws<1024>(x, g, window, tw, fb_packed, sch, dbd, stats, out, st)
                     : launch_mel_ws<512>(x, g, window, tw, fb_packed, sch, dbd, stats, out, st);
            if (rc) return rc;
            return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st) : 0;
        }
        switch (s->n_fft) {
            case 512:  rc = launch_mel_fast<256>(x, g, window, tw, fb_packed, sch, dbd, stats, out, st); break;
            case 1024: rc = launch_mel_fast<512>(x, g, window, tw, fb_packed, sch, dbd, stats, out, st); break;
            default:   rc = launch_mel_fast<1024>(x, g, window, tw, fb_packed, sch, dbd, stats, out, st); break;
        }
        if (rc) return rc;
        return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st) : 0;
    }
    // two-kernel path: STFT (complex, frame-contiguous) -> (|.| x filterbank) GEMM [+ dB]
    {
        const int64_t need2 = stats_region_bytes(s->batch) +
                              (int64_t)sizeof(float) * 2 * g.total_frames * g.K;
        if (workspace_bytes < need2)
            return fail(KPR_E_WORKSPACE, "mel workspace (unpacked filterbank path): need %lld bytes",
                        (long long)need2);
    }
    float* spec = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) +
                                           stats_region_bytes(s->batch));
    if (fast_nfft(s->n_fft)) {   // Stockham STFT (n_fft = 256, or no packed filterbank given)
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        Geom gc = g;
        gc.out_cl = 0;
        int rc;
        switch (s->n_fft) {
            case 256:  rc = launch_stft_fast<128>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
            case 512:  rc = launch_stft_fast<256>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
            case 1024: rc = launch_stft_fast<512>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
            default:   rc = launch_stft_fast<1024>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
        }
        if (rc) return rc;
    } else if (bluestein_ok(s)) {   // even non-power-of-two n_fft: chirp-z STFT, frame-contiguous
        Geom gc = g;
        gc.out_cl = 0;
        // wide packed filterbank: |X| rows straight into the fused kernel's MFMA consumers
        // (loader producers, FROM_MAG) instead of the complex spectrum + generic GEMM
        int slice_max = 0;
        for (int i = 0; i < 4; ++i) slice_max = std::max(slice_max, (int)sch.wave_nchunks[i]);
        if (fb_packed && n_filt > 64 && g.K <= 1025 && slice_max <= 64 && !g.out_cl &&
            g.total_frames < 0x7fffff00LL && mel_ws_lds_bytes(1024, sch.nseg) <= 160 * 1024) {
            if (int e = launch_stft_bs(x, gc, window, KPR_OUT_MAGNITUDE, spec, st)) return e;
            if (int e = launch_mel_ws<1024, true>(spec, g, nullptr, nullptr, fb_packed, sch, dbd, stats, out, st))
                return e;
            return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st) : 0;
        }
        if (int e = launch_stft_bs(x, gc, window, KPR_OUT_COMPLEX, spec, st)) return e;
    } else {
        if (int e = stft_gemm(x, s, g, window, spec, true, st)) return e;
    }
    GemmArgs ga{};
    ga.in = frames_contig_map(g, g.K);
    ga.out = frames_out_map(g, n_filt);
    ga.Kdim = g.K; ga.N = n_filt; ga.ldb = n_filt;
    ga.db = dbd; ga.stats = stats;
    if (fb_kranges_host) {
        ga.has_kr = 1;
        for (int t = 0; t < sch.ntiles; ++t) { ga.klo[t] = sch.klo[t]; ga.khi[t] = sch.khi[t]; }
    }
    if (dbd.enabled) {
        if (int e = run_gemm<A_CABS, E_DB>(spec, fb, ga, out, st)) return e;
        return db_clamp(out, s->batch, item_size, dbd.dyn, stats, st);
    }
    return run_gemm<A_CABS, E_PLAIN>(spec, fb, ga, out, st);
}

int kpr_filterbank_kranges(const float* fb_host, int n_freq, int n_filt, int32_t* out_host) {
    if (!fb_host || !out_host || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad arguments to kpr_filterbank_kranges");
    const int ntiles = (n_filt + 15) / 16;
    for (int t = 0; t < ntiles; ++t) {
        int lo = n_freq, hi = 0;
        for (int k = 0; k < n_freq; ++k)
            for (int m = t * 16; m < std::min(n_filt, t * 16 + 16); ++m) {
                float v = fb_host[(size_t)k * n_filt + m];
                if (v != 0.0f || v != v) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
            }
        if (lo >= hi) { lo = 0; hi = 0; }
        out_host[2 * t] = lo & ~3;
        out_host[2 * t + 1] = (hi + 3) & ~3;
    }
    return 0;
}

int kpr_abs_c64(const void* x, int64_t n, float* out, kpr_stream_t stream) {
    if (n < 0) return fail(KPR_E_BADARG, "negative size");
    if (n == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    hipLaunchKernelGGL(k_cplx_to_real, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (long long)n, 0, out);
    return launch_check("k_cplx_to_real");
}

int kpr_angle_c64(const void* x, int64_t n, float* out, kpr_stream_t stream) {
    if (n < 0) return fail(KPR_E_BADARG, "negative size");
    if (n == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    hipLaunchKernelGGL(k_cplx_to_real, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (long long)n, 1, out);
    return launch_check("k_cplx_to_real");
}

int kpr_apply_filterbank_f32(const float* x, int64_t batch, int channels, int64_t frames,
                             int n_freq, int layout, const float* fb, int n_filt,
                             const int32_t* fb_kranges_host, float* out, kpr_stream_t stream) {
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad sizes");
    if ((unsigned)layout > 1u) return fail(KPR_E_BADARG, "bad layout enum");
    const long long rows = batch * channels * frames;
    if (rows == 0) return 0;
    if (!x || !fb || !out) return fail(KPR_E_BADARG, "x / fb / out must not be NULL");
    const int ntiles = (n_filt + 15) / 16;
    if (ntiles > kMaxTiles) return fail(KPR_E_UNSUPPORTED, "n_filt too large");
    // narrow matrices on contiguous rows (LogmelToMFCC's DCT, small filterbanks): the thin GEMM
    const bool contiguous = layout == KPR_CHANNELS_FIRST || channels == 1;
    if (contiguous && ntiles <= 4 && n_freq <= 512 && (n_freq & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        const size_t lds = sizeof(float) * 4 * 64 * (size_t)ntiles * ((n_freq + 15) / 16);
        const unsigned grid = (unsigned)std::min<long long>((rows + 63) / 64, 256 * 8);
        hipStream_t st = (hipStream_t)stream;
        switch (ntiles) {
            case 1: hipLaunchKernelGGL(k_thin_gemm<1>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
            case 2: hipLaunchKernelGGL(k_thin_gemm<2>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
            case 3: hipLaunchKernelGGL(k_thin_gemm<3>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
            default: hipLaunchKernelGGL(k_thin_gemm<4>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
        }
        return launch_check("k_thin_gemm");
    }
    GemmArgs ga{};
    ga.in.rows = rows; ga.out.rows = rows;
    if (layout == KPR_CHANNELS_LAST) {
        // rows r = (b*F + f)*C + c
        ga.in.D0 = channels; ga.in.D1 = 1;
        ga.in.s2 = (long long)n_freq * channels; ga.in.s1 = 0; ga.in.s0 = 1; ga.in.es = channels;
        ga.out.D0 = channels; ga.out.D1 = 1;
        ga.out.s2 = (long long)n_filt * channels; ga.out.s1 = 0; ga.out.s0 = 1; ga.out.es = channels;
    } else {
        ga.in.D0 = 1; ga.in.D1 = 1; ga.in.s2 = n_freq; ga.in.s1 = 0; ga.in.s0 = 0; ga.in.es = 1;
        ga.out.D0 = 1; ga.out.D1 = 1; ga.out.s2 = n_filt; ga.out.s1 = 0; ga.out.s0 = 0; ga.out.es = 1;
    }
    ga.Kdim = n_freq; ga.N = n_filt; ga.ldb = n_filt;
    if (fb_kranges_host) {
        const int kp = (n_freq + 3) & ~3;
        ga.has_kr = 1;
        for (int t = 0; t < ntiles; ++t) {
            int lo = fb_kranges_host[2 * t], hi = fb_kranges_host[2 * t + 1];
            if (lo < 0 || hi > kp || lo > hi) return fail(KPR_E_BADARG, "bad k-range");
            ga.klo[t] = (short)lo; ga.khi[t] = (short)hi;
        }
    }
    return run_gemm<A_PLAIN, E_PLAIN>(x, fb, ga, out, (hipStream_t)stream);
}

int kpr_apply_filterbank_packed_f32(const float* x, int64_t batch, int channels, int64_t frames,
                                    int n_freq, int layout, const float* fb, const float* fb_packed,
                                    int n_filt, const int32_t* fb_kranges_host, float* out,
                                    kpr_stream_t stream) {
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad sizes");
    if ((unsigned)layout > 1u) return fail(KPR_E_BADARG, "bad layout enum");
    const long long rows = batch * channels * frames;
    const bool contiguous = layout == KPR_CHANNELS_FIRST || channels == 1;
    MelSched sch;
    if (fb_packed && x && out && contiguous && rows > 0 && rows < 0x7fffff00LL && n_freq <= 1025 &&
        n_filt > 64 /* narrow matrices: the thin GEMM */ && build_sched(n_freq, n_filt, fb_kranges_host, &sch) == 0) {
        int slice_max = 0;
        for (int i = 0; i < 4; ++i) slice_max = std::max(slice_max, (int)sch.wave_nchunks[i]);
        if (slice_max <= 64 && mel_ws_lds_bytes(1024, sch.nseg) <= 160 * 1024) {
            Geom g{};
            g.total_frames = rows; g.T = 0; g.F = (int)frames; g.C = channels;
            g.n_fft = 2 * (n_freq - 1); g.win = 0; g.hop = 0; g.pad_left = 0; g.K = n_freq;
            g.in_cl = 0; g.out_cl = 0; g.cfast = 0;
            DbDev dbd = make_db(nullptr);
            return launch_mel_ws<1024, true>(x, g, nullptr, nullptr, fb_packed, sch, dbd, nullptr, out,
                                             (hipStream_t)stream);
        }
    }
    return kpr_apply_filterbank_f32(x, batch, channels, frames, n_freq, layout, fb, n_filt, fb_kranges_host,
                                    out, stream);
}

int64_t kpr_db_workspace_bytes(int64_t n_items) {
    if (n_items < 0) return -1;
    return 256 + (int64_t)sizeof(unsigned) * 2 * std::max<int64_t>(1, n_items);
}

int kpr_mag_to_db_f32(const float* x, int64_t n_items, int64_t item_size, const kpr_db_params* db,
                      float* out, void* workspace, int64_t workspace_bytes, kpr_stream_t stream) {
    if (!db) return fail(KPR_E_BADARG, "db params are NULL");
    kpr_db_params p = *db;
    p.enabled = 1;
    if (int e = check_db(&p)) return e;
    if (n_items < 0 || item_size < 0) return fail(KPR_E_BADARG, "negative size");
    if (n_items == 0 || item_size == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    if (!workspace || workspace_bytes < kpr_db_workspace_bytes(n_items))
        return fail(KPR_E_WORKSPACE, "db workspace: need %lld bytes",
                    (long long)kpr_db_workspace_bytes(n_items));
    hipStream_t st = (hipStream_t)stream;
    DbDev dbd = make_db(&p);
    unsigned* stats = reinterpret_cast<unsigned*>(workspace);
    hipLaunchKernelGGL(k_stats_init, dim3(grid_1d(n_items, 256)), dim3(256), 0, st, stats,
                       (long long)n_items);
    if (int e = launch_check("k_stats_init")) return e;
    int chunks = (int)std::min<long long>(64, std::max<long long>(1, item_size / 4096));
    if ((item_size & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0)
        hipLaunchKernelGGL(k_db_log<4>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, x,
                           (long long)item_size, chunks, dbd, stats, out);
    else
        hipLaunchKernelGGL(k_db_log<1>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, x,
                           (long long)item_size, chunks, dbd, stats, out);
    if (int e = launch_check("k_db_log")) return e;
    return db_clamp(out, n_items, item_size, dbd.dyn, stats, st);
}

int64_t kpr_istft_workspace_bytes(const kpr_stft_geom* s, int64_t n_frames) {
    if (check_geom(s) || n_frames < 0) return -1;
    return 256 + (int64_t)sizeof(float) * s->batch * s->channels * n_frames * s->win_length;
}

int kpr_istft_f32(const void* spec, const kpr_stft_geom* s, int64_t n_frames,
                  const float* synth_window, float* out, void* workspace, int64_t workspace_bytes,
                  kpr_stream_t stream) {
    if (int e = check_geom(s)) return e;
    if (n_frames < 0) return fail(KPR_E_BADARG, "negative frame count");
    Geom g = make_geom(s, n_frames);
    g.pad_left = 0;
    if (g.total_frames == 0) return 0;
    if (!spec || !synth_window || !out) return fail(KPR_E_BADARG, "spec / window / out must not be NULL");
    const int64_t need = kpr_istft_workspace_bytes(s, n_frames);
    if (!workspace || workspace_bytes < need)
        return fail(KPR_E_WORKSPACE, "istft workspace: need %lld bytes", (long long)need);
    hipStream_t st = (hipStream_t)stream;
    float* frames = reinterpret_cast<float*>(workspace);
    if (fast_nfft(s->n_fft) && !getenv("KPR_ISTFT_TWO_KERNEL")) {
        // fused irFFT + window + overlap-add (no workspace traffic) whenever the frames overlap
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        bool launched = false;
        int rc;
        switch (s->n_fft) {      // wave-specialised ring kernel when its preconditions hold
            case 256:  rc = launch_istft_ws<128>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            case 512:  rc = launch_istft_ws<256>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            case 1024: rc = launch_istft_ws<512>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            default:   rc = launch_istft_ws<1024>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
        }
        if (rc) return rc;
        if (launched) return 0;
        switch (s->n_fft) {
            case 256:  rc = launch_istft_fused<128, 4>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            case 512:  rc = launch_istft_fused<256, 4>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            case 1024: rc = launch_istft_fused<512, 4>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            default:   rc = launch_istft_fused<1024, 8>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
        }
        if (rc) return rc;
        if (launched) return 0;
    }
    if (!fast_nfft(s->n_fft) && !getenv("KPR_ISTFT_TWO_KERNEL")) {
        // n_fft = 2^a 5^b: the ring kernel with mixed-radix producers
        bool launched = false;
        if (int e = launch_istft_ws_mr((const float2*)spec, s, n_frames, synth_window, out, st, &launched)) return e;
        if (launched) return 0;
    }
    if (fast_nfft(s->n_fft)) {
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        int rc;
        switch (s->n_fft) {
            case 256:  rc = launch_irfft_fast<128>((const float2*)spec, g, synth_window, tw, frames, st); break;
            case 512:  rc = launch_irfft_fast<256>((const float2*)spec, g, synth_window, tw, frames, st); break;
            case 1024: rc = launch_irfft_fast<512>((const float2*)spec, g, synth_window, tw, frames, st); break;
            default:   rc = launch_irfft_fast<1024>((const float2*)spec, g, synth_window, tw, frames, st); break;
        }
        if (rc) return rc;
    } else if (bluestein_ok(s)) {     // even non-power-of-two n_fft: inverse chirp-z, then the gather
        if (int e = launch_irfft_bs((const float2*)spec, g, synth_window, frames, st)) return e;
    } else {
        const float* idft = nullptr;
        if (int e = get_dft_inv(s->n_fft, &idft)) return e;
        GemmArgs ga{};
        ga.in = frames_out_map(g, g.K);          // spectrum in the caller's layout (complex units)
        ga.out = frames_contig_map(g, g.win);
        ga.Kdim = 2 * g.K; ga.N = std::min(g.n_fft, g.win); ga.ldb = g.n_fft;
        ga.window = synth_window; ga.win = g.win;
        if (int e = run_gemm<A_CPLX, E_WINDOW>((const float*)spec, idft, ga, frames, st)) return e;
        if (g.win > g.n_fft) {
            hipLaunchKernelGGL(k_fill_cols, dim3(grid_1d(g.total_frames * (g.win - g.n_fft), 256)),
                               dim3(256), 0, st, frames, g.total_frames, (long long)g.win, g.n_fft,
                               g.win);
            if (int e = launch_check("k_fill_cols")) return e;
        }
    }
    const long long t_out = (n_frames - 1) * (long long)s->hop_length + s->win_length;
    const long long n_sig = (long long)s->batch * s->channels;
    hipLaunchKernelGGL(k_ola, dim3(grid_1d(n_sig * t_out, 256)), dim3(256), 0, st, frames, n_sig,
                       (int)n_frames, s->channels, s->win_length, s->hop_length, t_out,
                       s->in_layout == KPR_CHANNELS_LAST ? 1 : 0, out);
    return launch_check("k_ola");
}

/* ---- Frame / Energy / Delta ------------------------------------------------------------------ */
int64_t kpr_frame_count(int64_t time, int frame_length, int hop_length, int pad_end) {
    if (time < 0 || frame_length <= 0 || hop_length <= 0) {
        fail(KPR_E_BADARG, "bad time/frame_length/hop_length (%lld, %d, %d)", (long long)time, frame_length,
             hop_length);
        return -1;
    }
    if (pad_end) return (time + hop_length - 1) / hop_length;
    return time < frame_length ? 0 : 1 + (time - frame_length) / hop_length;
}

static int frame_args(int64_t batch, int channels, int64_t time, int layout, int frame_length,
                      int hop_length, int pad_end, float pad_value, FrameArgs* a) {
    if (batch < 0 || channels <= 0 || (unsigned)layout > 1u)
        return fail(KPR_E_BADARG, "bad batch/channels/layout (%lld, %d, %d)", (long long)batch, channels, layout);
    const int64_t f = kpr_frame_count(time, frame_length, hop_length, pad_end);
    if (f < 0) return KPR_E_BADARG;
    if (f > 0x7fffffffLL) return fail(KPR_E_UNSUPPORTED, "too many frames per signal");
    a->n_sig = batch * channels; a->T = time; a->C = channels; a->F = (int)f; a->L = frame_length;
    a->hop = hop_length; a->cl = layout == KPR_CHANNELS_LAST && channels > 1; a->pad_value = pad_value;
    return 0;
}

int kpr_frame_f32(const float* x, int64_t batch, int channels, int64_t time, int layout,
                  int frame_length, int hop_length, int pad_end, float pad_value, float* out,
                  kpr_stream_t stream) {
    FrameArgs a;
    if (int e = frame_args(batch, channels, time, layout, frame_length, hop_length, pad_end, pad_value, &a))
        return e;
    const long long nrows = (a.cl ? (long long)batch : a.n_sig) * a.F;
    if (nrows == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    const int rowlen = a.cl ? a.L * a.C : a.L;
    const bool vec = (rowlen & 3) == 0 && (((uintptr_t)out) & 15) == 0;
    const long long work = nrows * (vec ? rowlen / 4 : rowlen);
    if (vec)
        hipLaunchKernelGGL(k_frame<4>, dim3(grid_1d(work, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x,
                           a, out, nrows);
    else
        hipLaunchKernelGGL(k_frame<1>, dim3(grid_1d(work, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x,
                           a, out, nrows);
    return launch_check("k_frame");
}

int kpr_energy_f32(const float* x, int64_t batch, int channels, int64_t time, int layout,
                   int frame_length, int hop_length, int pad_end, float pad_value, float scale,
                   float* out, kpr_stream_t stream) {
    FrameArgs a;
    if (int e = frame_args(batch, channels, time, layout, frame_length, hop_length, pad_end, pad_value, &a))
        return e;
    a.cl = layout == KPR_CHANNELS_LAST;            // output order depends on it even for one channel
    const long long nout = a.n_sig * a.F;
    if (nout == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    if (time == 0) return fail(KPR_E_UNSUPPORTED, "energy of an empty signal with pad_end");
    const int chunks = (a.F + kEnFrames - 1) / kEnFrames;
    const int q = frame_length / hop_length;
    const size_t lds = sizeof(float) * 2 * (size_t)(kEnFrames + q + 1);
    if (lds > 64 * 1024) return fail(KPR_E_UNSUPPORTED, "frame_length / hop_length = %d is too large", q);
    hipLaunchKernelGGL(k_energy, dim3((unsigned)std::min<long long>(a.n_sig * chunks, 1 << 16)), dim3(256), lds,
                       (hipStream_t)stream, x, a, scale, out, chunks);
    return launch_check("k_energy");
}

int kpr_delta_f32(const float* x, int64_t batch, int channels, int64_t frames, int n_freq, int layout,
                  int win_length, int pad_mode, float* out, kpr_stream_t stream) {
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || (unsigned)layout > 1u)
        return fail(KPR_E_BADARG, "bad batch/channels/frames/n_freq/layout");
    if (win_length < 3 || (win_length & 1) == 0)
        return fail(KPR_E_BADARG, "win_length must be odd and >= 3, got %d", win_length);
    if (pad_mode < 0 || pad_mode > 2) return fail(KPR_E_BADARG, "bad pad mode %d", pad_mode);
    const long long total = (long long)batch * channels * frames * n_freq;
    if (total == 0) return 0;
    if (!x || !out || x == out) return fail(KPR_E_BADARG, "x / out must not be NULL or aliased");
    const int n = (win_length - 1) / 2;
    double denom = 0;
    for (int i = 1; i <= n; ++i) denom += 2.0 * i * i;
    const long long outer = layout == KPR_CHANNELS_LAST ? batch : batch * channels;
    const long long inner = layout == KPR_CHANNELS_LAST ? (long long)n_freq * channels : n_freq;
    const bool vec = (inner & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(k_delta<4>, dim3(grid_1d(total / 4, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream,
                           x, outer, (long long)frames, inner, n, pad_mode, (float)(1.0 / denom), out);
    else
        hipLaunchKernelGGL(k_delta<1>, dim3(grid_1d(total, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x,
                           outer, (long long)frames, inner, n, pad_mode, (float)(1.0 / denom), out);
    return launch_check("k_delta");
}

}  // extern "C"
