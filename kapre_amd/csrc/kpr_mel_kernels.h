// kpr_mel_kernels.h -- the wave-specialised fused mel-spectrogram kernel k_mel_ws (FFT producer waves + MFMA consumer waves;
// FROM_MAG = stand-alone ApplyFilterbank) and the filterbank schedule it shares with k_mel_ts / k_mel_mr.  (The round-1 kernel
// k_mel_fused -- one 4-wave workgroup = FFT then GEMM -- was removed in round 5: dominated on every shape.)
// Part of the single translation unit kapre_hip.hip (included there, in this order; not stand-alone).
#pragma once

namespace kpr {

// ------------------------------------------------------------------------------------------
// fused mel kernel
// ------------------------------------------------------------------------------------------
#ifndef KPR_RING_DEPTH
#define KPR_RING_DEPTH 3
#endif
constexpr int kMaxTiles = 64;   // up to 1024 filters
constexpr int kWsSpinLimit = 1 << 22;   // polls of an LDS counter before a wave gives up waiting (>= 0.2 s)
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

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (a loop index usable in `if constexpr` and as
// an asm immediate)
template <int I, int N, class F>
KPR_DEV void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int kChunkRows = 32;   // MFMA loop granularity: 8 k-steps of 4 rows

__host__ __device__ inline int mel_row_cap(int K) { return (K + kChunkRows - 1) / kChunkRows * kChunkRows; }
__host__ __device__ inline int mel_row_stride(int K) {
    // S >= roundup(K,32) (tile k-ranges are padded to whole chunks and must stay inside the
    // zero-padded row), S % 16 == 2 -> conflict-free MFMA operand reads (banks 2j+h / 18j+h)
    return mel_row_cap(K) + 2;
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
template <int NC> struct WsSwzFor { typedef typename SwzFor<NC>::type type; };
template <> struct WsSwzFor<1024> { typedef SwzWide type; };      // 128-bit exchanges (kpr_fft.h)
// one ticket of k_mel_ws = G frames (one per lane group): gf_next is the first frame of the wave's
// next ticket (wave-uniform), lane group grp takes frame gf_next + grp
template <int NC>
KPR_DEV void ws_frame(const float* __restrict__ x, const Geom& g, FftTw<NC, typename WsSwzFor<NC>::type>& tw,
                      const f2* winl, float* row, float* xrow, int gf_next, int f_end, int fl, int grp, int lane, int K, int S,
                      f2 (&nz)[kPts], unsigned& nvm, f2 (&wv)[kPts], bool more, long long* dbgw, int& dbi) {
    constexpr int L = NC / kPts;
    typedef typename WsSwzFor<NC>::type WsSwz;
#ifdef KPR_FINE_STAMPS
#define KPR_FS() do { if (dbgw && lane == 0 && dbi < 32) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dbgw[dbi++] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define KPR_FS() do { (void)dbgw; (void)dbi; } while (0)
#endif
    KPR_FS();
    f2 z[kPts];
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = nz[m];
    mask_frame(z, nvm);
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = pmul(z[m], wv[m]);     // window: read from LDS at the end of the previous frame
    KPR_FS();
    if (gf_next < f_end) {                                  // wave-uniform
        const bool validn = gf_next + grp < f_end;
        FramePos pn = frame_pos(g, validn ? gf_next + grp : gf_next);
        nvm = fetch_frame<NC, true>(x, g, pn, validn, fl, nz);
    }
    {
        using Rx = Radix<NC>;
        tw.refresh();
        KPR_FS();
#ifndef KPR_FINE_STAMPS
        if constexpr (IsWide<WsSwz>::value) {
            cfft_forward_wide_planar(z, tw, xrow);
        } else
#endif
        {
#ifdef KPR_FINE_STAMPS
        {
            f2 o_[kPts];
            pass_compute<NC, 1, Rx::r1, 1, WsSwz>(z, tw, o_);
            KPR_FS();
            exchange_issue<NC, 1, Rx::r1, 1, WsSwz>(o_, z, tw, xrow);
            KPR_FS();
            pass_compute<NC, 2, Rx::r2, Rx::r1, WsSwz>(z, tw, o_);
            KPR_FS();
            exchange_issue<NC, 2, Rx::r2, Rx::r1, WsSwz>(o_, z, tw, xrow);
        }
#else
        fft_pass<NC, 1, Rx::r1, 1, WsSwz>(z, tw, xrow);
        KPR_FS();
        fft_pass<NC, 2, Rx::r2, Rx::r1, WsSwz>(z, tw, xrow);
#endif
        KPR_FS();
        if constexpr (Rx::r3 > 1) fft_pass<NC, 3, Rx::r3, Rx::r1 * Rx::r2, WsSwz>(z, tw, xrow);
        KPR_FS();
        }
    }
#define KPR_XSQRT(v_) __builtin_amdgcn_sqrtf(v_)
    if constexpr (L == 64 || L == 32) {
        // collect the magnitudes, then store bins fl + L m and NC - fl - L m as two runs with an L-word stride each --
        // hipcc merges them into ds_write2st64_b32 / ds_write2_b32 (8 LDS instructions instead of 16)
        float mk[kPts / 2], mp[kPts / 2];
        float mid = 0.0f;
        rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
            const float a = KPR_XSQRT(xk.x * xk.x + xk.y * xk.y);
            if (kp >= 0) {
                const int m = (k - fl) / L;                     // compile-time after unrolling
                mk[m] = a;
                mp[m] = KPR_XSQRT(xp.x * xp.x + xp.y * xp.y);
            } else mid = a;                                     // k = NC / 2 (lane 0 only)
        });
        float* lo = row + fl;
        float* hi = row + (NC - fl) - L * (kPts / 2 - 1);
#pragma unroll
        for (int m = 0; m < kPts / 2; ++m) lo[L * m] = mk[m];
#pragma unroll
        for (int m = 0; m < kPts / 2; ++m) hi[L * (kPts / 2 - 1 - m)] = mp[m];
        if (fl == 0) row[NC / 2] = mid;
    } else {
    rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
        row[k] = KPR_XSQRT(xk.x * xk.x + xk.y * xk.y);
        if (kp >= 0) row[kp] = KPR_XSQRT(xp.x * xp.x + xp.y * xp.y);
    });
    }
    // zero pad columns K .. S-1 (read by the last k-step; must be finite)
    for (int k = K + fl; k < S; k += L) row[k] = 0.0f;
#undef KPR_XSQRT
    // the NEXT frame's window values: z is dead here, and the LDS round trip then runs under the ticket /
    // publish code instead of at the head of the next frame (one exposed LDS latency less per frame)
    (void)more;
#pragma unroll
    for (int m = 0; m < kPts; ++m) wv[m] = winl[fl + L * m];
    KPR_FS();
#undef KPR_FS
}

// loader producers of k_mel_ws<NC, true> (see there): tickets of RPT rows, PER loads of 64 floats per
// row, NSET register sets (later tickets' rows are in flight while the current ones are written)
// Rows are contiguous (channels_first, or one channel) or, for channels_last spectrograms with C > 1, strided by C
// with channel-fastest row numbering (g.out_cl / g.cfast: the C rows that share the same cache lines sit in the same
// tile and are loaded at about the same time).
template <int RPT, int PER, int NSET>
KPR_DEV void ws_loader(const float* __restrict__ x, const Geom& g, int K, int S, int f_begin, int n_total, float* smem,
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
            const long long gf_ = f_begin + min(RPT * (n_) + r, n_total - 1);                    \
            const float* src_ = x + (g.out_cl ? spec_base(g, frame_pos(g, gf_), gf_, K) : gf_ * K); \
            const int es_ = spec_stride(g);                                                      \
            _Pragma("unroll") for (int u = 0; u < PER; ++u)                                      \
                if (64 * u < K) set_[r][u] = src_[min(lane + 64 * u, K - 1) * es_];  /* wave-uniform guard */ \
        }                                                                                        \
    } while (0)
#define WL_STORE(set_, n_)                                                                       \
    do {                                                                                         \
        const int q0_ = RPT * (n_), t_ = q0_ >> 4;                                               \
        if (t_ >= 2)  /* the group of buffer t & 1 has consumed t >> 1 tiles of it */            \
            while (__hip_atomic_load(&sync[5 + (t_ & 1)], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (t_ >> 1)) \
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
    // NSET tickets in flight per wave (static register sets, drained in ticket order): a dependent
    // round trip to HBM costs 3-4 us under load, and with four loader waves per CU two tickets each
    // were not enough rows in flight to feed two consumer groups
    float v[NSET][RPT][PER];
    int n[NSET];
#pragma unroll
    for (int q = 0; q < NSET; ++q) {
        WL_TICKET(n[q]);
        if (n[q] < n_tickets) WL_LOAD(v[q], n[q]);
    }
#pragma unroll 1
    for (;;) {
        bool done = false;
#pragma unroll
        for (int q = 0; q < NSET; ++q) {
            if (done || n[q] >= n_tickets) { done = true; continue; }      // tickets only grow: the later sets are done too
            WL_STORE(v[q], n[q]);
            WL_TICKET(n[q]);
            if (n[q] < n_tickets) WL_LOAD(v[q], n[q]);
        }
        if (done) break;
    }
#undef WL_TICKET
#undef WL_LOAD
#undef WL_STORE
}

constexpr int kWsConsPrio = 3;   // s_setprio of the consumer waves while they hold a tile
constexpr int kWsProd = 8;
constexpr int kWsThreads = 768;

// magnitude row stride of k_mel_ws: the row doubles as the skewed FFT exchange row (WsSwz needs
// NC + NC/32 + 24 words) and must keep S % 16 == 2 for the MFMA operand reads
__host__ __device__ inline int mel_ws_row_stride(int K) {
    const int NC = K - 1;
    bool skew = NC == 1024 || NC == 512;
    if (!skew) return mel_row_stride(K);
    if (NC == 1024) return (std::max(mel_row_cap(K), SwzWide::row_words(NC)) + 13) / 16 * 16 + 2;
    const int need = std::max(mel_row_cap(K), SwzSkew::row_words(NC));
    return (need + 13) / 16 * 16 + 2;
}

// ngrp = consumer groups (1: the fused kernel; 2: the FROM_MAG instance, see k_mel_ws)
__host__ __device__ inline size_t mel_ws_lds_bytes(int NC, int nseg, int ngrp = 1) {
    const int S = mel_ws_row_stride(NC + 1);
    return sizeof(float) * ((size_t)2 * kFT * S + (size_t)ngrp * nseg * 256) +
           (size_t)ngrp * kFT * (sizeof(long long) + sizeof(int)) + 8 * sizeof(int) +
           (ngrp > 1 ? 0 : (size_t)NC * 2 * sizeof(float))       // window pairs: FFT producers only
        ;
}

// FROM_MAG = true: the same kernel as a stand-alone ApplyFilterbank -- `x` holds magnitude rows
// (g.K floats per frame, contiguous) and the producers merely copy them into the tile; consumers,
// counters, tickets and the epilogue are shared.
// RES = true: every consumer wave's slice of the packed filterbank (<= kWsResident chunks) stays in registers for
// the whole kernel (the launcher checks the slice sizes); RES = false streams it from L2 per tile.
// (A split-bf16 form of the product -- hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, elementwise error <= 1.7e-5 --
//  was built, measured and removed in round 2: with the split done by the consumers it was not faster, and it had a
//  correctness problem that was never explained)
constexpr int kWsResident = 10;
// (A 32-points-per-lane producer -- one LDS exchange instead of two, four producer waves with 256 VGPRs -- was built
//  and measured in round 2, 62 us against 54: tools/probes/experiments/kpr_fft32.h.txt)
// LD8 (FROM_MAG only): eight loader waves and ONE consumer group instead of four + two.  Rows of more than 512 floats
// (n_fft 2048 spectrograms: 4.1 KB each) are loader-bound with four loaders -- in-kernel stamps (tools/stamps_fb.py): the
// consumers waited for rows 65 % of the time, a tile every 14 k cycles -- while short rows (speech: 201 floats) are bound by
// the consumers' fixed cost per tile, which is what the two groups are for.
template <int NC, bool FROM_MAG, bool RES = false, bool LD8 = false>
__global__ __launch_bounds__(kWsThreads) void k_mel_ws(const float* __restrict__ x, Geom g,
                                                       const float* __restrict__ window,
                                                       const float2* __restrict__ twtab,
                                                       const float* __restrict__ fbp, MelSched sch,
                                                       DbDev db, unsigned* __restrict__ item_stats,
                                                       float* __restrict__ out, int run_q, int run_r,
                                                       long long* __restrict__ dbg) {
    constexpr int PTS = kPts;
    constexpr int L = NC / PTS;        // lanes per frame
    constexpr int G = 64 / L;          // frames per wave per round
    constexpr int THREADS = kWsThreads;
    typedef typename WsSwzFor<NC>::type WsSwz;
    static_assert(!FROM_MAG || G == 1, "loader producers copy one row per wave");
    // FROM_MAG: copying rows is cheap and the consumers' fixed cost per tile (tile wait, ring refill
    // from L2, partial-sum exchange, stores: latency, the matrix pipe is ~15 % busy) bounds the
    // kernel, so the twelve waves are split 4 loaders + TWO consumer groups of four: group 0 takes the
    // even tiles (buffer 0), group 1 the odd ones (buffer 1), each with its own partial-sum area,
    // frame table, group barrier and per-buffer "tile consumed" counter.
    static_assert(!LD8 || FROM_MAG, "LD8 is a variant of the loader kernel");
    constexpr int NPROD = FROM_MAG ? (LD8 ? 8 : 4) : kWsProd;
    constexpr int NGRP = (FROM_MAG && !LD8) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = FROM_MAG ? g.K : NC + 1;
    const int S = mel_ws_row_stride(NC + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cgrp = (NGRP > 1 && wave >= NPROD) ? (wave - NPROD) >> 2 : 0;    // consumer group of this wave

    float* dpart = smem + 2 * kFT * S + cgrp * (sch.nseg * 256);         // [nseg][frame 16][filter 16] per group
    long long* fbase = reinterpret_cast<long long*>(smem + 2 * kFT * S + NGRP * sch.nseg * 256) + cgrp * kFT;
    int* fitem = reinterpret_cast<int*>(reinterpret_cast<long long*>(smem + 2 * kFT * S + NGRP * sch.nseg * 256) + NGRP * kFT) + cgrp * kFT;
    // monotonic LDS counters: sync[0], sync[1] rows written into mag buffer 0 / 1 (producers),
    // sync[2] consumer waves done reading a tile, sync[3] consumer-group barrier, sync[4] frame tickets;
    // FROM_MAG: sync[5], sync[6] consumer waves done with a tile of buffer 0 / 1, sync[7] barrier of group 1
    int* sync = reinterpret_cast<int*>(reinterpret_cast<long long*>(smem + 2 * kFT * S + NGRP * sch.nseg * 256) + NGRP * kFT) + NGRP * kFT;
    f2* winl = reinterpret_cast<f2*>(sync + 8);                          // (0.5 w[2n], 0.5 w[2n+1])
#define WS_SIGNAL_N(p_, n_) do { if (lane == 0) __hip_atomic_fetch_add((p_), (n_), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define WS_SIGNAL(p_) WS_SIGNAL_N(p_, 1)
// bounded (like kIwSpinLimit of the ISTFT ring): a protocol bug becomes a wrong result AND a bit in the device status word
// (kpr_common.h: the next API call fails with KPR_E_DEVICE), not a hung GPU
#define WS_SPIN_UNTIL(p_, n_, nap_) do { int spin_ = 0; for (; spin_ < kWsSpinLimit && __hip_atomic_load((p_), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (n_); ++spin_) __builtin_amdgcn_s_sleep(nap_); if (__builtin_expect(spin_ >= kWsSpinLimit, 0)) status_raise(kStMelWs); } while (0)

    int dbi = 0;
    // development aid: dbg[12*32] selects the workgroup whose waves record cycle stamps
    const bool stamp_me = dbg && (long long)blockIdx.x == dbg[12 * 32];
#ifdef KPR_DEV_STAMPS    /* tools/stamps.py needs a library built with -DKPR_DEV_STAMPS (tools/build_variant.py) */
#define KPR_STAMP() do { if (stamp_me && lane == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define KPR_STAMP() do { (void)stamp_me; (void)dbi; } while (0)
#endif
    KPR_STAMP();
    // A workgroup owns a CONTIGUOUS run of frames [f_begin, f_end), cut at ticket granularity (G
    // frames), so the runs differ by at most one ticket; it walks the run in tiles of 16 frames, the
    // last one possibly short.  Contiguous, not grid-strided: the next tile's samples overlap the
    // current one's and sit in the same pages.
    // (frame numbers fit in 32 bits here: the launcher falls back to k_mel_fused otherwise)
    // run_q, run_r = (tickets / workgroups, tickets % workgroups) from the host: the first run_r workgroups take
    // run_q + 1 tickets (two 64-bit divisions per wave used to sit on the prologue's critical path)
    const int bx = (int)blockIdx.x;
    const int f_begin = (run_q * bx + min(bx, run_r)) * G;
    const int f_end = (int)min(g.total_frames, (long long)(run_q * (bx + 1) + min(bx + 1, run_r)) * G);
    const int my = (f_end - f_begin + kFT - 1) / kFT;             // my tiles
    // Prologue: three independent groups of global loads -- the producers' first frame of samples and their
    // twiddles, and the window that all threads copy into LDS -- are ISSUED before anything waits, so the
    // workgroup pays one memory latency, not three in a row.  The first two tickets of every producer
    // wave are static (wave, wave + NPROD; the ticket counter starts at 2 NPROD), which is what lets the
    // fetch start before the LDS counters exist.
    [[maybe_unused]] f2 nz0[PTS];
    [[maybe_unused]] unsigned nvm0 = 0xffffffffu;
    [[maybe_unused]] FftTw<NC, WsSwz> tw0;
    [[maybe_unused]] float warm = 0.0f;
    if constexpr (!FROM_MAG) {
        if (wave < NPROD) {
            const int fl = lane & (L - 1), grp = lane / L;
            const int gf0 = f_begin + G * wave;
            if (gf0 < f_end) {
                const bool v0 = gf0 + grp < f_end;
                FramePos p0 = frame_pos(g, v0 ? gf0 + grp : gf0);
                nvm0 = fetch_frame<NC, true>(x, g, p0, v0, fl, nz0);
            }
            // (the twiddles are loaded after the barrier: any use of a loaded value before it -- even a register
            // copy hipcc makes of one -- would wait for the older sample loads as well)
        } else {
            // the CONSUMER waves copy the window into LDS: loads return in order, so a producer that waited for
            // window values queued behind its samples would hold the whole workgroup at the barrier below for
            // the HBM cold-start burst of the first frames (measured: prologue 6k -> 12k cycles)
            // (all loads first, clamped indices: one memory round trip, not one per loop iteration)
            constexpr int NCT = THREADS - NPROD * 64, WPT = (NC + NCT - 1) / NCT;
            const int c0 = tid - NPROD * 64;
            // ... and touch the twiddle table (2 NC float2, one 64-byte line per lane): the producers read it
            // right after the barrier, and a first touch after a kernel boundary costs a translation miss and an
            // HBM round trip (~3 us) that would otherwise sit on the first frame's critical path
            warm = reinterpret_cast<const float*>(twtab)[min(c0 * 16, 4 * NC - 1)];
            float wa[WPT], wb[WPT];
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const int n = 2 * min(c0 + u * NCT, NC - 1);
                wa[u] = window[min(n, g.win - 1)];
                wb[u] = window[min(n + 1, g.win - 1)];
            }
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const int i = c0 + u * NCT, n = 2 * i;
                if (i < NC) winl[i] = f2{(n < g.win) ? 0.5f * wa[u] : 0.0f, (n + 1 < g.win) ? 0.5f * wb[u] : 0.0f};
            }
        }
    }
    // ticket counter: the static tickets are taken (see the producers); SKEW: tickets 12 .. 15 of the first tile are
    // drawn dynamically and 16 .. 19 are static (drawn values >= 16 are shifted by 4)
    // SKEW needs every static ticket (up to 19) to exist: with a run of 13 .. 19 tickets a young wave would leave the loop
    // on its out-of-range static ticket while holding a valid drawn one (12 .. 15) -- that frame would never be produced
    // (ADVICE r02).  Short runs gain nothing from the skew anyway: they take the plain numbering.
    constexpr bool SKEW_OK = !FROM_MAG && G == 1 && NPROD == 8;
    const bool SKEW = SKEW_OK && (f_end - f_begin + G - 1) / G >= 20;
    if (tid < 8) sync[tid] = (tid == 4 && !FROM_MAG) ? (SKEW ? 12 : 2 * NPROD) : 0;
    __syncthreads();
    if (warm == 1.2345678e-30f) sync[7] = 1;      // keeps the warm-up load alive (a twiddle is never this value)

#ifdef KPR_FINE_STAMPS   /* stamps of workgroup 0 in tile 2 only (fits the 32-slot row) */
#define KPR_DO_FRAME(row_, gf_next_, more_) ws_frame<NC>(x, g, tw, winl, smem + (row_), smem + (((row_) + 3) & ~3), (gf_next_), f_end, fl, grp, lane, K, S, nz, nvm, wv, (more_), (stamp_me && t == 2) ? dbg + wave * 32 : nullptr, dbi)
#else
#define KPR_DO_FRAME(row_, gf_next_, more_) ws_frame<NC>(x, g, tw, winl, smem + (row_), smem + (((row_) + 3) & ~3), (gf_next_), f_end, fl, grp, lane, K, S, nz, nvm, wv, (more_), nullptr, dbi)
#endif

    if (wave < NPROD) {
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
            if (K <= 256) ws_loader<4, 4, 4>(x, g, K, S, f_begin, n_total, smem, sync, lane);
            else if (K <= 512) ws_loader<2, 8, 4>(x, g, K, S, f_begin, n_total, smem, sync, lane);
            else ws_loader<1, (NC + 1 + 63) / 64, 4>(x, g, K, S, f_begin, n_total, smem, sync, lane);
        } else {
        const int fl = lane & (L - 1), grp = lane / L;     // lane group grp owns frame G*ticket + grp
        FftTw<NC, WsSwz>& tw = tw0;
        tw.load(twtab, fl);
        f2 (&nz)[kPts] = nz0;
        unsigned nvm = nvm0;
        // Frames are handed out DYNAMICALLY (an LDS ticket counter): ticket n = the G frames
        // G*n .. G*n + G-1 of the run, frame q going to row q & 15 of tile q >> 4.  With a static
        // assignment the four older producer waves, which win the SIMD's issue arbitration, finish
        // early and idle a quarter of every tile; now they simply take more tickets.  A wave holds
        // its next ticket while it works on the current one, so the sample prefetch still runs one
        // ticket ahead.  (The first two tickets of a wave are static, see the prologue.)
        // (Also tried: some frames done by the consumers after their GEMM + epilogue -- 13 %
        // slower, a third FFT wave per SIMD does not raise the VALU utilisation.)
        // The loop keeps the LDS round trips that are not part of the FFT off the critical path: the
        // ticket after next is drawn right before the rows are published (its return and the row writes
        // are then ONE wait), the next frame's window values are read at the end of the current frame,
        // and the buffer-free counter is only polled when the wave enters a new tile.
        const int n_tickets = (n_total + G - 1) / G;
        // First tile (G == 1, eight producers): the four older waves (0-3) win the SIMD's issue arbitration and finish a
        // frame in ~6k cycles, the younger ones in ~11k.  With two static tickets each, the tile waited for a young
        // wave's second frame (ready after ~28k cycles).  Now the young waves' second static ticket lies in tile 1
        // (16 .. 19) and tickets 12 .. 15 are drawn dynamically -- by the old waves, which finish first: three frames
        // of the first tile for an old wave, one for a young one.
        int n = wave, n2 = (SKEW && wave >= 4) ? wave + 12 : wave + NPROD;
        f2 wv[kPts];
        if (n < n_tickets) {
#pragma unroll
            for (int m = 0; m < kPts; ++m) wv[m] = winl[fl + L * m];
        }
        KPR_STAMP();
        int t_free = 1;                                               // tiles 0 and 1 start free
#pragma unroll 1
        while (n < n_tickets) {
            const int q0 = G * n;                                     // first frame of the ticket
            const int t = q0 >> 4, j = (q0 & (kFT - 1)) + grp;
            // buffer t & 1 is free once all four consumers have read tile t - 2 (monotonic counter)
            if (t > t_free) { WS_SPIN_UNTIL(&sync[2], 4 * (t - 1), 2); t_free = t; }
            KPR_DO_FRAME((t & 1) * (kFT * S) + j * S, (n2 < n_tickets) ? f_begin + G * n2 : f_end, n2 < n_tickets);
            int n3;
            WS_TICKET(n3);
            if (SKEW && n3 >= 16) n3 += 4;
            WS_SIGNAL_N(&sync[t & 1], min(G, n_total - q0));          // rows written into this buffer
            KPR_STAMP();
            n = n2;
            n2 = n3;
        }
        }
#undef WS_TICKET
    } else {
        // ================================ consumers ==========================================
        const int cw = (wave - NPROD) & 3, ctid = tid - NPROD * 64 - cgrp * 256;
        int* const gbar = &sync[cgrp ? 7 : 3];                  // this group's barrier counter
        const int jcol = lane & 15, kq = lane >> 4;
        // The consumers issue few instructions (one MFMA per 32 matrix-pipe cycles) but each one
        // competes for the SIMD's VALU issue port with two producers that always have work ready;
        // at equal priority the port goes to the older (producer) waves and the GEMM runs 2.5x
        // slower than alone.  Raise the consumers' priority.
        __builtin_amdgcn_s_setprio(kWsConsPrio);
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
        // RESIDENT filterbank slice (RES): a wave's slice of the packed band is at most kWsResident chunks for the
        // usual mel banks (37 chunks at 1025 x 128, 20 at 513 x 80: 8 floats per lane and chunk), so it is
        // loaded ONCE into registers and every tile's GEMM then only reads magnitudes from LDS -- no L2 round
        // trips inside the GEMM (streamed, a chunk took ~580 cycles for 256 cycles of MFMA) and no per-tile
        // filterbank traffic.  Wider slices (dense / log banks) keep the streaming pipeline below.
        // (FROM_MAG + RES was tried in round 3: 110 -> 105 us on six-channel rows, but the ISA audit found an MFMA reading a
        //  register of the B ring before its counted wait in that instance -- tests/test_asm_audit.py -- so it stays off)
        static_assert(!(RES && FROM_MAG), "the resident slice is for the fused kernel");
        f32x4 ares[RES ? kWsResident : 1][2];
        if constexpr (RES) {
#pragma unroll
            for (int c = 0; c < kWsResident; ++c) {
                {
                    const float* p_ = fa + (long long)min(c, max(total - 1, 0)) * 512;
                    ares[c][0] = *reinterpret_cast<const f32x4*>(p_);
                    ares[c][1] = *reinterpret_cast<const f32x4*>(p_ + 256);
                }
            }
        }
        DbRun dbrun;
        dbrun.reset();
#pragma unroll 1
        for (int it = 1 + cgrp; it <= my; it += NGRP) {     // it - 1 = tile index; itg = this group's tile count
            const int itg = (it - 1 - cgrp) / NGRP + 1;
            {
                const int tile0 = f_begin + (it - 1) * kFT;
                const float* mag = smem + ((it - 1) & 1) * (kFT * S);
                // all rows of the tile written?  (rows of this buffer so far: 16 per earlier tile)
                // (poll rarely and at low priority: the producers need the issue slots)
                __builtin_amdgcn_s_setprio(0);
                WS_SPIN_UNTIL(&sync[(it - 1) & 1], kFT * ((it - 1) >> 1) + min(kFT, f_end - tile0), 8);
                __builtin_amdgcn_s_setprio(kWsConsPrio);
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
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sa[0]) : "v"(p_) : "memory");              \
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(sa[1]) : "v"(p_) : "memory");  \
        asm volatile("ds_read2_b32 %0, %1 offset1:4" : "=v"(sb[0]) : "v"(b_) : "memory");                \
        asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:12" : "=v"(sb[1]) : "v"(b_) : "memory");     \
        asm volatile("ds_read2_b32 %0, %1 offset0:16 offset1:20" : "=v"(sb[2]) : "v"(b_) : "memory");    \
        asm volatile("ds_read2_b32 %0, %1 offset0:24 offset1:28" : "=v"(sb[3]) : "v"(b_) : "memory");    \
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
                        if constexpr (RES) {
                            // B ring only (LDS reads, DB sets of 8 registers), fully unrolled over the slice
                            constexpr int DB = 4;
                            f2 bq[DB][4];
#define KPR_ISSUE_B(sb, chunk)                                                                 \
    do {                                                                                       \
        const int n_ = max(0, min((chunk), total - 1));                                        \
        const unsigned b_ = bbase + (unsigned)(__builtin_amdgcn_readlane(cinfo, n_) & 0xffff); \
        asm volatile("ds_read2_b32 %0, %1 offset1:4" : "=v"(sb[0]) : "v"(b_) : "memory");                \
        asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:12" : "=v"(sb[1]) : "v"(b_) : "memory");     \
        asm volatile("ds_read2_b32 %0, %1 offset0:16 offset1:20" : "=v"(sb[2]) : "v"(b_) : "memory");    \
        asm volatile("ds_read2_b32 %0, %1 offset0:24 offset1:28" : "=v"(sb[3]) : "v"(b_) : "memory");    \
    } while (0)
#pragma unroll
                            for (int u = 0; u < DB - 1; ++u) KPR_ISSUE_B(bq[u], u);
                            static_for<0, kWsResident>([&](auto C_) {
                                constexpr int c = decltype(C_)::value;
                                if (c < total) {                               // wave-uniform
                                    constexpr int AHEAD = (c + DB - 1 < kWsResident) ? DB - 1 : kWsResident - 1 - c;
                                    if constexpr (c + DB - 1 < kWsResident) KPR_ISSUE_B(bq[(c + DB - 1) % DB], c + DB - 1);
                                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(4 * AHEAD) : "memory");
                                    __builtin_amdgcn_sched_barrier(0);
                                    KPR_MMA(ares[c], bq[c % DB], c);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            });
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
#undef KPR_ISSUE_B
                        } else {
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
                        }
#undef KPR_ISSUE
#undef KPR_WAIT
#undef KPR_MMA
                    }
                }
                KPR_STAMP();
                // ---- consumer-group barrier (4 waves): LDS counter, monotonically increasing ----
                WS_SIGNAL(&sync[FROM_MAG ? 5 + ((it - 1) & 1) : 2]);   // this wave is done reading the mag buffer
                WS_SIGNAL(gbar);
                WS_SPIN_UNTIL(gbar, 8 * itg - 4, 1);         // all four GEMM slices are in dpart
                KPR_STAMP();
                // ---- epilogue: dB + fully coalesced stores of the staged 16 x M tile ----------
                {
                    const int q4 = sch.ntiles * 4;                      // float4 groups per frame
                    const int ostride = spec_stride(g);
                    for (int e0 = 0; e0 < kFT * q4; e0 += 256) {        // (wave-uniform trip count: db_account is a wave operation)
                        const int e = e0 + ctid;
                        const bool in = e < kFT * q4;
                        const int j = in ? e / q4 : 0, m4 = in ? e - j * q4 : 0;
                        const long long ob = in ? fbase[j] : -1;
                        const bool act = ob >= 0;                       // frame exists
                        const int t = m4 >> 2, off = (m4 & 3) * 4;
                        const int s0 = sch.t_s0[t], ns = sch.t_ns[t];
                        f32x4 v = *reinterpret_cast<const f32x4*>(dpart + s0 * 256 + j * 16 + off);
                        for (int u = 1; u < ns; ++u)                    // partials of a split tile, in order
                            v += *reinterpret_cast<const f32x4*>(dpart + (s0 + u) * 256 + j * 16 + off);
                        const int mel = 4 * m4;
                        if (db.enabled) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = to_db(v[r], db);
                            if ((sch.M & 3) == 0) {             // (wave-uniform) four filters exist together or not at all: no masks
                                db_account(dbrun, act && mel < sch.M, fitem[j], fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])),
                                           fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), item_stats, db);
                            } else {
                            float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (mel + r < sch.M) { vmax = fmaxf(vmax, v[r]); vmin = fminf(vmin, v[r]); }
                            db_account(dbrun, act, fitem[j], vmax, vmin, item_stats, db);
                            }
                        }
                        if (act) {
                            float* outc = out + ob;
                            if (!g.out_cl && (sch.M & 3) == 0 && mel + 3 < sch.M) {
                                *reinterpret_cast<float4*>(outc + mel) = make_float4(v[0], v[1], v[2], v[3]);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (mel + r < sch.M) outc[(long long)(mel + r) * ostride] = v[r];
                            }
                        }
                    }
                }
                // dpart / fbase are rewritten by the next tile: wait until all four waves are done
                WS_SIGNAL(gbar);
                WS_SPIN_UNTIL(gbar, 8 * itg, 1);
                KPR_STAMP();
            }
        }
        if (db.enabled) db_flush_wave(dbrun, item_stats, db);       // the running per-item extrema of this wave's lanes
    }
#undef KPR_STAMP
#undef WS_SIGNAL_N
#undef WS_SIGNAL
#undef WS_SPIN_UNTIL
#undef KPR_DO_FRAME
}

}  // namespace kpr
