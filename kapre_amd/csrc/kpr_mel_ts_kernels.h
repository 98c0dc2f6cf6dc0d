// kpr_mel_ts_kernels.h -- the tile-synchronous fused mel-spectrogram kernel k_mel_ts (round 3).
// Part of the single translation unit kapre_hip.hip (included after kpr_mel_kernels.h; not stand-alone).
//
// What the round-3 probes showed (tools/probes/fft_core.hip, profiles/r03_fft_core.md): the 1024-point producer
// stream of k_mel_ws runs at 1441 / 1312 / 1184 ns per frame and SIMD with 2 / 3 / 4 waves per SIMD when nothing
// else is on the CU (72-80 % of the vector ALU's issue time at 3-4 waves), while inside k_mel_ws -- two producer
// waves per SIMD next to a consumer wave, tickets, counters, a tile that waits for its slowest wave -- the same
// stream delivered a frame per 2540 ns and SIMD.  The loss was in the orchestration, not in the FFT.
//
// k_mel_ts drops the wave specialisation.  512-thread workgroups of eight EQUAL waves, TWO workgroups per CU (four
// waves per SIMD, <= 128 VGPRs, <= 80 KiB LDS each); a workgroup walks its run of frames in rounds of RF = 16 (n_fft
// 2048, 1024) or 32 (512) frames:
//   1. every wave transforms its share of the round (frame fetch -> window -> rFFT -> |X|) into rows of the magnitude
//      tile in LDS, then requests the NEXT round's samples (into the registers the FFT just freed);
//   2. barrier; the (frame tile x filter tile) products of the round are spread over the waves as whole items -- a
//      filter tile with many chunks is cut in two or three, so that the four SIMDs' matrix pipes get equal shares --
//      fp32 MFMA, filterbank fragments from L2, magnitudes from LDS; a wave finishes an item it owns completely
//      straight from its accumulators: 10 log10, per-item max / min, 16-byte stores;
//   3. barrier (magnitudes consumed); owners of a cut tile add the other parts' partial sums (fixed order:
//      deterministic) from LDS and finish likewise.
// Two s_barrier per round, no counters, no tickets, no priorities; the two workgroups of a CU drift apart by
// themselves, so one's GEMM / stores / sample requests run under the other's FFTs.
//
// Same arithmetic, in the same order, as k_mel_ws / k_mel_fused (the FFT building blocks of kpr_fft.h, the packed
// filterbank of kpr_filterbank_pack): composed.py:138-261 in one launch.
#pragma once

namespace kpr {

constexpr int kTsWaves = 8;           // waves per workgroup
constexpr int kTsMaxItems = 6;        // GEMM items per wave
constexpr int kTsMaxTiles = 16;       // filter tiles (<= 256 filters)
constexpr int kTsMaxSlots = 16;       // partial-sum slots (parts of cut filter tiles)

constexpr int kTsMaxEnt = 48;         // chunk entries per wave
struct MelItemTs {                    // host side only: a (frame tile, filter tile) product or a part of one
    unsigned char ft, t;              // frame tile of the round, filter tile
    unsigned char nch;                // chunks of this item
    unsigned char kind;               // 0: the wave finishes the tile (adds `nslots` partial sums first), 1: a part
    unsigned short c0;                // first chunk (index into the packed filterbank)
    unsigned char slot0, nslots;      // kind 0: slots [slot0, slot0 + nslots) to add; kind 1: slot0 = slot to write
};
// What the kernel gets: per wave a flat stream of chunk entries (device table, built once per filterbank geometry):
//   tab[w] = entries of wave w;  entry n of wave w at tab[8 + 3 (w kTsMaxEnt + n)]:
//     [0] chunk index in the packed filterbank | last chunk of its item << 31
//     [1] float offset of the chunk's first magnitude in the tile: 16 ft S + k0
//     [2] last chunks only: kind | t << 1 | ft << 5 | slot0 << 8 | nslots << 16
// Each lane keeps entry `lane` of its wave in three registers; v_readlane with the (wave-uniform) entry number feeds the
// software pipeline without a memory access.
struct MelSchedTs {
    int M, ntiles, FT, nslots;
    const unsigned* tab;              // device
};
// frames per round / tickets (G frames of one wave) per wave and round
// n_fft 1024 (NC 512): 32 frames = two tickets per wave and round (16-frame rounds measured 270 us on cfg5 against 241: a
// round's two barriers and its GEMM fill / drain are paid per round, whatever its size); n_fft 2048 cannot (its 16 rows are
// 70 KB of the 80 a workgroup may use); n_fft 512 with 64-frame rounds: 61 vs 65 us on the stereo + dB test shape but 29.7
// vs 27.6 on the mono one (84 frames per workgroup): 32 by default, k_mel_ts<256, 64> for runs of 64 k frames and more.
__host__ __device__ constexpr int mel_ts_rf(int NC) { return NC == 512 ? 32 : (kTsWaves * (64 / (NC / kPts)) < 16) ? 16 : kTsWaves * (64 / (NC / kPts)); }
__host__ __device__ inline size_t mel_ts_lds_bytes(int NC, int nslots, int RF = 0) {
    const int S = mel_ws_row_stride(NC + 1);
    if (!RF) RF = mel_ts_rf(NC);
    return sizeof(float) * ((size_t)RF * S + (size_t)nslots * 256) + (size_t)2 * RF * (sizeof(long long) + sizeof(int)) +
           (size_t)NC * 2 * sizeof(float);
}

// RF_ = frames per round (0: mel_ts_rf(NC)); the launcher picks 64 for long n_fft 512 runs
template <int NC, int RF_ = 0>
__global__ __launch_bounds__(kTsWaves * 64, 4) void k_mel_ts(const float* __restrict__ x, Geom g,
                                                            const float* __restrict__ window,
                                                            const float2* __restrict__ twtab,
                                                            const float* __restrict__ fbp, MelSchedTs sch, DbDev db,
                                                            unsigned* __restrict__ item_stats, float* __restrict__ out,
                                                            int run_q, int run_r, long long* __restrict__ dbg) {
    constexpr int L = NC / kPts;       // lanes per frame
    constexpr int G = 64 / L;          // frames per wave and ticket
    constexpr int RF = RF_ ? RF_ : mel_ts_rf(NC);  // frames (magnitude rows) per round
    static_assert(RF % (kTsWaves * G) == 0 && RF / 16 <= 8 && (RF / (kTsWaves * G) == 1 || RF / (kTsWaves * G) == 2), "one or two tickets per wave and round");
    constexpr int TPW = RF / (kTsWaves * G);   // tickets per wave and round (2 for n_fft 2048 and 1024, else 1)
    constexpr int THREADS = kTsWaves * 64;
    typedef typename WsSwzFor<NC>::type WsSwz;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = NC + 1;
    const int S = mel_ws_row_stride(K);
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef KPR_DEV_STAMPS    /* development: s_memtime stamps of the workgroup dbg[16 * 32] names, rounds 1 and 2 (tools/stamps.py) */
    int dbi = 0;
    const bool stamp_me = dbg && (long long)blockIdx.x == dbg[16 * 32];
#define TS_STAMP(cond_) do { if (stamp_me && (cond_) && lane0 == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TS_STAMP(cond_) do { (void)dbg; } while (0)
#endif
    TS_STAMP(true);
#ifdef KPR_DEV_STAMPS    /* every workgroup: start / end on the constant 100 MHz clock and on the shader clock (dbg[1024 + 4 bx ..]) */
    const unsigned long long wg_r0 = __builtin_amdgcn_s_memrealtime(), wg_c0 = __builtin_readcyclecounter();
#endif

    float* mag = smem;                                                    // [RF][S]
    float* dpart = smem + RF * S;                                         // [nslots][frame 16][filter 16]
    long long* fbase = reinterpret_cast<long long*>(dpart + sch.nslots * 256);   // [2][RF], by round parity
    int* fitem = reinterpret_cast<int*>(fbase + 2 * RF);                  // [2][RF]
    f2* winl = reinterpret_cast<f2*>(fitem + 2 * RF);                     // (0.5 w[2n], 0.5 w[2n+1])

    // contiguous run of frames per workgroup, cut at G-frame granularity (the next round's samples overlap the current
    // one's and sit in the same pages)
    const int bx = (int)blockIdx.x;
    const int f_begin = (run_q * bx + min(bx, run_r)) * G;
    const int f_end = (int)min(g.total_frames, (long long)(run_q * (bx + 1) + min(bx + 1, run_r)) * G);
    const int n_total = f_end - f_begin;
    const int nrounds = (n_total + RF - 1) / RF;

    // ---- prologue: window -> LDS, this wave's first frame(s), twiddles ------------------------------------------
    // All three sets of loads are REQUESTED before any of them is used (the window values are written to LDS further
    // down): the workgroup pays one cold memory latency, not two in a row.
    constexpr int WPT = (NC + THREADS - 1) / THREADS;
    float wa[WPT], wb[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int n = 2 * min(tid + u * THREADS, NC - 1);
        wa[u] = window[min(n, g.win - 1)];
        wb[u] = window[min(n + 1, g.win - 1)];
    }
    // the G frames of ticket tk of round r: run-relative index RF r + (tk kTsWaves + wave) G + grp
    // returns true (wave-uniform) when the registers hold the stereo pair form and need stereo_unswap() before use
    auto fetch_ticket = [&](int qt, int lane_, f2 (&dst)[kPts]) -> bool { // qt = first frame of the ticket (wave-uniform)
        bool sw = false;
        if (qt < n_total) {
            const int fl_ = lane_ & (L - 1), grp_ = (G == 1) ? 0 : lane_ / L;
            const int gf = f_begin + qt;
            const bool v = gf + grp_ < f_end;
            FramePos p = frame_pos(g, v ? gf + grp_ : gf);
            if constexpr (L == 16 || L == 32) fetch_frame_z<NC>(x, g, p, v, fl_, dst, lane_, &sw);
            else fetch_frame_z<NC>(x, g, p, v, fl_, dst);
        } else {
            // (no such frame: say so -- otherwise the registers have to survive a whole FFT for a ticket that, as far as
            // the compiler can tell, may still read them: 32 VGPRs)
#pragma unroll
            for (int m = 0; m < kPts; ++m) dst[m] = f2{0.0f, 0.0f};
        }
        return sw;
    };
    // rows no frame is written to feed the MFMAs too and must be finite: a run shorter than a round leaves some untouched
    // (in every other case round 0 writes all RF rows, pad columns included, before the first product)
    if (n_total < RF)
        for (int i = tid; i < RF * S; i += THREADS) mag[i] = 0.0f;
    // nz[0]: the first ticket of the coming round, requested before the previous round's hand-over barrier (arrives under
    // the GEMM and the stores); nz[1] (n_fft 2048: two tickets per wave and round): the second ticket, requested when the
    // round starts (arrives under the first ticket's FFT; the FFT leaves room for it: ~86 live VGPRs)
    // Requests are placed where the wave is about to wait anyway (a vector-memory instruction blocks the wave's issue
    // while the CU's address unit is busy: sixteen waves x 16-26 loads are ~3k cycles of it): nz[0] before the hand-over
    // barrier, nz[1] and the twiddles -- re-read every round (L1 / L2): held across the GEMM they would cost 20 of its
    // VGPRs -- before the second barrier.
    f2 nz[TPW][kPts];
    bool nsw[TPW] = {};                // nz[tk] holds the stereo pair form (fetch_frame_z)
    FftTw<NC, WsSwz> tw;
    tw.load(twtab, lane0 & (L - 1));
    nsw[0] = fetch_ticket(wave * G, lane0, nz[0]);
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + u * THREADS, n = 2 * i;
        if (i < NC) winl[i] = f2{(n < g.win) ? 0.5f * wa[u] : 0.0f, (n + 1 < g.win) ? 0.5f * wb[u] : 0.0f};
    }
    lds_barrier();
    TS_STAMP(true);

    const int n_ent = __builtin_amdgcn_readfirstlane((int)sch.tab[wave]);
    unsigned eA, eB, eF;              // entry `lane` of this wave's chunk stream
    {
        const unsigned* e = sch.tab + 8 + 3 * (wave * kTsMaxEnt + min(lane0, kTsMaxEnt - 1));
        eA = e[0]; eB = e[1]; eF = e[2];
    }

    DbRun dbrun;                                                         // running per-item extrema of this wave's lanes (dB)
    dbrun.reset();
#pragma unroll 1
    for (int r = 0; r < nrounds; ++r) {
        // The SIMD's issue arbitration is priority, then age: of the two workgroups of a CU the older one would run ahead
        // and leave the younger one to finish alone, on a half-empty CU (measured: 40 vs 54 us).  The workgroup with more
        // rounds left gets the higher priority, so the pair stays level.
        {
            const int left = nrounds - 1 - r;
            if (left >= 3) __builtin_amdgcn_s_setprio(3);
            else if (left == 2) __builtin_amdgcn_s_setprio(2);
            else if (left == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        TS_STAMP(r == 1 || r == 2);
        // ---- phase 1: frames -> |X| rows ---------------------------------------------------------------------------
#pragma unroll
        for (int tk = 0; tk < TPW; ++tk) {
            // per-lane quantities are re-derived from an opaque copy of the lane id in every phase: hoisted out of the
            // round loop they would all stay live across the FFT (the kernel has 128 VGPRs)
            int lane_f = lane0;
            asm volatile("" : "+v"(lane_f));
            const int lane = lane_f, fl = lane & (L - 1), grp = (G == 1) ? 0 : lane / L;
            const int slot = (tk * kTsWaves + wave) * G;                  // first row of the ticket
            const int q = RF * r + slot;
            if (q < n_total) {                                            // wave-uniform
                float* row = mag + (slot + grp) * S;
                float* xrow = mag + (((slot + grp) * S + 3) & ~3);
                f2 z[kPts];
                if constexpr (L == 16 || L == 32) {
                    if (nsw[tk]) stereo_unswap<NC>(nz[tk]);               // wave-uniform
                }
#pragma unroll
                for (int m = 0; m < kPts; ++m) z[m] = pmul(nz[tk][m], winl[fl + L * m]);
                tw.refresh();
                if constexpr (IsWide<WsSwz>::value) {
                    cfft_forward_wide_planar(z, tw, xrow);
                } else {
                    using Rx = Radix<NC>;
                    fft_pass<NC, 1, Rx::r1, 1, WsSwz>(z, tw, xrow);
                    fft_pass<NC, 2, Rx::r2, Rx::r1, WsSwz>(z, tw, xrow);
                    if constexpr (Rx::r3 > 1) fft_pass<NC, 3, Rx::r3, Rx::r1 * Rx::r2, WsSwz>(z, tw, xrow);
                }
                if constexpr (L == 64 || L == 32) {
                    float mk[kPts / 2], mp[kPts / 2];
                    float mid = 0.0f;
                    rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                        const float a = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y);
                        if (kp >= 0) {
                            const int m = (k - fl) / L;                   // compile-time after unrolling
                            mk[m] = a;
                            mp[m] = __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y);
                        } else mid = a;                                   // k = NC / 2 (lane 0 only)
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
                        row[k] = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y);
                        if (kp >= 0) row[kp] = __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y);
                    });
                }
                for (int k = K + fl; k < S; k += L) row[k] = 0.0f;        // pad columns read by the last k-step
            }
            TS_STAMP(r == 1 || r == 2);
            if constexpr (TPW == 2) {
                if (tk == 0) {                                            // the second ticket: requested here, used right away
                    int q1 = RF * r + (kTsWaves + wave) * G, lane_p = lane0;
                    asm volatile("" : "+s"(q1), "+v"(lane_p) :: "memory");
                    nsw[1] = fetch_ticket(q1, lane_p, nz[1]);
                }
            }
        }
        {   // the first ticket of the NEXT round: those samples arrive under the GEMM and the stores
            int qn = RF * (r + 1) + wave * G, lane_p = lane0;
            asm volatile("" : "+s"(qn), "+v"(lane_p) :: "memory");       // nothing of the fetch is computed above here
            nsw[0] = fetch_ticket(qn, lane_p, nz[0]);
        }
        if (tid < RF) {                                                   // output base / batch item of every row
            const int qf = RF * r + tid;
            const bool ok = qf < n_total;
            FramePos pc = frame_pos(g, ok ? f_begin + qf : 0);
            fbase[(r & 1) * RF + tid] = ok ? spec_base(g, pc, f_begin + qf, sch.M) : -1;
            fitem[(r & 1) * RF + tid] = pc.b;
        }
        // ---- phase 2: D[filter][frame] = sum_k fb[k][filter] |X|[frame][k] ----------------------------------------------
        // One software pipeline per wave over its chunk entries, both operands three entries ahead: filterbank fragments
        // from L2 (the first three are requested BEFORE the barrier), magnitudes from LDS (under the other workgroup's FFT
        // exchanges an LDS read returns after ~1k cycles; a chunk's eight MFMAs take 256).
        int lane_g = lane0;
        asm volatile("" : "+v"(lane_g));
        const int jcol = lane_g & 15, kq = lane_g >> 4;
        struct Ops { f32x4 a0, a1; float b[8]; };
        auto ldA = [&](int n, Ops& o) {
            const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)eA, min(n, n_ent - 1)) & 0xffffu;
            const float* q_ = fbp + (long long)a * 512 + lane_g * 4;
            o.a0 = *reinterpret_cast<const f32x4*>(q_);
            o.a1 = *reinterpret_cast<const f32x4*>(q_ + 256);
        };
        auto ldB = [&](int n, Ops& o) {
            const int boff = __builtin_amdgcn_readlane((int)eB, min(n, n_ent - 1));
            const float* b = mag + boff + jcol * S + kq;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.b[i] = b[4 * i];
        };
        Ops o0, o1, o2;
        if (n_ent > 0) { ldA(0, o0); ldA(1, o1); ldA(2, o2); }            // wave-uniform
        TS_STAMP(r == 1 || r == 2);
        lds_barrier();
        TS_STAMP(r == 1 || r == 2);

        const long long* fb_r = fbase + (r & 1) * RF;
        const int* fi_r = fitem + (r & 1) * RF;
        // finish a 16 x 16 tile from the accumulators: lane holds D[filter 16 t + 4 kq + e][frame 16 ft + jcol]
        auto finish = [&](int ft, int t, f32x4 v) {
            const int j = 16 * ft + jcol;
            const long long ob = fb_r[j];
            const int mel = 16 * t + 4 * kq;
            if (db.enabled) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = to_db(v[e], db);
                if ((sch.M & 3) == 0) {
                    // (wave-uniform) a lane's four filters exist together or not at all: no per-value masks -- those were
                    // eight v_cndmask on VCC per call, ~20 cycles each on gfx950
                    const float vmax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    const float vmin = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
                    const bool have = ob >= 0 && mel < sch.M;
                    db_account(dbrun, have, have ? fi_r[j] : -1, vmax, vmin, item_stats, db);
                } else {
                    float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (mel + e < sch.M) { vmax = fmaxf(vmax, v[e]); vmin = fminf(vmin, v[e]); }
                    db_account(dbrun, ob >= 0, (ob >= 0) ? fi_r[j] : -1, vmax, vmin, item_stats, db);
                }
            }
            if (ob >= 0) {
                float* outc = out + ob;
                if (!g.out_cl && (sch.M & 3) == 0 && mel + 3 < sch.M) {
                    *reinterpret_cast<float4*>(outc + mel) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    const int ostride = spec_stride(g);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (mel + e < sch.M) outc[(long long)(mel + e) * ostride] = v[e];
                }
            }
        };
        f32x4 held = {0.f, 0.f, 0.f, 0.f};
        int held_ft = 0, held_t = 0, held_s0 = 0, held_ns = 0;
        if (n_ent > 0) {                                                  // wave-uniform
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            auto step = [&](int n, Ops& o) {                              // entry n: eight MFMAs, item end, refill the set
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0[0], o.b[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0[1], o.b[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0[2], o.b[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0[3], o.b[3], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1[0], o.b[4], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1[1], o.b[5], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1[2], o.b[6], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1[3], o.b[7], acc1, 0, 0, 0);
                if (n + 3 < n_ent) { ldA(n + 3, o); ldB(n + 3, o); }
                if (__builtin_amdgcn_readlane((int)eA, n) < 0) {          // last chunk of its item (bit 31)
                    const unsigned fin = (unsigned)__builtin_amdgcn_readlane((int)eF, n);
                    const int kind = fin & 1, t = (fin >> 1) & 15, ft = (fin >> 5) & 7, slot0 = (fin >> 8) & 255, ns = (fin >> 16) & 15;
                    const f32x4 d = acc0 + acc1;
                    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (kind == 1) *reinterpret_cast<f32x4*>(dpart + slot0 * 256 + jcol * 16 + 4 * kq) = d;
                    else if (ns == 0) finish(ft, t, d);
                    else { held = d; held_ft = ft; held_t = t; held_s0 = slot0; held_ns = ns; }
                }
            };
            ldB(0, o0); ldB(1, o1); ldB(2, o2);
#pragma unroll 1
            for (int n = 0; n < n_ent; n += 3) {
                step(n, o0);
                if (n + 1 < n_ent) step(n + 1, o1);
                if (n + 2 < n_ent) step(n + 2, o2);
            }
        }
        {   // next round: twiddle factors and the second ticket (see the prologue)
            const float2* tt = twtab;
            int lane_t = lane0, q1 = RF * (r + 1) + (kTsWaves + wave) * G;
            asm volatile("" : "+s"(tt), "+v"(lane_t), "+s"(q1) :: "memory");
            tw.load(tt, lane_t & (L - 1));
            (void)q1;
        }
        TS_STAMP(r == 1 || r == 2);
        lds_barrier();                                                    // magnitudes consumed, partial sums written
        TS_STAMP(r == 1 || r == 2);
        if (held_ns > 0) {                                                // wave-uniform
            for (int u = 0; u < held_ns; ++u)                             // partials of a cut tile, in order
                held += *reinterpret_cast<const f32x4*>(dpart + (held_s0 + u) * 256 + jcol * 16 + 4 * kq);
            finish(held_ft, held_t, held);
        }
        TS_STAMP(r == 1 || r == 2);
        // (dpart is rewritten only after the next round's first barrier, which this wave's reads precede)
    }
    if (db.enabled) db_flush_wave(dbrun, item_stats, db);
#ifdef KPR_DEV_STAMPS
    if (dbg && tid == 0 && blockIdx.x < 4096) {
        long long* e = dbg + 1024 + 4 * (long long)blockIdx.x;
        e[0] = (long long)wg_r0; e[1] = (long long)__builtin_amdgcn_s_memrealtime();
        e[2] = (long long)wg_c0; e[3] = (long long)__builtin_readcyclecounter();
    }
#endif
#undef TS_STAMP
}

}  // namespace kpr
