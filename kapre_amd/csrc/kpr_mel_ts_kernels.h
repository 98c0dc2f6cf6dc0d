// kpr_mel_ts_kernels.h -- the tile-synchronous fused mel-spectrogram kernel k_mel_ts (round 3).
// Part of the single translation unit kapre_hip.hip (included after kpr_mel_kernels.h; not stand-alone).
//
// What the round-3 probes showed (tools/probes/fft_core.hip, profiles/r03_fft_core.md): the 1024-point producer
// stream of k_mel_ws runs at 1441 / 1312 / 1184 ns per frame and SIMD with 2 / 3 / 4 waves per SIMD when nothing
// else is on the CU (72-80 % of the vector ALU's issue time at 3-4 waves), while inside k_mel_ws -- two producer
// waves per SIMD next to a consumer wave, tickets, counters, a tile that waits for its slowest wave -- the same
// stream delivered a frame per 2540 ns and SIMD.  The loss was in the orchestration, not in the FFT.
//
// k_mel_ts therefore drops the wave specialisation: ONE 1024-thread workgroup per CU, sixteen equal waves (four per
// SIMD, <= 128 VGPRs), and a round is
//   1. every wave transforms its G = 64 / L frames (frame fetch -> window -> rFFT -> |X|) into rows of the
//      magnitude tile in LDS -- 16 G frames = G MFMA tiles per round --, then issues the NEXT round's sample loads
//      (into the registers the FFT just freed) and the filterbank fragments of its GEMM slice;
//   2. barrier; every wave multiplies its 1/16 slice of the banded chunk stream (fp32 MFMA, fragments from L2,
//      magnitudes from LDS) and leaves partial 16 x 16 results in LDS;
//   3. barrier; all 1024 threads add the partials in a fixed order (deterministic), apply the optional
//      10 log10, collect the per-item max / min and store coalesced rows.
// Two s_barrier per round, no counters, no tickets, no priorities.  The GEMM phase is short because sixteen waves
// share it (38 chunks at 1025 x 128: 2-3 chunks per wave) and the samples of the next round arrive under it.
//
// Same arithmetic, in the same order, as k_mel_ws / k_mel_fused (the FFT building blocks of kpr_fft.h, the packed
// filterbank of kpr_filterbank_pack, the epilogue): composed.py:138-261 in one launch.
#pragma once

namespace kpr {

constexpr int kTsWaves = 16;          // waves per workgroup
constexpr int kTsMaxFt = 4;           // frame tiles per round (G: 1 for n_fft 2048, 2 for 1024, 4 for 512)
constexpr int kTsMaxTiles = 16;       // filter tiles (<= 256 filters)
constexpr int kTsPre = 3;             // chunks of a wave's slice whose fragments are requested before the hand-over barrier
constexpr int kTsMaxSegs = 96;

struct MelSchedTs {
    int M, ntiles, total, G;              // filters, filter tiles, chunks per frame tile, frame tiles per round
    int nseg;                             // partial-sum slots over all frame tiles of a round
    short klo[kTsMaxTiles];               // first magnitude row of filter tile t
    unsigned short chunk0[kTsMaxTiles + 1];   // first chunk of filter tile t in the packed filterbank
    // the G * total chunk items of a round (frame tile major) are cut into 16 contiguous slices, one per wave; an item
    // run inside one (frame tile, filter tile) is a segment = one partial-sum slot
    unsigned short cut[kTsWaves + 1];
    unsigned char wave_seg0[kTsWaves];
    unsigned char ts0[kTsMaxFt][kTsMaxTiles], tns[kTsMaxFt][kTsMaxTiles];   // slots of (frame tile, filter tile)
};

__host__ __device__ inline int mel_ts_rows(int NC) { return kTsWaves * (64 / (NC / kPts)); }
__host__ __device__ inline size_t mel_ts_lds_bytes(int NC, int nseg) {
    const int S = mel_ws_row_stride(NC + 1), RF = mel_ts_rows(NC);
    return sizeof(float) * ((size_t)RF * S + (size_t)nseg * 256) + (size_t)RF * (sizeof(long long) + sizeof(int)) +
           (size_t)NC * 2 * sizeof(float);
}

template <int NC>
__global__ __launch_bounds__(kTsWaves * 64) void k_mel_ts(const float* __restrict__ x, Geom g,
                                                         const float* __restrict__ window,
                                                         const float2* __restrict__ twtab,
                                                         const float* __restrict__ fbp, MelSchedTs sch, DbDev db,
                                                         unsigned* __restrict__ item_stats, float* __restrict__ out,
                                                         int run_q, int run_r, long long* __restrict__ dbg) {
    constexpr int L = NC / kPts;       // lanes per frame
    constexpr int G = 64 / L;          // frames per wave and round = frame tiles per round
    constexpr int RF = kTsWaves * G;   // frames (magnitude rows) per round
    constexpr int THREADS = kTsWaves * 64;
    typedef typename WsSwzFor<NC>::type WsSwz;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = NC + 1;
    const int S = mel_ws_row_stride(K);
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef KPR_DEV_STAMPS    /* development: s_memtime stamps of the workgroup dbg[16 * 32] names, rounds 1 and 2 (tools/stamps.py) */
    int dbi = 0;
    const bool stamp_me = dbg && (long long)blockIdx.x == dbg[kTsWaves * 32];
#define TS_STAMP(cond_) do { if (stamp_me && (cond_) && lane0 == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TS_STAMP(cond_) do { (void)dbg; } while (0)
#endif
    TS_STAMP(true);

    float* mag = smem;                                                    // [RF][S]
    float* dpart = smem + RF * S;                                         // [nseg][frame 16][filter 16]
    long long* fbase = reinterpret_cast<long long*>(dpart + sch.nseg * 256);
    int* fitem = reinterpret_cast<int*>(fbase + RF);
    f2* winl = reinterpret_cast<f2*>(fitem + RF);                         // (0.5 w[2n], 0.5 w[2n+1])

    // contiguous run of frames per workgroup, cut at G-frame granularity (as k_mel_ws: the next round's samples
    // overlap the current one's and sit in the same pages)
    const int bx = (int)blockIdx.x;
    const int f_begin = (run_q * bx + min(bx, run_r)) * G;
    const int f_end = (int)min(g.total_frames, (long long)(run_q * (bx + 1) + min(bx + 1, run_r)) * G);
    const int n_total = f_end - f_begin;
    const int nrounds = (n_total + RF - 1) / RF;

    // ---- prologue: window -> LDS, this wave's first frame(s), twiddles ------------------------------------------
    {
        constexpr int WPT = (NC + THREADS - 1) / THREADS;
        float wa[WPT], wb[WPT];
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int n = 2 * min(tid + u * THREADS, NC - 1);
            wa[u] = window[min(n, g.win - 1)];
            wb[u] = window[min(n + 1, g.win - 1)];
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int i = tid + u * THREADS, n = 2 * i;
            if (i < NC) winl[i] = f2{(n < g.win) ? 0.5f * wa[u] : 0.0f, (n + 1 < g.win) ? 0.5f * wb[u] : 0.0f};
        }
    }
    f2 nz[kPts];
    if (G * wave < n_total) {                                             // wave-uniform
        const int fl = lane0 & (L - 1), grp = lane0 / L;
        const int gf0 = f_begin + G * wave;
        const bool v0 = gf0 + grp < f_end;
        FramePos p0 = frame_pos(g, v0 ? gf0 + grp : gf0);
        fetch_frame_z<NC>(x, g, p0, v0, fl, nz);
    }
    FftTw<NC, WsSwz> tw;
    tw.load(twtab, lane0 & (L - 1));
    lds_barrier();
    TS_STAMP(true);

    const int total = sch.total;
    const int i0 = __builtin_amdgcn_readfirstlane((int)sch.cut[wave]);
    const int i1 = __builtin_amdgcn_readfirstlane((int)sch.cut[wave + 1]);
    // (frame tile, chunk, filter tile) of the first item of this wave's slice
    const int ft0 = i0 / max(total, 1), c0 = i0 - ft0 * total;
    int t0 = 0;
    while (t0 + 1 < sch.ntiles && c0 >= (int)sch.chunk0[t0 + 1]) ++t0;

#pragma unroll 1
    for (int r = 0; r < nrounds; ++r) {
        const int q = RF * r + G * wave;                                  // first frame of this wave's ticket (run-relative)
        // per-lane quantities are re-derived from an opaque copy of the lane id in every phase: hoisted out of the round
        // loop they would all stay live across the FFT (the kernel has 128 VGPRs)
        int lane_f = lane0;
        asm volatile("" : "+v"(lane_f));
        const int lane = lane_f, fl = lane & (L - 1), grp = lane / L;
        TS_STAMP(r == 1 || r == 2);
        // ---- phase 1: frame -> |X| row ---------------------------------------------------------------------------
        if (q < n_total) {                                                // wave-uniform
            float* row = mag + (G * wave + grp) * S;
            float* xrow = mag + (((G * wave + grp) * S + 3) & ~3);
            f2 z[kPts];
#pragma unroll
            for (int m = 0; m < kPts; ++m) z[m] = nz[m];
#pragma unroll
            for (int m = 0; m < kPts; ++m) z[m] = pmul(z[m], winl[fl + L * m]);
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
                        const int m = (k - fl) / L;                       // compile-time after unrolling
                        mk[m] = a;
                        mp[m] = __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y);
                    } else mid = a;                                       // k = NC / 2 (lane 0 only)
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
            for (int k = K + fl; k < S; k += L) row[k] = 0.0f;            // pad columns read by the last k-step
        }
        // the next round's samples: requested now, they arrive under the GEMM and the epilogue
        TS_STAMP(r == 1 || r == 2);
        int qn = q + RF, lane_p = lane0;
        asm volatile("" : "+s"(qn), "+v"(lane_p) :: "memory");           // nothing of the prefetch is computed above here
        if (qn < n_total) {                                               // wave-uniform
            const int flp = lane_p & (L - 1), grpp = lane_p / L;
            const int gfn = f_begin + qn;
            const bool validn = gfn + grpp < f_end;
            FramePos pn = frame_pos(g, validn ? gfn + grpp : gfn);
            fetch_frame_z<NC>(x, g, pn, validn, flp, nz);
        } else {
            // (no frame in the next round: say so -- otherwise nz has to survive this round's FFT for a next round
            // that, as far as the compiler can tell, may still read it: 32 VGPRs)
#pragma unroll
            for (int m = 0; m < kPts; ++m) nz[m] = f2{0.0f, 0.0f};
        }
        // ... and the first filterbank fragments of this wave's GEMM slice (L2)
        f32x4 apre[kTsPre][2];
        const float* fbr = fbp;
        asm volatile("" : "+s"(fbr) :: "memory");                         // (per round: loop-invariant loads would be hoisted and stay live)
        {
            int c = c0, ft = ft0;
#pragma unroll
            for (int n = 0; n < kTsPre; ++n) {
                const float* p_ = fbr + (long long)c * 512 + lane_p * 4;
                if (i0 + n < i1) {
                    apre[n][0] = *reinterpret_cast<const f32x4*>(p_);
                    apre[n][1] = *reinterpret_cast<const f32x4*>(p_ + 256);
                }
                if (++c == total) { c = 0; ++ft; }
            }
            (void)ft;
        }
        TS_STAMP(r == 1 || r == 2);
        lds_barrier();
        TS_STAMP(r == 1 || r == 2);

        int lane_g = lane0, tid_g = tid;
        asm volatile("" : "+v"(lane_g), "+v"(tid_g));
        const int jcol = lane_g & 15, kq = lane_g >> 4;
        // ---- phase 2: D[filter][frame] = sum_k fb[k][filter] |X|[frame][k], this wave's slice of the chunk items ---
        if (tid_g < RF) {                                                 // output base / batch item of every row
            const int qf = RF * r + tid_g;
            const bool ok = qf < n_total;
            FramePos pc = frame_pos(g, ok ? f_begin + qf : 0);
            fbase[tid_g] = ok ? spec_base(g, pc, f_begin + qf, sch.M) : -1;
            fitem[tid_g] = pc.b;
        }
        if (i0 < i1) {                                                    // wave-uniform
            int c = c0, ft = ft0, t = t0;
            int seg = __builtin_amdgcn_readfirstlane((int)sch.wave_seg0[wave]);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            auto item = [&](const f32x4 (&sa)[2], bool last_of_slice) {
                const int k0 = (int)sch.klo[t] + kChunkRows * (c - (int)sch.chunk0[t]);
                const float* bp = mag + (16 * ft + jcol) * S + kq + k0;
                const float b0 = bp[0], b1 = bp[4], b2 = bp[8], b3 = bp[12], b4 = bp[16], b5 = bp[20], b6 = bp[24], b7 = bp[28];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][0], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][1], b1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][2], b2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][3], b3, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][0], b4, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][1], b5, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][2], b6, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][3], b7, acc1, 0, 0, 0);
                bool close = last_of_slice;
                if (++c == (int)sch.chunk0[t + 1]) { ++t; close = true; }
                if (c == total) { c = 0; t = 0; ++ft; }
                if (close) {     // lane holds D[filter 4 kq + r][frame jcol] (partial sum of this segment)
                    *reinterpret_cast<f32x4*>(dpart + seg * 256 + jcol * 16 + 4 * kq) = acc0 + acc1;
                    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                    ++seg;
                }
            };
#pragma unroll
            for (int n = 0; n < kTsPre; ++n)
                if (i0 + n < i1) item(apre[n], i0 + n + 1 == i1);
#pragma unroll 1
            for (int i = i0 + kTsPre; i < i1; ++i) {                      // wider slices (dense / log banks): streamed
                const float* p_ = fbp + (long long)c * 512 + lane * 4;
                f32x4 sa[2];
                sa[0] = *reinterpret_cast<const f32x4*>(p_);
                sa[1] = *reinterpret_cast<const f32x4*>(p_ + 256);
                item(sa, i + 1 == i1);
            }
        }
        TS_STAMP(r == 1 || r == 2);
        lds_barrier();
        TS_STAMP(r == 1 || r == 2);

        // ---- phase 3: partial sums -> dB -> coalesced stores of the RF x M tile --------------------------------------
        {
            const int q4 = sch.ntiles * 4;                                // float4 groups per frame
            const int ostride = spec_stride(g);
            float wmax = -INFINITY, wmin = INFINITY;
            int my_b = -1;
            int tid_e = tid;
            asm volatile("" : "+v"(tid_e));
            for (int e = tid_e; e < RF * q4; e += THREADS) {
                const int j = e / q4, m4 = e - j * q4;
                const long long ob = fbase[j];
                if (ob < 0) continue;                                     // frame beyond the end
                const int t = m4 >> 2, off = (m4 & 3) * 4, ftj = j >> 4;
                const int s0 = sch.ts0[ftj][t], ns = sch.tns[ftj][t];
                f32x4 v = *reinterpret_cast<const f32x4*>(dpart + s0 * 256 + (j & 15) * 16 + off);
                for (int u = 1; u < ns; ++u)                              // partials of a split tile, in order
                    v += *reinterpret_cast<const f32x4*>(dpart + (s0 + u) * 256 + (j & 15) * 16 + off);
                const int mel = 4 * m4;
                if (db.enabled) {
                    const int b_here = fitem[j];
                    if (my_b >= 0 && my_b != b_here && wmax >= wmin) {    // rare: thread spans items
                        atomicMax(&item_stats[2 * my_b], enc_f(wmax));
                        atomicMin(&item_stats[2 * my_b + 1], enc_f(wmin));
                        wmax = -INFINITY; wmin = INFINITY;
                    }
                    my_b = b_here;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        v[rr] = to_db(v[rr], db);
                        if (mel + rr < sch.M) { wmax = fmaxf(wmax, v[rr]); wmin = fminf(wmin, v[rr]); }
                    }
                }
                float* outc = out + ob;
                if (!g.out_cl && (sch.M & 3) == 0 && mel + 3 < sch.M) {
                    *reinterpret_cast<float4*>(outc + mel) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (mel + rr < sch.M) outc[(long long)(mel + rr) * ostride] = v[rr];
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
                    if ((tid_e & 63) == 0 && wmax >= wmin) {
                        atomicMax(&item_stats[2 * b0], enc_f(wmax));
                        atomicMin(&item_stats[2 * b0 + 1], enc_f(wmin));
                    }
                } else if (my_b >= 0 && wmax >= wmin) {
                    atomicMax(&item_stats[2 * my_b], enc_f(wmax));
                    atomicMin(&item_stats[2 * my_b + 1], enc_f(wmin));
                }
            }
        }
        TS_STAMP(r == 1 || r == 2);
        // no barrier here: the next round's phase 1 only writes magnitude rows (every MFMA read of them is behind the
        // second barrier), dpart / fbase are rewritten only after the next round's first barrier
    }
#undef TS_STAMP
}

}  // namespace kpr
