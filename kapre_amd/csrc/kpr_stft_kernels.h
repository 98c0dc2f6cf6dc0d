// kpr_stft_kernels.h -- forward STFT kernels: k_stft (powers of two), k_stft_bs (Bluestein), k_stft_mr (mixed radix 2^a 5^b).
// Part of the single translation unit kapre_hip.hip (included there, in this order; not stand-alone).
#pragma once

namespace kpr {

// ------------------------------------------------------------------------------------------
// stand-alone STFT kernel (complex / magnitude / phase epilogue)
// ------------------------------------------------------------------------------------------
#define KPR_STFT_STORE(p_, v_) (*(p_) = (v_))
#ifndef KPR_STFT_WAVES
#define KPR_STFT_WAVES 4
#endif
constexpr int kStftMinBlocks = 3;   // workgroups per CU the complex / magnitude channels_first instances are register-budgeted for
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
__global__ __launch_bounds__(64 * KPR_STFT_WAVES, (MODE == KPR_OUT_PHASE || OUT_CL) ? 2 : kStftMinBlocks) void k_stft(const float* __restrict__ x, Geom g,
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
    int n = wave;                                           // first ticket is static: no sync needed
#define KPR_FETCH(n_)                                                                            \
    do {                                                                                         \
        const long long gf_ = (g_begin + (n_)) * G + grp;                                        \
        const bool valid_ = gf_ < g.total_frames;                                                \
        FramePos p_ = frame_pos(g, valid_ ? gf_ : 0);                                            \
        if constexpr (L == 16 || L == 32) fetch_frame_z<NC>(x, g, p_, valid_, fl, nz, lane, &nzsw);   /* (stereo pair form, see there) */ \
        else fetch_frame_z<NC>(x, g, p_, valid_, fl, nz);   /* no validity mask: zero fill + loads under EXEC at signal edges */ \
    } while (0)
    bool nzsw = false;                                      // nz holds the stereo pair form (fetch_frame_z): unswap at use
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
        if constexpr (L == 16 || L == 32) {
            if (nzsw) stereo_unswap<NC>(nz);                // wave-uniform
        }
#pragma unroll
        for (int m = 0; m < kPts; ++m) z[m] = pmul(nz[m], winl[fl + L * m]);
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
                KPR_LDS_FENCE_W();                 // (kpr_fft.h: the row is handed from lane to lane without a barrier)
                rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                    st2[k] = xk;
                    if (kp >= 0) st2[kp] = (kp == NC) ? f2{xp.x, 0.0f} : xp;
                });
                KPR_LDS_FENCE_R();
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
                KPR_LDS_FENCE_X();
                KPR_STAMP();
            } else {
                KPR_LDS_FENCE_W();
                rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                    stage[k] = (MODE == KPR_OUT_MAGNITUDE)
                                   ? __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y)
                                   : atan2f(k == 0 ? 0.0f : xk.y, xk.x);
                    if (kp >= 0)
                        stage[kp] = (MODE == KPR_OUT_MAGNITUDE)
                                        ? __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y)
                                        : atan2f(kp == NC ? 0.0f : xp.y, xp.x);
                });
                KPR_LDS_FENCE_R();
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
                KPR_LDS_FENCE_X();
            }
            continue;
        }
        // channels_last: bins of one frame are C elements apart -> narrow strided stores
        const long long ob = spec_base(g, p, gf, K);
        if constexpr (MODE == KPR_OUT_COMPLEX && (L == 16 || L == 32)) {
            // ... except interleaved STEREO with channel-fastest frame numbering (round 3): the two channel-frames of one
            // (item, frame) are neighbouring lane groups of this wave and their K x 2 complex values are ONE contiguous
            // block of the output.  Both spectra go to their LDS rows as in the channels_first path, then the 2 L lanes of
            // the pair write (X0[k], X1[k]) as 16-byte stores: 9 wide stores per lane instead of 17 eight-byte stores
            // 16 bytes apart.  Wave-uniform; a wave with a missing partner frame (end of the run) takes the path below.
            if (g.C == 2 && g.cfast && __all(valid && p.c == (grp & 1))) {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                f2* st2 = reinterpret_cast<f2*>(stage);
                KPR_LDS_FENCE_W();
                rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                    st2[k] = f2{xk.x, k == 0 ? 0.0f : xk.y};
                    if (kp >= 0) st2[kp] = f2{xp.x, kp == NC ? 0.0f : xp.y};
                });
                KPR_LDS_FENCE_R();
                const f2* r0 = reinterpret_cast<const f2*>(smem + (wave * G + (grp & ~1)) * (2 * NC + 8));
                const f2* r1 = r0 + (2 * NC + 8) / 2;
                f4u* o4 = reinterpret_cast<f4u*>(reinterpret_cast<float2*>(outv) + (ob - p.c));
                for (int i = lane & (2 * L - 1); i < K; i += 2 * L) {
                    const f2 a = r0[i], b = r1[i];
                    o4[i] = f4u{a.x, a.y, b.x, b.y};
                }
                KPR_LDS_FENCE_X();
                continue;
            }
        }
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
// k_stft3 (round 4): the channels_first / interleaved-pair STFT with the work split of k_mel_pw -- 128 VGPRs (sixteen waves per
// CU), no validity masks (fetch_frame_z), the next group's samples requested one group ahead.  (Round 3 had this kernel with a
// static run of frame groups per wave, k_stft2: dominated everywhere in tools/sweep_dispatch.py and removed in round 5.)
// In-kernel stamps of k_mel_pw showed what a static run per wave costs on this hardware: the SIMD's issue arbitration is oldest-first, so the four waves of a SIMD finish equal
// shares at very different times (59 k / 70 k / 83 k / 104 k cycles there) and the tail runs on a quarter-full CU.  Here
// ONE sixteen-wave workgroup per CU draws frame groups from an LDS counter: old waves simply take more groups.  The
// ticket of the next group is drawn at the top of a frame (the atomic's return rides under the window reads) and its
// samples are requested before the FFT; the twiddle set is gathered by one wave and handed over through
// LDS (sixteen waves x ten gathers at kernel start sit in the address unit's queue for microseconds).
// Same arithmetic, same store path, bit-identical output.
// ------------------------------------------------------------------------------------------
constexpr int kStft3Waves = 16;
constexpr int kStft3TwRegs = 10;      // FftTw<NC>::kNumTw <= 10
__host__ __device__ inline size_t stft3_lds_bytes(int NC) {
    const int G = 64 / (NC / kPts);
    return sizeof(float) * ((size_t)kStft3Waves * G * (2 * NC + 8) + 2 * (size_t)NC + 4 + 2 * 64 * (size_t)kStft3TwRegs);
}
template <int NC, int MODE, bool CL = false>     // CL: an interleaved side with several channels (n_fft 1024): channel-pair fetch,
                                                 // channels_last store of the wave's G channel-frames as neighbours
__global__ __launch_bounds__(64 * kStft3Waves, 4) void k_stft3(const float* __restrict__ x, Geom g,
                                                              const float* __restrict__ window,
                                                              const float2* __restrict__ twtab,
                                                              void* __restrict__ outv, int run_q, int run_r) {
    constexpr int L = NC / kPts;
    constexpr int G = 64 / L;
    constexpr int WAVES = kStft3Waves;
    typedef typename SwzFor<NC>::type SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & (L - 1), grp = (G == 1) ? 0 : lane / L;
    const int K = NC + 1;
    float* stage = smem + (wave * G + grp) * (2 * NC + 8);                 // exchange row, then the finished spectrum (16B aligned)
    float* row = stage;
    f2* winl = reinterpret_cast<f2*>(smem + WAVES * G * (2 * NC + 8));      // (0.5 w[2n], 0.5 w[2n+1])
    int* ctr = reinterpret_cast<int*>(winl + NC);                          // the workgroup's ticket counter
    f2* twl = reinterpret_cast<f2*>(ctr + 4);                              // [kNumTw][64]
    // the workgroup's contiguous run of frame groups: [n_wg0, n_wg0 + n_wg)
    const int bx = (int)blockIdx.x;
    const long long n_wg0 = (long long)run_q * bx + min(bx, run_r);
    const int n_wg = run_q + (bx < run_r ? 1 : 0);
    f2 nz[kPts];
    bool nsw = false;                                                      // nz holds the channel-pair form (fetch_frame_z)
    auto fetch = [&](int tk) {                                             // tk: group of this workgroup (wave-uniform)
        if (tk < n_wg) {
            const long long gf = (n_wg0 + tk) * G + grp;
            const bool valid = gf < g.total_frames;
            FramePos p = frame_pos(g, valid ? gf : 0);
            if constexpr (CL && L == 32) fetch_frame_z<NC>(x, g, p, valid, fl, nz, lane, &nsw);
            else fetch_frame_z<NC>(x, g, p, valid, fl, nz);                // (16 lanes per frame: the pair form costs spills here)
        }
    };
    int cur = wave;
    fetch(cur);
    static_assert(FftTw<NC, SW>::kNumTw <= kStft3TwRegs, "LDS staging area of the twiddle set");
    if (wave == 0) {
        FftTw<NC, SW> t0;
        t0.load(twtab, fl);
        t0.for_each_tw([&](f2& v, int i) { twl[i * 64 + lane] = v; });
    }
    for (int i = tid; i < NC; i += 64 * WAVES) {
        const int m = 2 * i;
        const float a = window[min(m, g.win - 1)], b = window[min(m + 1, g.win - 1)];
        winl[i] = f2{(m < g.win) ? 0.5f * a : 0.0f, (m + 1 < g.win) ? 0.5f * b : 0.0f};
    }
    if (tid == 0) *ctr = WAVES;
    lds_barrier();
    FftTw<NC, SW> tw;
    tw.for_each_tw([&](f2& v, int i) { v = twl[i * 64 + lane]; });
    tw.set_addresses(fl);
#pragma unroll 1
    while (cur < n_wg) {
        int drawn = 0;
        if (lane == 0) drawn = atomicAdd(ctr, 1);                          // ds_add_rtn_u32: returns under the window reads
        const long long gf = (n_wg0 + cur) * G + grp;
        const bool valid = gf < g.total_frames;
        FramePos p = frame_pos(g, valid ? gf : 0);
        f2 z[kPts];
        if constexpr (CL && L == 32) {
            if (nsw) stereo_unswap<NC>(nz);                                // wave-uniform
        }
#pragma unroll
        for (int m = 0; m < kPts; ++m) z[m] = pmul(nz[m], winl[fl + L * m]);
        const int nxt = __builtin_amdgcn_readfirstlane(drawn);
        // development knock-outs (tools/knockout_stft.sh; VERDICT r04 item 4: is it the stores, the loads or the transform?):
        //   KPR_STFT3_KO 1 = stores only (no sample loads, no transform: the rows keep whatever they hold; same address stream)
        //                2 = loads + transform, nothing stored        3 = loads only
#ifdef KPR_STFT3_KO
        constexpr int KO = KPR_STFT3_KO;
#else
        constexpr int KO = 0;
#endif
        if constexpr (KO != 1) fetch(nxt);                                 // next group's samples, one ahead
        asm volatile("" ::: "memory");                                     // (pins the loads here: hipcc otherwise sinks them behind the stores)
        if constexpr (KO == 3) {
#pragma unroll
            for (int m = 0; m < kPts; ++m) asm volatile("" :: "v"(z[m]));
            cur = nxt;
            continue;
        }
        tw.refresh();
        if constexpr (KO != 1) cfft_forward<NC, SW>(z, tw, row);
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        if constexpr (MODE == KPR_OUT_COMPLEX) {
            f2* st2 = reinterpret_cast<f2*>(stage);
            KPR_LDS_FENCE_W();                     // (kpr_fft.h: the rows are handed from lane to lane without a barrier)
            if constexpr (KO != 1)
            rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                st2[k] = xk;
                if (kp >= 0) st2[kp] = (kp == NC) ? f2{xp.x, 0.0f} : xp;
            });
            KPR_LDS_FENCE_R();
            if constexpr (KO == 2) { cur = nxt; continue; }
            // frames numbered (b, c, f) [or one channel]: the G rows of a wave are consecutive rows of a channels_first output
            const bool rows_adjacent = !CL && !g.out_cl && (!g.cfast || g.C == 1);
            // frames numbered channel-fastest and a channels_last output, G divides C: the G frames of a wave are neighbouring
            // CHANNELS of the same (item, frame)
            const bool chans_adjacent = CL && g.out_cl && g.cfast && (g.C % G) == 0;
            if (rows_adjacent) {                                           // (G = 1, n_fft 2048, since round 5: its rows start 8 bytes further into a line each)
                // The G rows of a wave are ONE contiguous run of the output (channels_first: row gf at (gf) K complex words),
                // G * 8 K bytes: written as such -- 64 lanes x 16 bytes = 1 KiB of consecutive addresses per instruction --
                // instead of G separate 512-byte pieces per instruction whose partial cache lines (a row is 8 K = 4104
                // bytes, no multiple of a line) are completed by other instructions at other times.  A row is a multiple
                // of 8 bytes, not of 16: every 16-byte chunk is fetched from LDS as two 8-byte halves, each from whichever
                // row it falls in.
                constexpr int RB = 8 * (NC + 1), SB = 4 * (2 * NC + 8);     // bytes per row: in the output / between LDS rows
                const long long gf0 = (n_wg0 + cur) * G;
                const int nrows = (int)min((long long)G, g.total_frames - gf0);     // (wave-uniform; the last group may be short)
                if (nrows > 0) {
                    FramePos p0 = frame_pos(g, gf0);
                    char* outb = reinterpret_cast<char*>(reinterpret_cast<float*>(outv) + 2 * spec_base(g, p0, gf0, K));
                    const char* stb = reinterpret_cast<const char*>(smem + (wave * G) * (2 * NC + 8));
                    const int total = nrows * RB;
                    struct __attribute__((aligned(8))) f2u8 { float x, y; };
                    // Round 5: every store instruction covers 1 KiB that STARTS ON A 128-BYTE LINE.  The run starts wherever row
                    // gf0 does (a multiple of 8 K = 4104 bytes: 16 bytes further into a line with every group), so instructions
                    // that start at the run's first byte each straddle nine lines and leave two of them half written until the
                    // next instruction of the wave arrives; the store-only knock-out of this kernel (tools/knockout_stft.sh) ran
                    // at 3.9 TB/s that way -- 61 of the kernel's 72 us.  Lane j of instruction q takes the 16 bytes at
                    // 16 (j + 64 q) - mis, mis = the run's offset into its line; the lanes that fall in front of the run idle.
#ifdef KPR_STFT3_NOALIGN      /* development: A/B of the store alignment */
                    const int mis = 0;
#else
                    const int mis = (int)__builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<unsigned long long>(outb) & 127ull));
#endif
#pragma unroll 3
                    for (int q = 0; q < (G * RB + 127 + 1023) / 1024; ++q) {
                        const int o = 16 * (lane + 64 * q) - mis;                // (mis is a multiple of 8)
                        const int o1 = o + 8;
                        const bool va = o >= 0 && o < total, vb = o1 >= 0 && o1 < total;
                        if (va || vb) {
                            const int oa = va ? o : o1, ob = vb ? o1 : o;
                            const int r0 = oa / RB, r1 = ob / RB;             // (compile-time divisor)
                            const f2u8 a = *reinterpret_cast<const f2u8*>(stb + r0 * SB + (oa - r0 * RB));
                            const f2u8 b = *reinterpret_cast<const f2u8*>(stb + r1 * SB + (ob - r1 * RB));
                            if (va && vb) {
                                typedef float f4nt __attribute__((ext_vector_type(4), aligned(4)));
#ifdef KPR_STFT3_PLAINST      /* development: plain instead of non-temporal wide stores */
                                *reinterpret_cast<f4nt*>(outb + o) = f4nt{a.x, a.y, b.x, b.y};
#else
                                __builtin_nontemporal_store(f4nt{a.x, a.y, b.x, b.y}, reinterpret_cast<f4nt*>(outb + o));
#endif
                            } else *reinterpret_cast<f2u8*>(outb + oa) = a;     // (va != vb: a == b, the one valid half)
                        }
                    }
                }
            } else if (chans_adjacent) {                                   // (G = 1, n_fft 2048: one channel per wave, stride C)
                // channels_last output (b, f, k, c): element idx = k G + j of the wave's G K values goes to channel c0 + j of
                // frequency k -- consecutive lanes write consecutive channels: G x 8 contiguous bytes per k (the whole run is
                // contiguous when G = C) where one frame per lane group writes 8 bytes every 8 C
                constexpr int SB = 4 * (2 * NC + 8);                          // bytes between the LDS rows of a wave
                const long long gf0 = (n_wg0 + cur) * G;
                const int nrows = (int)min((long long)G, g.total_frames - gf0);   // (wave-uniform)
                if (nrows > 0) {
                    FramePos p0 = frame_pos(g, gf0);
                    float* out0 = reinterpret_cast<float*>(outv) + 2 * spec_base(g, p0, gf0, K);   // (b, f, 0, c0)
                    const char* stb = reinterpret_cast<const char*>(smem + (wave * G) * (2 * NC + 8));
                    const int C = g.C;
                    const int j = lane & (G - 1), k0 = lane / G;                // (G is a power of two)
                    const char* sl = stb + j * SB + 8 * k0;
                    float* ol = out0 + 2 * (k0 * C + j);
#pragma unroll 2
                    for (int q = 0; q < (G * K + 63) / 64; ++q) {
                        const int k = k0 + (64 / G) * q;
                        if (k < K && j < nrows) {
                            const f2 v = *reinterpret_cast<const f2*>(sl + 8 * (64 / G) * q);
                            // (plain stores: the other channels of these cache lines come from other waves and meet them in the L2;
                            //  non-temporal ones went out as masked partial lines: 175 vs 125 us for 32 x 4 x 110250)
                            *reinterpret_cast<f2*>(ol + 2 * (64 / G) * q * C) = v;
                        }
                    }
                }
            } else if (valid) {
                float* out = reinterpret_cast<float*>(outv) + 2 * spec_base(g, p, gf, K);
#pragma unroll
                for (int q = 0; q < (2 * NC / 4) / L; ++q) {
                    const int i4 = fl + L * q;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * i4);
                    KPR_STFT_STORE(reinterpret_cast<f4u*>(out + 4 * i4), v);
                    if (q & 1) __builtin_amdgcn_sched_barrier(0);          // two at a time (register budget)
                }
                if (fl == 0) { out[2 * NC] = stage[2 * NC]; out[2 * NC + 1] = 0.0f; }
            }
        } else {
            KPR_LDS_FENCE_W();
            rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                stage[k] = (MODE == KPR_OUT_MAGNITUDE) ? __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y)
                                                       : atan2f(k == 0 ? 0.0f : xk.y, xk.x);
                if (kp >= 0)
                    stage[kp] = (MODE == KPR_OUT_MAGNITUDE) ? __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y)
                                                            : atan2f(kp == NC ? 0.0f : xp.y, xp.x);
            });
            KPR_LDS_FENCE_R();
            if (CL && g.out_cl && g.cfast && (g.C % G) == 0) {
                // channels_last output, the G frames of the wave = neighbouring channels of one (item, frame): G x 4 contiguous
                // bytes per frequency (see the complex branch)
                const long long gf0 = (n_wg0 + cur) * G;
                const int nrows = (int)min((long long)G, g.total_frames - gf0);   // (wave-uniform)
                if (nrows > 0) {
                    FramePos p0 = frame_pos(g, gf0);
                    float* out0 = reinterpret_cast<float*>(outv) + spec_base(g, p0, gf0, K);
                    const int j = lane & (G - 1), k0 = lane / G, C = g.C;
                    const float* sl = smem + (wave * G + j) * (2 * NC + 8) + k0;
                    float* ol = out0 + (k0 * C + j);
#pragma unroll 4
                    for (int q = 0; q < (G * K + 63) / 64; ++q) {
                        const int k = k0 + (64 / G) * q;
                        if (k < K && j < nrows) ol[(64 / G) * q * C] = sl[(64 / G) * q];
                    }
                }
            } else if (valid) {
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
        KPR_LDS_FENCE_X();
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------
// STFT for n_fft = 4096 / 8192 (NB = n_fft/2 = R * 1024 complex points, R = 2 / 4): one wave per
// frame runs R sub-FFTs of 1024 points (decimation in time: sub-sequence r = points r, r + R, ...)
// with the same cfft_forward<1024> as the n_fft 2048 kernels -- their results sit in the same
// lane / register (k = fl + 64 m) for every r --, multiplies by W_NB^{r k} (per-lane base x
// compile-time W_32 / W_64 steps), a radix-R butterfly across r in registers gives Z[k + 1024 s],
// and the spectrum goes to the frame's LDS row where the real-FFT pairing works in place on pairs
// (as in k_stft_mr) before the whole wave copies it out.  Without this kernel these sizes took the
// DFT-as-GEMM path: 2.9 ms instead of ~0.1 ms for 64 x 44100 samples at n_fft 4096.
// Replaces tf.signal.stft as called at kapre/time_frequency.py:174-182.
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(R == 2 ? 256 : 128, 2) void k_stft_big(const float* __restrict__ x, Geom g,
                                                                     const float* __restrict__ window,
                                                                     const float2* __restrict__ tw2048,
                                                                     const float2* __restrict__ twbig, int mode,
                                                                     void* __restrict__ outv) {
    constexpr int NC = 1024, NB = R * NC, K = NB + 1, L = 64;
    constexpr int NW = (R == 2) ? 4 : 2;                         // waves (frames in flight) per workgroup
    constexpr int RSF = NB + 1;                                  // row stride (complex words), odd
    typedef typename SwzFor<NC>::type SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fl = lane;
    f2* zrow = reinterpret_cast<f2*>(smem) + wave * RSF;         // E_r[k] at r * 1024 + k, then Z, then X
    FftTw<NC, SW> tw;
    tw.load(tw2048, fl);
    const int ostride = spec_stride(g);
#pragma unroll 1
    for (long long gf = (long long)blockIdx.x * NW + wave; gf < g.total_frames; gf += (long long)gridDim.x * NW) {
        FramePos p = frame_pos(g, gf);
        const float* sig = x + p.sig_off;
        const int es = p.es, omax = (int)(g.T - 1) * es;
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            f2 z[kPts];
            // point j = fl + 64 m of sub-sequence r is complex point n = r + R j: samples 2n, 2n + 1
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                const int n = r + R * (fl + L * m);
                const int o0 = ((int)p.s0 + 2 * n) * es, o1 = o0 + es;
                const float a = sig[min(max(o0, 0), omax)], b = sig[min(max(o1, 0), omax)];
                const float wa = window[min(2 * n, g.win - 1)], wb = window[min(2 * n + 1, g.win - 1)];
                const bool ka = 2 * n < g.win && (unsigned)o0 <= (unsigned)omax;
                const bool kb = 2 * n + 1 < g.win && (unsigned)o1 <= (unsigned)omax;
                z[m] = f2{ka ? 0.5f * a * wa : 0.0f, kb ? 0.5f * b * wb : 0.0f};
                if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            tw.refresh();
            // the block that receives E_r is free until then: it is the sub-FFT's exchange row (1080 of its 2048 floats)
            cfft_forward<NC, SW>(z, tw, reinterpret_cast<float*>(zrow + NC * r));
#pragma unroll
            for (int m = 0; m < kPts; ++m) zrow[NC * r + fl + L * m] = z[m];     // E_r[k], k = fl + 64 m
        }
        // Z[k + 1024 s] = sum_r W_R^{rs} W_NB^{rk} E_r[k]: twiddle (base W_NB^{r fl} x W_{NB/64}^{r m},
        // NB/64 = 32 or 64) and a radix-R butterfly across r, in place at the R positions of every k
#pragma unroll
        for (int m = 0; m < kPts; ++m) {
            f2 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = zrow[NC * r + fl + L * m];
            // W_NB^{r k}, k = fl + 64 m, straight from the table (L1-resident).  Until round 5 this was a per-lane base times a
            // compile-time W_32 / W_64 step held in SGPR pairs: seventeen more 64-bit scalar constants than the register file
            // holds next to the sub-FFT's own, which hipcc spilled to VGPR lanes and reloaded with v_readlane directly in front
            // of the inline-asm multiply that reads them -- 0 of the 2 wait states "VALU writes SGPR -> VALU reads it" asks
            // for, invisible to hipcc's hazard recognizer because the reader is asm (tools/hazard_scan.py).
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const float2 t = twbig[2 * r * (fl + L * m)];
                v[r] = cmul(v[r], f2{t.x, t.y});
            }
            Dft<R>::run(v);
#pragma unroll
            for (int sft = 0; sft < R; ++sft) zrow[NC * sft + fl + L * m] = v[sft];
        }
        KPR_LDS_FENCE_X();      // (kpr_fft.h) the pairing reads other lanes' words; each pair (k, NB - k) is then one lane's own
        // pairing in place: the pair (k, NB-k) -> X[k], X[NB-k]; k = 0 -> X[0], X[NB]  (row holds Z/2)
        for (int k = lane; 2 * k <= NB; k += 64) {
            const int kp = (k == 0) ? 0 : NB - k;
            const f2 zk = zrow[k], zp = zrow[kp];
            const float2 t2 = twbig[k];
            const f2 e = cadd_conj(zk, zp), d = csub_conj(zk, zp);
            const f2 td = cmul(d, f2{t2.x, t2.y});
            f2 xk = cadd_mi(e, td);                               // e - i t d
            f2 xq = cadd_pi(e, td);                               // e + i t d, conjugated below
            xq.y = -xq.y;
            if (k == 0) { xk.y = 0.0f; xq.y = 0.0f; }             // DC and Nyquist are real
            zrow[k] = xk;
            if (2 * k != NB) zrow[NB - k] = xq;
        }
        KPR_LDS_FENCE_R();
        const long long ob = spec_base(g, p, gf, K);
        if (mode == KPR_OUT_COMPLEX) {
            float2* out = reinterpret_cast<float2*>(outv) + ob;
            for (int k = lane; k < K; k += 64) { const f2 v = zrow[k]; out[(long long)k * ostride] = make_float2(v.x, v.y); }
        } else {
            float* out = reinterpret_cast<float*>(outv) + ob;
            for (int k = lane; k < K; k += 64) {
                const f2 v = zrow[k];
                out[(long long)k * ostride] = (mode == KPR_OUT_MAGNITUDE) ? __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y)
                                                                          : atan2f(v.y, v.x);
            }
        }
        KPR_LDS_FENCE_X();
    }
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
        KPR_LDS_FENCE_X();
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
        KPR_LDS_FENCE_R();          // (kpr_fft.h: the copy below reads other lanes' words)
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
        KPR_LDS_FENCE_X();
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
// exchange / spectrum row of the mixed-radix kernels: room for the FFT's exchange (F::ROW complex
// words) and for the finished spectrum (N + 1), odd stride
template <class F>
__host__ __device__ constexpr int mr_row_stride() { return ((F::ROW > F::N + 1) ? F::ROW : F::N + 1) | 1; }

template <class F>
__global__ __launch_bounds__(256, 3) void k_stft_mr(const float* __restrict__ x, Geom g,
                                                    const float* __restrict__ window,
                                                    const float2* __restrict__ twtab, int mode,
                                                    void* __restrict__ outv, long long ngroups) {
    constexpr int P = F::P, L = F::L, N = F::N, G = 64 / L, K = N + 1;
    constexpr int PIN = F::PIN, LIN = F::LIN;                     // lane l < LIN holds x[l + LIN m], m < PIN
    constexpr int RSF = mr_row_stride<F>();                       // row stride (complex words), odd
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool active = lane < G * L;                             // lanes beyond the last whole frame idle along
    const int grp = active ? lane / L : 0, l = active ? lane - grp * L : 0;
    const int li = min(l, LIN - 1);                               // lanes beyond LIN hold no input: clamped, masked
    const bool has_in = active && l < LIN;
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
    // raw samples of group `gi_` into zr / vm (unconditional loads from clamped offsets, masked when
    // the group is transformed) and its output base into ob_; issued one group ahead
#define MR_FETCH(gi_, ob_)                                                                       \
    do {                                                                                         \
        const long long gf_ = (gi_) * G + grp;                                                   \
        const bool valid_ = has_in && gf_ < g.total_frames;                                      \
        FramePos p_ = frame_pos(g, (active && gf_ < g.total_frames) ? gf_ : 0);                                            \
        ob_ = (active && gf_ < g.total_frames) ? spec_base(g, p_, gf_, K) : -1;                                            \
        const float* sig_ = x + p_.sig_off;                                                      \
        const bool easy_ = valid_ && p_.es == 1 && p_.s0 >= 0 && p_.s0 + 2 * N <= g.T && g.win >= 2 * N && \
                           (((unsigned long long)(sig_ + p_.s0)) & 7ull) == 0;                   \
        if (__all(easy_ || !has_in)) {         /* whole frames inside the signal: one dwordx2 per point */ \
            const float2* fp_ = reinterpret_cast<const float2*>(sig_ + (valid_ ? p_.s0 : 0)) + li; \
            _Pragma("unroll") for (int m = 0; m < PIN; ++m) { const float2 v_ = fp_[LIN * m]; zr[m] = f2{v_.x, v_.y}; } \
            vm = valid_ ? ~0ull : 0ull;                                                          \
        } else {                                                                                 \
            const int es_ = p_.es, omax_ = (int)(g.T - 1) * es_;                                 \
            const int o_base_ = ((int)p_.s0 + 2 * li) * es_;                                      \
            vm = 0;                                                                              \
            _Pragma("unroll") for (int m = 0; m < PIN; ++m) {                                    \
                const int n_ = 2 * (li + LIN * m);                                                  \
                const int o0_ = o_base_ + m * (2 * LIN) * es_, o1_ = o0_ + es_;                    \
                zr[m] = f2{sig_[min(max(o0_, 0), omax_)], sig_[min(max(o1_, 0), omax_)]};        \
                vm |= (valid_ && n_ < g.win && (unsigned)o0_ <= (unsigned)omax_) ? (1ull << (2 * m)) : 0ull; \
                vm |= (valid_ && n_ + 1 < g.win && (unsigned)o1_ <= (unsigned)omax_) ? (2ull << (2 * m)) : 0ull; \
                if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);                             \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    f2 zr[PIN];
    unsigned long long vm = 0;
    long long ob_next = -1;
    long long grpi = (long long)blockIdx.x * 4 + wave;
    if (grpi < ngroups) MR_FETCH(grpi, ob_next);
#pragma unroll 1
    for (; grpi < ngroups; grpi += (long long)gridDim.x * 4) {
        const long long ob = ob_next;
        f2 z[P];
        if (__all(vm == ~0ull || !has_in)) {
#pragma unroll
            for (int m = 0; m < PIN; ++m) z[m] = pmul(zr[m], winl[li + LIN * m]);
        } else {
#pragma unroll
            for (int m = 0; m < PIN; ++m) {
                const unsigned kx = (unsigned)(-(int)((vm >> (2 * m)) & 1ull));
                const unsigned ky = (unsigned)(-(int)((vm >> (2 * m + 1)) & 1ull));
                const f2 v = f2{__uint_as_float(__float_as_uint(zr[m].x) & kx), __uint_as_float(__float_as_uint(zr[m].y) & ky)};
                z[m] = pmul(v, winl[li + LIN * m]);
            }
        }
        {   // the next group's samples travel while this one is transformed
            const long long gn = grpi + (long long)gridDim.x * 4;
            if (gn < ngroups) MR_FETCH(gn, ob_next);
        }
        // ---- Z/2 = FFT_N(z / 2), left in the row in natural order ---------------------------------
        F::run(z, l, active, row, tab);
        if (active) {
#pragma unroll
            for (int r = 0; r < P; ++r)
                if (F::holds(l, r)) row[F::bin(l, r)] = z[r];
        }
        KPR_LDS_FENCE_X();          // (kpr_fft.h) the pairing reads other lanes' words; a pair (k, N - k) is then one lane's own
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
        KPR_LDS_FENCE_R();
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
        KPR_LDS_FENCE_X();
    }
#undef MR_FETCH
}

}  // namespace kpr
