// kpr_mel_mr_kernels.h -- fused mel-spectrogram kernel for the mixed-radix transform sizes (n_fft = 2^a 5^b: 160, 200, 320,
// 400, 640, 800, 1000 -- the 10 ... 50 ms speech front ends at 16 kHz -- and the sizes with a factor 3: 96 ... 960, e.g. the
// 10 / 20 ms frames at 48 kHz): k_mel_mr<F>, F = MrFft<R2, R3> or TwoPassFft<N1, N2> (kpr_fft_mr.h).
// Part of the single translation unit kapre_hip.hip (included after kpr_mel_ts_kernels.h; not stand-alone).
//
// Before this kernel these sizes took two launches -- k_stft_mr writing |X| rows to HBM, then k_mel_ws<.., FROM_MAG>
// reading them back -- 149 + 152 us for 256 x 10 s @16 kHz, n_fft 400, hop 160, 80 mels (profiles/r03): 206 MB out and in
// again for 246 MB of algorithmic traffic, and a consumer-latency-bound second kernel (16-frame tiles of a 201-row
// product are ~65 MFMAs each).
//
// Structure: k_mel_ts with the mixed-radix FFT of kpr_fft_mr.h as the producer.  256-thread workgroups of FOUR equal waves,
// up to three per CU (MrFft holds 20 points per lane + 20 prefetched: 168 VGPRs = three waves per SIMD; ~45 KiB LDS); a
// workgroup walks its run of frames in rounds of RF = 4 G frames (G = 64 / L frames per wave, L = N / 20 lanes per frame;
// n_fft 400: L = 10, G = 6, RF = 24):
//   1. every wave transforms its G frames: samples (requested a round ahead) x window -> N-point complex FFT through the
//      frame's LDS row -> real-FFT pairing read out of the row into registers -> |X[k]| written back over the SAME row
//      (the row is the exchange buffer first and the magnitude row of the GEMM afterwards: 2 (N + 1) floats);
//   2. barrier; the (frame tile x filter tile) products of the round are spread over the four waves by the k_mel_ts
//      schedule (MelSchedTs, built for this kernel's frame tiles, row stride and wave count), fp32 MFMA, filterbank
//      fragments from L2 (requested before the barrier), magnitudes from LDS; a wave finishes the tiles it owns straight
//      from its accumulators (dB, per-item extrema, stores).  RF need not be a multiple of the MFMA's 16 frames: the
//      columns of the last tile beyond RF read rows 0 .. of the round again and are never stored;
//   3. barrier (magnitudes consumed; cut tiles summed).
// The workgroups of a CU drift apart, so one's GEMM / stores / sample requests run under the others' FFTs (the first
// version -- one 8-wave workgroup per CU, 48-frame rounds -- spent 45 of its 155 us in a GEMM phase during which the vector
// ALUs idled, and 94 in an FFT phase with two waves per SIMD).
// Same FFT and pairing arithmetic as k_stft_mr, the packed filterbank product of k_mel_ws with the k_mel_ts order of
// partial sums (fixed per filterbank: results are deterministic, not bit-identical to the two-launch path):
// composed.py:138-261 in one launch for these n_fft.  n_fft 400 has a lane-local pairing (see phase 1 below).
#pragma once

namespace kpr {

constexpr int kMrWaves = 4;           // waves per workgroup (six-wave workgroups, two per CU, 36-frame rounds: 210 vs 131 us)
template <class F>
__host__ __device__ constexpr int mel_mr_rf() { return kMrWaves * (64 / F::L); }            // frames per round
template <class F>
__host__ __device__ constexpr int mel_mr_nt() { return (mel_mr_rf<F>() + 15) / 16; }        // 16-frame MFMA tiles per round
// magnitude / exchange row stride in floats: >= 2 (N + 1) (the complex spectrum during the pairing), >= the padded row
// the MFMA k-ranges may touch, S % 16 == 2 (conflict-free MFMA operand reads), even (the row is addressed as f2 too)
template <class F>
__host__ __device__ constexpr int mel_mr_row_stride() {
    constexpr int cw = (F::ROW > F::N + 1) ? F::ROW : F::N + 1;          // complex words: FFT exchange, then the spectrum
    constexpr int need = (2 * cw > (F::N + 1 + kChunkRows - 1) / kChunkRows * kChunkRows)
                             ? 2 * cw : (F::N + 1 + kChunkRows - 1) / kChunkRows * kChunkRows;
    return (need - 2 + 15) / 16 * 16 + 2;
}
template <class F>
__host__ __device__ inline size_t mel_mr_lds_bytes(int nslots) {
    constexpr int RF = mel_mr_rf<F>(), S = mel_mr_row_stride<F>(), NT = mel_mr_nt<F>();
    return sizeof(float) * ((size_t)RF * S + (size_t)nslots * 256) + (size_t)2 * 16 * NT * (sizeof(long long) + sizeof(int)) +
           (size_t)3 * F::N * 2 * sizeof(float);                        // window pairs (N) + twiddle table (2 N)
}

template <class F>
__global__ __launch_bounds__(kMrWaves * 64, 3) void k_mel_mr(const float* __restrict__ x, Geom g,
                                                            const float* __restrict__ window,
                                                            const float2* __restrict__ twtab,
                                                            const float* __restrict__ fbp, MelSchedTs sch, DbDev db,
                                                            unsigned* __restrict__ item_stats, float* __restrict__ out,
                                                            int run_q, int run_r, long long* __restrict__ dbg) {
    constexpr int P = F::P, L = F::L, N = F::N, G = 64 / L, K = N + 1;
    constexpr int RF = mel_mr_rf<F>(), S = mel_mr_row_stride<F>(), NT = mel_mr_nt<F>(), RT = 16 * NT;
    constexpr int THREADS = kMrWaves * 64;
    constexpr int NIT = (N / 2) / L + 1;                                  // pairing steps of a lane: k = l + L i, 2 k <= N
    constexpr int KCAP = (K + kChunkRows - 1) / kChunkRows * kChunkRows;  // columns the MFMA k-ranges may read
    static_assert(NT <= 8 && RT - RF <= RF && RT <= THREADS, "frame tile in 3 bits; the last tile's spare columns alias rows 0 ..");
    constexpr int PIN = F::PIN, LIN = F::LIN;                             // lane l < LIN holds x[l + LIN m], m < PIN (TwoPassFft: < L, P)
    static_assert(2 * PIN <= 64, "one validity bit per sample of a lane");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef KPR_DEV_STAMPS    /* development: s_memtime stamps of the workgroup dbg[16 * 32] names, rounds 2 and 3 (tools/stamps.py) */
    int dbi = 0;
    const bool stamp_me = dbg && (long long)blockIdx.x == dbg[16 * 32];
#define MR_STAMP(cond_) do { if (stamp_me && (cond_) && lane0 == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
    const unsigned long long wg_r0 = __builtin_amdgcn_s_memrealtime(), wg_c0 = __builtin_readcyclecounter();
#else
#define MR_STAMP(cond_) do { (void)dbg; } while (0)
#endif

    float* mag = smem;                                                    // [RF][S]: exchange row, then |X| row
    float* dpart = smem + RF * S;                                         // [nslots][frame 16][filter 16]
    long long* fbase = reinterpret_cast<long long*>(dpart + sch.nslots * 256);   // [2][RT], by round parity
    int* fitem = reinterpret_cast<int*>(fbase + 2 * RT);                  // [2][RT]
    f2* winl = reinterpret_cast<f2*>(fitem + 2 * RT);                     // (0.5 w[2n], 0.5 w[2n+1])
    f2* tab = winl + N;                                                   // exp(-2 pi i j / n_fft), j < n_fft

    const int bx = (int)blockIdx.x;
    const int f_begin = (run_q * bx + min(bx, run_r)) * G;
    const int f_end = (int)min(g.total_frames, (long long)(run_q * (bx + 1) + min(bx + 1, run_r)) * G);
    const int n_total = f_end - f_begin;
    const int nrounds = (n_total + RF - 1) / RF;

    // (the window / table copies and the zero fill are issued AFTER this wave's first samples have been requested, see below)
    auto fill_tables = [&]() {
        for (int i = tid; i < N; i += THREADS) {
            const int n = 2 * i;
            const float a = window[min(n, g.win - 1)], b = window[min(n + 1, g.win - 1)];
            winl[i] = f2{(n < g.win) ? 0.5f * a : 0.0f, (n + 1 < g.win) ? 0.5f * b : 0.0f};
        }
        for (int i = tid; i < 2 * N; i += THREADS) { const float2 t = twtab[i]; tab[i] = f2{t.x, t.y}; }
        // rows no frame is written to feed the MFMAs too and must be finite: a run shorter than a round leaves some untouched
        // (otherwise round 0 writes every row -- exchange data, then magnitudes and pad columns -- before the first product)
        if (n_total < RF)
            for (int i = tid; i < RF * S; i += THREADS) mag[i] = 0.0f;
    };

    // this wave's G frames of round r: run-relative index RF r + wave G + grp.  Raw samples into zr, validity bits into vm
    // (bit 2m / 2m+1: sample 2 (l + L m) / + 1 lies inside the window and the signal); requested one round ahead.
    f2 zr[PIN];
    unsigned long long vm = 0;
    auto fetch = [&](int q0, int lane_) {                                 // q0 = first frame of the wave's group (wave-uniform)
        const bool act0 = lane_ < G * L;
        const int grp_ = act0 ? lane_ / L : 0, lf_ = act0 ? lane_ - grp_ * L : 0;
        const bool act = act0 && lf_ < LIN;                               // lanes that hold input (all of a frame's for MrFft)
        const int l_ = min(lf_, LIN - 1);
        const int gf = f_begin + q0 + grp_;
        const bool valid = act && gf < f_end;
        FramePos p = frame_pos(g, valid ? gf : f_begin);
        const float* sig = x + p.sig_off;
        const bool easy = valid && p.es == 1 && p.s0 >= 0 && p.s0 + 2 * N <= g.T && g.win >= 2 * N &&
                          (((unsigned long long)(sig + p.s0)) & 7ull) == 0;
        if (__all(easy || !act)) {                                        // whole frames inside the signal: one dwordx2 per point
            const float2* fp = reinterpret_cast<const float2*>(sig + (valid ? p.s0 : 0)) + l_;
#pragma unroll
            for (int m = 0; m < PIN; ++m) { const float2 v = fp[LIN * m]; zr[m] = f2{v.x, v.y}; }
            vm = valid ? ~0ull : 0ull;
        } else if (__all((valid && p.s0 >= 0 && p.s0 + 2 * N <= g.T && g.win >= 2 * N) || !act)) {
            // whole frames inside the signal, samples strided by the channel count (channels_last, C > 1): two plain loads
            // per point, no clamps, no validity bits (lanes without input read the head of a signal that is long enough,
            // or nothing of theirs is used)
            const int es = p.es;
            const float* q = sig + (valid ? ((long long)p.s0 + 2 * l_) * es : 0);
#pragma unroll
            for (int m = 0; m < PIN; ++m) zr[m] = f2{q[(2 * LIN * m) * es], q[(2 * LIN * m + 1) * es]};
            vm = valid ? ~0ull : 0ull;
        } else {
            const int es = p.es, omax = (int)(g.T - 1) * es;
            const int o_base = ((int)p.s0 + 2 * l_) * es;
            vm = 0;
#pragma unroll
            for (int m = 0; m < PIN; ++m) {
                const int n = 2 * (l_ + LIN * m);
                const int o0 = o_base + m * (2 * LIN) * es, o1 = o0 + es;
                zr[m] = f2{sig[min(max(o0, 0), omax)], sig[min(max(o1, 0), omax)]};
                vm |= (valid && n < g.win && (unsigned)o0 <= (unsigned)omax) ? (1ull << (2 * m)) : 0ull;
                vm |= (valid && n + 1 < g.win && (unsigned)o1 <= (unsigned)omax) ? (2ull << (2 * m)) : 0ull;
                if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (wave * G < n_total) fetch(wave * G, lane0);
    fill_tables();
    lds_barrier();

    const int n_ent = __builtin_amdgcn_readfirstlane((int)sch.tab[wave]);
    unsigned eA, eB, eF;              // entry `lane` of this wave's chunk stream (see MelSchedTs)
    {
        const unsigned* e = sch.tab + 8 + 3 * (wave * kTsMaxEnt + min(lane0, kTsMaxEnt - 1));   // (table rows of eight waves; four used)
        eA = e[0]; eB = e[1]; eF = e[2];
    }

    DbRun dbrun;
    dbrun.reset();
#pragma unroll 1
    for (int r = 0; r < nrounds; ++r) {
        // The SIMD's issue arbitration is priority, then age: of the workgroups of a CU the oldest would run ahead and leave
        // the youngest to finish alone on a third-empty CU.  Priority by rounds left keeps them level (as in k_mel_ts).
        {
            const int left = nrounds - 1 - r;
            if (left >= 3) __builtin_amdgcn_s_setprio(3);
            else if (left == 2) __builtin_amdgcn_s_setprio(2);
            else if (left == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // ---- phase 1: G frames -> |X| rows -------------------------------------------------------------------------------
        MR_STAMP(r == 2 || r == 3);
        {
            int lane_f = lane0;
            asm volatile("" : "+v"(lane_f));                              // (per-phase lane quantities: not hoisted over the GEMM)
            const int lane = lane_f;
            const bool active = lane < G * L;
            const int grp = active ? lane / L : 0, l = active ? lane - grp * L : 0;
            const int slot = wave * G;
            if (RF * r + slot < n_total) {                                // wave-uniform
                float* rowf = mag + (slot + grp) * S;
                f2* row = reinterpret_cast<f2*>(rowf);
                f2 z[P];
                const int li = min(l, LIN - 1);
                const bool has_in = active && l < LIN;
#pragma unroll
                for (int m = PIN; m < P; ++m) z[m] = f2{0.0f, 0.0f};     // (TwoPassFft with N1 < N2: not read by its first pass)
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
                {   // the next round's samples travel under this round's FFT, GEMM and stores
                    int qn = RF * (r + 1) + slot, lane_p = lane0;
                    asm volatile("" : "+s"(qn), "+v"(lane_p) :: "memory");
                    if (qn < n_total) fetch(qn, lane_p);
                }
                MR_STAMP(r == 2 || r == 3);
                if constexpr (F::L == 10 && F::N == 200) {
                    // n_fft 400 (MrFft<10, 1>): the last pass with the columns paired inside the lane.  Pass 2 is a DFT-10 over
                    // each of the 20 columns k1 of the exchanged data, X[k1 + 20 kb]; the partner of that bin, N - k1 - 20 kb,
                    // sits in column 20 - k1 (kb' = 9 - kb).  A lane that takes the columns l and 20 - l (lane 0: 0 and 10,
                    // which pair with themselves) holds every pair (k, N - k) it needs in its own registers: no natural-order
                    // write of the spectrum, no pair reads -- 42 LDS instructions and one LDS round trip less per wave and
                    // round than the generic path below.
                    Dft<P>::run(z);
#pragma unroll
                    for (int k1 = 1; k1 < P; ++k1) z[k1] = cmul(z[k1], tab[2 * l * k1]);       // W_N^{l k1}
                    KPR_LDS_FENCE_W();                                    // (kpr_fft.h: lane-to-lane hand-over without a barrier)
                    if (active) {
#pragma unroll
                        for (int k1 = 0; k1 < P; ++k1) row[l + L * k1] = z[k1];
                    }
                    KPR_LDS_FENCE_R();
                    MR_STAMP(r == 2 || r == 3);
                    const bool l0 = l == 0;
                    const int ca = l, cb = l0 ? 10 : 20 - l;
                    f2 ta[10], tb[10];
#pragma unroll
                    for (int bq = 0; bq < 10; ++bq) { ta[bq] = row[bq + L * ca]; tb[bq] = row[bq + L * cb]; }
                    Dft<10>::run(ta);                                     // ta[kb] = X[ca + 20 kb]
                    Dft<10>::run(tb);                                     // tb[kb] = X[cb + 20 kb]
                    // pair i of a lane l >= 1: (ta[i], tb[9 - i]), bins l + 20 i and N - l - 20 i.  Lane 0: X[0] with itself (-> X[0], X[N]),
                    // (ta[i], ta[10 - i]) for i = 1 .. 4, X[100] with itself, (tb[i], tb[9 - i]) for i = 0 .. 4 -- eleven.
                    float mk[11], mq[11];
                    int kk[11];
#pragma unroll
                    for (int i = 0; i < 11; ++i) {
                        f2 zk, zp;
                        int k;
                        // (the pair is always evaluated from its lower bin, k <= N / 2, as the generic path and k_stft_mr do:
                        //  same operands, same twiddle, the same magnitudes)
                        if (i <= 4) {
                            zk = ta[i];
                            const f2 p0 = ta[(10 - i) % 10];              // lane 0: ta[0], ta[9] .. ta[6]
                            zp = l0 ? p0 : tb[9 - i];
                            k = l0 ? 20 * i : l + 20 * i;
                        } else if (i == 5) {
                            zk = l0 ? ta[5] : tb[4];
                            zp = ta[5];
                            k = l0 ? 100 : 100 - l;
                        } else if (i <= 9) {
                            zk = l0 ? tb[i - 6] : tb[9 - i];
                            zp = l0 ? tb[15 - i] : ta[i];
                            k = l0 ? 20 * (i - 6) + 10 : 200 - l - 20 * i;
                        } else {                                          // lane 0 only
                            zk = tb[4]; zp = tb[5]; k = 90;
                        }
                        const f2 e = cadd_conj(zk, zp), d = csub_conj(zk, zp);
                        const f2 td = cmul(d, tab[k]);
                        f2 xk = cadd_mi(e, td);                           // e - i t d
                        f2 xq = cadd_pi(e, td);                           // conj(X[N - k])
                        if (k == 0) { xk.y = 0.0f; xq.y = 0.0f; }         // DC and Nyquist are real
                        mk[i] = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y);
                        mq[i] = __builtin_amdgcn_sqrtf(xq.x * xq.x + xq.y * xq.y);
                        kk[i] = k;
                    }
                    KPR_LDS_FENCE_W();
                    if (active) {
#pragma unroll
                        for (int i = 0; i < 11; ++i) {
                            if (i < 10 || l0) {
                                rowf[kk[i]] = mk[i];
                                if (2 * kk[i] != N) rowf[N - kk[i]] = mq[i];
                            }
                        }
                    }
                    KPR_LDS_FENCE_X();
                } else {
                    F::run(z, l, active, row, tab);                           // Z / 2 = FFT_N(z / 2)
                    MR_STAMP(r == 2 || r == 3);
                    KPR_LDS_FENCE_W();
                    if (active) {
#pragma unroll
                        for (int rr = 0; rr < P; ++rr)
                            if (F::holds(l, rr)) row[F::bin(l, rr)] = z[rr];          // natural order
                    }
                    KPR_LDS_FENCE_R();
                    // pairing (k, N - k) -> |X[k]|, |X[N - k]| (k = 0 -> X[0], X[N]); all reads of the complex row first, then
                    // the magnitudes over the same floats (LDS executes a wave's accesses in program order)
                    float mk[NIT], mq[NIT];
#pragma unroll
                    for (int i = 0; i < NIT; ++i) {
                        const int k = min(l + L * i, N / 2);
                        const int kp = (k == 0) ? 0 : N - k;
                        const f2 zk = row[k], zp = row[kp];
                        const f2 e = cadd_conj(zk, zp), d = csub_conj(zk, zp);
                        const f2 td = cmul(d, tab[k]);
                        f2 xk = cadd_mi(e, td);                               // e - i t d
                        f2 xq = cadd_pi(e, td);                               // conj(X[N - k])
                        if (k == 0) { xk.y = 0.0f; xq.y = 0.0f; }             // DC and Nyquist are real
                        mk[i] = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y);
                        mq[i] = __builtin_amdgcn_sqrtf(xq.x * xq.x + xq.y * xq.y);
                    }
                    KPR_LDS_FENCE_W();
                    if (active) {
#pragma unroll
                        for (int i = 0; i < NIT; ++i) {
                            const int k = l + L * i;
                            if (2 * k <= N) {
                                rowf[k] = mk[i];
                                if (2 * k != N) rowf[N - k] = mq[i];
                            }
                        }
                    }
                }
                if (active) {
                    for (int k = K + l; k < KCAP; k += L) rowf[k] = 0.0f; // pad columns read by the last k-step
                }
                KPR_LDS_FENCE_X();
            }
        }
        if (tid < RT) {                                                   // output base / batch item of every MFMA column
            const int qf = RF * r + tid;
            const bool ok = tid < RF && qf < n_total;
            FramePos pc = frame_pos(g, ok ? f_begin + qf : 0);
            fbase[(r & 1) * RT + tid] = ok ? spec_base(g, pc, f_begin + qf, sch.M) : -1;
            fitem[(r & 1) * RT + tid] = pc.b;
        }
        // ---- phase 2: D[filter][frame] = sum_k fb[k][filter] |X|[frame][k]  (as k_mel_ts) ---------------------------------
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
            const int eb = __builtin_amdgcn_readlane((int)eB, min(n, n_ent - 1));   // first row k0 | frame tile << 16
            int rowi = 16 * (eb >> 16) + jcol;
            if (RT != RF && rowi >= RF) rowi -= RF;                       // spare columns of the last tile: any finite row
            const float* b = mag + rowi * S + (eb & 0xffff) + kq;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.b[i] = b[4 * i];
        };
        Ops o0, o1, o2;
        if (n_ent > 0) { ldA(0, o0); ldA(1, o1); ldA(2, o2); }            // wave-uniform; requested before the barrier
        MR_STAMP(r == 2 || r == 3);
        lds_barrier();
        MR_STAMP(r == 2 || r == 3);

        const long long* fb_r = fbase + (r & 1) * RT;
        const int* fi_r = fitem + (r & 1) * RT;
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
        MR_STAMP(r == 2 || r == 3);
        lds_barrier();                                                    // magnitudes consumed, partial sums written
        MR_STAMP(r == 2 || r == 3);
        if (held_ns > 0) {                                                // wave-uniform
            for (int u = 0; u < held_ns; ++u)                             // partials of a cut tile, in order
                held += *reinterpret_cast<const f32x4*>(dpart + (held_s0 + u) * 256 + jcol * 16 + 4 * kq);
            finish(held_ft, held_t, held);
        }
    }
    if (db.enabled) db_flush_wave(dbrun, item_stats, db);
#ifdef KPR_DEV_STAMPS
    if (dbg && tid == 0 && blockIdx.x < 4096) {
        long long* e = dbg + 1024 + 4 * (long long)blockIdx.x;
        e[0] = (long long)wg_r0; e[1] = (long long)__builtin_amdgcn_s_memrealtime();
        e[2] = (long long)wg_c0; e[3] = (long long)__builtin_readcyclecounter();
    }
#endif
#undef MR_STAMP
}

}  // namespace kpr
