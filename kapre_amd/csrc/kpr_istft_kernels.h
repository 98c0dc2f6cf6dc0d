// kpr_istft_kernels.h -- inverse kernels: k_irfft / k_irfft_bs / k_irfft_mr + k_ola (two-kernel path), k_istft_fused (barrier kernel),
// k_istft_ws / k_istft_ws_mr (ring kernels: FFT producer waves + overlap-add consumer wave).
// Part of the single translation unit kapre_hip.hip (included there, in this order; not stand-alone).
#pragma once

namespace kpr {

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
template <class F>
__global__ __launch_bounds__(256, 2) void k_irfft_mr(const float2* __restrict__ spec, Geom g,
                                                     const float* __restrict__ synth,
                                                     const float2* __restrict__ twtab,
                                                     float* __restrict__ frames, long long ngroups) {
    constexpr int P = F::P, L = F::L, N = F::N, G = 64 / L, K = N + 1;
    constexpr int PIN = F::PIN, LIN = F::LIN;                     // lane l < LIN holds Z[l + LIN m], m < PIN
    constexpr int RSF = mr_row_stride<F>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool active = lane < G * L;
    const int grp = active ? lane / L : 0, l = active ? lane - grp * L : 0;
    const int li = min(l, LIN - 1);
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
        for (int m = 0; m < PIN; ++m) {            // unconditional loads, masked below
            const int k = li + LIN * m;            // < N
            float2 a = sp[(long long)k * ostride], b = sp[(long long)(N - k) * ostride];
            if (k == 0) { a.y = 0.0f; b.y = 0.0f; }                        // irfft ignores Im of DC / Nyquist
            const f2 xk = f2{a.x, a.y}, xp = f2{b.x, -b.y};                 // X[k], conj X[N-k]
            const f2 e = cadd(xk, xp), d = csub(xk, xp);
            const f2 tc = tab[k];
            const f2 od = cmul(d, f2{tc.x, -tc.y});                        // (X - conj X') conj(t)
            f2 zc = f2{e.x - od.y, -(e.y + od.x)};                         // conj(2 Z) = conj(E + i O)
            if (!valid || l >= LIN) zc = f2{0.0f, 0.0f};
            z[m] = zc;
        }
        F::run(z, l, active, row, tab);                                    // Y = FFT_N(conj 2Z)
        if (!valid) continue;
        float* fo = frames + gf * (long long)g.win;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            if (!F::holds(l, r)) continue;
            const int n = F::bin(l, r);                                    // z[n] = conj(Y[n]) / (2N)
            const f2 w = winl[n];
            if (2 * n < g.win) fo[2 * n] = z[r].x * w.x;
            if (2 * n + 1 < g.win) fo[2 * n + 1] = -z[r].y * w.y;
        }
        // win_length > n_fft: the irfft output is right-padded with zeros
        for (int n = 2 * N + l; n < g.win; n += L) fo[n] = 0.0f;
    }
}

// Inverse counterpart of k_stft_big (InverseSTFT at n_fft 4096 / 8192, NB = R * 1024): the spectrum row
// goes to LDS, the inverse pairing works in place on pairs (k, NB-k) -> conj(2 Z[k]), conj(2 Z[NB-k]),
// the NB-point FFT is R sub-FFTs of 1024 points + the in-place radix-R combine of k_stft_big (input
// row and result row are separate: every sub-FFT reads the whole input), and the wave writes
// conj(Y) x synthesis window / n_fft into the [total_frames][win] buffer that k_ola gathers from.
// Replaces tf.signal.inverse_stft as called at kapre/time_frequency.py:307-314.
template <int R>
__global__ __launch_bounds__(R == 2 ? 256 : 128, 1) void k_irfft_big(const float2* __restrict__ spec, Geom g,
                                                                      const float* __restrict__ synth,
                                                                      const float2* __restrict__ tw2048,
                                                                      const float2* __restrict__ twbig,
                                                                      float* __restrict__ frames) {
    constexpr int NC = 1024, NB = R * NC, K = NB + 1, L = 64;
    constexpr int NW = (R == 2) ? 4 : 2;
    constexpr int RSF = NB + 1;
    typedef typename SwzFor<NC>::type SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fl = lane;
    f2* arow = reinterpret_cast<f2*>(smem) + (2 * wave) * RSF;       // X, then conj(2 Z) in natural order
    f2* zrow = arow + RSF;                                           // E_r[k] at r * 1024 + k, then Y
    FftTw<NC, SW> tw;
    tw.load(tw2048, fl);
    const int ostride = spec_stride(g);
    const float sc = 0.5f / (float)NB;                               // 1/2 of the pairing, 1/NB of the inverse DFT
#pragma unroll 1
    for (long long gf = (long long)blockIdx.x * NW + wave; gf < g.total_frames; gf += (long long)gridDim.x * NW) {
        FramePos p = frame_pos(g, gf);
        const float2* sp = spec + spec_base(g, p, gf, K);
        for (int k = lane; k < K; k += 64) { const float2 v = sp[(long long)k * ostride]; arow[k] = f2{v.x, v.y}; }
        KPR_LDS_FENCE_X();      // (kpr_fft.h) the pairing reads other lanes' words; a pair (k, NB - k) is then one lane's own
        // inverse pairing in place; irfft ignores the imaginary parts of DC and Nyquist
        for (int k = lane; 2 * k <= NB; k += 64) {
            f2 xk = arow[k], xq = arow[NB - k];
            if (k == 0) { xk.y = 0.0f; xq.y = 0.0f; }
            const float2 t2 = twbig[k];
            const f2 xp = f2{xq.x, -xq.y};                                  // conj X[NB-k]
            const f2 e = cadd(xk, xp), d = csub(xk, xp);
            const f2 od = cmul(d, f2{t2.x, -t2.y});                         // (X - conj X') conj(t)
            arow[k] = f2{e.x - od.y, -(e.y + od.x)};                        // conj(2 Z[k])
            if (k != 0 && 2 * k != NB) arow[NB - k] = f2{e.x + od.y, e.y - od.x};   // conj(2 Z[NB-k])
        }
        KPR_LDS_FENCE_X();
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            f2 z[kPts];
#pragma unroll
            for (int m = 0; m < kPts; ++m) z[m] = arow[r + R * (fl + L * m)];
            tw.refresh();
            cfft_forward<NC, SW>(z, tw, reinterpret_cast<float*>(zrow + NC * r));
#pragma unroll
            for (int m = 0; m < kPts; ++m) zrow[NC * r + fl + L * m] = z[m];
        }
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
        // x[2n] + i x[2n+1] = conj(Y[n]) / (2 NB), times the synthesis window; win > n_fft: zeros behind
        float* fo = frames + gf * (long long)g.win;
        for (int n = lane; n < NB; n += 64) {
            const f2 y = zrow[n];
            if (2 * n < g.win) fo[2 * n] = y.x * sc * synth[2 * n];
            if (2 * n + 1 < g.win) fo[2 * n + 1] = -y.y * sc * synth[2 * n + 1];
        }
        for (int n = 2 * NB + lane; n < g.win; n += 64) fo[n] = 0.0f;
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
            KPR_LDS_FENCE_R();      // (kpr_fft.h: X[k] and X[NC - k] were staged by other lanes of this wave)
            const float2* st2 = reinterpret_cast<const float2*>(stage);
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                const int k = fl + L * m;
                float2 a = st2[k], b = st2[NC - k];
                if (!valid) { a = make_float2(0.f, 0.f); b = a; }
                if (k == 0) { a.y = 0.0f; b.y = 0.0f; }   // irfft ignores Im of DC / Nyquist
                z[m] = irfft_pair_one<NC>(f2{a.x, a.y}, f2{b.x, b.y}, tw, m);
            }
            KPR_LDS_FENCE_X();
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
// VEC = samples per lane and group: 4 (ds_read_b128, 16-byte stores; hop, win multiples of 4) or 2
// (b64, 8-byte stores; hop, win even -- the default hop n_fft/4 of n_fft = 1000, 600, 360)
template <int RJ, int VEC = 4>
struct IwPass {
    static constexpr int IT = kIwReads / RJ;
    typedef float vt __attribute__((ext_vector_type(VEC)));
    vt v[IT][RJ];
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

template <int RJ, int VEC = 4>
KPR_DEV void iw_issue(IwPass<RJ, VEC>& s, const IwCtx& c, int cq, int qe, int lane,
                      const int (&qk)[IwPass<RJ, VEC>::IT], const int (&o4k)[IwPass<RJ, VEC>::IT], bool with_flag) {
    constexpr int IT = IwPass<RJ, VEC>::IT;
    typedef typename IwPass<RJ, VEC>::vt vt;
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
    const int n4 = (min(qe * c.hop, c.t_out) - cq * c.hop) / VEC;
    if (with_flag)
        s.full = c.regular && n4 == 64 * IT && cq - (RJ - 1) >= c.fa && qe - 1 <= c.f_last;
    if (s.full) {                                                      // wave-uniform
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int base = cq + qk[u] - c.fa;
#pragma unroll
            for (int jj = RJ - 1; jj >= 0; --jj)
                s.v[u][jj] = *reinterpret_cast<const vt*>(
                    c.smem + ((base - jj) & c.rmask) * c.RS + jj * c.hop + VEC * o4k[u]);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int fh = (lane + 64 * u < n4) ? cq + qk[u] : -(1 << 20);   // beyond the batch: no row matches
#pragma unroll
        for (int jj = RJ - 1; jj >= 0; --jj) {
            int addr;
            (void)iw_use<RJ>(c, fh, VEC * o4k[u], jj, addr);
            s.v[u][jj] = *reinterpret_cast<const vt*>(c.smem + addr);
        }
    }
}

template <int RJ, int VEC = 4>
KPR_DEV void iw_consume(IwPass<RJ, VEC>& s, const IwCtx& c, float* __restrict__ osig, int* emitted, int lane,
                        const int (&qk)[IwPass<RJ, VEC>::IT], const int (&o4k)[IwPass<RJ, VEC>::IT]) {
    constexpr int IT = IwPass<RJ, VEC>::IT;
    typedef typename IwPass<RJ, VEC>::vt vt;
    if (!__all(s.flag >= s.want)) {
        // the producers are behind: wait for the frames, then read the rows again
        const int* flag = &c.done[(s.want - 1) & c.rmask];
        int spin = 0;
        for (; spin < kIwSpinLimit; ++spin) {
            const bool ok = s.flag == 0x7fffffff ||
                __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= s.want;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (__builtin_expect(spin >= kIwSpinLimit, 0)) status_raise(kStIstftWsCons);
        iw_issue<RJ, VEC>(s, c, s.cq, s.qe, lane, qk, o4k, false);
    }
    const int n4 = (min(s.qe * c.hop, c.t_out) - s.cq * c.hop) / VEC;
    float* const op = osig + (long long)s.cq * c.hop;
    const vt zero = vt(0.0f);
    if (s.full) {
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            vt acc = zero;
#pragma unroll
            for (int jj = RJ - 1; jj >= 0; --jj) acc += s.v[u][jj];   // descending j = ascending frame
            *reinterpret_cast<vt*>(op + VEC * (lane + 64 * u)) = acc;
        }
    } else
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const bool here = lane + 64 * u < n4;
        const int fh = here ? s.cq + qk[u] : -(1 << 20);
        vt acc = zero;
#pragma unroll
        for (int jj = RJ - 1; jj >= 0; --jj) {          // descending j = ascending frame
            int addr;
            acc += iw_use<RJ>(c, fh, VEC * o4k[u], jj, addr) ? s.v[u][jj] : zero;
        }
        if (here) *reinterpret_cast<vt*>(op + VEC * (lane + 64 * u)) = acc;
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
                if (need > 0) {
                    int spin = 0;
                    for (; spin < kIwSpinLimit &&
                         __hip_atomic_load(&sync[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spin)
                        __builtin_amdgcn_s_sleep(2);
                    if (__builtin_expect(spin >= kIwSpinLimit, 0)) status_raise(kStIstftWsProd);
                }
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
template <class F, int RJ, int VEC>
__global__ __launch_bounds__(kIwThreads) void k_istft_ws_mr(const float2* __restrict__ spec,
                                                            IstftWsPlan pl,
                                                            const float* __restrict__ synth,
                                                            const float2* __restrict__ twtab,
                                                            float* __restrict__ out, int nitems) {
    constexpr int P = F::P, L = F::L, N = F::N, G = 64 / L, K = N + 1;
    constexpr int PIN = F::PIN, LIN = F::LIN;                  // lane l < LIN holds Z[l + LIN m], m < PIN
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
        const int li = min(l, LIN - 1);
        float2 xa[PIN], xb[PIN];
#define IW_TICKET(dst_)                                                                          \
    do {                                                                                         \
        int v_ = 0;                                                                              \
        if (lane == 0) v_ = __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        dst_ = __builtin_amdgcn_readfirstlane(v_);                                               \
    } while (0)
#define IW_LOAD(n_)                                                                              \
    do {                                                                                         \
        const int p_ = G * (n_) + grp;                                                           \
        const float2* sp_ = sp0 + (long long)(fa + (p_ < nframes ? p_ : 0)) * K + li;            \
        const float2* sq_ = sp_ + (N - 2 * li);   /* X[N - k]: a second base and immediate offsets (as N - 2 li - LIN m   */ \
        _Pragma("unroll") for (int m = 0; m < PIN; ++m) {      /* hipcc kept PIN 64-bit per-lane offsets in registers)     */ \
            xa[m] = sp_[LIN * m];                                                                \
            xb[m] = sq_[-LIN * m];                                                               \
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
                for (int m = 0; m < PIN; ++m) {
                    const int k = li + LIN * m;                                // < N
                    float2 a = xa[m], b = xb[m];
                    if (k == 0) { a.y = 0.0f; b.y = 0.0f; }                    // irfft ignores Im of DC / Nyquist
                    const f2 xk = f2{a.x, a.y}, xp = f2{b.x, -b.y};             // X[k], conj X[N-k]
                    const f2 e = cadd(xk, xp), d = csub(xk, xp);
                    const f2 tc = tab[k];
                    const f2 od = cmul(d, f2{tc.x, -tc.y});                    // (X - conj X') conj(t)
                    f2 zc = f2{e.x - od.y, -(e.y + od.x)};                     // conj(2 Z) = conj(E + i O)
                    if (!valid || l >= LIN) zc = f2{0.0f, 0.0f};
                    z[m] = zc;
                    // four points at a time: scheduled freely, hipcc issues all PIN twiddle reads and keeps every e / d / od live
                    // at once (240 live VGPRs here in the 20-point plans: the spills of VERDICT r03)
                    if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                const int need = fa + min(G * n + G - 1, nframes - 1) - pl.NR + pl.R - q0;
                if (need > 0) {
                    int spin = 0;
                    for (; spin < kIwSpinLimit &&
                         __hip_atomic_load(&sync[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spin)
                        __builtin_amdgcn_s_sleep(2);
                    if (__builtin_expect(spin >= kIwSpinLimit, 0)) status_raise(kStIstftWsProd);
                }
                float* row = smem + ((valid ? p : 0) & rmask) * pl.RS;
                F::run(z, l, valid, reinterpret_cast<f2*>(row), tab);           // Y = FFT_N(conj 2Z)
                // next ticket's rows: requested AFTER the FFT (round 4).  In flight during the FFT -- 4 PIN registers on top of
                // its working set -- every one of the 27 instances spilled 3 ... 44 VGPRs to scratch; they land
                // under the window / store pass and the other producers' FFTs instead.
                if (n2 < n_tickets) {
                    IW_LOAD(n2);
                } else {        // (defined on both paths: otherwise the 4 PIN registers count as live around the whole loop body)
#pragma unroll
                    for (int m = 0; m < PIN; ++m) xa[m] = xb[m] = make_float2(0.0f, 0.0f);
                }
                if (valid) {
#pragma unroll
                    for (int r = 0; r < P; ++r) {               // win is even: samples t, t+1 share the test
                        if (!F::holds(l, r)) continue;
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
        int qk[IwPass<RJ, VEC>::IT], o4k[IwPass<RJ, VEC>::IT];
        {
            const int nq4 = pl.hop / VEC;
#pragma unroll
            for (int u = 0; u < IwPass<RJ, VEC>::IT; ++u) {
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
            IwPass<RJ, VEC> pa, pb;
            iw_issue<RJ, VEC>(pa, c, q0, min(q0 + pl.QB, q1), lane, qk, o4k, true);
#pragma unroll 1
            for (;;) {
                const bool more_b = pa.qe < q1;
                if (more_b) iw_issue<RJ, VEC>(pb, c, pa.qe, min(pa.qe + pl.QB, q1), lane, qk, o4k, true);
                iw_consume<RJ, VEC>(pa, c, osig, &sync[1], lane, qk, o4k);
                if (!more_b) break;
                const bool more_a = pb.qe < q1;
                if (more_a) iw_issue<RJ, VEC>(pa, c, pb.qe, min(pb.qe + pl.QB, q1), lane, qk, o4k, true);
                iw_consume<RJ, VEC>(pb, c, osig, &sync[1], lane, qk, o4k);
                if (!more_a) break;
            }
            __syncthreads();
        }
    }
#undef IW_ITEM_PARAMS
#undef IW_ITEM_SYNC
}

// overlap-add as a gather: out[t] = sum_{f : f*hop <= t < f*hop + win} frames[f][t - f*hop]
template <class T>
__global__ void k_ola(const T* __restrict__ frames, long long n_sig, int F, int C, int win,
                      int hop, long long t_out, int out_cl, T* __restrict__ out) {
    const long long total = n_sig * t_out;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long bc = i / t_out;
        const long long t = i - bc * t_out;
        long long f_hi = t / hop;
        if (f_hi > F - 1) f_hi = F - 1;
        long long f_lo = (t - win + hop) / hop;      // ceil((t - win + 1) / hop) for t-win+1 > 0
        if (t - win + 1 <= 0) f_lo = 0;
        T acc = 0;
        for (long long f = f_lo; f <= f_hi; ++f)      // ascending frame order == tf overlap_and_add
            acc += frames[(bc * F + f) * win + (t - f * hop)];
        long long o;
        if (out_cl) { long long b = bc / C, c = bc - b * C; o = (b * t_out + t) * C + c; }
        else o = i;
        out[o] = acc;
    }
}

}  // namespace kpr
