// k_istft_pw (round 4): the fused inverse STFT with every wave a complete worker -- spectrum rows -> inverse pairing ->
// FFT -> synthesis window -> overlap-add -> waveform -- for hops that are a multiple of 2 L samples (L = n_fft / 32 lanes per
// frame): hop = n_fft / 4 (Kapre's default, S = 4 below), n_fft / 2 (S = 8), n_fft / 8 (S = 2).
//
// Why: the ring kernel k_istft_ws runs 7 producer waves + 1 consumer per CU at 256 registers -- fewer than two waves per
// SIMD, so its inverse FFTs run at little more than single-wave speed (cfg4: 77 us where the bytes take 52 at the copy rate
// of this part and the transforms ~31 at four waves per SIMD).  Sixteen waves of <= 128 registers hide each other's
// latencies the way k_stft3 / k_mel_pw do.  What made the ring necessary was the overlap-add: a sample is the sum of
// R = n_fft / hop frames, in ascending frame order (tf.signal.overlap_and_add; kapre/time_frequency.py:307-314).  Here the
// sum never leaves the lane:
//   * lane fl of a frame holds samples t = 2 (fl + L m), t + 1 in register slot m (the FFT's own layout), so with
//     hop = 2 L S the next frame's sample at the same time index sits in the SAME lane, S slots further down.  A STREAM =
//     one lane group (L lanes) walking a run of consecutive frames of one signal keeps 16 slots of running sums: add the
//     frame, the lowest S slots are a finished hop block (stored: L lanes x 8 bytes contiguous per slot), shift by S.
//     Interior samples are summed in exactly the reference's order.
//   * runs are static (W x G streams per workgroup split a SEGMENT of one signal evenly, +-1 frame); what a static split
//     loses to the SIMD's oldest-first issue arbitration (profiles/r04_mel_pw.md section 3) is taken back by rotating the
//     waves' priorities every frame (s_setprio).
//   * run boundaries: the first R-1 blocks of a run lack the predecessor's frames, its last R-1 slots-groups ("tail") lack
//     the successor's.  The successor stores its first R-1 blocks as PARTIAL sums straight into the waveform and raises
//     an LDS flag; the predecessor, when its run ends, reads them back (L2), adds its tail -- (earlier frames) + (later
//     frames), deterministic, at most two roundings away from the sequential order -- and stores the final values.  No
//     workspace, no extra HBM traffic; the only wait is bounded and its producer never waits for anyone.
//   * segment boundaries (between workgroups) recompute R-1 halo frames, as the ring kernel does.
#pragma once

namespace kpr {

struct IstftPwPlan {
    long long t_out;     // (F - 1) hop + win
    int F, win, hop;
    int segs;            // segments per signal; segment j = frames [j F / segs, (j + 1) F / segs)
    int nitems;          // signals x segs
};
constexpr int kIpwTwRegs = 10;           // FftTw<NC>::kNumTw <= 10
constexpr int kIpwSpinLimit = 1 << 22;   // every wait is bounded: a protocol error must end as a wrong result, not a hang

__host__ __device__ constexpr int ipw_row_words(int NC) { return NC >= 512 ? ((SwzSkew::row_words(NC) + 3) & ~3) : NC; }
__host__ __device__ inline size_t ipw_lds_bytes(int NC, int W) {
    const int G = 64 / (NC / kPts);
    return sizeof(float) * ((size_t)W * G * ipw_row_words(NC) + 2 * (size_t)NC + 2 * 64 * (size_t)kIpwTwRegs) +
           sizeof(int) * ((size_t)W * G + 4);
}

template <int NC, int S, int W>
__global__ __launch_bounds__(W * 64, W == 12 ? 3 : 4) void k_istft_pw(const float2* __restrict__ spec, IstftPwPlan pl,
                                                        const float* __restrict__ synth,
                                                        const float2* __restrict__ twtab, float* __restrict__ out) {
    constexpr int L = NC / kPts, G = 64 / L, K = NC + 1, R = kPts / S, NSTR = W * G, RW = ipw_row_words(NC);
    constexpr int TAIL = kPts - S;      // slots of running sums that outlive the run: blocks rb .. rb + R - 2
    static_assert(S == 2 || S == 4 || S == 8, "hop = n_fft S / 16");
    typedef typename SwzFor<NC>::type SW;
    enum { FINAL = 0, PARTIAL = 1, DISCARD = 2, RMW = 3 };
    struct __attribute__((aligned(4))) float2u { float x, y; };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & (L - 1), grp = (G == 1) ? 0 : lane / L;
    float* row = smem + (wave * G + grp) * RW;
    f2* winl = reinterpret_cast<f2*>(smem + W * G * RW);                  // (s w[2n], -s w[2n+1]), s = 1 / n_fft
    f2* twl = winl + NC;                                                  // [kNumTw][64]
    int* flags = reinterpret_cast<int*>(twl + 64 * kIpwTwRegs);           // [NSTR]: item + 1 once the stream's partial blocks of
                                                                          // that item are out (items ascend: never reset)

    static_assert(FftTw<NC, SW>::kNumTw <= kIpwTwRegs, "LDS staging area of the twiddle set");
    if (wave == 0) {
        FftTw<NC, SW> t0;
        t0.load(twtab, fl);
        t0.for_each_tw([&](f2& v, int i) { twl[i * 64 + lane] = v; });
    }
    {
        // irfft's 1 / n_fft and the conjugation after the forward FFT (IFFT(z) = conj(FFT(conj z))) folded into the window
        const float sc = 1.0f / (float)(2 * NC);
        for (int i = tid; i < NC; i += W * 64) {
            const int n = 2 * i;
            const float a = synth[min(n, pl.win - 1)], b = synth[min(n + 1, pl.win - 1)];
            winl[i] = f2{(n < pl.win) ? sc * a : 0.0f, (n + 1 < pl.win) ? -sc * b : 0.0f};
        }
    }
    if (tid < NSTR) flags[tid] = 0;
    lds_barrier();
    const int s_id = wave * G + grp;                                      // this lane group's stream

#pragma unroll 1
    for (int item = blockIdx.x; item < pl.nitems; item += gridDim.x) {
        const int sig = item / pl.segs, seg = item - sig * pl.segs;
        const int f0 = (int)((long long)seg * pl.F / pl.segs), f1 = (int)((long long)(seg + 1) * pl.F / pl.segs);
        const int fa = max(0, f0 - (R - 1));                              // halo: R - 1 frames of the previous segment
        const int n = f1 - fa, base = n / NSTR, rem = n - base * NSTR;    // (the plan guarantees base >= R - 1)
        const int ra = fa + s_id * base + min(s_id, rem), rb = ra + base + (s_id < rem ? 1 : 0);
        const int nit = base + (rem ? 1 : 0);                             // workgroup-uniform; runs are aligned at their END
        const float2* sp0 = spec + (long long)sig * pl.F * K;
        float* osig = out + (long long)sig * pl.t_out;
        const int t_out = (int)pl.t_out;
        // the run's first R - 1 blocks: complete at the start of a signal, the previous segment's at a halo, else partial
        const int head_kind = (ra == 0) ? FINAL : (s_id == 0 ? DISCARD : PARTIAL);
        // its tail: final at the end of the signal, recomputed by the next segment's halo, else completed from the
        // successor's partial blocks
        const int tail_kind = (rb == pl.F) ? FINAL : (s_id == NSTR - 1 ? DISCARD : RMW);

        f2 acc[kPts];
#pragma unroll
        for (int m = 0; m < kPts; ++m) acc[m] = f2{0.0f, 0.0f};
        float2 xa[kPts], xb[kPts];
#define IPW_LOAD(f_)                                                                                          \
    do {                                                                                                      \
        const float2* sp_ = sp0 + (long long)min(max((f_), ra), pl.F - 1) * K + fl;                           \
        const float2* sq_ = sp_ + (NC - 2 * fl);        /* X[NC - k]: one more base, immediate offsets */      \
        _Pragma("unroll") for (int m = 0; m < kPts; ++m) {                                                    \
            xa[m] = sp_[L * m];                                                                               \
            xb[m] = sq_[-L * m];                                                                              \
        }                                                                                                     \
    } while (0)
        IPW_LOAD(rb - nit);
#pragma unroll 1
        for (int i = 0; i < nit; ++i) {
            // the four waves of a SIMD take turns at the top priority (issue arbitration is oldest-first otherwise)
            switch ((i + (wave >> 2)) & 3) {
                case 0:  __builtin_amdgcn_s_setprio(0); break;
                case 1:  __builtin_amdgcn_s_setprio(1); break;
                case 2:  __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
            const int f = rb - nit + i;
            const bool active = f >= ra;                                  // (only i = 0 of the shorter runs is idle)
            // The twiddle set is read from LDS in every frame (an opaque copy of the lane id keeps hipcc from hoisting the
            // reads): 20 registers that are not live while the 64 of the spectrum rows and the 32 running sums are.
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            FftTw<NC, SW> tw;
            tw.pp = twl[(FftTw<NC, SW>::kNumTw - 1) * 64 + lane_o];
            f2 z[kPts];
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                float2 a = xa[m], bb = xb[m];
                if (fl + L * m == 0) { a.y = 0.0f; bb.y = 0.0f; }         // irfft ignores Im of DC / Nyquist
                z[m] = irfft_pair_one<NC>(f2{a.x, a.y}, f2{bb.x, bb.y}, tw, m);
            }
            __builtin_amdgcn_sched_barrier(0);
            tw.for_each_tw([&](f2& v, int i) { v = twl[i * 64 + lane_o]; });
            tw.set_addresses(lane_o & (L - 1));
            cfft_forward<NC, SW>(z, tw, row);
            __builtin_amdgcn_sched_barrier(0);
            const float on = active ? 1.0f : 0.0f;
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                const f2 y = pmul(z[m], winl[fl + L * m]);
                acc[m] = f2{fmaf(on, y.x, acc[m].x), fmaf(on, y.y, acc[m].y)};
            }
            // the loads stay below the FFT and below the sums (64 registers: nothing of the frame may be live next to them)
#pragma unroll
            for (int m = 0; m < kPts; ++m) asm volatile("" : "+v"(acc[m].x), "+v"(acc[m].y));
            asm volatile("" ::: "memory");
            if (i + 1 < nit) {
                IPW_LOAD(f + 1);                                          // next frame's rows: in flight under the stores
            } else {
                // (defined on both paths: otherwise the 64 registers count as live around the whole loop body)
#pragma unroll
                for (int m = 0; m < kPts; ++m) xa[m] = xb[m] = make_float2(0.0f, 0.0f);
            }
            // block f is complete as far as this run goes
            const int j = f - ra;
            const int kind = (j < R - 1) ? head_kind : FINAL;
            if (active && kind != DISCARD) {
                const int t0 = f * pl.hop + 2 * fl;
#pragma unroll
                for (int m = 0; m < S; ++m) {
                    const int t = t0 + 2 * L * m;
                    if (t + 1 < t_out) *reinterpret_cast<float2u*>(osig + t) = float2u{acc[m].x, acc[m].y};
                    else if (t < t_out) osig[t] = acc[m].x;
                }
            }
            if (__any(active && j == R - 2 && head_kind == PARTIAL)) {    // the partial blocks are out: tell the predecessor
                // (both parties are waves of this workgroup: the stores are acknowledged -- vmcnt(0) -- before the flag goes up;
                //  a system-scope fence here wrote the L2 back once per stream: 228 us instead of 60-odd)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (active && j == R - 2 && head_kind == PARTIAL && fl == 0)
                    __hip_atomic_store(&flags[s_id], item + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int m = 0; m < TAIL; ++m) acc[m] = acc[m + S];
#pragma unroll
            for (int m = TAIL; m < kPts; ++m) acc[m] = f2{0.0f, 0.0f};
        }
#undef IPW_LOAD
        __builtin_amdgcn_s_setprio(0);
        // ---- the tail: slots 0 .. TAIL-1 = blocks rb .. rb + R - 2 without the successor's frames ----------------------
        if (tail_kind == RMW) {
            for (int spin = 0; spin < kIpwSpinLimit &&
                 __hip_atomic_load(&flags[s_id + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < item + 1; ++spin)
                __builtin_amdgcn_s_sleep(2);
            float* ob = osig + (long long)rb * pl.hop + 2 * fl;           // (rb < F: all of it inside the waveform)
            float px[TAIL], py[TAIL];
#pragma unroll
            for (int m = 0; m < TAIL; ++m) {                              // device-scope loads: served by the L2, not this CU's L1
                px[m] = __hip_atomic_load(ob + 2 * L * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                py[m] = __hip_atomic_load(ob + 2 * L * m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int m = 0; m < TAIL; ++m)
                *reinterpret_cast<float2u*>(ob + 2 * L * m) = float2u{acc[m].x + px[m], acc[m].y + py[m]};
        } else if (tail_kind == FINAL) {
            const int t0 = rb * pl.hop + 2 * fl;
#pragma unroll
            for (int m = 0; m < TAIL; ++m) {
                const int t = t0 + 2 * L * m;
                if (t + 1 < t_out) *reinterpret_cast<float2u*>(osig + t) = float2u{acc[m].x, acc[m].y};
                else if (t < t_out) osig[t] = acc[m].x;
            }
        }
    }
}

}  // namespace kpr
