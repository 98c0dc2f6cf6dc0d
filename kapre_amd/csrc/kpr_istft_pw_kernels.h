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
//   * run boundaries: the first R-1 blocks of a run lack the predecessor's frames, its last R-1 slot groups ("tail") lack
//     the successor's.  The successor leaves its first R-1 blocks as PARTIAL sums in an LDS stash (as many streams as the
//     160 KB hold: 27 of 31 at n_fft 1024 / hop 256; the others store them straight into the waveform) and raises an LDS
//     flag; the predecessor, when its run ends, reads them back, adds its tail -- (earlier frames) + (later frames),
//     deterministic, at most two roundings away from the sequential order -- and stores the final values.  The only
//     wait is bounded and its producer never waits for anyone.  (First version, every partial block through the
//     waveform: 1.41x the output bytes written, 1.11x the input read -- rocprofv3 -- at 5.2 TB/s of fabric traffic.)
//   * segment boundaries (between workgroups) recompute R-1 halo frames, as the ring kernel does.
#pragma once

namespace kpr {

struct IstftPwPlan {
    long long t_out;     // (F - 1) hop + win
    int F, win, hop;
    int segs;            // segments per signal; segment j = frames [j q + min(j, r), (j + 1) q + min(j + 1, r)), F = segs q + r
    int seg_q, seg_r;
    int nitems;          // signals x segs
    int n_stash;         // the first n_stash streams that have a predecessor keep their partial head blocks in LDS
    // interleaved layouts (IL instances; channels_last with C > 1 on either side): C a power of two <= the streams of a
    // workgroup, the streams of a workgroup = (frame run, channel) with the channel fastest -- neighbouring lane groups and
    // waves then read / write neighbouring bytes; an item is a segment of one BATCH ITEM (all its channels)
    int C, in_cl, out_cl;
};
constexpr int kIpwTwRegs = 10;           // FftTw<NC>::kNumTw <= 10
constexpr int kIpwSpinLimit = 1 << 22;   // every wait is bounded: a protocol error must end as a wrong result + KPR_E_DEVICE, not a hang

__host__ __device__ constexpr int ipw_row_words(int NC) { return NC >= 512 ? ((SwzSkew::row_words(NC) + 3) & ~3) : NC; }
__host__ __device__ inline size_t ipw_lds_bytes(int NC, int W) {           // without the stashes
    const int G = 64 / (NC / kPts);
    return sizeof(float) * ((size_t)W * G * ipw_row_words(NC) + 2 * (size_t)NC + 2 * 64 * (size_t)kIpwTwRegs) +
           sizeof(int) * ((size_t)W * G + 4);
}
__host__ __device__ constexpr size_t ipw_stash_bytes(int NC, int S) { return sizeof(float) * 2 * (size_t)(kPts - S) * (NC / kPts); }

template <int NC, int S, int W, bool IL = false>
__global__ __launch_bounds__(W * 64, 4) void k_istft_pw(const float2* __restrict__ spec, IstftPwPlan pl,
                                                        const float* __restrict__ synth,
                                                        const float2* __restrict__ twtab, float* __restrict__ out) {
    constexpr int L = NC / kPts, G = 64 / L, K = NC + 1, R = kPts / S, NSTR = W * G, RW = ipw_row_words(NC);
    constexpr int TAIL = kPts - S;      // slots of running sums that outlive the run: blocks rb .. rb + R - 2
    static_assert(S == 2 || S == 4 || S == 8, "hop = n_fft S / 16");
    typedef typename SwzFor<NC>::type SW;
    enum { FINAL = 0, PARTIAL = 1, DISCARD = 2, RMW = 3 };
    struct __attribute__((aligned(4))) float2u { float x, y; };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f2* winl = reinterpret_cast<f2*>(smem + W * G * RW);                  // (s w[2n], -s w[2n+1]), s = 1 / n_fft
    f2* twl = winl + NC;                                                  // [kNumTw][64]
    int* flags = reinterpret_cast<int*>(twl + 64 * kIpwTwRegs);           // [NSTR]: item + 1 once the stream's partial blocks of
                                                                          // that item are out (items ascend: never reset)
    f2* stash0 = twl + 64 * kIpwTwRegs + (W * G + 4) / 2;                 // [n_stash][TAIL][L] behind the flags

    // Everything a lane knows about its stream is RE-DERIVED from the lane id in every phase (a dozen integer
    // instructions): kept in registers it would sit next to the 32 running sums, the 64 prefetched spectrum values and
    // the FFT's own ~70 and push the kernel over the 128 of four waves per SIMD.  (asm volatile: not merged by hipcc.)
    auto lane_now = []() {
        int x;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
        return x;
    };
    struct Run { int fl, sid, rr, c, ra, rb; };                           // rr: index of the frame run, c: channel (IL)
    struct Item { int sig, fa, base, rem, nit; };                         // a segment of one signal / batch item (workgroup-uniform)
    const int t_out_ = (int)pl.t_out;
    const int CS = IL ? pl.C : 1;                                         // streams between a run and its successor
    const int RUNS = IL ? NSTR / pl.C : NSTR;                             // frame runs per item
    const int es_in = (IL && pl.in_cl) ? pl.C : 1, es_out = (IL && pl.out_cl) ? pl.C : 1;   // element strides
    auto item_of = [&](int item) {
        Item it;
        it.sig = item / pl.segs;
        const int seg = item - it.sig * pl.segs;
        const int f0 = seg * pl.seg_q + min(seg, pl.seg_r), f1 = (seg + 1) * pl.seg_q + min(seg + 1, pl.seg_r);
        it.fa = max(0, f0 - (R - 1));                                     // halo: R - 1 frames of the previous segment
        const int n = f1 - it.fa;
        it.base = n / RUNS;                                               // (the plan guarantees base >= R - 1)
        it.rem = n - it.base * RUNS;
        it.nit = it.base + (it.rem ? 1 : 0);                              // runs are aligned at their END
        return it;
    };
    auto run_of = [&](const Item& it, int lane_) {                        // this lane group's stream and its frames [ra, rb)
        Run r;
        r.fl = lane_ & (L - 1);
        r.sid = wave * G + ((G == 1) ? 0 : lane_ / L);
        r.c = IL ? (r.sid & (pl.C - 1)) : 0;
        r.rr = IL ? r.sid / pl.C : r.sid;
        r.ra = it.fa + r.rr * it.base + min(r.rr, it.rem);
        r.rb = r.ra + it.base + (r.rr < it.rem ? 1 : 0);
        return r;
    };
    // spectrum row f of the stream's signal (float2 units, element stride es_in) / its waveform (floats, stride es_out)
    auto in_row = [&](const Item& it, const Run& r, int f) -> const float2* {
        // (the workgroup-uniform part first: one scalar product, one per-lane multiply-add)
        if constexpr (!IL) return spec + (long long)it.sig * pl.F * K + (long long)f * K;
        else return pl.in_cl ? spec + (long long)it.sig * pl.F * K * pl.C + ((long long)f * K * pl.C + r.c)
                             : spec + (long long)it.sig * pl.C * pl.F * K + ((long long)r.c * pl.F + f) * K;
    };
    auto out_sig = [&](const Item& it, const Run& r) -> float* {
        if constexpr (!IL) return out + (long long)it.sig * pl.t_out;
        else return pl.out_cl ? out + (long long)it.sig * pl.t_out * pl.C + r.c
                              : out + ((long long)it.sig * pl.C + r.c) * pl.t_out;
    };
    // development knock-outs (tools/knockout_stft.sh): KPR_IPW_KO 1 = loads only (no transform, nothing
    // stored), 2 = loads + transform (nothing stored), 3 = stores only (no spectrum loads, no transform)
#ifdef KPR_IPW_KO
    constexpr int KO = KPR_IPW_KO;
#else
    constexpr int KO = 0;
#endif
    auto store2 = [&](float* o, int t, f2 v) {                            // samples t, t + 1 of the stream's waveform
        if constexpr (KO == 1 || KO == 2) { asm volatile("" :: "v"(v), "v"(o), "v"(t)); return; }
        if (IL && es_out != 1) {
            if (t < t_out_) o[(long long)t * es_out] = v.x;
            if (t + 1 < t_out_) o[(long long)(t + 1) * es_out] = v.y;
        } else {
            if (t + 1 < t_out_) *reinterpret_cast<float2u*>(o + t) = float2u{v.x, v.y};
            else if (t < t_out_) o[t] = v.x;
        }
    };
    float2 xa[kPts], xb[kPts];
    // development switch (tools/kbench_ab.sh, profiles/r06_istft_prefetch.md): KPR_IPW_PRE = n > 0 requests the first n of the next
    // frame's 16 row pairs right after the pairing pass -- when the registers of the current frame's rows are free but the
    // transform has not started -- into 4 n registers that stay live across the transform; the other 16 - n follow behind the sums
    // as before.  (A whole frame of early prefetch needs 168 registers: three waves per SIMD, measured slower in round 4.)
#ifdef KPR_IPW_PRE
    constexpr int PRE = KPR_IPW_PRE;
#else
    constexpr int PRE = 0;
#endif
    float2 pa[PRE > 0 ? PRE : 1], pb[PRE > 0 ? PRE : 1];
#define IPW_LOAD_RANGE(it_, r_, f_, M0_, M1_, DA_, DB_, OFF_)                                                 \
    do {                                                                                                      \
        const float2* sp_ = in_row((it_), (r_), min(max((f_), (r_).ra), pl.F - 1)) + (r_).fl * es_in;         \
        const float2* sq_ = sp_ + (NC - 2 * (r_).fl) * es_in;                                                 \
        _Pragma("unroll") for (int m = (M0_); m < (M1_); ++m) {                                               \
            DA_[m - (OFF_)] = sp_[(L * m) * es_in];                                                           \
            DB_[m - (OFF_)] = sq_[-(L * m) * es_in];                                                          \
        }                                                                                                     \
    } while (0)
#define IPW_LOAD(it_, r_, f_)                                                                                 \
    do {                                                                                                      \
        const float2* sp_ = in_row((it_), (r_), min(max((f_), (r_).ra), pl.F - 1)) + (r_).fl * es_in;         \
        const float2* sq_ = sp_ + (NC - 2 * (r_).fl) * es_in;   /* X[NC - k]: one more base, immediate offsets */ \
        _Pragma("unroll") for (int m = 0; m < kPts; ++m) {                                                    \
            if constexpr (KO == 3) { xa[m] = make_float2(1.0f, 0.5f); xb[m] = make_float2(0.25f, 2.0f); asm volatile("" :: "v"(sp_), "v"(sq_)); continue; } \
            xa[m] = sp_[(L * m) * es_in];   /* (plain loads: nontemporal ones cost 20 %, 85 vs 70.8 us on cfg4 -- the  */ \
            xb[m] = sq_[-(L * m) * es_in];  /*  256-byte pieces of a row straddle lines the next piece needs again)   */ \
        }                                                                                                     \
    } while (0)
    static_assert(FftTw<NC, SW>::kNumTw <= kIpwTwRegs, "LDS staging area of the twiddle set");
    if (wave == 0) {
        FftTw<NC, SW> t0;
        t0.load(twtab, tid & (L - 1));
        t0.for_each_tw([&](f2& v, int i) { twl[i * 64 + (tid & 63)] = v; });
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
    // the first rows of the first item are requested before the barrier (after the table loads: 64 registers): they travel
    // while the workgroup gathers
    // (n_fft 2048: the loads need two more 64-bit bases -- offsets beyond the immediate range -- and spilled them here)
    constexpr bool EARLY = NC <= 512;
    int item = blockIdx.x;
    // (an item's parameters -- a 32-bit division -- are computed before its rows are requested: at kernel start and at the END
    //  of the previous item, never with the 64 prefetch registers live)
    Item it = item_of(item < pl.nitems ? item : 0);
    if (EARLY && item < pl.nitems) {
        const Run r = run_of(it, lane_now());
        IPW_LOAD(it, r, r.rb - it.nit);
    }
    lds_barrier();

#pragma unroll 1
    while (item < pl.nitems) {
        const int nit = it.nit;
        // the run's first R - 1 blocks: complete at the start of a signal, the previous segment's at a halo, else partial
        auto head_kind_of = [&](const Run& r) { return (r.ra == 0) ? FINAL : (r.rr == 0 ? DISCARD : PARTIAL); };

        f2 acc[kPts];
#pragma unroll
        for (int m = 0; m < kPts; ++m) acc[m] = f2{0.0f, 0.0f};
        if (!EARLY || item != (int)blockIdx.x) {
            const Run r = run_of(it, lane_now());
            IPW_LOAD(it, r, r.rb - nit);
        }
#pragma unroll 1
        for (int i = 0; i < nit; ++i) {
            // the four waves of a SIMD take turns at the top priority (issue arbitration is oldest-first otherwise: cfg4
            // 71.5 us with the rotation, 75.7 without; starting the waves a quarter period apart instead: 73 ... 79)
            switch ((i + (wave >> 2)) & 3) {
                case 0:  __builtin_amdgcn_s_setprio(0); break;
                case 1:  __builtin_amdgcn_s_setprio(1); break;
                case 2:  __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
            // The twiddle set is read from LDS in every frame: 20 registers that are not live while the 64 of the spectrum
            // rows and the 32 running sums are.
            const int lane_o = lane_now();
            FftTw<NC, SW> tw;
            tw.pp = twl[(FftTw<NC, SW>::kNumTw - 1) * 64 + lane_o];
            f2 z[kPts];
            {
                const int fl = lane_o & (L - 1);
#pragma unroll
                for (int m = 0; m < kPts; ++m) {
                    float2 a = xa[m], bb = xb[m];
                    if (fl + L * m == 0) { a.y = 0.0f; bb.y = 0.0f; }     // irfft ignores Im of DC / Nyquist
                    z[m] = irfft_pair_one<NC>(f2{a.x, a.y}, f2{bb.x, bb.y}, tw, m);
                }
            }
            // (Round 4 had scheduling fences around the transform: without them the <512, 2> instance produced wrong values
            //  in the lanes fl mod 16 < 2.  Root cause, round 5: hipcc had hoisted the first LDS store of the exchange's second
            //  component above the last load of its first -- legal for every single lane, fatal across lanes.  The exchange now
            //  carries its own compiler-level ordering, kpr_fft.h KPR_LDS_FENCE_*; profiles/r05_hazard_rootcause.md.)
            if constexpr (PRE > 0) {
                // (the rows of frame i are in z: xa / xb are free.  Frame i + 1's first PRE row pairs, requested before the transform)
                if (i + 1 < nit) {
                    const Run rp_ = run_of(it, lane_o);
                    IPW_LOAD_RANGE(it, rp_, rp_.rb - nit + i + 1, 0, PRE, pa, pb, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < PRE; ++m) asm volatile("" : "=v"(pa[m].x), "=v"(pa[m].y), "=v"(pb[m].x), "=v"(pb[m].y));
                }
            }
            tw.for_each_tw([&](f2& v, int i) { v = twl[i * 64 + lane_o]; });
            tw.set_addresses(lane_o & (L - 1));
            if constexpr (KO != 1 && KO != 3) cfft_forward<NC, SW>(z, tw, smem + (wave * G + ((G == 1) ? 0 : lane_o / L)) * RW);
            const Run r = run_of(it, lane_now());
            const int f = r.rb - nit + i;
            const bool active = f >= r.ra;                                // (only i = 0 of the shorter runs is idle)
            const float on = active ? 1.0f : 0.0f;
#pragma unroll
            for (int m = 0; m < kPts; ++m) {
                const f2 y = pmul(z[m], winl[r.fl + L * m]);
                acc[m] = f2{fmaf(on, y.x, acc[m].x), fmaf(on, y.y, acc[m].y)};
            }
            // the loads stay below the FFT and below the sums (64 registers: nothing of the frame may be live next to them)
#pragma unroll
            for (int m = 0; m < kPts; ++m) asm volatile("" : "+v"(acc[m].x), "+v"(acc[m].y));
            asm volatile("" ::: "memory");
            if (i + 1 < nit) {
                if constexpr (PRE > 0) {
                    IPW_LOAD_RANGE(it, r, f + 1, PRE, kPts, xa, xb, 0);
#pragma unroll
                    for (int m = 0; m < PRE; ++m) { xa[m] = pa[m]; xb[m] = pb[m]; }
                } else {
                    IPW_LOAD(it, r, f + 1);                                    // next frame's rows: in flight under the stores
                }
            } else {
                // (defined on both paths -- by empty asm statements, no instructions: otherwise the 64 registers count as live
                //  around the whole loop body)
#pragma unroll
                for (int m = 0; m < kPts; ++m) asm volatile("" : "=v"(xa[m].x), "=v"(xa[m].y), "=v"(xb[m].x), "=v"(xb[m].y));
            }
            // block f is complete as far as this run goes
            const int j = f - r.ra;
            const int head_kind = head_kind_of(r);
            const int kind = (j < R - 1) ? head_kind : FINAL;
            if (active && kind == PARTIAL && r.sid - CS < pl.n_stash) {
                f2* st = stash0 + (r.sid - CS) * (TAIL * L) + r.fl + j * (S * L);
#pragma unroll
                for (int m = 0; m < S; ++m) st[m * L] = acc[m];
            } else if (active && kind != DISCARD) {
                float* osig = out_sig(it, r);
                const int t0 = f * pl.hop + 2 * r.fl;
#pragma unroll
                for (int m = 0; m < S; ++m) store2(osig, t0 + 2 * L * m, acc[m]);
            }
            if (__any(active && j == R - 2 && head_kind == PARTIAL)) {    // the partial blocks are out: tell the predecessor
                // (both parties are waves of this workgroup: global stores are acknowledged -- vmcnt(0) -- before the flag goes
                //  up, LDS executes a wave's operations in order; a system-scope fence here wrote the L2 back once per
                //  stream: 228 us instead of 72)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (active && j == R - 2 && head_kind == PARTIAL && r.fl == 0)
                    __hip_atomic_store(&flags[r.sid], item + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int m = 0; m < TAIL; ++m) acc[m] = acc[m + S];
#pragma unroll
            for (int m = TAIL; m < kPts; ++m) acc[m] = f2{0.0f, 0.0f};
        }
#undef IPW_LOAD
#undef IPW_LOAD_RANGE
        __builtin_amdgcn_s_setprio(0);
        // ---- the tail: slots 0 .. TAIL-1 = blocks rb .. rb + R - 2 without the successor's frames ----------------------
        const Run r = run_of(it, lane_now());
        // final at the end of the signal, recomputed by the next segment's halo, else completed from the successor's
        // partial blocks
        const int tail_kind = (r.rb == pl.F) ? FINAL : (r.rr == RUNS - 1 ? DISCARD : RMW);
        float* osig = out_sig(it, r);
        if (tail_kind == RMW) {
            int spin = 0;
            for (; spin < kIpwSpinLimit &&
                 __hip_atomic_load(&flags[r.sid + CS], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < item + 1; ++spin)
                __builtin_amdgcn_s_sleep(2);
            if (__builtin_expect(spin >= kIpwSpinLimit, 0)) status_raise(kStIstftPw);
            const int tb = r.rb * pl.hop + 2 * r.fl;                      // (rb < F: all of it inside the waveform)
            float* ob = osig + (long long)tb * es_out;
            float px[TAIL], py[TAIL];
            if (r.sid < pl.n_stash) {
                const f2* st = stash0 + r.sid * (TAIL * L) + r.fl;        // the successor's
#pragma unroll
                for (int m = 0; m < TAIL; ++m) {
                    const f2 v = st[m * L];
                    px[m] = v.x;
                    py[m] = v.y;
                }
            } else {
#pragma unroll
                for (int m = 0; m < TAIL; ++m) {                          // device-scope loads: served by the L2, not this CU's L1
                    px[m] = __hip_atomic_load(ob + (2 * L * m) * es_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    py[m] = __hip_atomic_load(ob + (2 * L * m + 1) * es_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int m = 0; m < TAIL; ++m) store2(osig, tb + 2 * L * m, f2{acc[m].x + px[m], acc[m].y + py[m]});
        } else if (tail_kind == FINAL) {
            const int t0 = r.rb * pl.hop + 2 * r.fl;
#pragma unroll
            for (int m = 0; m < TAIL; ++m) store2(osig, t0 + 2 * L * m, acc[m]);
        }
        // (nothing of the prefetch registers is carried into the next item)
#pragma unroll
        for (int m = 0; m < kPts; ++m) asm volatile("" : "=v"(xa[m].x), "=v"(xa[m].y), "=v"(xb[m].x), "=v"(xb[m].y));
        item += gridDim.x;
        if (item < pl.nitems) {
            it = item_of(item);
            // A workgroup that takes a second item: a stream that is ahead would write the new item's partial blocks into its
            // stash while its predecessor has not read the old item's yet (found by tools/fuzz_parity.py: 144 signals x 3
            // segments on 256 workgroups, relative error 0.74; single-item launches were all the tests had).  One barrier
            // per item; the flags stay monotonic.
            lds_barrier();
        }
    }
}

}  // namespace kpr
