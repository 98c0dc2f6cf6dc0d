// kpr_mel_pw_kernels.h -- the per-wave fused mel-spectrogram kernel k_mel_pw (round 4).
// Part of the single translation unit kapre_hip.hip (included after kpr_mel_ts_kernels.h; not stand-alone).
//
// Why.  k_mel_ws (producer / consumer waves, 16-frame tiles behind tickets) and k_mel_ts (equal waves, two barriers per
// round) both put every wave of a CU into the same phase at the same time: knock-out builds (profiles/r03_fft_core.md
// section 4) measured FFT-only 28.0 us + everything-else 19.7 us ~= together 49.2 us on the north-star shape -- sample
// requests, filterbank fragments from L2, magnitude rows back from LDS, stores and the MFMA GEMM never ran under the
// other waves' FFTs, because there were no "other" waves in another phase.
//
// k_mel_pw has no shared phase.  A wave owns its frames END TO END:
//     samples (prefetched one frame ahead) -> window -> rFFT -> |X| into the wave's OWN LDS row
//     -> banded mel sums straight from that row -> [10 log10] -> one coalesced store per 64 filters.
// No consumer waves, no 16-frame tile, no tickets, no filterbank fragments, no MFMA, ONE barrier (after the
// prologue's table copy).  Sixteen independent waves per CU (<= 128 VGPRs) drift apart by themselves, so one wave's
// memory latency is covered by the other three waves of its SIMD, which is what the hardware scheduler is for.
//
// The mel product without a GEMM.  A mel (or any triangular) filterbank has at most two non-zeros per frequency bin,
// in neighbouring filters: bin k feeds filter a(k) with weight w0[k] and filter a(k)+1 with w1[k], a(k) non-decreasing
// (kpr_filterbank_pack checks exactly this; any other matrix keeps the MFMA kernels).  Bins with equal a(k) form a
// SEGMENT; with S0 / S1 the w0- / w1-weighted magnitude sums of a segment,
//     out[m] = S0[segment a = m] + S1[segment a = m - 1] + fb[Nyquist][m] |X[Nyquist]|.
// Stage 1: lane fl of a frame owns the 16 contiguous bins [16 fl, 16 fl + 16) (four ds_read_b128 from the row, whose
//   layout k + 4 (k >> 6) makes them conflict free), multiplies by its 32 weights (eight ds_read_b128 from a table
//   shared by the workgroup) with one v_pk_fma_f32 per bin into a running (S0, S1) pair, and wherever a segment or
//   the lane's range ends it appends the pair to a compact list in LDS -- the list reuses the start of the row, whose
//   magnitudes are in registers by then.  "Where" is static: 16 lane masks E_i (SGPR pairs, one per bin position) are
//   installed as EXEC around a ds_write_b64 + accumulator reset + pointer bump.
// Stage 2: lane fl finishes filters fl, fl + L, ... : it adds the <= 4 CMQ partial sums of its two segments in a FIXED
//   order (the table holds their LDS offsets; unused slots point at a zero word), so results are deterministic.
// ~2 K multiply-adds per frame instead of 38 K issued MFMA flops, and -- the point -- nothing to wait for.
//
// Same arithmetic as composed.py:138-261 (STFT -> Magnitude -> ApplyFilterbank [-> MagnitudeToDecibel]) in one launch;
// FFT building blocks of kpr_fft.h, frame fetch of kpr_common.h.
#pragma once

namespace kpr {

constexpr int kPwMaxRounds = 8;      // filters per lane: n_filt <= 8 L
constexpr int kPwMaxCmq = 4;         // partial sums per segment <= 4 * kPwMaxCmq (a segment may span 16 lanes = 256 bins)
constexpr int kPwEmaskWords = 32;    // 16 x 64-bit lane masks
constexpr int kPwTwRegs = 10;        // FftTw<NC>::kNumTw <= 10 for NC = 128 ... 1024 (staged through LDS once per workgroup)

// per-frame LDS row: FFT exchange row, then magnitudes at word k + 4 (k >> 6), then the partial-sum list (from word 0),
// and at the very end a pair of zero words nothing ever writes (target of unused stage-2 slots)
__host__ __device__ constexpr int pw_exch_words(int NC) {
    return NC == 1024 ? SwzWide::row_words(NC) : NC == 512 ? SwzSkew::row_words(NC) : NC;
}
__host__ __device__ constexpr int pw_mag_word(int k) { return k + 4 * (k >> 6); }
__host__ __device__ constexpr int pw_zero_word(int NC) {
    const int need = pw_exch_words(NC) > pw_mag_word(NC) + 1 ? pw_exch_words(NC) : pw_mag_word(NC) + 1;
    return (need + 3) & ~3;
}
__host__ __device__ constexpr int pw_row_words(int NC) { return pw_zero_word(NC) + 4; }

// band plan section of the packed filterbank (kpr_filterbank_pack writes it, see build_band_plan in kapre_hip.hip):
//   emask[16] (u64) | T1[8][L] float4 | P[L] u32 | WN[NR][L] float | T2[NR][CMQ][L] uint4
struct PwPlan {
    int L, NR, CMQ, nlist;
    int M;
    const unsigned* sec;               // device: the section (starts with the masks)
    // the blob's header as the device holds it NOW, and where the host's cached copy of it puts the section: k_mel_pw compares
    // hdr[6 .. 10] = (band_off, L, NR, CMQ, nlist) with this plan before it trusts a single table offset (a caller may have
    // written another packed filterbank to the same address without kpr_filterbank_forget: ADVICE r04)
    const unsigned* hdr;
    unsigned band_off;
    // PAIR form, channels_last output with C >= 4 (round 5): cl_slots > 0 = the M x C block of one (item, frame)
    // is collected in one of cl_slots LDS slots by the C / 2 waves that hold its channel pairs and written by the last of them as
    // ONE contiguous run of 4 M C bytes; cl_blk = floats per slot (M C rounded up to 4).  0 = 8-byte (c, c + 1) stores per filter
    int cl_slots, cl_blk;
};
constexpr int kPwSlotSpinLimit = 1 << 22;
__host__ __device__ inline int pw_table_words(int L, int NR, int CMQ) { return 32 * L + L + NR * L + 4 * NR * CMQ * L; }
// the part of the tables a workgroup keeps in LDS: P | WN | T2 (the 32 weights per lane, T1, are read from global memory
// -- the L1 -- once per frame: the LDS pipe is the busiest unit of this kernel, the vector-memory path the idlest)
__host__ __device__ inline int pw_lds_table_words(int L, int NR, int CMQ) { return L + NR * L + 4 * NR * CMQ * L; }
// (pair_hold: the PAIR form keeps the first channel's results of a pair, kPwMaxRounds x 64 floats per wave, until the second
//  channel's exist -- channels_last outputs are then written as 8-byte (c, c + 1) pieces)
// (cl_slots / cl_blk: the slot ring of the staged channels_last store shares that area: slots, then two ints per slot)
__host__ __device__ inline size_t pw_pair_area_words(int W, int cl_slots, int cl_blk) {
    const size_t hold = (size_t)W * 64 * kPwMaxRounds, ring = (size_t)cl_slots * cl_blk + 2 * (size_t)cl_slots;
    return hold > ring ? hold : ring;
}
__host__ __device__ inline size_t pw_lds_bytes(int NC, int W, int NR, int CMQ, bool pair_hold = false, int cl_slots = 0, int cl_blk = 0) {
    const int L = NC / kPts, G = 64 / L;
    return sizeof(float) * ((size_t)W * G * pw_row_words(NC) + (size_t)pw_lds_table_words(L, NR, CMQ) + 2 * (size_t)NC + 4 +
                            2 * 64 * (size_t)kPwTwRegs + (pair_hold ? pw_pair_area_words(W, cl_slots, cl_blk) : 0));
}

// The banded mel sums of ONE frame whose magnitudes sit in `row` (layout pw_mag_word): stage 1 + stage 2 of the header
// comment.  Called by all lanes of the wave with full EXEC; `sec` = the plan section in global memory (masks through the
// scalar cache), wq = the lane's 32 weights (pw_load_weights), `tab` = the workgroup's LDS copy of P | WN | T2.
// emit(r, value) receives filter fl + L r.
// CONTRACT: full EXEC on entry -- stage 1 installs its lane masks with s_mov_b64 exec and restores exec to -1, not to
// the incoming mask (k_mel_pw calls it outside every divergent region; a caller under a partial mask must save / restore it).
// Non-finite magnitudes: a bin whose two weights are both zero contributes 0 * |X| -- NaN for an Inf / NaN magnitude, as in the
// dense product of the reference (tensordot), but attributed to the segment the bin lies in, not to every filter of the row.
// Also the body of tools/probes/mel_epilogue.hip (cycles per frame of exactly this code on LDS-resident rows).
// the 32 weights of lane fl (T1), requested from global memory; a caller that also prefetches samples issues this FIRST:
// vector-memory loads complete in order, so whatever is requested before the weights is waited for with them
template <int NC>
KPR_DEV void pw_load_weights(const unsigned* __restrict__ sec, int fl, f4 (&wq)[8]) {
    constexpr int L = NC / kPts;
    const f4* t1 = reinterpret_cast<const f4*>(sec + kPwEmaskWords) + fl;
#pragma unroll
    for (int j = 0; j < 8; ++j) wq[j] = t1[j * L];
}
// EMIT_LDS: emit() itself stores to LDS (the PAIR form parks the first channel's results in the lane's own words): the load group
// of stage 2 is closed around every call, so that the ISA audit's rule "no LDS store inside a load group" stays exact.
// the sixteen lane masks of stage 1 (constant address space = scalar loads: two s_load_dwordx16)
typedef unsigned long long pw_u64x8 __attribute__((ext_vector_type(8)));
struct PwMasks { pw_u64x8 lo, hi; };
KPR_DEV PwMasks pw_load_masks(const unsigned* __restrict__ sec) {
    typedef const pw_u64x8 __attribute__((address_space(4))) * ConstU64x8;
    unsigned long long ema = (unsigned long long)sec;
    asm volatile("" : "+s"(ema));
    PwMasks em;
    em.lo = ((ConstU64x8)ema)[0];
    em.hi = ((ConstU64x8)ema)[1];
    return em;
}
// Stage 1 + stage 2 on magnitudes that are in REGISTERS: m0 .. m3 = the lane's 16 contiguous bins [16 fl, 16 fl + 16), magn =
// |X[Nyquist]|, ptr = LDS byte address of the lane's first list entry (row + P[fl]).  `row` only holds the partial-sum list and the
// zero words here.  Shared by k_mel_pw (through pw_band_sums, which reads the registers back from the wave's magnitude row) and by
// the stand-alone ApplyFilterbank kernel k_fb_pw (kpr_fb_pw_kernels.h: the bins come straight from global memory), so that both
// produce bit-identical mel rows from the same magnitudes.
// IL = 0: q[4] = the lane's 16 contiguous bins; IL = 1 / 2 (round 6, k_fb_pw on interleaved stereo rows): q[8] = the lane's 16 bins
// of BOTH channels, (bin, channel) interleaved as the (item, frame, bin, channel) layout has them -- the packed multiply-add
// broadcasts one half of a register pair anyway, so channel IL - 1 is picked by the same op_sel that picks the odd bin of a pair
// in the contiguous form: no de-interleaving moves.
// wq(j), j < 8: the lane's weight quads (w0, w1 of bins 2 j, 2 j + 1) -- an array element for the kernels that hold them in
// registers, an LDS read for the ST instances of k_fb_pw, which have no 32 registers to spare (WLATE: each pair of quads is
// fetched when its four bins are due, a scheduling barrier keeps hipcc from hoisting all eight reads to the top)
template <int NC, bool EMIT_LDS, bool GATHER32, int IL, int NQ, bool WLATE = false, class WQ, class Emit>
KPR_DEV void pw_band_core_w(float* row, int fl, const PwMasks& em, WQ&& wq, const float* tab, int NR, int CMQ,
                            const f4 (&mm)[NQ], float magn, unsigned ptr, Emit&& emit) {
    static_assert((IL == 0 && NQ == 4) || (IL > 0 && NQ == 8), "16 bins per lane: four quads, or eight with two channels interleaved");
    constexpr int L = NC / kPts;
    // ---- stage 1: this lane's 16 bins -> (S0, S1) partial sums, appended to the list at the start of the row (LDS executes
    // a wave's operations in order: every read above is issued before the first list write -- the fence keeps hipcc's reads
    // on their side of the asm stores)
    f2 acc = f2{0.0f, 0.0f};
    auto step = [&](f2 mpair, int hi_half, f2 wpair, int i) {
        if (hi_half) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(mpair), "v"(wpair));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(mpair), "v"(wpair));
        const unsigned long long e = i < 8 ? em.lo[i & 7] : em.hi[i & 7];
#ifdef KPR_PW_KO_APPEND         /* development knock-out (counters only, wrong sums): no list appends (tools/lds_conflicts.sh) */
        if (false) {
#else
        if (e != 0ull) {                                                  // wave-uniform
#endif
#ifdef KPR_PW_KO_EXEC           /* development knock-out (timing only, wrong sums): the append without its EXEC switches */
            asm volatile("ds_write_b64 %1, %0\n\t"
                         "v_mov_b64 %0, 0\n\t"
                         "v_add_u32 %1, 8, %1"
                         : "+v"(acc), "+v"(ptr) : "s"(e) : "memory");
#else
            asm volatile("s_mov_b64 exec, %2\n\t"
                         "ds_write_b64 %1, %0\n\t"
                         "v_mov_b64 %0, 0\n\t"
                         "v_add_u32 %1, 8, %1\n\t"
                         "s_mov_b64 exec, -1"
                         : "+v"(acc), "+v"(ptr) : "s"(e) : "memory");
#endif
        }
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f4 wa = wq(2 * c), wb = wq(2 * c + 1);
        if constexpr (IL == 0) {
            step(quad_pair<0>(mm[c]), 0, quad_pair<0>(wa), 4 * c);
            step(quad_pair<0>(mm[c]), 1, quad_pair<2>(wa), 4 * c + 1);
            step(quad_pair<2>(mm[c]), 0, quad_pair<0>(wb), 4 * c + 2);
            step(quad_pair<2>(mm[c]), 1, quad_pair<2>(wb), 4 * c + 3);
        } else {                                                          // bin i: quad i / 2, pair i % 2; the half = the channel
            step(quad_pair<0>(mm[2 * c]), IL - 1, quad_pair<0>(wa), 4 * c);
            step(quad_pair<2>(mm[2 * c]), IL - 1, quad_pair<2>(wa), 4 * c + 1);
            step(quad_pair<0>(mm[2 * c + 1]), IL - 1, quad_pair<0>(wb), 4 * c + 2);
            step(quad_pair<2>(mm[2 * c + 1]), IL - 1, quad_pair<2>(wb), 4 * c + 3);
        }
        if constexpr (WLATE) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- stage 2: filters fl + L r: the partial sums of segment a = m (S0 halves) and a = m - 1 (S1 halves), fixed order.
    // Two forms, the same values added in the same order:
    //   GATHER32 (k_mel_pw, rounds 4-6): every lane gathers the S0 words of segment m and the S1 words of segment m - 1 with two
    //     ds_read_b32 per list entry (lane stride two words: each a two-way bank conflict);
    //   else (k_fb_pw, round 6): the lane of filter m reads the (S0, S1) PAIRS of its own segment -- one conflict-free
    //     ds_read_b64 per entry -- sums both halves, and takes the S1 sum of segment m - 1 from the neighbouring lane (the
    //     previous round's last lane for fl = 0) with ONE ds_bpermute per round.  Half the LDS instructions, but the bpermute
    //     is a dependent LDS round trip per round in a wave's serial chain: same-box A/B (profiles/r06_fb_pw.md) k_fb_pw
    //     19.9 -> 19.2 us, k_mel_pw<512> (three rounds, cfg5) 209 -> 213.5 us, the headline within its +-1.5 us of noise --
    //     so the fused kernel keeps the first form.
    KPR_LDS_FENCE_R();                                                    // (the list entries of other lanes)
    const float* wn = tab + L + fl;
    const uint4* t2 = reinterpret_cast<const uint4*>(tab + (1 + NR) * L) + fl;
    const char* rowc = reinterpret_cast<const char*>(row);
    float t_prev = 0.0f;                                                  // this lane's S1 sum of the round before
#ifdef KPR_PW_KO_STAGE2         /* development knock-out (counters only, wrong sums): no gathers (tools/lds_conflicts.sh) */
    for (int r = 0; r < NR; ++r) emit(r, fmaf(wn[r * L], magn, acc.x));
    NR = 0;
#endif
    for (int r = 0; r < NR; ++r) {                                        // wave-uniform trip count
        float u = 0.0f, t = 0.0f;
        if constexpr (GATHER32) {
            // (also: the PAIR instance of n_fft 2048 sits at its 168 registers and spilled four with the other form)
            for (int q = 0; q < CMQ; ++q) {
                const uint4 o = t2[(r * CMQ + q) * L];
                const unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u += *reinterpret_cast<const float*>(rowc + (ow[e] & 0xffffu));
                    t += *reinterpret_cast<const float*>(rowc + (ow[e] >> 16));
                }
            }
            const float wr_ = wn[r * L];
            if constexpr (EMIT_LDS) KPR_LDS_FENCE_X();
            emit(r, fmaf(wr_, magn, u + t));
            if constexpr (EMIT_LDS) KPR_LDS_FENCE_R();
            continue;
        }
        for (int q = 0; q < CMQ; ++q) {
            const uint4 o = t2[(r * CMQ + q) * L];
            const unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f2 pr = *reinterpret_cast<const f2a*>(rowc + (ow[e] & 0xffffu));
                u += pr.x;
                t += pr.y;
            }
        }
        // filter m = fl + L r takes S1 of segment m - 1: lane fl - 1 of this round, lane L - 1 of the round before for fl = 0
        // (0 for filter 0: t_prev starts at 0)
        // (the source lane is re-derived per round -- three instructions -- instead of held across the sums: the PAIR instance
        //  of n_fft 2048 has no register to spare)
        int src4;                                                         // 4 x the lane whose S1 sum this lane's filter takes
        if constexpr (L == 64) src4 = 4 * ((fl + 63) & 63);
        else {
            int lane_s;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
            src4 = 4 * ((lane_s & ~(L - 1)) | ((fl + (L - 1)) & (L - 1)));
        }
        const float give = (fl == L - 1) ? t_prev : t;
        const float d = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, give)));
        t_prev = t;
        const float wr_ = wn[r * L];
        if constexpr (EMIT_LDS) KPR_LDS_FENCE_X();
        emit(r, fmaf(wr_, magn, u + d));
        if constexpr (EMIT_LDS) KPR_LDS_FENCE_R();
    }
    KPR_LDS_FENCE_X();
}
template <int NC, bool EMIT_LDS, bool GATHER32, int IL, int NQ, class Emit>
KPR_DEV void pw_band_core_q(float* row, int fl, const PwMasks& em, const f4 (&wq)[8], const float* tab, int NR, int CMQ,
                            const f4 (&mm)[NQ], float magn, unsigned ptr, Emit&& emit) {
    pw_band_core_w<NC, EMIT_LDS, GATHER32, IL, NQ>(row, fl, em, [&](int j) { return wq[j]; }, tab, NR, CMQ, mm, magn, ptr, emit);
}
template <int NC, bool EMIT_LDS = false, bool GATHER32 = false, class Emit>
KPR_DEV void pw_band_core(float* row, int fl, const PwMasks& em, const f4 (&wq)[8], const float* tab, int NR, int CMQ,
                          f4 m0, f4 m1, f4 m2, f4 m3, float magn, unsigned ptr, Emit&& emit) {
    const f4 mm[4] = {m0, m1, m2, m3};
    pw_band_core_q<NC, EMIT_LDS, GATHER32, 0, 4>(row, fl, em, wq, tab, NR, CMQ, mm, magn, ptr, emit);
}
template <int NC, bool EMIT_LDS = false, bool GATHER32 = false, class Emit>
KPR_DEV void pw_band_sums(float* row, int fl, const unsigned* __restrict__ sec, const f4 (&wq)[8], const float* tab, int NR, int CMQ,
                          Emit&& emit) {
    // all sixteen masks are requested at once, ahead of the LDS reads they share a counter with, and re-read per frame
    // (32 SGPRs held across the FFT otherwise)
    const PwMasks em = pw_load_masks(sec);
    const unsigned rowb = (unsigned)(size_t)row;                          // LDS byte address of the row
    // (the magnitudes were written by other lanes of this wave: kpr_fft.h, lds_wave_fence)
    KPR_LDS_FENCE_R();
    const f4a* mq = reinterpret_cast<const f4a*>(row + 16 * fl + 4 * (fl >> 2));
    const f4 m0 = mq[0], m1 = mq[1], m2 = mq[2], m3 = mq[3];
    const float magn = row[pw_mag_word(NC)];                              // |X[Nyquist]| (one address: a broadcast)
    const unsigned ptr = rowb + reinterpret_cast<const unsigned*>(tab)[fl];
    KPR_LDS_FENCE_X();
    pw_band_core<NC, EMIT_LDS, GATHER32>(row, fl, em, wq, tab, NR, CMQ, m0, m1, m2, m3, magn, ptr, emit);
}

// W = waves per workgroup (any of them is a complete worker; W only sets how many share one copy of the tables)
//
// PAIR (round 4, interleaved waveforms -- channels_last with an even channel count): a ticket is G channel PAIRS and the wave
// computes the two frames of a pair one after the other from ONE fetch: (x[t][c], x[t][c+1]) are 8 contiguous bytes, so a
// pair costs 32 dwordx2 loads where two single frames cost 64 dword loads over the same cache lines.  Every wave of the
// plain kernel pulls all lines of its frame's time span through the CU's L1 to use 4 bytes of every 4 C: at C = 6 that is
// 48 KB per frame against 64 B per clock from the L2 -- 770 cycles per frame and CU next to 750 cycles of arithmetic
// (cfg3: 184 us channels_last vs 122 us channels_first, and splitting the loads so that every line is touched once per
// wave -- lower half of the lanes the even, upper half the odd samples, one v_permlane32_swap per point at use -- moved
// nothing, 186 vs 189 us: it is the fill traffic, not the tag lookups).  The second channel's samples wait in
// 32 registers while the first is transformed, which does not fit under the 128 of four waves per SIMD: PAIR runs three
// waves per SIMD (W = 12, 168 registers).
template <int NC, int W, bool PAIR = false>
__global__ __launch_bounds__(W * 64, PAIR ? 3 : 4) void k_mel_pw(const float* __restrict__ x, Geom g,
                                                      const float* __restrict__ window,
                                                      const float2* __restrict__ twtab, PwPlan pl, DbDev db,
                                                      unsigned* __restrict__ item_stats, float* __restrict__ out,
                                                      int run_q, int run_r, long long* __restrict__ dbg) {
    // (run_q, run_r count UNITS of tickets: 1, or C / 2 with the staged channels_last store -- a workgroup then owns whole
    //  (item, frame) blocks)
    constexpr int L = NC / kPts;       // lanes per frame
    constexpr int G = 64 / L;          // frames per wave and ticket
    constexpr int THREADS = W * 64;
    constexpr int RWD = pw_row_words(NC);
    typedef typename WsSwzFor<NC>::type WsSwz;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef KPR_DEV_STAMPS    /* development: s_memtime stamps of the workgroup dbg[16 * 32] names (tools/stamps_pw.py) */
    int dbi = 0;
    const bool stamp_me = dbg && (long long)blockIdx.x == dbg[16 * 32];
#define PW_STAMP() do { if (stamp_me && lane0 == 0 && dbi < 32) dbg[wave * 32 + dbi++] = (long long)__builtin_readcyclecounter(); } while (0)
    const unsigned long long wg_r0 = __builtin_amdgcn_s_memrealtime(), wg_c0 = __builtin_readcyclecounter();
#else
#define PW_STAMP() do { (void)dbg; } while (0)
#endif
    PW_STAMP();
    // the header words of the blob, requested first (scalar loads; compared before the first table access, below)
    typedef const unsigned __attribute__((address_space(4)))* ConstU32;
    unsigned long long hdra = (unsigned long long)pl.hdr;
    asm volatile("" : "+s"(hdra));
    const unsigned h6 = ((ConstU32)hdra)[6], h7 = ((ConstU32)hdra)[7], h8 = ((ConstU32)hdra)[8], h9 = ((ConstU32)hdra)[9],
                   h10 = ((ConstU32)hdra)[10];

    float* rows = smem;                                                   // [W * G][RWD]
    float* tab = smem + W * G * RWD;                                      // P | WN | T2 (as in the section, after T1)
    f2* winl = reinterpret_cast<f2*>(tab + pw_lds_table_words(L, pl.NR, pl.CMQ));   // (0.5 w[2n], 0.5 w[2n+1])
    int* ctr = reinterpret_cast<int*>(winl + NC);                         // the workgroup's ticket counter

    // ---- work split.  A ticket = G consecutive frames.  The WORKGROUP owns a contiguous run of tickets (even split over
    // the grid); inside it the waves draw tickets from an LDS counter.  A static split per wave does not work here: the
    // SIMD's issue arbitration is oldest-first, so of the four waves of a SIMD the oldest runs at nearly single-wave speed
    // and the youngest at a third of it -- in-kernel stamps (tools/stamps_pw.py) showed the waves of one SIMD finishing equal
    // shares at 59 k, 70 k, 83 k and 104 k cycles, the last 20 k with one wave left.  With tickets the old waves simply
    // take more frames.  A wave draws its next ticket only after its FFT, when it requests that ticket's samples (they land
    // under the band sums and the stores): drawn at the start of the frame, a young wave's ticket waited 20 k cycles for
    // its owner while older waves had run out of work (stamps: ends spread over 17 k cycles; now one frame's sums).
    // (run_q, run_r = tickets / grid, tickets % grid from the host: a 64-bit division is ~150 instructions per wave)
    const int bx = (int)blockIdx.x;
    // the staged store exists in the one-pair-per-ticket instance (n_fft 2048) only: with two pairs per wave the block index is
    // per lane group and the <512> instance, at its 168 registers, spilled for it
    constexpr bool STAGE = PAIR && G == 1;
    const int run_unit = (STAGE && pl.cl_slots > 0) ? (g.C >> 1) : 1;
    const int t_wg0 = (run_q * bx + min(bx, run_r)) * run_unit;
    const int n_wg = (run_q + (bx < run_r ? 1 : 0)) * run_unit;

    // ---- prologue: everything is REQUESTED before anything is used (one cold memory latency, not four in a row) ------
    auto fetch_ticket = [&](int tk, int lane_, f2 (&dst)[kPts]) -> bool {   // tk = ticket of this workgroup, wave-uniform
        bool sw = false;
        if (tk < n_wg) {
            const int fl_ = lane_ & (L - 1), grp_ = (G == 1) ? 0 : lane_ / L;
            const long long gf = (long long)(t_wg0 + tk) * G;
            const bool v = gf + grp_ < g.total_frames;
            FramePos p = frame_pos32(g, (unsigned)(v ? gf + grp_ : gf));     // (the launcher keeps total_frames below 2^31)
            if constexpr (L == 16 || L == 32) fetch_frame_z<NC>(x, g, p, v, fl_, dst, lane_, &sw);
            else fetch_frame_z<NC>(x, g, p, v, fl_, dst);
        } else {
#pragma unroll
            for (int m = 0; m < kPts; ++m) dst[m] = f2{0.0f, 0.0f};
        }
        return sw;
    };
    // PAIR: ticket tk = G pairs; pair q = frames 2 q, 2 q + 1 (channel-fastest numbering, even C: same item, same frame index,
    // channels c, c + 1).  Raw form: ra[m] = (x[2n][c], x[2n][c+1]), rb[m] = (x[2n+1][c], x[2n+1][c+1]), n = fl + L m.
    auto fetch_pair = [&](int tk, int lane_, f2 (&ra)[kPts], f2 (&rb)[kPts]) {
        if (tk < n_wg) {
            const int fl_ = lane_ & (L - 1), grp_ = (G == 1) ? 0 : lane_ / L;
            const long long q = (long long)(t_wg0 + tk) * G;
            const bool v = 2 * (q + grp_) < g.total_frames;
            FramePos p = frame_pos32(g, (unsigned)(2 * (v ? q + grp_ : q)));
            const bool inside = v && p.s0 >= 0 && (p.s0 + 2 * NC) <= g.T;
            if (__all(inside)) {
                struct __attribute__((aligned(4))) float2u { float x, y; };
                const int es = p.es;
                const float* fp = x + p.sig_off + (p.s0 + 2 * fl_) * es;
#pragma unroll
                for (int m = 0; m < kPts; ++m) {
                    const float2u a = *reinterpret_cast<const float2u*>(fp + (2 * L * m) * es);
                    const float2u b = *reinterpret_cast<const float2u*>(fp + (2 * L * m + 1) * es);
                    ra[m] = f2{a.x, a.y};
                    rb[m] = f2{b.x, b.y};
                    if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (addresses four points at a time)
                }
            } else {                                                      // signal edges: frame by frame (waited for there)
                FramePos p1 = p;
                p1.sig_off += 1; p1.c += 1; p1.bc += 1;
                fetch_frame_z<NC>(x, g, p, v, fl_, ra);
                fetch_frame_z<NC>(x, g, p1, v, fl_, rb);
#pragma unroll
                for (int m = 0; m < kPts; ++m) {                          // (x0, x1), (y0, y1) -> the raw form
                    const float t = ra[m].y;
                    ra[m].y = rb[m].x;
                    rb[m].x = t;
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < kPts; ++m) ra[m] = rb[m] = f2{0.0f, 0.0f};
        }
    };
    f2 nz[kPts];
    f2 nz2[PAIR ? kPts : 1];                                              // PAIR: the pair's second frame
    int cur = wave;                                                       // the first ticket of every wave is static
    bool nsw = false;
    if constexpr (PAIR) fetch_pair(cur, lane0, nz, nz2);
    else nsw = fetch_ticket(cur, lane0, nz);
    PW_STAMP();
    constexpr int WPT = (NC + THREADS - 1) / THREADS;
    float wa[WPT], wb[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int n = 2 * min(tid + u * THREADS, NC - 1);
        wa[u] = window[min(n, g.win - 1)];
        wb[u] = window[min(n + 1, g.win - 1)];
    }
    // The twiddle set of a lane is the same in every wave: ONE wave gathers it (ten vector-memory instructions) and hands
    // it over through LDS.  At kernel start the CU's address unit is the bottleneck -- stamps: sixteen waves x (16 sample
    // + 10 twiddle + 2 window loads) took 11 k cycles to ISSUE, the youngest wave got its first samples requested last.
    static_assert(FftTw<NC, WsSwz>::kNumTw <= kPwTwRegs, "LDS staging area of the twiddle set");
    f2* twl = reinterpret_cast<f2*>(ctr + 4);                             // [kNumTw][64]
    float* hold = reinterpret_cast<float*>(twl + 64 * kPwTwRegs) + wave * (64 * kPwMaxRounds);   // PAIR: [kPwMaxRounds][64]
    // PAIR, staged channels_last store: the same area as a ring of cl_slots blocks of cl_blk floats, then per slot the number of
    // channel pairs that have arrived and the number of blocks the slot has seen off (a block's generation)
    // (all three re-derived where they are used: scalar registers are what this instance is short of)
    auto slots_at = [&]() { return reinterpret_cast<float*>(twl + 64 * kPwTwRegs); };
    auto slot_cnt_at = [&]() { return reinterpret_cast<int*>(slots_at() + pl.cl_slots * pl.cl_blk); };
    auto slot_done_at = [&]() { return slot_cnt_at() + pl.cl_slots; };
    if constexpr (STAGE) {
        if (tid < 2 * pl.cl_slots) slot_cnt_at()[tid] = 0;
    }
    if (wave == 0) {
        FftTw<NC, WsSwz> t0;
        t0.load(twtab, lane0 & (L - 1));
        t0.for_each_tw([&](f2& v, int i) { twl[i * 64 + lane0] = v; });
    }
    PW_STAMP();
    // (compared BEFORE the first access through pl.sec -- ADVICE r05: a replaced blob of another layout must not be read at the
    //  old offsets; the scalar loads were requested at kernel entry and have arrived while the samples were being requested)
    if (h6 != pl.band_off || h7 != (unsigned)pl.L || h8 != (unsigned)pl.NR || h9 != (unsigned)pl.CMQ || h10 != (unsigned)pl.nlist) {
        // not the plan this launch was sized for (workgroup-uniform): nothing is computed, the next API call fails (KPR_E_DEVICE)
        if (tid == 0) status_raise(kStStalePlan);
        return;
    }
    {
        const int nt = pw_lds_table_words(L, pl.NR, pl.CMQ);              // multiple of 4
        const uint4* src = reinterpret_cast<const uint4*>(pl.sec + kPwEmaskWords + 32 * L);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < nt / 4; i += THREADS) dst[i] = src[i];
    }
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + u * THREADS, n = 2 * i;
        if (i < NC) winl[i] = f2{(n < g.win) ? 0.5f * wa[u] : 0.0f, (n + 1 < g.win) ? 0.5f * wb[u] : 0.0f};
    }
    if (lane0 < 4 * G) rows[(wave * G + (lane0 >> 2)) * RWD + pw_zero_word(NC) + (lane0 & 3)] = 0.0f;   // the zero words
    if (tid == 0) *ctr = W;
    PW_STAMP();
    lds_barrier();
    FftTw<NC, WsSwz> tw;
    tw.for_each_tw([&](f2& v, int i) { v = twl[i * 64 + lane0]; });
    tw.set_addresses(lane0 & (L - 1));
    PW_STAMP();

    DbRun dbrun;                                                         // running per-item extrema of this wave's lanes (dB)
    dbrun.reset();
    const int ostride = spec_stride(g);

    // The end of a short run.  The SIMD issues oldest-first: when the tickets run out its four waves finish one after the other
    // and the last one runs alone at ~43 % of the vector ALU (stamps: 5.6 k, 1.4 k, 4.6 k cycles apart).  A wave whose draw
    // finds no ticket has only this frame's sums left and steps back (priority 0 against 1): the waves with a whole frame to
    // go get the issue slots, all four end closer together.  Same-box A/B (profiles/r06_mel_tail.md): headline 37.8 -> 37.4 us,
    // cfg2 13.7 -> 13.3; runs of many tickets per wave and the instances with several frames per wave do not gain (cfg5
    // + 0.5 %, n_fft 512 + 1 %): they keep the hardware's order.  (Priorities by stage in every frame: 1.5 ... 5 % slower.)
    const bool tail_prio = G == 1 && n_wg < 8 * W;
    if (tail_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    while (cur < n_wg) {
      if constexpr (PAIR) {                                               // raw form -> (channel c frame, channel c + 1 frame)
#pragma unroll
          for (int m = 0; m < kPts; ++m) {
              const float t = nz[m].y;
              nz[m].y = nz2[m].x;
              nz2[m].x = t;
          }
      }
      int nxt = 0;
      // (both frames of a pair are spelled out -- 31 KB of code instead of 17: as a loop, the second frame's registers would
      //  count as live around the whole body, 15 registers more than there are)
#pragma unroll
      for (int sub = 0; sub < (PAIR ? 2 : 1); ++sub) {
        // per-lane quantities are re-derived from an opaque copy of the lane id in every phase: hoisted out of the
        // frame loop they would all stay live across the FFT (the kernel has 128 VGPRs)
        int lane_f = lane0;
        asm volatile("" : "+v"(lane_f));
        const int lane = lane_f, fl = lane & (L - 1), grp = (G == 1) ? 0 : lane / L;
        float* row = rows + (wave * G + grp) * RWD;
        f4 wq[8];
        {
            // ---- samples -> window -> rFFT -> |X| row ---------------------------------------------------------------
            f2 z[kPts];
            if constexpr (L == 16 || L == 32) {
                if (nsw) stereo_unswap<NC>(nz);                           // wave-uniform
            }
            // (the window from global memory -- the L1 -- instead of LDS, 32 LDS cycles per frame less, was built and measured:
            //  47.0 vs 44.5 us on the north-star shape, the round trip at the start of every frame costs more than it saves;
            //  likewise ds_write_addtid_b32 for the magnitudes and the partial sums: tools/probes/experiments/kpr_mel_pw_addtid.h.txt)
#pragma unroll
            for (int m = 0; m < kPts; ++m) z[m] = pmul(nz[m], winl[fl + L * m]);
            if constexpr (PAIR) {
                if (sub == 0) {
#pragma unroll
                    for (int m = 0; m < kPts; ++m) nz[m] = nz2[m];
                }
            }
            tw.refresh();
            if constexpr (IsWide<WsSwz>::value) {
                cfft_forward_wide_planar(z, tw, row);
            } else {
                using Rx = Radix<NC>;
                fft_pass<NC, 1, Rx::r1, 1, WsSwz>(z, tw, row);
                fft_pass<NC, 2, Rx::r2, Rx::r1, WsSwz>(z, tw, row);
                if constexpr (Rx::r3 > 1) fft_pass<NC, 3, Rx::r3, Rx::r1 * Rx::r2, WsSwz>(z, tw, row);
            }
            // the 32 mel weights of this lane: requested here, a pairing pass ahead of the sample prefetch below.  Vector
            // memory completes in order and hipcc counts conservatively across the branches of the fetch, so weights
            // requested together with the samples were waited for WITH them (an HBM round trip inside every frame).
            pw_load_weights<NC>(pl.sec, fl, wq);
            float mk[kPts / 2], mp[kPts / 2];
            float mid = 0.0f;
            rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
                const float a = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y);
                if (kp >= 0) {
                    const int m = (k - fl) / L;                           // compile-time after unrolling
                    mk[m] = a;
                    mp[m] = __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y);
                } else mid = a;                                           // k = NC / 2 (lane 0 only)
            });
            // bin k lives at word k + 4 (k >> 6): k = fl + L m -> fl + L m + 4 ((L m) >> 6); k' = NC - k = L (16 - m) - fl
            // -> k' + 4 ((L (15 - m)) >> 6) for fl >= 1, and 4 more on lane 0 wherever L (16 - m) is a multiple of 64
            float* lo = row + fl;
            float* hi = row + (NC - fl);
            float* hi0 = hi + ((fl == 0) ? 4 : 0);
            KPR_LDS_FENCE_W();
#pragma unroll
            for (int m = 0; m < kPts / 2; ++m) lo[L * m + 4 * ((L * m) >> 6)] = mk[m];
#pragma unroll
            for (int m = 0; m < kPts / 2; ++m) {
                const int off = -L * m + 4 * ((L * (15 - m)) >> 6);
                if ((L * (16 - m)) % 64 == 0) hi0[off] = mp[m]; else hi[off] = mp[m];
            }
            if (fl == 0) row[pw_mag_word(NC / 2)] = mid;
        }
        if (!PAIR || sub == 1) {   // the next ticket: drawn now, its samples requested now -- they land under this frame's sums and stores
            int drawn = 0, lane_p = lane0;
            asm volatile("" : "+v"(lane_p) :: "memory");                 // nothing of the fetch is computed above here
            if (lane_p == 0) drawn = atomicAdd(ctr, 1);                   // ds_add_rtn_u32
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(wq[j]));   // (the weights have landed: waited for HERE)
            nxt = __builtin_amdgcn_readfirstlane(drawn);
            if constexpr (PAIR) fetch_pair(nxt, lane_p, nz, nz2);
            else nsw = fetch_ticket(nxt, lane_p, nz);
            if (tail_prio && nxt >= n_wg) __builtin_amdgcn_s_setprio(0);  // (see tail_prio)
        }
        PW_STAMP();
        // ---- banded mel sums of the row, [10 log10], stores ---------------------------------------------------------------
        {
            const long long gf = PAIR ? 2 * ((long long)(t_wg0 + cur) * G + grp) + sub : (long long)(t_wg0 + cur) * G + grp;
            const bool fvalid = gf < g.total_frames;
            // channels_first output with frames numbered (b, c, f), or one channel: frame gf's row starts at gf M
            const bool lin_out = (!g.out_cl && !g.cfast) || g.C == 1;     // wave-uniform
            long long obase = gf * pl.M;
            int item_b = 0;
            int skey = 0;                                                 // staged store: 64 x (block index inside the workgroup) + channel
            if (!lin_out || db.enabled) {
                FramePos pc = frame_pos32(g, (unsigned)(fvalid ? gf : 0));
                if (!lin_out) obase = spec_base(g, pc, gf, pl.M);
                item_b = pc.b;
                if constexpr (STAGE) {
                    // (staged: t_wg0 is a multiple of C / 2 tickets, the workgroup's first pair is channel pair 0 of block bf0)
                    const int bf0 = (run_q * (int)blockIdx.x + min((int)blockIdx.x, run_r)) * G;   // the workgroup's first block
                    skey = 64 * (pc.b * g.F + pc.f - bf0) + pc.c;          // (the launcher stages only for C <= 64)
                }
            }
            float* outc = out + obase;
            // PAIR + channels_last + C >= 4, staged (round 5): the M x C block of an (item, frame) is ONE contiguous run of the
            // output.  Its C / 2 channel pairs are consecutive tickets, i.e. in the hands of C / 2 waves at about the same time:
            // each writes its two columns into slot lbf mod cl_slots, the last to arrive stores the block -- 1 KiB per
            // instruction, whole cache lines -- and hands the slot to block lbf + cl_slots.  (WRITE_SIZE of cfg3: 92 MB with
            // 4-byte stores, 54 MB with the 8-byte pairs below, the output itself is 33 MB.)
            const bool cl_stage = STAGE && g.out_cl && pl.cl_slots > 0;   // wave-uniform
            // (slot index / generation / address are re-derived from skey where they are used: one register across the sums, the
            //  <512> instance has none to spare)
            auto slot_idx = [&](int key) { return (key >> 6) & (pl.cl_slots - 1); };
            auto slot_gen = [&](int key) { return (key >> 6) / pl.cl_slots; };          // (a power of two; key >= 0)
            if constexpr (STAGE) {
                if (cl_stage) {
                    if (sub == 0) {
                        const int sidx = slot_idx(skey), sgen = slot_gen(skey);
                        KPR_LDS_FENCE_X();                                // (the magnitude stores above are a closed group: what follows polls)
                        // the slot's previous block (lbf - cl_slots) must have left: its pairs were drawn before this one and their
                        // waves wait for nobody who waits for us -- bounded all the same (KPR_E_DEVICE, kpr_common.h)
                        int spin = 0;
                        for (; spin < kPwSlotSpinLimit; ++spin) {
                            const bool ok = !fvalid || __hip_atomic_load(&slot_done_at()[sidx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= sgen;
                            if (__all(ok)) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (__builtin_expect(spin >= kPwSlotSpinLimit, 0)) status_raise(kStMelPwSlot);
                    }
                }
            }
            // PAIR + channels_last output (round 5): the results of channel c wait in LDS (this lane's own words) until those
            // of channel c + 1 exist, then (c, c + 1) leave as ONE 8-byte store per filter: half the store instructions, and
            // 8 instead of 4 bytes of every 4 C-byte period written at a time (cfg3, C = 6: profiles/r05_cl_output.md)
            const bool pair_cl = PAIR && g.out_cl;                        // wave-uniform
#ifdef KPR_PW_GATHER64          /* development: the pair-gather form of stage 2 in the fused kernel (tools/lds_conflicts.sh) */
            constexpr bool G32 = PAIR && NC == 1024;
#else
            constexpr bool G32 = true;
#endif
            pw_band_sums<NC, PAIR, G32>(row, fl, pl.sec, wq, tab, pl.NR, pl.CMQ, [&](int r, float v) {
                const int mel = fl + L * r;
                const bool have = fvalid && mel < pl.M;
                if (db.enabled) {
                    v = to_db(v, db);
                    db_account(dbrun, have, have ? item_b : -1, v, v, item_stats, db);
                }
                if (STAGE && cl_stage) {
                    if (have) slots_at()[slot_idx(skey) * pl.cl_blk + mel * g.C + (skey & 63)] = v;
                } else if (PAIR && pair_cl) {
                    if (sub == 0) hold[64 * r + lane] = v;
                    else if (have) {
                        struct __attribute__((aligned(8))) float2a { float x, y; };
                        *reinterpret_cast<float2a*>(outc - 1 + (long long)mel * ostride) = float2a{hold[64 * r + lane], v};
                    }
                } else if (have) outc[(long long)mel * ostride] = v;
            });
            if constexpr (STAGE) {
                if (cl_stage && sub == 1) {
                    // both columns of this pair are in the slot (LDS executes a wave's operations in order: the counter moves
                    // after them); whoever brings the count to C / 2 finds every other pair's columns there as well
                    const int sidx = slot_idx(skey), sgen = slot_gen(skey), chan = skey & 63;
                    const float* slot = slots_at() + sidx * pl.cl_blk;
                    int old = 0;
                    if (fl == 0 && fvalid)
                        old = __hip_atomic_fetch_add(&slot_cnt_at()[sidx], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    old = __builtin_amdgcn_readfirstlane(old);             // (G = 1: one block per wave)
                    if (fvalid && old == (g.C >> 1) - 1) {
                        KPR_LDS_FENCE_R();
                        typedef float f4nt __attribute__((ext_vector_type(4), aligned(16)));
                        const f4nt* src = reinterpret_cast<const f4nt*>(slot);
                        f4nt* dst = reinterpret_cast<f4nt*>(outc - chan);                // (b, f, 0, 0): 4 M C contiguous bytes
                        const int n4 = (pl.M * g.C) >> 2;
#pragma unroll 1
                        for (int i = fl; i < n4; i += L) __builtin_nontemporal_store(src[i], dst + i);
                        KPR_LDS_FENCE_X();
                        if (fl == 0) {                                                  // (behind the reads in this wave's LDS order)
                            __hip_atomic_store(&slot_cnt_at()[sidx], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(&slot_done_at()[sidx], sgen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
        }
        PW_STAMP();
      }
      cur = nxt;
    }
    if (db.enabled) db_flush_wave(dbrun, item_stats, db);
#ifdef KPR_DEV_STAMPS    /* every workgroup: start / end on the constant 100 MHz clock and on the shader clock (dbg[1024 + 4 bx ..]) */
    if (dbg && tid == 0 && blockIdx.x < 4096) {
        long long* e = dbg + 1024 + 4 * (long long)blockIdx.x;
        e[0] = (long long)wg_r0; e[1] = (long long)__builtin_amdgcn_s_memrealtime();
        e[2] = (long long)wg_c0; e[3] = (long long)__builtin_readcyclecounter();
    }
#endif
#undef PW_STAMP
}

}  // namespace kpr
