// fft_core.hip -- stand-alone probe of the 1024-point producer instruction stream of k_mel_ws (n_fft 2048).
//
// "find what caps the FFT core at ~45-50 % VALU issue -- with a microbenchmark, not inside
// k_mel_ws".  WPS waves per SIMD (one workgroup of 4*WPS waves per CU, 256 workgroups) loop the producer's per-frame
// work on REGISTER/LDS-resident data: no global memory, no consumers, no tickets.  What a frame does is selected by
// template flags so that one binary holds every knock-out; the arithmetic is the product's (kpr_fft.h is included
// unmodified).  Per configuration the program prints wall time per frame and SIMD, the shader clock measured inside
// the loop (s_memtime against the constant 100 MHz s_memrealtime) and cycles per frame and SIMD (wall time x clock).
//
// Build: tools/probes/build.sh     Run on the GPU box: tools/probes/run_probes.sh
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <type_traits>
#include <vector>

#include "../../kapre_amd/csrc/kpr_fft.h"

using namespace kpr;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

enum : unsigned {
    F_EXCH   = 1u,       // LDS exchanges between the passes (off: passes hand over in registers -- wrong FFT, timing only)
    F_PAIR   = 2u,       // cross-lane reads of the real-FFT pairing (ds_bpermute)
    F_MAGW   = 4u,       // magnitude row writes
    F_WIN    = 8u,       // window values re-read from LDS every frame
    F_SQRT   = 16u,      // v_sqrt_f32 in the magnitudes
    F_NARROW = 32u,      // 32-bit skewed exchange (SwzSkew) instead of the 128-bit planar one
    F_PASSES = 64u,      // the three butterfly passes (off: only window / pairing / magnitudes remain)
    F_PAIRA  = 128u,     // pairing arithmetic + magnitudes (off: the frame ends after the FFT)
    F_LOAD2  = 256u,     // the next frame's samples from global memory (L2-resident), 16 x dwordx2 per lane, a frame ahead
    F_LOAD4  = 512u,     // the same bytes as 8 x dwordx4 per lane (+ 16 v_permlane32_swap to regroup: timing only here)
    F_STORE  = 1024u,    // the frame's complex spectrum to global memory: 8 x dwordx4 per lane, 1 KiB per instruction (8 KiB per frame)
    F_STORENT= 2048u,    // the same with nontemporal stores
    F_STAGE  = 4096u,    // k_stft's epilogue: the complex spectrum goes to the LDS row (17 x 8-byte writes), is read back with
                         // 8 x ds_read_b128 and stored as 8 x dwordx4 (1 KiB per instruction) -- instead of the magnitudes
    F_COLD   = 8192u,    // loads and stores walk fresh memory (every wave its own run of frames in a 2.4 GB buffer) instead of
                         // re-using a cache-resident window
    F_COLDRD = 16384u,   // only the loads walk fresh memory
    F_COLDWR = 32768u,   // only the stores walk fresh memory
    F_FULL   = F_EXCH | F_PAIR | F_MAGW | F_WIN | F_SQRT | F_PASSES | F_PAIRA,
};

constexpr int NC = 1024;
constexpr int ROWS = 2064;         // (also holds a frame's complex spectrum for the staged-store variants: 2 NC + 8 floats)
constexpr int ROWS_OLD = 1088;         // row stride: >= SwzSkew::row_words(1024) = 1080 and SwzWide 1028, rows stay 16-byte aligned

// one frame of the producer; returns nothing, leaves magnitudes in `row`
template <unsigned FL, class SW>
__device__ __forceinline__ void probe_frame(f2 (&nz)[kPts], f2 (&wv)[kPts], FftTw<NC, SW>& tw, const f2* winl, float* row,
                                            float* xrow, int fl, int lane, float& acc, const float* __restrict__ gsrc,
                                            const float* __restrict__ gwarm, const float* __restrict__ cold_base = nullptr) {
    constexpr int L = NC / kPts;
    f2 z[kPts];
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = nz[m];
    if constexpr ((FL & F_LOAD4) != 0) {
#pragma unroll
        for (int j = 0; j < kPts / 2; ++j) {
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(z[2 * j].x), "+v"(z[2 * j + 1].x));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(z[2 * j].y), "+v"(z[2 * j + 1].y));
        }
    }
#pragma unroll
    for (int m = 0; m < kPts; ++m) z[m] = pmul(z[m], wv[m]);
    // stands in for the sample prefetch of the next frame: the next frame's input is made opaque so that nothing is
    // hoisted out of the frame loop
    if constexpr ((FL & F_LOAD2) != 0) {
        // (F_COLD: the frame that starts 512 floats further on in the wave's own region: fresh lines every frame)
        // cold reads: the wave's own signal, hop 512 (gsrc advances 2048 floats per frame: the read region sits 2^29 floats
        // higher and advances 512)
        const float2* p = reinterpret_cast<const float2*>(((FL & (F_COLD | F_COLDRD)) != 0)
                              ? gsrc + (1u << 29) - (size_t)(gsrc - cold_base) / 4 * 3 : gwarm) + fl;
#pragma unroll
        for (int m = 0; m < kPts; ++m) { const float2 v = p[L * m]; nz[m] = f2{v.x, v.y}; }
    } else if constexpr ((FL & F_LOAD4) != 0) {
        const float4* p = reinterpret_cast<const float4*>(gsrc) + lane;
#pragma unroll
        for (int j = 0; j < kPts / 2; ++j) {
            const float4 v = p[64 * j];
            nz[2 * j] = f2{v.x, v.y};
            nz[2 * j + 1] = f2{v.z, v.w};
        }
    } else {
#pragma unroll
        for (int m = 0; m < kPts; ++m) asm volatile("" : "+v"(nz[m]));
    }
    tw.refresh();
    if constexpr ((FL & F_PASSES) != 0) {
        if constexpr ((FL & F_EXCH) != 0) {
            if constexpr (IsWide<SW>::value) cfft_forward_wide_planar(z, tw, xrow);
            else cfft_forward<NC, SW>(z, tw, xrow);
        } else {
            // the same butterflies and twiddles, registers handed over in place (no LDS traffic)
            using Rx = Radix<NC>;
            f2 o[kPts];
            pass_compute<NC, 1, Rx::r1, 1, SW>(z, tw, o);
            pass_compute<NC, 2, Rx::r2, Rx::r1, SW>(o, tw, z);
            pass_compute<NC, 3, Rx::r3, Rx::r1 * Rx::r2, SW>(z, tw, o);
#pragma unroll
            for (int m = 0; m < kPts; ++m) z[m] = o[m];
        }
    }
    if constexpr ((FL & F_STAGE) != 0) {
        f2* st2 = reinterpret_cast<f2*>(row);
        rfft_pair<NC>(z, tw, fl, lane, [&](int k, f2 xk, int kp, f2 xp) {
            st2[k] = xk;
            if (kp >= 0) st2[kp] = (kp == NC) ? f2{xp.x, 0.0f} : xp;
        });
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v* dst = ((FL & (F_COLD | F_COLDWR)) != 0) ? reinterpret_cast<f4v*>(const_cast<float*>(gsrc))
                                                     : reinterpret_cast<f4v*>(const_cast<float*>(gwarm) + (1u << 23));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i4 = fl + 64 * q;
            const f4v v = *reinterpret_cast<const f4v*>(row + 4 * i4);
            dst[i4] = v;
        }
    } else if constexpr ((FL & F_PAIRA) != 0) {
        float mk[kPts / 2], mp[kPts / 2];
        float mid = 0.0f;
        auto mag = [&](f2 v) {
            const float s = v.x * v.x + v.y * v.y;
            if constexpr ((FL & F_SQRT) != 0) return __builtin_amdgcn_sqrtf(s);
            else return s;
        };
        auto emit = [&](int k, f2 xk, int kp, f2 xp) {
            const float a = mag(xk);
            if (kp >= 0) {
                const int m = (k - fl) / L;
                mk[m] = a;
                mp[m] = mag(xp);
            } else mid = a;
        };
        if constexpr ((FL & F_PAIR) != 0) {
            rfft_pair<NC>(z, tw, fl, lane, emit);
        } else {
            // pairing arithmetic on the lane's own slots (no cross-lane reads)
            const f2 ppmi = f2{tw.pp.y, -tw.pp.x};
#pragma unroll
            for (int m = 0; m < kPts / 2; ++m) {
                const f2 zp = z[kPts - 1 - m];
                const f2 e = cadd_conj(z[m], zp);
                const f2 t = cmul(cmul_w32(csub_conj(z[m], zp), m), ppmi);
                emit(fl + L * m, cadd(e, t), NC - fl - L * m, csub(e, t));
            }
        }
        if constexpr ((FL & F_MAGW) != 0) {
            float* lo = row + fl;
            float* hi = row + (NC - fl) - L * (kPts / 2 - 1);
#pragma unroll
            for (int m = 0; m < kPts / 2; ++m) lo[L * m] = mk[m];
#pragma unroll
            for (int m = 0; m < kPts / 2; ++m) hi[L * (kPts / 2 - 1 - m)] = mp[m];
            if (fl == 0) row[NC / 2] = mid;
        } else {
#pragma unroll
            for (int m = 0; m < kPts / 2; ++m) acc += mk[m] + mp[m];
            acc += mid;
        }
    } else {
#pragma unroll
        for (int m = 0; m < kPts; ++m) acc += z[m].x + z[m].y;
    }
    if constexpr ((FL & (F_STORE | F_STORENT)) != 0) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v* dst = reinterpret_cast<f4v*>(const_cast<float*>(gsrc) + (1u << 23)) + lane;     // second half of the buffer
#pragma unroll
        for (int j = 0; j < kPts / 2; ++j) {
            const f4v v = {z[2 * j].x, z[2 * j].y, z[2 * j + 1].x, z[2 * j + 1].y};
            if constexpr ((FL & F_STORENT) != 0) __builtin_nontemporal_store(v, dst + 64 * j);
            else dst[64 * j] = v;
        }
    }
    if constexpr ((FL & F_WIN) != 0) {
#pragma unroll
        for (int m = 0; m < kPts; ++m) wv[m] = winl[fl + L * m];
    } else {
#pragma unroll
        for (int m = 0; m < kPts; ++m) asm volatile("" : "+v"(wv[m]));
    }
}

// stamps[wave_global][4] = {memtime0, realtime0, memtime1, realtime1}
template <int WPS, unsigned FL>
__global__ __launch_bounds__(WPS * 256) void k_core(const float2* __restrict__ twtab, const float* __restrict__ window,
                                                    float* __restrict__ sink, int frames, unsigned long long* __restrict__ stamps,
                                                    const float* __restrict__ gsrc) {
    typedef typename std::conditional<(FL & F_NARROW) != 0, SwzSkew, SwzWide>::type SW;
    constexpr int L = NC / kPts;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fl = lane & (L - 1);
    f2* winl = reinterpret_cast<f2*>(smem);                     // NC pairs
    float* row = smem + 2 * NC + wave * ROWS;                   // this wave's magnitude / exchange row
    float* xrow = row;
    for (int i = tid; i < NC; i += WPS * 256) winl[i] = f2{0.5f * window[2 * i], 0.5f * window[2 * i + 1]};
    FftTw<NC, SW> tw;
    tw.load(twtab, fl);
    __syncthreads();
    f2 nz[kPts], wv[kPts];
#pragma unroll
    for (int m = 0; m < kPts; ++m) {
        const int n = fl + L * m;
        nz[m] = f2{__sinf(0.37f * n + 0.11f * blockIdx.x), __cosf(0.23f * n + wave)};
        wv[m] = winl[n];
    }
    float acc = 0.0f;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    // every wave walks its own 128 KB of a 32 MB buffer with hop 512: frames overlap 4x, as in the target workload
    const float* gw = gsrc + ((size_t)(blockIdx.x * (WPS * 4) + wave) % 256) * 32768;
    // F_COLD: the wave's frames are consecutive frames of one long signal: 512 new samples in, 2048 floats out per frame
    const size_t wglob = (size_t)blockIdx.x * (WPS * 4) + wave;
    const float* gcold = gsrc + (1u << 24) + wglob * (size_t)frames * 2048;
#pragma unroll 1
    for (int it = 0; it < frames; ++it) {
        if constexpr ((FL & (F_COLD | F_COLDRD | F_COLDWR)) != 0) probe_frame<FL, SW>(nz, wv, tw, winl, row, xrow, fl, lane, acc, gcold + (size_t)it * 2048, gw + (size_t)(it & 31) * 512, gsrc + (1u << 24));
        else probe_frame<FL, SW>(nz, wv, tw, winl, row, xrow, fl, lane, acc, gw + (size_t)(it & 31) * 512, gw + (size_t)(it & 31) * 512);
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        unsigned long long* s = stamps + 4ull * (blockIdx.x * (WPS * 4) + wave);
        s[0] = t0; s[1] = r0; s[2] = t1; s[3] = r1;
    }
    // keep everything alive
    float v = acc + row[(lane * 17) & 1023];
    if (v == 1.2345e-30f) sink[blockIdx.x] = v;
}

struct Result { std::string name; int wps; double us; double mhz; double cyc_per_frame_simd; double ns_per_frame_simd; };

static std::vector<float2> make_twiddles(int nfft) {
    std::vector<float2> t(nfft);
    for (int j = 0; j < nfft; ++j) {
        const double a = -2.0 * M_PI * j / nfft;
        t[j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    return t;
}

static const float* g_src = nullptr;
template <int WPS, unsigned FL>
static Result run(const char* name, const float2* d_tw, const float* d_win, float* d_sink, unsigned long long* d_st, int frames, int reps) {
    const int grid = 256, threads = WPS * 256;
    const size_t lds = sizeof(float) * (2 * NC + (size_t)(WPS * 4) * ROWS);
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_core<WPS, FL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    k_core<WPS, FL><<<grid, threads, lds>>>(d_tw, d_win, d_sink, frames, d_st, g_src);       // warm-up
    HIP_OK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        HIP_OK(hipEventRecord(e0));
        k_core<WPS, FL><<<grid, threads, lds>>>(d_tw, d_win, d_sink, frames, d_st, g_src);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3);
    }
    std::vector<unsigned long long> st(4ull * grid * WPS * 4);
    HIP_OK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // clock: shader cycles per 100 MHz tick, averaged over the waves; in-loop duration = max over waves
    double mhz = 0; int n = 0; double cyc = 0;
    for (size_t w = 0; w < st.size() / 4; ++w) {
        const double dc = (double)(st[4 * w + 2] - st[4 * w]), dr = (double)(st[4 * w + 3] - st[4 * w + 1]);
        if (dr > 0) { mhz += dc / dr * 100.0; ++n; }
        cyc += dc;
    }
    mhz /= std::max(n, 1);
    cyc /= std::max<size_t>(st.size() / 4, 1);
    Result R;
    R.name = name; R.wps = WPS; R.us = best; R.mhz = mhz;
    // throughput comes from the WALL clock: the SIMD's issue arbitration is oldest-first, the waves of a SIMD finish one
    // after another, and the mean per-wave loop time understates the time the SIMD needed (valu_micro.hip, calibration rows)
    R.ns_per_frame_simd = best * 1e3 / ((double)frames * WPS);
    R.cyc_per_frame_simd = R.ns_per_frame_simd * mhz * 1e-3;
    (void)cyc;
    HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
    return R;
}

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 200;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    std::vector<float2> tw = make_twiddles(2 * NC);
    std::vector<float> win(2 * NC);
    for (int i = 0; i < 2 * NC; ++i) win[i] = 0.5f - 0.5f * (float)std::cos(2.0 * M_PI * i / (2 * NC));
    float2* d_tw; float* d_win; float* d_sink; unsigned long long* d_st;
    HIP_OK(hipMalloc(&d_tw, tw.size() * sizeof(float2)));
    HIP_OK(hipMalloc(&d_win, win.size() * sizeof(float)));
    HIP_OK(hipMalloc(&d_sink, 4096 * sizeof(float)));
    HIP_OK(hipMalloc(&d_st, 4ull * 256 * 16 * sizeof(unsigned long long)));
    HIP_OK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_win, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
    {
        float* d_src;
        const size_t bytes = ((size_t)(1u << 29) + (size_t)(1u << 24) + (size_t)4096 * 56 * 512 + (1 << 20)) * sizeof(float);   // 2.6 GB
        HIP_OK(hipMalloc(&d_src, bytes));
        HIP_OK(hipMemset(d_src, 0x3c, bytes));
        g_src = d_src;
    }
    std::vector<Result> rs;
#define RUN(W, F, NAME) rs.push_back(run<W, (F)>(NAME, d_tw, d_win, d_sink, d_st, frames, reps))
#define SWEEP(F, NAME) do { RUN(1, F, NAME); RUN(2, F, NAME); RUN(3, F, NAME); RUN(4, F, NAME); } while (0)
    SWEEP(F_FULL, "full (wide planar exchange)");
    SWEEP(F_FULL | F_LOAD2, "full + next frame's samples from L2, 16 x dwordx2");
    SWEEP(F_FULL | F_LOAD4, "full + next frame's samples from L2, 8 x dwordx4 + 16 permlane32_swap");
    SWEEP(F_FULL | F_LOAD2 | F_STORE, "full + sample loads + 8 KiB of stores per frame (8 x dwordx4)");
    SWEEP(F_FULL | F_LOAD2 | F_STORENT, "full + sample loads + 8 KiB of nontemporal stores per frame");
    SWEEP((F_FULL & ~(F_MAGW | F_SQRT | F_PAIRA)) | F_LOAD2 | F_STAGE, "STFT frame: loads + FFT + pairing -> LDS row -> 8 x ds_read_b128 + dwordx4 stores");
    if (frames <= 56) {
        SWEEP((F_FULL & ~(F_MAGW | F_SQRT | F_PAIRA)) | F_LOAD2 | F_STAGE | F_COLD, "STFT frame, fresh memory (8 KB written, 2 KB new samples per frame)");
        SWEEP((F_FULL & ~(F_MAGW | F_SQRT | F_PAIRA)) | F_LOAD2 | F_STAGE | F_COLDRD, "STFT frame, fresh loads only");
        SWEEP((F_FULL & ~(F_MAGW | F_SQRT | F_PAIRA)) | F_LOAD2 | F_STAGE | F_COLDWR, "STFT frame, fresh stores only");
    }
    SWEEP(F_FULL | F_NARROW, "full, 32-bit skewed exchange");
    SWEEP(F_FULL & ~F_EXCH, "no LDS exchange");
    SWEEP(F_FULL & ~F_PAIR, "no pairing bpermute");
    SWEEP(F_FULL & ~F_MAGW, "no magnitude row writes");
    SWEEP(F_FULL & ~F_WIN, "no window re-read");
    SWEEP(F_FULL & ~F_SQRT, "no v_sqrt");
    SWEEP(F_FULL & ~(F_EXCH | F_PAIR | F_WIN | F_MAGW), "no LDS at all (VALU stream only)");
    SWEEP(F_PASSES | F_EXCH, "FFT passes + exchanges only");
    SWEEP(F_PASSES, "FFT passes only (no LDS)");
    SWEEP(F_PAIR | F_MAGW | F_WIN | F_SQRT | F_PAIRA, "no FFT passes (window, pairing, magnitudes)");
    std::printf("| configuration | waves/SIMD | kernel us | sclk MHz (in loop) | ns / frame / SIMD (wall) | cycles / frame / SIMD (wall x clock) | VALU issue share (1796-cycle stream) |\n|---|---|---|---|---|---|---|\n");
    for (const Result& r : rs)
        std::printf("| %s | %d | %.1f | %.0f | %.0f | %.0f | %s |\n", r.name.c_str(), r.wps, r.us, r.mhz, r.ns_per_frame_simd, r.cyc_per_frame_simd,
                    r.name.rfind("full", 0) == 0 ? (std::to_string((int)(100.0 * 1796.0 / r.cyc_per_frame_simd + 0.5)) + " %").c_str() : "");
    return 0;
}
