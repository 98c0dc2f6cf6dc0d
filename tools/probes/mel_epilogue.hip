// mel_epilogue.hip -- stand-alone probe of k_mel_pw's banded mel sums ("probe first").
//
// The exact device function the product kernel calls (pw_band_sums, kapre_amd/csrc/kpr_mel_pw_kernels.h, included
// unmodified) runs on LDS-resident magnitude rows: WPS waves per SIMD (one workgroup of 4*WPS waves per CU, 256
// workgroups), every wave its own row; the only global memory inside the loop is what the function itself reads per
// frame (the 32 weights per lane, L1-resident, and the sixteen masks through the scalar cache).  Modes:
//     sums      stage 1 + stage 2 only (the magnitudes are written once, before the loop)
//     row+sums  + the 17 magnitude row writes a frame does (so that every frame's sums start from real magnitudes)
//     row       the row writes alone
//     empty     the loop skeleton
// Prints wall time per frame and SIMD, the shader clock measured in the loop and cycles per frame and SIMD, after
// checking the sums of one frame against the dense product on the host (mel bank 44.1 kHz / n_fft 2048 / 128 mels, built
// here with the Slaney formulas; the plan comes from kpr_filterbank_pack of the product library).
// Go / no-go of the verdict: <= +350 VALU cycles and <= +150 LDS-pipe cycles per frame (rocprofv3 --pmc passes over
// this binary: tools/probes/run_probes.sh).
//
// Build: tools/probes/build.sh mel_epilogue   (links kapre_amd/lib/libkapre_hip.so)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kapre_hip.h"
#include "../../kapre_amd/csrc/kpr_fft.h"
#include "../../kapre_amd/csrc/kpr_fft_mr.h"
#include "../../kapre_amd/csrc/kpr_common.h"
#include "../../kapre_amd/csrc/kpr_mel_kernels.h"
#include "../../kapre_amd/csrc/kpr_mel_ts_kernels.h"
#include "../../kapre_amd/csrc/kpr_mel_pw_kernels.h"

using namespace kpr;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

constexpr int NC = 1024, L = 64;
constexpr int RWD = pw_row_words(NC);
enum { M_SUMS = 1, M_ROW = 2 };

template <int MODE>
__global__ __launch_bounds__(1024) void probe_epi(const unsigned* __restrict__ sec, int NR, int CMQ, int iters,
                                                  const float* __restrict__ mags, float* __restrict__ outv,
                                                  unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    float* rows = smem;
    float* tab = smem + nw * RWD;                              // P | WN | T2, as k_mel_pw keeps them
    {
        const int nt = pw_lds_table_words(L, NR, CMQ);
        const uint4* src = reinterpret_cast<const uint4*>(sec + kPwEmaskWords + 32 * L);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < nt / 4; i += blockDim.x) dst[i] = src[i];
    }
    float* row = rows + wave * RWD;
    for (int i = lane; i < RWD; i += 64) row[i] = 0.0f;
    float mk[8], mp[8], mid = mags[NC / 2];
#pragma unroll
    for (int m = 0; m < 8; ++m) { mk[m] = mags[lane + L * m]; mp[m] = mags[NC - lane - L * m]; }
    auto write_row = [&]() {                                   // the magnitude writes of k_mel_pw, same addresses
        float* lo = row + lane;
        float* hi = row + (NC - lane);
        float* hi0 = hi + ((lane == 0) ? 4 : 0);
#pragma unroll
        for (int m = 0; m < 8; ++m) lo[L * m + 4 * ((L * m) >> 6)] = mk[m];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int off = -L * m + 4 * ((L * (15 - m)) >> 6);
            if ((L * (16 - m)) % 64 == 0) hi0[off] = mp[m]; else hi[off] = mp[m];
        }
        if (lane == 0) row[pw_mag_word(NC / 2)] = mid;
    };
    write_row();
    __syncthreads();
    float keep = 0.0f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr ((MODE & M_ROW) != 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) { asm volatile("" : "+v"(mk[m]), "+v"(mp[m])); }
            write_row();
        }
        if constexpr ((MODE & M_SUMS) != 0) {
            int fl = lane;
            asm volatile("" : "+v"(fl));
            f4 wq[8];
            pw_load_weights<NC>(sec, fl, wq);
            pw_band_sums<NC>(row, fl, sec, wq, tab, NR, CMQ, [&](int r, float v) {
                keep += v;
                if (iters == 1) outv[((size_t)blockIdx.x * nw + wave) * (L * NR) + fl + L * r] = v;
            });
        }
        asm volatile("" : "+v"(keep));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (keep == 12345.678f) outv[0] = keep;
    if (lane == 0 && clk) {
        clk[2 * ((size_t)blockIdx.x * nw + wave)] = c1 - c0;
        clk[2 * ((size_t)blockIdx.x * nw + wave) + 1] = r1 - r0;
    }
}

static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

template <int MODE>
static void run(const char* name, const unsigned* sec, int NR, int CMQ, int iters, const float* mags, float* outv,
                unsigned long long* clk, int wps) {
    const int nw = 4 * wps;
    const size_t lds = sizeof(float) * ((size_t)nw * RWD + pw_lds_table_words(L, NR, CMQ));
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_epi<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe_epi<MODE>, dim3(256), dim3(64 * nw), lds, 0, sec, NR, CMQ, iters / 10 + 2, mags, outv, clk);   // warm-up
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe_epi<MODE>, dim3(256), dim3(64 * nw), lds, 0, sec, NR, CMQ, iters, mags, outv, clk);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipDeviceSynchronize());
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * 256 * nw);
    HIP_OK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (size_t i = 0; i < h.size(); i += 2) { cyc += (double)h[i]; rt += (double)h[i + 1]; }
    const double mhz = cyc / rt * 100.0;                                  // s_memrealtime ticks at 100 MHz
    const double ns = ms * 1e6 / ((double)iters * wps);                   // per frame and SIMD: wps frames per SIMD per iteration
    std::printf("| %s | %d | %.1f | %.0f | %.1f | %.0f |\n", name, wps, ms * 1e3, mhz, ns, ns * mhz * 1e-3);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int K = NC + 1, M = 128;
    const double sr = 44100.0;
    // librosa.filters.mel (Slaney scale, slaney norm), float64 then float32 -- the bank of the north-star workload
    std::vector<float> fb((size_t)K * M, 0.0f);
    {
        std::vector<double> mel_f(M + 2);
        const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(sr / 2);
        for (int i = 0; i < M + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (M + 1));
        for (int i = 0; i < M; ++i)
            for (int k = 0; k < K; ++k) {
                const double f = k * sr / (2.0 * NC);
                const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]), upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
                const double w = std::max(0.0, std::min(lower, upper)) * 2.0 / (mel_f[i + 2] - mel_f[i]);
                fb[(size_t)k * M + i] = (float)w;
            }
    }
    std::vector<int32_t> kr(2 * ((M + 15) / 16));
    if (kpr_filterbank_kranges(fb.data(), K, M, kr.data())) { std::fprintf(stderr, "kranges: %s\n", kpr_last_error()); return 2; }
    const int64_t nfl = kpr_filterbank_pack_floats(K, M, kr.data());
    std::vector<float> blob((size_t)nfl);
    if (kpr_filterbank_pack(fb.data(), K, M, kr.data(), blob.data())) { std::fprintf(stderr, "pack: %s\n", kpr_last_error()); return 2; }
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(blob.data());
    if (!hdr[6]) { std::fprintf(stderr, "no band plan in the packed filterbank\n"); return 2; }
    const int NR = (int)hdr[8], CMQ = (int)hdr[9], nlist = (int)hdr[10], words = (int)hdr[11];
    std::printf("band plan: L %u, NR %d, CMQ %d, %d partial sums per frame, %d words\n\n", hdr[7], NR, CMQ, nlist, words);

    std::vector<float> mags(K);
    unsigned s = 12345u;
    for (int k = 0; k < K; ++k) { s = s * 1664525u + 1013904223u; mags[k] = (float)((s >> 8) & 0xffff) / 65536.0f + 0.01f; }
    unsigned* d_sec; float *d_mags, *d_out; unsigned long long* d_clk;
    HIP_OK(hipMalloc(&d_sec, sizeof(uint32_t) * words));
    HIP_OK(hipMemcpy(d_sec, hdr + hdr[6], sizeof(uint32_t) * words, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&d_mags, sizeof(float) * K));
    HIP_OK(hipMemcpy(d_mags, mags.data(), sizeof(float) * K, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&d_out, sizeof(float) * 256 * 16 * L * NR));
    HIP_OK(hipMalloc(&d_clk, 8 * 2 * 256 * 16));

    // ---- correctness: one frame, every wave, against the dense product
    {
        const int nw = 16;
        const size_t lds = sizeof(float) * ((size_t)nw * RWD + pw_lds_table_words(L, NR, CMQ));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_epi<M_SUMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(probe_epi<M_SUMS>, dim3(256), dim3(64 * nw), lds, 0, d_sec, NR, CMQ, 1, d_mags, d_out, d_clk);
        HIP_OK(hipDeviceSynchronize());
        std::vector<float> got((size_t)256 * nw * L * NR);
        HIP_OK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int m = 0; m < M; ++m) {
            double want = 0;
            for (int k = 0; k < K; ++k) want += (double)mags[k] * fb[(size_t)k * M + m];
            for (int w = 0; w < 256 * nw; w += 97) worst = std::max(worst, std::fabs(got[(size_t)w * L * NR + m] - want) / std::max(1e-30, std::fabs(want)));
        }
        std::printf("band sums vs dense product (float64), worst relative error over 128 filters: %.3g %s\n\n", worst, worst < 2e-6 ? "OK" : "MISMATCH");
        if (!(worst < 2e-6)) return 1;
    }
    std::printf("| configuration | waves/SIMD | kernel us | sclk MHz (in loop) | ns / frame / SIMD (wall) | cycles / frame / SIMD (wall x clock) |\n|---|---|---|---|---|---|\n");
    for (int wps = 1; wps <= 4; ++wps) run<M_SUMS>("sums (stage 1 + 2)", d_sec, NR, CMQ, iters, d_mags, d_out, d_clk, wps);
    for (int wps = 1; wps <= 4; ++wps) run<M_SUMS | M_ROW>("row writes + sums", d_sec, NR, CMQ, iters, d_mags, d_out, d_clk, wps);
    for (int wps = 1; wps <= 4; ++wps) run<M_ROW>("row writes", d_sec, NR, CMQ, iters, d_mags, d_out, d_clk, wps);
    for (int wps = 1; wps <= 4; ++wps) run<0>("empty loop", d_sec, NR, CMQ, iters, d_mags, d_out, d_clk, wps);
    return 0;
}
