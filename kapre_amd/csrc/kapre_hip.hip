// kapre_hip.hip -- gfx950 kernels + C ABI (include/kapre_hip.h) for Kapre's time-frequency path.
//
// One translation unit; the kernels live in the headers included below, one per kernel family:
//   kpr_fft.h, kpr_fft_mr.h   FFT building blocks: packed-f32 complex arithmetic, in-register DFTs,
//                             LDS exchange policies, the power-of-two Stockham passes, the mixed-radix
//                             (2^a 5^b) passes, real-FFT pairing
//   kpr_common.h              errors, frame geometry, sample fetch
//   kpr_mel_kernels.h         k_mel_ws / k_mel_fused: waveform -> [frame + window + rFFT -> |X| -> (K x M)
//                             filterbank on fp32 MFMA -> optional 10 log10], the whole Sequential of
//                             composed.py:138-261 in one launch; FROM_MAG: stand-alone ApplyFilterbank
//   kpr_mel_ts_kernels.h      k_mel_ts: the same Sequential, tile-synchronous (16 equal waves, two barriers per round) --
//                             the default fused mel kernel since round 3
//   kpr_mel_pw_kernels.h      k_mel_pw: the same Sequential with every wave owning its frames end to end and the mel product
//                             as banded sums on the vector ALU (round 4; banks with <= 2 non-zeros per bin)
//   kpr_stft_kernels.h        k_stft / k_stft_big / k_stft_bs / k_stft_mr: frame + window + rFFT with complex /
//                             magnitude / phase epilogue (time_frequency.py:164-185 [+ :359 / :402])
//   kpr_istft_kernels.h       k_istft_ws / k_istft_ws_mr / k_istft_fused, k_irfft* + k_ola
//                             (time_frequency.py:304-317)
//   kpr_generic_kernels.h     k_stft_gen / k_irfft_gen (run-time mixed-radix FFT for every transform size without a tuned
//                             plan, float32 and float64) and the float64 layer chain (time_frequency.py:155)
//   kpr_signal_kernels.h      k_frame, k_energy, k_delta, k_thin_gemm (signal.py, time_frequency.py:563-644)
//   kpr_misc_kernels.h        Magnitude / Phase, k_db_* (backend.py:126-194: log pass with per-item max/min
//                             statistics, then the dynamic-range clamp), k_gemm (generic fp32-MFMA GEMM:
//                             dense filterbanks and the DFT-as-GEMM path for transform sizes no FFT
//                             kernel covers -- the idea of the reference's kapre/tflite_compatible_stft.py:14-75)
//   kpr_grad_kernels.h        backward passes of the elementwise layers (tf.abs / tf.math.angle on complex data, the
//                             decibel map, the bin scaling that turns the inverse-STFT launch into STFT^T and back)
// This file: table caches, launch plans, argument validation and the C ABI.
//
// gfx950 only: wave64, v_mfma_f32_16x16x4_f32, 160 KiB LDS.  No CUDA/compat paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <type_traits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kapre_hip.h"
#include "kpr_fft.h"
#include "kpr_fft_mr.h"

#include "kpr_common.h"
#include "kpr_mel_kernels.h"
#include "kpr_mel_ts_kernels.h"
#include "kpr_mel_pw_kernels.h"
#include "kpr_fb_pw_kernels.h"
#include "kpr_mel_mr_kernels.h"
#include "kpr_signal_kernels.h"
#include "kpr_stft_kernels.h"
#include "kpr_istft_kernels.h"
#include "kpr_istft_pw_kernels.h"
#include "kpr_generic_kernels.h"
#include "kpr_misc_kernels.h"
#include "kpr_grad_kernels.h"

namespace kpr {

// ------------------------------------------------------------------------------------------
// host side: table caches
// ------------------------------------------------------------------------------------------
static std::mutex g_mu;

// Process-wide tuning switches (kpr_set_option): plain atomics, read on the launch path.  The
// library never reads the process environment.
enum { OPT_MEL_VARIANT, OPT_ISTFT_PATH, OPT_MIXED_RADIX, OPT_DB_CHUNKS, OPT_VERBOSE, OPT_STFT_VARIANT, OPT_DB_SLOTS, OPT_MEL_CL_STAGE, OPT_FB_VARIANT, OPT_COUNT };
static std::atomic<int> g_opt[OPT_COUNT] = {{0}, {0}, {1}, {0}, {0}, {0}, {0}, {1}, {0}};
static inline int opt(int id) { return g_opt[id].load(std::memory_order_relaxed); }
static std::map<std::pair<int, int>, float2*> g_tw;           // (device, n_fft) -> twiddles
static std::map<std::pair<int, int>, float*> g_dft_fwd;       // (device, n_fft) -> [n_fft][2K]
static std::map<std::pair<int, int>, float*> g_dft_inv;       // (device, n_fft) -> [2K][n_fft]

static int cur_device(int* dev) {
    KPR_HIP(hipGetDevice(dev));
    return 0;
}

static int get_twiddles(int n_fft, const float2** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tw.find({dev, n_fft});
    if (it == g_tw.end()) {
        std::vector<float2> h(n_fft);
        for (int j = 0; j < n_fft; ++j) {
            double a = -2.0 * M_PI * (double)j / (double)n_fft;
            h[j] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        KPR_HIP(hipMalloc(&d, sizeof(float2) * n_fft));
        KPR_HIP(hipMemcpy(d, h.data(), sizeof(float2) * n_fft, hipMemcpyHostToDevice));
        it = g_tw.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// Bluestein tables for an even n_fft that is not a power of two (k_stft_bs): M, then
// [w: M][Bt: M][t: NCr + 1] as float2; Bt = FFT_M(chirp) / (2M) computed in double precision
static int bluestein_m(int n_fft) {
    if (n_fft < 4 || (n_fft & 1)) return 0;
    const int ncr = n_fft / 2;
    int m = 128;
    while (m < 2 * ncr - 1) m *= 2;
    return m <= 1024 ? m : 0;
}

static std::map<std::pair<int, int>, float2*> g_bs;

static int get_bluestein(int n_fft, const float2** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_bs.find({dev, n_fft});
    if (it == g_bs.end()) {
        const int ncr = n_fft / 2, m = bluestein_m(n_fft);
        std::vector<double> wr(ncr), wi(ncr), br(m, 0.0), bi(m, 0.0);
        for (int n = 0; n < ncr; ++n) {
            const long long n2 = ((long long)n * n) % (2LL * ncr);           // exact angle reduction
            const double a = -M_PI * (double)n2 / (double)ncr;
            wr[n] = std::cos(a); wi[n] = std::sin(a);
        }
        for (int n = 0; n < ncr; ++n) { br[n] = wr[n]; bi[n] = -wi[n]; }
        for (int n = 1; n < ncr; ++n) { br[m - n] = wr[n]; bi[m - n] = -wi[n]; }
        // O(M^2) DFT of the chirp in double precision (once per n_fft and device; M <= 1024)
        std::vector<float2> h(2 * (size_t)m + ncr + 1);
        for (int n = 0; n < m; ++n) h[n] = n < ncr ? make_float2((float)wr[n], (float)wi[n]) : make_float2(0.f, 0.f);
        for (int k = 0; k < m; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < m; ++n) {
                if (br[n] == 0.0 && bi[n] == 0.0) continue;
                const double a = -2.0 * M_PI * (double)(((long long)k * n) % m) / (double)m;
                const double c = std::cos(a), sn = std::sin(a);
                sr += br[n] * c - bi[n] * sn;
                si += br[n] * sn + bi[n] * c;
            }
            h[m + k] = make_float2((float)(sr / (2.0 * m)), (float)(si / (2.0 * m)));
        }
        for (int k = 0; k <= ncr; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n_fft;
            h[2 * (size_t)m + k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        KPR_HIP(hipMalloc(&d, sizeof(float2) * h.size()));
        KPR_HIP(hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
        it = g_bs.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// forward DFT matrix [n_fft rows n][2K cols]: col 2k = cos(2 pi k n/N), col 2k+1 = -sin(...)
static int get_dft_fwd(int n_fft, const float** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dft_fwd.find({dev, n_fft});
    if (it == g_dft_fwd.end()) {
        const int K = n_fft / 2 + 1;
        std::vector<float> h((size_t)n_fft * 2 * K);
        for (int n = 0; n < n_fft; ++n)
            for (int k = 0; k < K; ++k) {
                long long kn = ((long long)k * n) % n_fft;     // exact angle reduction
                double a = 2.0 * M_PI * (double)kn / (double)n_fft;
                h[(size_t)n * 2 * K + 2 * k] = (float)std::cos(a);
                h[(size_t)n * 2 * K + 2 * k + 1] = (float)(-std::sin(a));
            }
        float* d = nullptr;
        KPR_HIP(hipMalloc(&d, h.size() * sizeof(float)));
        KPR_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        it = g_dft_fwd.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// inverse real DFT matrix [2K rows][n_fft cols]: row 2k = c_k cos(2 pi k n/N)/N,
// row 2k+1 = -c_k sin(2 pi k n/N)/N, c_k = 1 for DC (and Nyquist when N even) else 2
static int get_dft_inv(int n_fft, const float** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dft_inv.find({dev, n_fft});
    if (it == g_dft_inv.end()) {
        const int K = n_fft / 2 + 1;
        std::vector<float> h((size_t)2 * K * n_fft);
        for (int k = 0; k < K; ++k) {
            const bool edge = (k == 0) || ((n_fft % 2 == 0) && k == n_fft / 2);
            const double ck = (edge ? 1.0 : 2.0) / (double)n_fft;
            for (int n = 0; n < n_fft; ++n) {
                long long kn = ((long long)k * n) % n_fft;
                double a = 2.0 * M_PI * (double)kn / (double)n_fft;
                h[(size_t)(2 * k) * n_fft + n] = (float)(ck * std::cos(a));
                h[(size_t)(2 * k + 1) * n_fft + n] = edge ? 0.0f : (float)(-ck * std::sin(a));
            }
        }
        float* d = nullptr;
        KPR_HIP(hipMalloc(&d, h.size() * sizeof(float)));
        KPR_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        it = g_dft_inv.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// ------------------------------------------------------------------------------------------
// host side: geometry / validation
// ------------------------------------------------------------------------------------------
static bool fast_nfft(int n_fft) {
    return n_fft == 256 || n_fft == 512 || n_fft == 1024 || n_fft == 2048;
}

static long long frames_of(const kpr_stft_geom* s) {
    long long t = s->time + (s->pad_begin ? (s->n_fft - s->hop_length) : 0);
    if (s->pad_end) return (t + s->hop_length - 1) / s->hop_length;
    if (t < s->win_length) return 0;
    return 1 + (t - s->win_length) / s->hop_length;
}

// Forward transforms with win_length > n_fft (time_frequency.py:174-182 hands both to tf.signal.stft): frames are cut with
// frame_length = win_length -- frames_of() and the right padding keep the caller's value -- and windowed, then rfft(fft_length)
// CROPS them to their first n_fft samples.  Everything behind the frame count therefore sees win_length = n_fft and the first n_fft
// entries of the caller's window: every FFT family takes these calls (through round 5 float32 fell back to the DFT-as-GEMM path and
// float64 refused them).
static kpr_stft_geom forward_geom(const kpr_stft_geom* s) {
    kpr_stft_geom e = *s;
    e.win_length = std::min(s->win_length, s->n_fft);
    return e;
}

static int check_geom(const kpr_stft_geom* s) {
    if (!s) return fail(KPR_E_BADARG, "geometry is NULL");
    if (s->batch < 0 || s->channels <= 0 || s->time < 0)
        return fail(KPR_E_BADARG, "bad batch/channels/time (%lld, %d, %lld)", (long long)s->batch,
                    s->channels, (long long)s->time);
    if (s->n_fft < 2 || s->win_length < 1 || s->hop_length < 1)
        return fail(KPR_E_BADARG, "bad n_fft/win_length/hop_length (%d, %d, %d)", s->n_fft,
                    s->win_length, s->hop_length);
    if ((unsigned)s->in_layout > 1u || (unsigned)s->out_layout > 1u)
        return fail(KPR_E_BADARG, "bad layout enum");
    if (s->pad_begin && s->n_fft < s->hop_length)
        return fail(KPR_E_BADARG, "pad_begin needs n_fft >= hop_length");
    // the kernels address one (batch item, channel) signal with 32-bit element offsets
    if (s->time * (long long)s->channels >= (1LL << 30))
        return fail(KPR_E_UNSUPPORTED, "time * channels = %lld elements per batch item: 2^30 or more is not supported",
                    (long long)(s->time * (long long)s->channels));
    return 0;
}

static Geom make_geom(const kpr_stft_geom* s, long long F) {
    Geom g;
    g.F = (int)F;
    g.C = s->channels;
    g.T = s->time;
    g.total_frames = s->batch * s->channels * F;
    g.n_fft = s->n_fft;
    g.win = s->win_length;
    g.hop = s->hop_length;
    g.pad_left = s->pad_begin ? (s->n_fft - s->hop_length) : 0;
    g.K = s->n_fft / 2 + 1;
    // with one channel the two layouts are the same memory image: take the contiguous paths
    // (Kapre's default is channels_last, so this is the common case)
    g.in_cl = s->in_layout == KPR_CHANNELS_LAST && s->channels > 1;
    g.out_cl = s->out_layout == KPR_CHANNELS_LAST && s->channels > 1;
    g.cfast = 0;
    geom_set_magic(g);
    return g;
}

// statistics slots per item for a batch of n items (DbDev::slot_mask): as many as keep slots x items <= 2048,
// at most 32, one for batches of 256 items and more (every word then collects a handful of atomics anyway)
// (the statistics region of the workspace is sized for db_slots_cap, whatever the option says: a workspace sized under one
//  "db_slots" value stays valid under any other -- ADVICE r03)
static int db_slots_cap(long long n_items) { return n_items <= 8192 ? 32 : 1; }
static int db_slots(long long n_items) {
    if (opt(OPT_DB_SLOTS) > 0) {                                 // forced (A/B runs, tests): rounded down to a power of two
        int f = 1;
        while (2 * f <= opt(OPT_DB_SLOTS)) f *= 2;
        return std::min(f, db_slots_cap(n_items));
    }
    int s = 1;
    while (s < 32 && (long long)(2 * s) * n_items <= 2048 && n_items < 256) s *= 2;
    return s;
}

static DbDev make_db(const kpr_db_params* db) {
    DbDev d{0, 1e-5f, 0.0f, 80.0f, 0, 0};
    if (db && db->enabled) {
        d.enabled = 1;
        // (to_db feeds max(x, amin) to v_log_f32, which has no denormal support: an amin below the smallest normal float --
        //  a floor under -379 dB -- is raised to it)
        d.amin = std::max(db->amin, 1.17549435e-38f);
        d.ref_term = (float)(10.0 * std::log10(std::max((double)db->amin, (double)db->ref_value)));
        d.dyn = db->dynamic_range;
    }
    return d;
}

static int check_db(const kpr_db_params* db) {
    if (db && db->enabled) {
        // same checks (and order) as backend.py:168-173
        if (!(db->ref_value > 0)) return fail(KPR_E_BADARG, "ref_value must be positive");
        if (!(db->amin > 0)) return fail(KPR_E_BADARG, "amin must be positive");
        if (!(db->dynamic_range > 0)) return fail(KPR_E_BADARG, "dynamic_range must be positive");
    }
    return 0;
}

static int grid_1d(long long n, int block, int cap = 256 * 16) {
    long long b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// Names of the kernels this thread's most recent API call launched, in order ("k_stats_init + k_mel_pw<1024> + k_db_clamp");
// kpr_last_launches() hands it to diagnostics (bench.py prints it as roofline.kernel instead of a table of its own).
// The entry points that launch the hot kernels clear it on entry (launch_log_begin).
static thread_local std::string g_launches;
static void launch_log_begin() { g_launches.clear(); }

// ---- the device status word (kpr_common.h): one word of mapped, coherent host memory per process ------------------------------
struct StatusWord {
    std::mutex mu;
    std::atomic<unsigned*> host{nullptr};   // what the host reads (volatile); written once under `mu`, read without it by every call
    bool installed[64] = {false};     // g_status_word of that device points at it
};
static StatusWord g_status;
// called by the launchers of the kernels that can raise it; everything is done once per device
static int status_word_ready() {
    int dev = 0;
    KPR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(g_status.mu);
    if (g_status.installed[dev]) return 0;
    // (a first call under stream capture: the allocation and the symbol copy are not stream work)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    int rc = 0;
    do {
        unsigned* hostp = g_status.host.load(std::memory_order_acquire);
        if (!hostp) {
            void* h = nullptr;
            hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent);
            if (e != hipSuccess) { rc = fail(KPR_E_HIP, "hipHostMalloc (status word) failed: %s", hipGetErrorString(e)); break; }
            *static_cast<volatile unsigned*>(h) = 0u;
            hostp = static_cast<unsigned*>(h);
            g_status.host.store(hostp, std::memory_order_release);
        }
        void* d = nullptr;
        hipError_t e = hipHostGetDevicePointer(&d, hostp, 0);
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_status_word), &d, sizeof d);
        if (e != hipSuccess) { rc = fail(KPR_E_HIP, "installing the status word failed: %s", hipGetErrorString(e)); break; }
        g_status.installed[dev] = true;
    } while (0);
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    return rc;
}
static const char* status_text(unsigned bits) {
    static thread_local char buf[384];
    snprintf(buf, sizeof buf, "0x%08x:%s%s%s%s%s%s%s", bits, (bits & kStMelWs) ? " k_mel_ws(bounded wait ran out)" : "",
             (bits & kStIstftWsCons) ? " k_istft_ws(consumer: bounded wait ran out)" : "",
             (bits & kStIstftWsProd) ? " k_istft_ws(producer: bounded wait ran out)" : "",
             (bits & kStIstftPw) ? " k_istft_pw(bounded wait ran out)" : "",
             (bits & kStMelPwSlot) ? " k_mel_pw_pair(bounded wait ran out)" : "",
             (bits & kStStalePlan) ? " k_mel_pw / k_fb_pw(the packed filterbank changed under a cached band plan: kpr_filterbank_forget)" : "",
             (bits & kStSelfTest) ? " self-test(bounded wait ran out)" : "");
    return buf;
}
// entry of every API call that launches hot kernels: the launch log restarts, and a status word raised by an EARLIER call's
// kernels fails this one (sticky until kpr_device_status reads it -- like a HIP sticky error, but recoverable)
static int api_enter() {
    launch_log_begin();
    if (unsigned* hostp = g_status.host.load(std::memory_order_acquire)) {
        const unsigned bits = *static_cast<volatile unsigned*>(hostp);
        if (bits)
            return fail(KPR_E_DEVICE, "a kernel of an earlier call raised the device status word (%s): its results are wrong; "
                        "kpr_device_status() reads and clears the condition", status_text(bits));
    }
    return 0;
}
// `detail`: the template arguments beyond the transform size that pick the INSTANCE ("s4", "w16", "rj4,v2" ...): the fuzz gate of
// tests/test_fuzz_gate.py asserts that every instance of the large-launch kernels was reached, not just every family
static int launch_check(const char* what, int tag = 0, const char* detail = nullptr) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(KPR_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    if (g_launches.size() < 200) {
        if (!g_launches.empty()) g_launches += " + ";
        g_launches += what;
        if (tag) {
            char b[48];
            if (detail) snprintf(b, sizeof b, "<%d,%s>", tag, detail);
            else snprintf(b, sizeof b, "<%d>", tag);
            g_launches += b;
        }
    }
    return 0;
}

// frame-row maps for GEMM paths: rows are global frames g = (b*C + c)*F + f
static RowMap frames_out_map(const Geom& g, long long Q) {
    RowMap m;
    m.rows = g.total_frames; m.D0 = g.F; m.D1 = g.C;
    if (g.out_cl) { m.s2 = (long long)g.F * Q * g.C; m.s1 = 1; m.s0 = Q * g.C; m.es = g.C; }
    else { m.s2 = (long long)g.C * g.F * Q; m.s1 = (long long)g.F * Q; m.s0 = Q; m.es = 1; }
    return m;
}
static RowMap frames_contig_map(const Geom& g, long long Q) {
    RowMap m;
    m.rows = g.total_frames; m.D0 = g.F; m.D1 = g.C;
    m.s2 = (long long)g.C * g.F * Q; m.s1 = (long long)g.F * Q; m.s0 = Q; m.es = 1;
    return m;
}

template <int AMODE, int EPI>
static int run_gemm(const float* a, const float* bm, const GemmArgs& ga, float* out,
                    hipStream_t st) {
    if (ga.in.rows <= 0 || ga.N <= 0) return 0;
    dim3 grid((unsigned)((ga.in.rows + 63) / 64), (unsigned)((ga.N + 63) / 64));
    hipLaunchKernelGGL((k_gemm<AMODE, EPI>), grid, dim3(256), 0, st, a, bm, ga, out);
    return launch_check("k_gemm");
}

// STFT of every frame into `out` (complex64), through the DFT-as-GEMM path
static int stft_gemm(const float* x, const kpr_stft_geom* s, const Geom& g, const float* window,
                     float* out_cplx, bool out_contig, hipStream_t st) {
    const float* dft = nullptr;
    if (int e = get_dft_fwd(g.n_fft, &dft)) return e;
    GemmArgs ga{};
    ga.in.rows = g.total_frames; ga.in.D0 = g.F; ga.in.D1 = g.C;
    if (g.in_cl) { ga.in.s2 = g.T * g.C; ga.in.s1 = 1; ga.t_es = g.C; }
    else { ga.in.s2 = (long long)g.C * g.T; ga.in.s1 = g.T; ga.t_es = 1; }
    ga.in.s0 = 0; ga.in.es = 0;
    ga.out = out_contig ? frames_contig_map(g, g.K) : frames_out_map(g, g.K);
    ga.Kdim = std::min(g.win, g.n_fft);
    ga.N = 2 * g.K;
    ga.ldb = 2 * g.K;
    ga.T = g.T; ga.hop = g.hop; ga.pad_left = g.pad_left;
    ga.window = window; ga.win = g.win;
    (void)s;
    return run_gemm<A_FRAME, E_CPLX>(x, dft, ga, out_cplx, st);
}

static int device_cus(int* cus);

// Kernels that use more than 64 KiB of dynamic LDS must opt in, once per (kernel, device).
// (benign race: the call is idempotent)
struct LdsOptIn { bool done[64] = {}; };
static int allow_big_lds(LdsOptIn& st, const void* fn) {
    int dev = 0;
    KPR_HIP(hipGetDevice(&dev));
    const bool slot = dev >= 0 && dev < 64;
    if (!slot || !st.done[dev]) {
        KPR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (slot) st.done[dev] = true;
    }
    return 0;
}
static long long* g_debug_stamps = nullptr;   // development aid: kpr_debug_stamps()

template <int NC, int NW>
static int launch_istft_fused(const float2* spec, const kpr_stft_geom* s, long long F,
                              const float* synth, const float2* tw, float* out, hipStream_t st,
                              bool* launched) {
    constexpr int L = NC / kPts, G = 64 / L;
    *launched = false;
    const int win = s->win_length, hop = s->hop_length;
    if (hop > win || F < 1) return 0;                       // gaps between frames: two-kernel path
    const int R = (win + hop - 1) / hop;
    const int RS = ((std::max(win, NC) + 3) & ~3) + 4;
    const int spare = (G > 1) ? NW * (G - 1) : 0;           // scratch rows for idle frame slots
    // rows: as many as fit next to a second workgroup on the CU (80 KiB each), at most 16; if that
    // leaves too few new frames per block, take the whole CU (160 KiB) instead
    // (an 8-wave workgroup at ~190 VGPRs fills the CU's register file on its own)
    int NR = (NW == 8) ? 0 : std::min(16, (int)(80 * 1024 / (sizeof(float) * RS)) - spare);
    if (NR < R + 3) NR = std::min(16, (int)(160 * 1024 / (sizeof(float) * RS)) - spare);
    if (NR < R + 1) return 0;
    IstftPlan pl;
    pl.n_sig = (long long)s->batch * s->channels;
    pl.t_out = (F - 1) * (long long)hop + win;
    pl.F = (int)F; pl.C = s->channels; pl.win = win; pl.hop = hop;
    pl.NR = NR; pl.R = R; pl.FB = NR - R + 1;
    pl.RS = RS;
    pl.chunks = (int)((pl.t_out + (long long)pl.FB * hop - 1) / ((long long)pl.FB * hop));
    pl.spec_cl = s->out_layout == KPR_CHANNELS_LAST;
    pl.wave_cl = s->in_layout == KPR_CHANNELS_LAST;
    pl.vec4 = hop % 4 == 0 && win % 4 == 0 && !(pl.wave_cl && s->channels > 1) &&
              (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const size_t lds = sizeof(float) * (size_t)(NR + spare) * pl.RS;
    if (lds > 160 * 1024) return 0;
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_istft_fused<NC, NW>))) return e;
    const long long nblocks = pl.n_sig * pl.chunks;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const int per_cu = std::max(1, (int)(160 * 1024 / lds));
    const unsigned grid = (unsigned)std::min<long long>(nblocks, (long long)std::min(per_cu, 2) * cus);
    hipLaunchKernelGGL((k_istft_fused<NC, NW>), dim3(grid), dim3(NW * 64), lds, st, spec, pl, synth,
                       tw, out, nblocks);
    *launched = true;
    return launch_check("k_istft_fused", NC);
}


static int device_cus(int* cus) {
    int dev = 0;
    KPR_HIP(hipGetDevice(&dev));
    static int cached[64] = {0};
    int v = 256;
    if (dev >= 0 && dev < 64) {
        if (!cached[dev]) {
            int q = 0;
            KPR_HIP(hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev));
            cached[dev] = q > 0 ? q : 256;
        }
        v = cached[dev];
    }
    *cus = v;
    return 0;
}

// segments per signal for k_istft_ws: rounds of `cus` workgroups x (blocks + halo frames) per segment
static int istft_ws_segments(long long n_sig, int Q, int R, int cus) {
    // cost of a schedule in frame times: rounds of `cus` workgroups x (hop blocks + halo frames + the
    // fixed start of a segment: first spectrum rows from HBM, pipeline fill -- about 20 frames' worth)
    int best = 1;
    double best_cost = 1e300;
    const int smax = std::max(1, std::min(4096, Q / (4 * R)));
    for (int sg = 1; sg <= smax; ++sg) {
        const long long rounds = (n_sig * sg + cus - 1) / cus;
        const double cost = (double)rounds * ((Q + sg - 1) / sg + R - 1 + 20);
        if (cost < best_cost * 0.999) { best_cost = cost; best = sg; }
        if (n_sig * sg > 8LL * cus) break;          // more segments than that only add halos
    }
    return best;
}

template <int NC, int RJ>
static int launch_istft_ws_inst(const float2* spec, const IstftWsPlan& pl, size_t lds, unsigned grid,
                                const float* synth, const float2* tw, float* out, int nitems, hipStream_t st) {
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_istft_ws<NC, RJ>))) return e;
    if (int e = status_word_ready()) return e;                  // (the kernel's bounded waits report there)
    hipLaunchKernelGGL((k_istft_ws<NC, RJ>), dim3(grid), dim3(kIwThreads), lds, st, spec, pl, synth, tw, out,
                       nitems, g_debug_stamps);
    return launch_check("k_istft_ws", NC, RJ == 2 ? "rj2" : RJ == 4 ? "rj4" : "rj8");
}

// Plan of the ring kernels (k_istft_ws, k_istft_ws_mr); false when they do not apply.
//   row_min: floats a ring row needs as FFT exchange buffer, G: frames per producer ticket,
//   spare: extra rows (exchange rows of idle frame slots), extra: LDS bytes behind rows / flags / counters
//   vec_min: 4 = only groups of four samples per consumer lane (k_istft_ws), 2 = pairs allowed as well
//   *vec: samples per consumer lane and group the plan uses (4 or 2)
static bool istft_ws_plan(const kpr_stft_geom* s, long long F, const float* out, int row_min, int G, int spare,
                          size_t extra, int cus, int vec_min, IstftWsPlan* plo, size_t* lds, int* rj, int* vec,
                          long long* nitems) {
    const int win = s->win_length, hop = s->hop_length;
    if (hop > win || F < 1 || opt(OPT_ISTFT_PATH) == 1 || opt(OPT_ISTFT_PATH) == 2) return false;
    // contiguous waveform and contiguous spectrogram rows (channels_first, or one channel)
    if ((s->in_layout == KPR_CHANNELS_LAST && s->channels > 1) || (s->out_layout == KPR_CHANNELS_LAST && s->channels > 1))
        return false;
    // VEC samples per lane in the overlap-add: hop, win multiples of VEC, 4 * VEC byte aligned waveform
    int VEC = 0;
    if (hop % 4 == 0 && win % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) VEC = 4;
    else if (vec_min <= 2 && hop % 2 == 0 && win % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) VEC = 2;
    if (!VEC) return false;
    const long long n_sig = (long long)s->batch * s->channels;
    const long long t_out = (F - 1) * (long long)hop + win;
    if (n_sig * 4096 >= (1LL << 31) || t_out + hop >= (1LL << 31)) return false;
    IstftWsPlan pl;
    pl.t_out = t_out;
    pl.F = (int)F; pl.C = s->channels; pl.win = win; pl.hop = hop;
    pl.R = (win + hop - 1) / hop;
    const int RJ = pl.R <= 2 ? 2 : pl.R <= 4 ? 4 : 8;         // rows read per sample group
    if (pl.R > 8) return false;
    if (VEC == 2 && RJ != 4) return false;                     // pairs: only the four-row instance is built
    const int per_pass = 64 * (kIwReads / RJ);                 // sample groups per consumer pass
    if (hop / VEC > per_pass) return false;                    // a hop block must fit one pass
    pl.RS = ((std::max(win, row_min) + 3) & ~3) + 4;
    pl.Q = (int)F - 1 + pl.R;
    pl.QB = std::min(16, per_pass / (hop / VEC));
    auto bytes = [&](int nr) { return sizeof(float) * (size_t)(nr + spare) * pl.RS + sizeof(int) * (size_t)(nr + 8) + extra; };
    int NR = 128;
    while (NR > 1 && bytes(NR) > 160 * 1024) NR >>= 1;
    // room for the frames of the two passes in flight (R-1+2*QB), the producers' tickets and slack
    if (NR < pl.R - 1 + 2 * pl.QB + 2 * G + 1 || pl.R - 1 + pl.QB > 64) return false;
    pl.NR = NR;
    pl.segs = istft_ws_segments(n_sig, pl.Q, pl.R, cus);
    pl.QS = (pl.Q + pl.segs - 1) / pl.segs;
    pl.segs = (pl.Q + pl.QS - 1) / pl.QS;                       // no empty segment
    *plo = pl;
    *lds = bytes(NR);
    *rj = RJ;
    *vec = VEC;
    *nitems = n_sig * pl.segs;
    return true;
}

template <int NC>
static int launch_istft_ws(const float2* spec, const kpr_stft_geom* s, long long F, const float* synth,
                           const float2* tw, float* out, hipStream_t st, bool* launched) {
    constexpr int L = NC / kPts, G = 64 / L;
    *launched = false;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    IstftWsPlan pl;
    size_t lds;
    int RJ, VEC;
    long long nitems;
    if (!istft_ws_plan(s, F, out, NC, G, kIwProd * (G - 1), 0, cus, 4, &pl, &lds, &RJ, &VEC, &nitems)) return 0;
    const unsigned grid = (unsigned)std::min<long long>(nitems, cus);
    *launched = true;
    switch (RJ) {
        case 2:  return launch_istft_ws_inst<NC, 2>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
        case 4:  return launch_istft_ws_inst<NC, 4>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
        default: return launch_istft_ws_inst<NC, 8>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
    }
}

// ---- k_istft_pw: every wave a complete worker, the overlap-add in registers (kpr_istft_pw_kernels.h) ---------------------
template <int NC, int S, bool IL>
static int launch_istft_pw_inst(const float2* spec, const IstftPwPlan& pl_in, unsigned grid, const float* synth,
                                const float2* tw, float* out, hipStream_t st) {
    constexpr int W = 16, NSTR = W * (64 / (NC / kPts));
    IstftPwPlan pl = pl_in;
    // as many LDS stashes for the partial head blocks as fit behind the exchange rows and tables (the streams of the first
    // frame run -- one per channel -- have no predecessor)
    pl.n_stash = (int)std::min<size_t>(NSTR - (IL ? pl.C : 1), (160 * 1024 - ipw_lds_bytes(NC, W)) / ipw_stash_bytes(NC, S));
    const size_t lds = ipw_lds_bytes(NC, W) + pl.n_stash * ipw_stash_bytes(NC, S);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_istft_pw<NC, S, W, IL>))) return e;
    if (opt(OPT_VERBOSE))
        fprintf(stderr, "[kapre_hip] k_istft_pw<%d,%d,%s>: grid %u, lds %zu B (%d stashes), %d segments per signal, %d items\n", NC, S,
                IL ? "interleaved" : "contiguous", grid, lds, pl.n_stash, pl.segs, pl.nitems);
    if (int e = status_word_ready()) return e;                  // (the kernel's bounded waits report there)
    hipLaunchKernelGGL((k_istft_pw<NC, S, W, IL>), dim3(grid), dim3(W * 64), lds, st, spec, pl, synth, tw, out);
    return launch_check(IL ? "k_istft_pw_il" : "k_istft_pw", NC, S == 2 ? "s2" : S == 4 ? "s4" : "s8");
}
template <int NC>
static int launch_istft_pw(const float2* spec, const kpr_stft_geom* s, long long F, const float* synth,
                           const float2* tw, float* out, hipStream_t st, bool* launched) {
    constexpr int L = NC / kPts, G = 64 / L, NSTR = 16 * G;
    *launched = false;
    const int win = s->win_length, hop = s->hop_length;
    // hop = S x (2 L samples), S = 2 / 4 / 8: the next frame's samples sit S register slots further down in the same lane
    if (hop % (2 * L) || win > 2 * NC || hop > win || F < 1) return 0;
    const int S = hop / (2 * L);
    if (S != 2 && S != 4 && S != 8) return 0;
    // contiguous waveform and contiguous spectrogram rows (channels_first, or one channel): a stream = a frame run of one
    // signal.  An interleaved side (channels_last, C > 1; round 4): the IL instances, streams = (frame run, channel) with
    // the channel fastest, an item = a segment of one batch item -- C a power of two that divides the streams of a workgroup,
    // hop = n_fft / 4 or / 2 (the instances that are built)
    const bool il = (s->in_layout == KPR_CHANNELS_LAST && s->channels > 1) || (s->out_layout == KPR_CHANNELS_LAST && s->channels > 1);
    const int C = s->channels;
    if (il && ((C & (C - 1)) != 0 || C > NSTR || S == 2)) return 0;
    const int runs = il ? NSTR / C : NSTR;                                       // frame runs per item
    const long long n_sig = il ? (long long)s->batch : (long long)s->batch * s->channels;
    const long long t_out = (F - 1) * (long long)hop + win;
    if (n_sig * 4096 >= (1LL << 31) || t_out + 2LL * NC >= (1LL << 31) || F >= (1LL << 30)) return 0;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    // segments per signal: every stream of a segment needs R - 1 frames of its own (run boundaries are two-party sums);
    // cost of a schedule = rounds of `cus` workgroups x (frames per stream + a fixed start)
    const int R = kPts / S;
    const long long need = (long long)runs * (R - 1);
    const int segs_max = (int)std::min<long long>(4096, F / need);
    if (segs_max < 1) return 0;
    int segs = 1;
    double best = 1e300;
    for (int sg = 1; sg <= segs_max; ++sg) {
        const long long rounds = (n_sig * sg + cus - 1) / cus;
        const double cost = (double)rounds * ((double)((F + sg - 1) / sg + R - 1 + runs - 1) / runs + 3.0);
        if (cost < best * 0.999) { best = cost; segs = sg; }
        if (n_sig * sg > 8LL * cus) break;
    }
    // fewer items than three quarters of the CUs: the ring kernel (32 x 434 frames at n_fft 1024 = 128 items: 30.8 vs 31.5 us)
    if (opt(OPT_ISTFT_PATH) == 0 && n_sig * segs * 4 < 3LL * cus) return 0;
    IstftPwPlan pl;
    pl.t_out = t_out; pl.F = (int)F; pl.win = win; pl.hop = hop; pl.segs = segs; pl.nitems = (int)(n_sig * segs);
    pl.seg_q = (int)(F / segs); pl.seg_r = (int)(F % segs);
    pl.n_stash = 0;
    pl.C = il ? C : 1;
    // (kpr_stft_geom: in_layout = the WAVEFORM's layout, out_layout = the SPECTROGRAM's -- for the inverse as well)
    pl.in_cl = (il && s->out_layout == KPR_CHANNELS_LAST) ? 1 : 0;              // the kernel's input: the spectrogram
    pl.out_cl = (il && s->in_layout == KPR_CHANNELS_LAST) ? 1 : 0;              // its output: the waveform
    const unsigned grid = (unsigned)std::min<long long>(pl.nitems, cus);
    *launched = true;
    if (il) return S == 4 ? launch_istft_pw_inst<NC, 4, true>(spec, pl, grid, synth, tw, out, st)
                          : launch_istft_pw_inst<NC, 8, true>(spec, pl, grid, synth, tw, out, st);
    switch (S) {
        case 2:  return launch_istft_pw_inst<NC, 2, false>(spec, pl, grid, synth, tw, out, st);
        case 4:  return launch_istft_pw_inst<NC, 4, false>(spec, pl, grid, synth, tw, out, st);
        default: return launch_istft_pw_inst<NC, 8, false>(spec, pl, grid, synth, tw, out, st);
    }
}

template <int NC, int MODE, bool OUT_CL>
static int launch_stft_inst(const float* x, const Geom& g, const float* window, const float2* tw,
                            void* out, hipStream_t st) {
    constexpr int L = NC / kPts, G = 64 / L;
    const long long ngroups = (g.total_frames + G - 1) / G;          // wave-loads of G frames
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = stft_lds_bytes(NC);
    // channels_last output with several channels (round 4): k_stft3 writes the G channel-frames of a wave as neighbours
    // (n_fft 1024, complex output, even channel count); everything else of that layout stays on k_stft
    const bool cl_ok = OUT_CL && MODE != KPR_OUT_PHASE && NC >= 512 && g.cfast && (g.C % G) == 0;
    if constexpr (MODE != KPR_OUT_PHASE && NC >= 512) if (!OUT_CL || cl_ok) {      // (the n_fft 256 / 512 instances spill at 128 VGPRs)
        // k_stft3: one sixteen-wave workgroup per CU drawing frame groups from an LDS counter (stft_variant 0 = automatic, from
        // 8 groups per CU up -- tools/sweep_dispatch.py stft: 13.5 groups per CU 12.2 vs 14.0 us, 2.6 per CU 10.3 vs 7.0 us
        // against k_stft; 3 = always; 1 = k_stft).  The round-3 kernel with static runs per wave (k_stft2) lost to one of the
        // two on every shape of the sweep (profiles/r05_sweep_before_prune.log) and was removed in round 5.
        if (opt(OPT_STFT_VARIANT) == 3 || (opt(OPT_STFT_VARIANT) == 0 && ngroups >= 8LL * cus)) {
            const size_t lds3 = stft3_lds_bytes(NC);
            static LdsOptIn lds_opt_in3;
            const unsigned grid3 = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + kStft3Waves - 1) / kStft3Waves, cus));
            {
                // the CL instance (channel-pair fetch at n_fft 1024; channels_last store) whenever a side is interleaved and
                // the G frames of a wave are channels of one (item, frame) -- also for interleaved input with
                // channels_first output (per-row stores there); n_fft 2048 (one frame per wave): channels_last output only
                if (g.cfast && (g.C % G) == 0 && (NC == 512 || OUT_CL)) {
                    static LdsOptIn lds_opt_in3c;
                    if (int e = allow_big_lds(lds_opt_in3c, reinterpret_cast<const void*>(&k_stft3<NC, MODE, true>))) return e;
                    hipLaunchKernelGGL((k_stft3<NC, MODE, true>), dim3(grid3), dim3(64 * kStft3Waves), lds3, st, x, g, window, tw, out,
                                       (int)(ngroups / grid3), (int)(ngroups % grid3));
                    return launch_check("k_stft3_cl", NC, MODE == KPR_OUT_COMPLEX ? "complex" : "magnitude");
                }
            }
            if (int e = allow_big_lds(lds_opt_in3, reinterpret_cast<const void*>(&k_stft3<NC, MODE, false>))) return e;
            hipLaunchKernelGGL((k_stft3<NC, MODE, false>), dim3(grid3), dim3(64 * kStft3Waves), lds3, st, x, g, window, tw, out,
                               (int)(ngroups / grid3), (int)(ngroups % grid3));
            return launch_check("k_stft3", NC, MODE == KPR_OUT_COMPLEX ? "complex" : "magnitude");
        }
    }
    // workgroups the hardware can keep resident per CU (registers + LDS), asked from the runtime
    static int resident_dev[64] = {0};
    int dev = 0;
    KPR_HIP(hipGetDevice(&dev));
    int& resident = resident_dev[(dev >= 0 && dev < 64) ? dev : 0];
    if (!resident) {
        KPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stft<NC, MODE, OUT_CL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int nb = 0;
        KPR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_stft<NC, MODE, OUT_CL>,
                                                             64 * KPR_STFT_WAVES, lds));
        resident = std::max(1, nb);
        if (opt(OPT_VERBOSE))
            fprintf(stderr, "[kapre_hip] k_stft<%d,%d,%d>: %d resident workgroups per CU (lds %zu B)\n", NC, MODE,
                    (int)OUT_CL, resident, lds);
    }
    // at least one group per wave when there is enough work
    const unsigned grid = (unsigned)std::max<long long>(
        1, std::min<long long>((ngroups + KPR_STFT_WAVES - 1) / KPR_STFT_WAVES, (long long)resident * cus));
    hipLaunchKernelGGL((k_stft<NC, MODE, OUT_CL>), dim3(grid), dim3(64 * KPR_STFT_WAVES), lds, st, x, g,
                       window, tw, out, ngroups, g_debug_stamps);
    return launch_check("k_stft", NC, OUT_CL ? (MODE == KPR_OUT_COMPLEX ? "complex,cl" : MODE == KPR_OUT_MAGNITUDE ? "magnitude,cl" : "phase,cl")
                                             : (MODE == KPR_OUT_COMPLEX ? "complex" : MODE == KPR_OUT_MAGNITUDE ? "magnitude" : "phase"));
}

template <int NC>
static int launch_stft_fast(const float* x, const Geom& g, const float* window, const float2* tw,
                            int mode, void* out, hipStream_t st) {
    const bool cl = g.out_cl != 0;
    switch (mode) {
        case KPR_OUT_COMPLEX:
            return cl ? launch_stft_inst<NC, KPR_OUT_COMPLEX, true>(x, g, window, tw, out, st)
                      : launch_stft_inst<NC, KPR_OUT_COMPLEX, false>(x, g, window, tw, out, st);
        case KPR_OUT_MAGNITUDE:
            return cl ? launch_stft_inst<NC, KPR_OUT_MAGNITUDE, true>(x, g, window, tw, out, st)
                      : launch_stft_inst<NC, KPR_OUT_MAGNITUDE, false>(x, g, window, tw, out, st);
        default:
            return cl ? launch_stft_inst<NC, KPR_OUT_PHASE, true>(x, g, window, tw, out, st)
                      : launch_stft_inst<NC, KPR_OUT_PHASE, false>(x, g, window, tw, out, st);
    }
}

// n_fft 4096 / 8192: R = 2 / 4 sub-FFTs of 1024 points per frame (k_stft_big)
static bool big_nfft(int n_fft) { return n_fft == 4096 || n_fft == 8192; }

template <int R>
static int launch_stft_big_inst(const float* x, const Geom& g, const float* window, int mode, void* out, hipStream_t st) {
    constexpr int NW = (R == 2) ? 4 : 2;
    const float2 *tw2048 = nullptr, *twbig = nullptr;
    if (int e = get_twiddles(2048, &tw2048)) return e;
    if (int e = get_twiddles(g.n_fft, &twbig)) return e;
    const size_t lds = sizeof(float) * 2 * (size_t)NW * (R * 1024 + 1);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_stft_big<R>))) return e;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((g.total_frames + NW - 1) / NW, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_stft_big<R>), dim3(grid), dim3(64 * NW), lds, st, x, g, window, tw2048, twbig, mode, out);
    return launch_check("k_stft_big");
}

static int launch_stft_big(const float* x, const Geom& g, const float* window, int mode, void* out, hipStream_t st) {
    return g.n_fft == 4096 ? launch_stft_big_inst<2>(x, g, window, mode, out, st)
                           : launch_stft_big_inst<4>(x, g, window, mode, out, st);
}

template <int R>
static int launch_irfft_big_inst(const float2* spec, const Geom& g, const float* synth, float* frames, hipStream_t st) {
    constexpr int NW = (R == 2) ? 4 : 2;
    const float2 *tw2048 = nullptr, *twbig = nullptr;
    if (int e = get_twiddles(2048, &tw2048)) return e;
    if (int e = get_twiddles(g.n_fft, &twbig)) return e;
    const size_t lds = sizeof(float) * 2 * (size_t)(2 * NW) * (R * 1024 + 1);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_irfft_big<R>))) return e;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((g.total_frames + NW - 1) / NW, cus));
    hipLaunchKernelGGL((k_irfft_big<R>), dim3(grid), dim3(64 * NW), lds, st, spec, g, synth, tw2048, twbig, frames);
    return launch_check("k_irfft_big");
}

static int launch_irfft_big(const float2* spec, const Geom& g, const float* synth, float* frames, hipStream_t st) {
    return g.n_fft == 4096 ? launch_irfft_big_inst<2>(spec, g, synth, frames, st)
                           : launch_irfft_big_inst<4>(spec, g, synth, frames, st);
}

// Bluestein STFT (even n_fft that is not a power of two, n_fft <= 1024, win_length <= n_fft)
static bool bluestein_ok(const kpr_stft_geom* s) {
    return !fast_nfft(s->n_fft) && bluestein_m(s->n_fft) > 0 && s->win_length <= s->n_fft;
}

template <int M>
static int launch_stft_bs_m(const float* x, const Geom& g, const float* window, const float2* tw,
                            const float2* bs, int mode, void* out, hipStream_t st) {
    constexpr int L = M / kPts, G = 64 / L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = sizeof(float) * ((size_t)4 * G * bs_slot_words(M, g.n_fft / 2) + 2 * (size_t)(3 * M + g.n_fft / 2 + 2));
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_stft_bs<M>))) return e;
    const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_stft_bs<M>), dim3(grid), dim3(256), lds, st, x, g, window, tw, bs, mode, out, ngroups);
    return launch_check("k_stft_bs");
}

// Mixed-radix plans (kpr_fft_mr.h).  n_fft = 2^a 5^b: MrFft<R2, R3>, N = n_fft / 2 = 20 * R2 * R3;
// n_fft with a factor 3: TwoPassFft<N1, N2>, N = N1 * N2.
typedef MrFft<4, 1> Fft160;    typedef MrFft<5, 1> Fft200;    typedef MrFft<4, 2> Fft320;
typedef MrFft<10, 1> Fft400;   typedef MrFft<4, 4> Fft640;    typedef MrFft<20, 1> Fft800;
typedef MrFft<5, 5> Fft1000;
typedef TwoPassFft<8, 6> Fft96;     typedef TwoPassFft<4, 15> Fft120;   typedef TwoPassFft<8, 12> Fft192;
typedef TwoPassFft<8, 15> Fft240;   typedef TwoPassFft<12, 15> Fft360;  typedef TwoPassFft<16, 12> Fft384;
typedef TwoPassFft<16, 15> Fft480;  typedef TwoPassFft<20, 15> Fft600;  typedef TwoPassFft<15, 24> Fft720;
typedef TwoPassFft<16, 24> Fft768;  typedef TwoPassFft<20, 24> Fft960;
// (the smaller factor first where it matters: N1 values per lane are prefetched one ticket ahead, twice
//  over in the inverse kernels, and <24, .> spilled there)

// 1: MrFft plan (forward, inverse and ring-ISTFT kernels), 2: TwoPassFft plan (forward and inverse), 0: none
static int mixed_radix_plan(int n_fft) {
    switch (n_fft) {
        case 160: case 200: case 320: case 400: case 640: case 800: case 1000: return 1;
        case 96: case 120: case 192: case 240: case 360: case 384: case 480: case 600: case 720: case 768:
        case 960: return 2;
        default: return 0;
    }
}
#define KPR_MR_CASES(FN, ...)                                                                      \
    case 160: return FN<Fft160>(__VA_ARGS__);   case 200: return FN<Fft200>(__VA_ARGS__);          \
    case 320: return FN<Fft320>(__VA_ARGS__);   case 400: return FN<Fft400>(__VA_ARGS__);          \
    case 640: return FN<Fft640>(__VA_ARGS__);   case 800: return FN<Fft800>(__VA_ARGS__);          \
    case 1000: return FN<Fft1000>(__VA_ARGS__);
#define KPR_2P_CASES(FN, ...)                                                                      \
    case 96: return FN<Fft96>(__VA_ARGS__);     case 120: return FN<Fft120>(__VA_ARGS__);          \
    case 192: return FN<Fft192>(__VA_ARGS__);   case 240: return FN<Fft240>(__VA_ARGS__);          \
    case 360: return FN<Fft360>(__VA_ARGS__);   case 384: return FN<Fft384>(__VA_ARGS__);          \
    case 480: return FN<Fft480>(__VA_ARGS__);   case 600: return FN<Fft600>(__VA_ARGS__);          \
    case 720: return FN<Fft720>(__VA_ARGS__);   case 768: return FN<Fft768>(__VA_ARGS__);          \
    case 960: return FN<Fft960>(__VA_ARGS__);

template <class FF>
static size_t mr_lds_bytes() {
    constexpr int G = 64 / FF::L;
    return sizeof(float) * 2 * ((size_t)4 * G * mr_row_stride<FF>() + 3 * (size_t)FF::N);
}

template <class FF>
static int launch_stft_mr_inst(const float* x, const Geom& g, const float* window, const float2* tw, int mode,
                               void* out, hipStream_t st) {
    constexpr int G = 64 / FF::L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = mr_lds_bytes<FF>();
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_stft_mr<FF>))) return e;
    const int per_cu = std::max(1, std::min(3, (int)(160 * 1024 / lds)));   // ~150 VGPRs: three workgroups per CU
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_stft_mr<FF>), dim3(grid), dim3(256), lds, st, x, g, window, tw, mode, out, ngroups);
    return launch_check("k_stft_mr", FF::N);
}

static int launch_stft_mr(const float* x, const Geom& g, const float* window, int mode, void* out, hipStream_t st) {
    const float2* tw = nullptr;
    if (int e = get_twiddles(g.n_fft, &tw)) return e;
    switch (g.n_fft) {
        KPR_MR_CASES(launch_stft_mr_inst, x, g, window, tw, mode, out, st)
        KPR_2P_CASES(launch_stft_mr_inst, x, g, window, tw, mode, out, st)
        default: return fail(KPR_E_UNSUPPORTED, "no mixed-radix plan for n_fft %d", g.n_fft);
    }
}

static int launch_stft_bs(const float* x, const Geom& g, const float* window, int mode, void* out,
                          hipStream_t st) {
    // sizes with a mixed-radix plan: one N-point FFT per frame instead of two chirp-z FFTs
    if (mixed_radix_plan(g.n_fft) && opt(OPT_MIXED_RADIX)) return launch_stft_mr(x, g, window, mode, out, st);
    const int m = bluestein_m(g.n_fft);
    const float2 *tw = nullptr, *bs = nullptr;
    if (int e = get_twiddles(2 * m, &tw)) return e;
    if (int e = get_bluestein(g.n_fft, &bs)) return e;
    switch (m) {
        case 128:  return launch_stft_bs_m<128>(x, g, window, tw, bs, mode, out, st);
        case 256:  return launch_stft_bs_m<256>(x, g, window, tw, bs, mode, out, st);
        case 512:  return launch_stft_bs_m<512>(x, g, window, tw, bs, mode, out, st);
        default:   return launch_stft_bs_m<1024>(x, g, window, tw, bs, mode, out, st);
    }
}

template <int M>
static int launch_irfft_bs_m(const float2* spec, const Geom& g, const float* synth, const float2* tw,
                             const float2* bs, float* frames, hipStream_t st) {
    constexpr int L = M / kPts, G = 64 / L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = sizeof(float) * ((size_t)4 * G * ((M + M / 32 + 24 + 3) / 4 * 4) +
                                        2 * (size_t)(3 * M + g.n_fft / 2 + 2));
    const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_irfft_bs<M>), dim3(grid), dim3(256), lds, st, spec, g, synth, tw, bs, frames, ngroups);
    return launch_check("k_irfft_bs");
}

template <class FF>
static int launch_irfft_mr_inst(const float2* spec, const Geom& g, const float* synth, const float2* tw,
                                float* frames, hipStream_t st) {
    constexpr int G = 64 / FF::L;
    const long long ngroups = (g.total_frames + G - 1) / G;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const size_t lds = mr_lds_bytes<FF>();
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_irfft_mr<FF>))) return e;
    const int per_cu = std::max(1, std::min(2, (int)(160 * 1024 / lds)));
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + 3) / 4, (long long)per_cu * cus));
    hipLaunchKernelGGL((k_irfft_mr<FF>), dim3(grid), dim3(256), lds, st, spec, g, synth, tw, frames, ngroups);
    return launch_check("k_irfft_mr");
}

static int launch_irfft_mr(const float2* spec, const Geom& g, const float* synth, float* frames, hipStream_t st) {
    const float2* tw = nullptr;
    if (int e = get_twiddles(g.n_fft, &tw)) return e;
    switch (g.n_fft) {
        KPR_MR_CASES(launch_irfft_mr_inst, spec, g, synth, tw, frames, st)
        KPR_2P_CASES(launch_irfft_mr_inst, spec, g, synth, tw, frames, st)
        default: return fail(KPR_E_UNSUPPORTED, "no mixed-radix plan for n_fft %d", g.n_fft);
    }
}

template <class FF, int RJ, int VEC>
static int launch_istft_ws_mr_inst(const float2* spec, const IstftWsPlan& pl, size_t lds, unsigned grid,
                                   const float* synth, const float2* tw, float* out, int nitems, hipStream_t st) {
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_istft_ws_mr<FF, RJ, VEC>))) return e;
    if (int e = status_word_ready()) return e;                  // (the kernel's bounded waits report there)
    hipLaunchKernelGGL((k_istft_ws_mr<FF, RJ, VEC>), dim3(grid), dim3(kIwThreads), lds, st, spec, pl, synth, tw,
                       out, nitems);
    return launch_check("k_istft_ws_mr", FF::N, RJ == 2 ? (VEC == 4 ? "rj2,v4" : "rj2,v2") : RJ == 4 ? (VEC == 4 ? "rj4,v4" : "rj4,v2")
                                                                                   : (VEC == 4 ? "rj8,v4" : "rj8,v2"));
}

template <class FF>
static int launch_istft_ws_mr_plan(const float2* spec, const kpr_stft_geom* s, long long F, const float* synth,
                                   const float2* tw, float* out, hipStream_t st, bool* launched) {
    constexpr int G = 64 / FF::L;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    IstftWsPlan pl;
    size_t lds;
    int RJ, VEC;
    long long nitems;
    // rows double as exchange rows (ROW complex words); window pairs and the twiddle table behind the counters
    if (!istft_ws_plan(s, F, out, 2 * FF::ROW, G, 0, sizeof(float) * 2 * 3 * (size_t)FF::N, cus, 2, &pl, &lds, &RJ, &VEC,
                       &nitems))
        return 0;
    if (RJ > 4) return 0;                                       // more than four overlapping frames: two-kernel path
    // hop or win_length not a multiple of four: the consumer would sum and store 8 bytes per lane.  Those 18 instances
    // (k_istft_ws_mr<FF, 4, 2>) were removed in round 6: on 64 x 10 s @ 16 kHz they beat irFFT + overlap-add as two kernels by at
    // most 1.4x (n_fft 200 / hop 50: 156 vs 214 us, 1000 / 250: 154 vs 189, 120 / 30: 156 vs 187) and LOST at 360 / 90 (205 vs
    // 138) and 600 / 150 (245 vs 199) -- profiles/r06_sweep_before_prune.log; VERDICT r05 item 7 asked for > 1.5x.
    if (VEC == 2) return 0;
    const unsigned grid = (unsigned)std::min<long long>(nitems, cus);
    *launched = true;
    if (RJ == 2) return launch_istft_ws_mr_inst<FF, 2, 4>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
    return launch_istft_ws_mr_inst<FF, 4, 4>(spec, pl, lds, grid, synth, tw, out, (int)nitems, st);
}

// ring kernel for the mixed-radix transform sizes; *launched stays false when it does not apply
static int launch_istft_ws_mr(const float2* spec, const kpr_stft_geom* s, long long F, const float* synth,
                              float* out, hipStream_t st, bool* launched) {
    *launched = false;
    if (!mixed_radix_plan(s->n_fft) || !opt(OPT_MIXED_RADIX) || s->win_length > s->n_fft) return 0;
    const float2* tw = nullptr;
    if (int e = get_twiddles(s->n_fft, &tw)) return e;
    switch (s->n_fft) {
        KPR_MR_CASES(launch_istft_ws_mr_plan, spec, s, F, synth, tw, out, st, launched)
        KPR_2P_CASES(launch_istft_ws_mr_plan, spec, s, F, synth, tw, out, st, launched)
        default: return 0;
    }
}

static int launch_irfft_bs(const float2* spec, const Geom& g, const float* synth, float* frames,
                           hipStream_t st) {
    if (mixed_radix_plan(g.n_fft) && opt(OPT_MIXED_RADIX)) return launch_irfft_mr(spec, g, synth, frames, st);
    const int m = bluestein_m(g.n_fft);
    const float2 *tw = nullptr, *bs = nullptr;
    if (int e = get_twiddles(2 * m, &tw)) return e;
    if (int e = get_bluestein(g.n_fft, &bs)) return e;
    switch (m) {
        case 128:  return launch_irfft_bs_m<128>(spec, g, synth, tw, bs, frames, st);
        case 256:  return launch_irfft_bs_m<256>(spec, g, synth, tw, bs, frames, st);
        case 512:  return launch_irfft_bs_m<512>(spec, g, synth, tw, bs, frames, st);
        default:   return launch_irfft_bs_m<1024>(spec, g, synth, tw, bs, frames, st);
    }
}

template <int NC>
static int launch_irfft_fast(const float2* spec, const Geom& g, const float* synth,
                             const float2* tw, float* frames, hipStream_t st) {
    constexpr int L = NC / kPts, G = 64 / L;
    const long long nblocks = (g.total_frames + 4 * G - 1) / (4 * G);
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const unsigned grid = (unsigned)std::min<long long>(nblocks, 2LL * cus);
    const size_t lds = sizeof(float) * (4 * G * NC + 4 * G * (2 * NC + 8));
    hipLaunchKernelGGL((k_irfft<NC>), dim3(grid), dim3(256), lds, st, spec, g, synth, tw, frames,
                       nblocks);
    return launch_check("k_irfft");
}


// Per 16-filter tile: the k range [lo, hi) the fused kernel walks, padded to whole chunks of
// kChunkRows rows inside [0, mel_row_cap(K)] (rows outside the caller's exact-zero range hold
// zeros for this tile, rows >= K do not exist and are packed as zeros).
static int tile_ranges(int K, int M, const int32_t* kr_host, int* lo_out, int* hi_out) {
    const int ntiles = (M + 15) / 16;
    if (ntiles > kMaxTiles)
        return fail(KPR_E_UNSUPPORTED, "n_filt=%d exceeds the %d-filter limit", M, kMaxTiles * 16);
    const int kp = (K + 3) & ~3;
    const int cap = mel_row_cap(K);
    for (int t = 0; t < ntiles; ++t) {
        int lo = 0, hi = kp;
        if (kr_host) {
            lo = kr_host[2 * t]; hi = kr_host[2 * t + 1];
            if (lo < 0 || hi > kp || lo > hi || (lo & 3) || (hi & 3))
                return fail(KPR_E_BADARG, "bad filterbank k-range for tile %d: [%d,%d)", t, lo, hi);
        }
        lo &= ~7;       // whole 8-row blocks (a tile's first row stays 32-byte aligned within the magnitude row)
        int need = std::max(kChunkRows, (hi - lo + kChunkRows - 1) / kChunkRows * kChunkRows);
        hi = std::min(cap, lo + need);
        lo = std::max(0, hi - need);
        if (hi - lo != need)
            return fail(KPR_E_UNSUPPORTED, "n_freq=%d too small for the fused kernel", K);
        lo_out[t] = lo; hi_out[t] = hi;
    }
    return 0;
}

static int build_sched(int K, int M, const int32_t* kr_host, MelSched* sch) {
    int lo[kMaxTiles], hi[kMaxTiles];
    if (int e = tile_ranges(K, M, kr_host, lo, hi)) return e;
    const int ntiles = (M + 15) / 16;
    sch->M = M;
    sch->ntiles = ntiles;
    int total = 0;
    for (int t = 0; t < ntiles; ++t) {
        sch->klo[t] = (short)lo[t]; sch->khi[t] = (short)hi[t];
        sch->chunk0[t] = (unsigned short)total;          // packed layout: tiles in natural order
        total += (hi[t] - lo[t]) / kChunkRows;
    }
    // 4 equal contiguous slices of the chunk stream; tiles straddling a cut are split
    int cut[5];
    for (int w = 0; w <= 4; ++w) cut[w] = (int)(((long long)total * w + 2) / 4);
    cut[0] = 0; cut[4] = total;
    int nseg = 0, w = 0;
    sch->wave_seg0[0] = 0;
    for (int t = 0; t < ntiles; ++t) {
        int c = sch->chunk0[t];
        const int cend = c + (hi[t] - lo[t]) / kChunkRows;
        sch->t_s0[t] = (unsigned char)nseg;
        sch->t_ns[t] = 0;
        while (c < cend) {
            while (w < 3 && c >= cut[w + 1]) { ++w; sch->wave_seg0[w] = nseg; }
            const int e = std::min(cend, cut[w + 1] > c ? cut[w + 1] : cend);
            if (nseg >= kMaxSegs) return fail(KPR_E_UNSUPPORTED, "too many filterbank segments");
            sch->seg_tile[nseg] = (unsigned char)t;
            sch->seg_nch[nseg] = e - c;
            sch->seg_k0[nseg] = lo[t] + (c - sch->chunk0[t]) * kChunkRows;
            ++sch->t_ns[t];
            ++nseg;
            c = e;
        }
    }
    while (w < 4) { ++w; sch->wave_seg0[w] = nseg; }
    sch->nseg = nseg;
    for (int i = 0; i < 4; ++i) {
        sch->wave_chunk0[i] = (unsigned short)cut[i];
        sch->wave_nchunks[i] = (unsigned short)(cut[i + 1] - cut[i]);
    }
    return 0;
}

// ---- packed filterbank: 64-float header + MFMA fragments + band plan -----------------------------
// header words (uint32): [0] magic 'KPFB' [1] n_freq [2] n_filt [3] tiles [4] chunks [5] k-range hash
//   [6] float offset of the band plan section from the start of the blob (0: the matrix has no band plan)
//   [7] L (lanes per frame the plan is laid out for) [8] NR [9] CMQ [10] partial sums per frame [11] section words
constexpr int kPackHeaderFloats = 64;
constexpr uint32_t kPackMagic = 0x4b504642u;

// Band plan of k_mel_pw (kpr_mel_pw_kernels.h has the arithmetic): possible when n_freq - 1 is 128 ... 1024 (the fused
// power-of-two sizes) and every bin below Nyquist has at most two non-zeros, in neighbouring filters a(k), a(k) + 1 with
// a(k) non-decreasing -- mel banks of any scale / normalisation, any triangular bank.  Returns the section size in words
// (0: no plan) and fills `sec` (capacity pw_section_cap(K) words) and the header fields.
// lanes per row of a band plan for K = n_freq rows: the K - 1 bins below Nyquist in groups of 16 per lane, L a power of two (8 ...
// 64).  The fused kernel k_mel_pw needs K - 1 == 16 L exactly (n_fft 256 ... 2048); the stand-alone kernel k_fb_pw takes any K - 1
// that is a multiple of four (round 6: n_fft 400 -> 200 bins on 16 lanes, 13 of them in use; the bins beyond K - 1 have zero weights
// and are never loaded).  0: no plan for this K.
static int pw_plan_lanes(int K) {
    const int nb = K - 1;
    if (nb < 4 || nb > 1024 || (nb & 3)) return 0;
    int L = 8;
    while (kPts * L < nb) L *= 2;
    return L;
}
static int pw_section_cap(int K) {
    const int L = pw_plan_lanes(K);
    return L ? kPwEmaskWords + pw_table_words(L, kPwMaxRounds, kPwMaxCmq) : 0;
}
static int build_band_plan(const float* fb, int K, int M, uint32_t* sec, uint32_t* hdr_fields /* [7..11] */) {
    const int cap = pw_section_cap(K);
    if (!cap) return 0;
    const int NB = K - 1;                                             // bins below Nyquist
    const int L = pw_plan_lanes(K), NC = kPts * L, G = 64 / L;       // NC >= NB: the plan's padded bin count
    const int NR = (M + L - 1) / L;
    if (NR > kPwMaxRounds) return 0;
    std::vector<int> a(NC);
    std::vector<float> w0(NC, 0.0f), w1(NC, 0.0f);
    int prev = 0;
    for (int k = 0; k < NC; ++k) {
        int idx[3], n = 0;
        for (int m = 0; k < NB && m < M && n < 3; ++m) {              // (bins NB ... NC - 1 do not exist: no weights)
            const float v = fb[(size_t)k * M + m];
            if (v != 0.0f || v != v) idx[n++] = m;
        }
        if (n > 2) return 0;
        if (n == 2 && idx[1] != idx[0] + 1) return 0;
        if (n == 0) a[k] = prev;
        else if (n == 2) {
            if (idx[0] < prev) return 0;
            a[k] = idx[0];
            w0[k] = fb[(size_t)k * M + idx[0]];
            w1[k] = fb[(size_t)k * M + idx[1]];
        } else {
            const int m = idx[0];
            const float v = fb[(size_t)k * M + m];
            if (m == prev) { a[k] = prev; w0[k] = v; }
            else if (m == prev + 1) { a[k] = prev; w1[k] = v; }       // (keeps the running segment going)
            else if (m > prev + 1) { a[k] = m; w0[k] = v; }
            else return 0;
        }
        prev = a[k];
    }
    // partial sums: a lane's 16 bins, cut where a(k) changes; listed lane by lane = in bin order, so the partial sums of one
    // segment are neighbours in the list
    unsigned long long emask[16] = {0};
    std::vector<int> first(M + 1, 0), cnt(M + 1, 0), P(L, 0);
    int nlist = 0;
    for (int fl = 0; fl < L; ++fl) {
        P[fl] = 8 * nlist;
        bool any = false;                                             // the running piece has a non-zero weight
        for (int i = 0; i < 16; ++i) {
            const int k = 16 * fl + i;
            any = any || w0[k] != 0.0f || w1[k] != 0.0f;
            // a piece whose weights are all zero leaves the accumulators at zero: nothing to append (bins outside
            // [f_min, f_max] would otherwise form one piece per lane of a very long "segment")
            if ((i == 15 || a[k + 1] != a[k]) && any) {
                emask[i] |= 1ull << fl;
                if (cnt[a[k]] == 0) first[a[k]] = nlist;
                ++cnt[a[k]];
                ++nlist;
                any = false;
            }
        }
    }
    // the list grows from the start of the row (every magnitude, Nyquist included, is in registers by then) and must stop
    // short of the zero words at its end
    if (2 * nlist > pw_zero_word(NC)) return 0;
    int cm = 1;
    for (int m = 0; m < M; ++m) cm = std::max(cm, cnt[m]);
    const int CMQ = (cm + 3) / 4;
    if (CMQ > kPwMaxCmq) return 0;
    for (int i = 0; i < 16; ++i) {
        unsigned long long e = 0;
        for (int gq = 0; gq < G; ++gq) e |= emask[i] << (L * gq);
        sec[2 * i] = (uint32_t)(e & 0xffffffffull);
        sec[2 * i + 1] = (uint32_t)(e >> 32);
    }
    uint32_t* tab = sec + kPwEmaskWords;
    auto putf = [](uint32_t* p, float v) { std::memcpy(p, &v, 4); };
    for (int fl = 0; fl < L; ++fl)
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * fl + 2 * j + (e >> 1);
                putf(&tab[(j * L + fl) * 4 + e], (e & 1) ? w1[k] : w0[k]);
            }
    for (int fl = 0; fl < L; ++fl) tab[32 * L + fl] = (uint32_t)P[fl];
    const uint32_t zoff = 4u * (uint32_t)pw_zero_word(NC);
    for (int r = 0; r < NR; ++r)
        for (int fl = 0; fl < L; ++fl) {
            const int m = fl + L * r;
            putf(&tab[33 * L + r * L + fl], m < M ? fb[(size_t)NB * M + m] : 0.0f);     // the Nyquist row
            for (int q = 0; q < CMQ; ++q)
                for (int e = 0; e < 4; ++e) {
                    const int sidx = 4 * q + e;
                    uint32_t ou = zoff, od = zoff;
                    if (m < M && sidx < cnt[m]) ou = 8u * (uint32_t)(first[m] + sidx);
                    if (m >= 1 && m < M && sidx < cnt[m - 1]) od = 8u * (uint32_t)(first[m - 1] + sidx) + 4u;
                    tab[(33 + NR) * L + (((r * CMQ + q) * L) + fl) * 4 + e] = ou | (od << 16);
                }
        }
    hdr_fields[0] = (uint32_t)L; hdr_fields[1] = (uint32_t)NR; hdr_fields[2] = (uint32_t)CMQ; hdr_fields[3] = (uint32_t)nlist;
    hdr_fields[4] = (uint32_t)(kPwEmaskWords + pw_table_words(L, NR, CMQ));
    return (int)hdr_fields[4];
}

static uint32_t kranges_hash(int K, int M, const int32_t* kr_host) {
    uint32_t h = 0x811c9dc5u;
    auto mix = [&h](uint32_t v) { for (int i = 0; i < 4; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 16777619u; } };
    mix((uint32_t)K); mix((uint32_t)M);
    if (kr_host) for (int i = 0; i < 2 * ((M + 15) / 16); ++i) mix((uint32_t)kr_host[i]);
    else mix(0xdeadbeefu);
    return h;
}

struct SchedKey {
    int K, M; uint32_t h;
    bool operator<(const SchedKey& o) const { return K != o.K ? K < o.K : M != o.M ? M < o.M : h < o.h; }
};
static std::map<SchedKey, MelSched> g_sched;                          // built once per filterbank geometry
struct PackInfo { uint32_t band_off, L, NR, CMQ, nlist; };          // what the verified header says about the band plan
static std::map<std::pair<const void*, SchedKey>, PackInfo> g_pack_ok;    // packed blobs already verified

// schedule of (K, M, k-ranges): cached, so the steady-state call does no host work beyond a lookup
static int get_sched(int K, int M, const int32_t* kr_host, MelSched* out, uint32_t* hash_out = nullptr) {
    if ((M + 15) / 16 > kMaxTiles || K > 32000)            // 64 tiles; row bounds are stored as 16-bit values
        return fail(KPR_E_UNSUPPORTED, "filterbank %d x %d exceeds the packed schedule (%d filters, 32000 rows)", K, M,
                    kMaxTiles * 16);
    const SchedKey key{K, M, kranges_hash(K, M, kr_host)};
    if (hash_out) *hash_out = key.h;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_sched.find(key);
    if (it == g_sched.end()) {
        MelSched sch;
        if (int e = build_sched(K, M, kr_host, &sch)) return e;
        if (g_sched.size() > 256) g_sched.clear();
        it = g_sched.emplace(key, sch).first;
    }
    *out = it->second;
    return 0;
}

// A packed blob must describe the SAME matrix geometry and k-ranges as the call's arguments: its header
// (written by kpr_filterbank_pack) is read back from the device ONCE per (pointer, geometry, k-ranges)
// -- a 32-byte copy on the call's stream on first use, waited for, so that it is ordered after an upload of the
// blob on that stream (not legal during stream capture: warm up first, as for the table uploads) -- and compared.
// Mismatch = BADARG.  Best effort by design: the cache is keyed on the device address, so a different buffer that
// later lands on the same address with the same geometry arguments is not re-read.
static int verify_packed(const float* fb_packed, int K, int M, const int32_t* kr_host, const MelSched& sch,
                         hipStream_t st, PackInfo* info = nullptr) {
    const SchedKey key{K, M, kranges_hash(K, M, kr_host)};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_pack_ok.find({fb_packed, key});
        if (it != g_pack_ok.end()) { if (info) *info = it->second; return 0; }
    }
    uint32_t hdr[12] = {0};
    KPR_HIP(hipMemcpyAsync(hdr, fb_packed, sizeof(hdr), hipMemcpyDeviceToHost, st));
    KPR_HIP(hipStreamSynchronize(st));
    int chunks = 0;
    for (int t = 0; t < sch.ntiles; ++t) chunks += (sch.khi[t] - sch.klo[t]) / kChunkRows;
    if (hdr[0] != kPackMagic)
        return fail(KPR_E_BADARG, "fb_packed does not start with a kpr_filterbank_pack header");
    if ((int)hdr[1] != K || (int)hdr[2] != M || (int)hdr[3] != sch.ntiles || (int)hdr[4] != chunks || hdr[5] != key.h)
        return fail(KPR_E_BADARG, "fb_packed was packed for another filterbank (%u x %u, %u tiles, %u chunks, "
                    "k-range hash %08x) than this call describes (%d x %d, %d tiles, %d chunks, hash %08x)",
                    hdr[1], hdr[2], hdr[3], hdr[4], hdr[5], K, M, sch.ntiles, chunks, key.h);
    PackInfo pi{hdr[6], hdr[7], hdr[8], hdr[9], hdr[10]};
    if (pi.band_off) {                                          // a band plan this build cannot run = no band plan
        const int NC = kPts * pw_plan_lanes(K);                       // (the plan's padded bin count; 0: no plan for this K)
        const bool sane = pw_section_cap(K) && (int)pi.L == pw_plan_lanes(K) && (int)pi.NR == (M + (int)pi.L - 1) / (int)pi.L &&
                          pi.CMQ >= 1 && (int)pi.CMQ <= kPwMaxCmq && pi.band_off == (uint32_t)(kPackHeaderFloats + chunks * 512) &&
                          hdr[11] == (uint32_t)(kPwEmaskWords + pw_table_words((int)pi.L, (int)pi.NR, (int)pi.CMQ)) &&
                          2 * (int)pi.nlist <= pw_zero_word(NC);                 // (the partial-sum list fits in front of the zero words)
        if (!sane) pi = PackInfo{0, 0, 0, 0, 0};
    }
    if (info) *info = pi;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pack_ok.size() > 1024) g_pack_ok.clear();
    g_pack_ok[{fb_packed, key}] = pi;
    return 0;
}

template <int NC, bool FROM_MAG, bool RES, bool LD8 = false>
static int launch_mel_ws_inst(const float* x, const Geom& g, const float* window, const float2* tw,
                              const float* fbp, const MelSched& sch, const DbDev& db, unsigned* stats,
                              float* out, hipStream_t st) {
    const size_t lds = mel_ws_lds_bytes(NC, sch.nseg, FROM_MAG ? 2 : 1);     // (the LD8 form uses one group's worth less)
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_mel_ws<NC, FROM_MAG, RES, LD8>))) return e;
    const long long ntiles = (g.total_frames + kFT - 1) / kFT;
    if (ntiles > 0x7fffffffLL) return fail(KPR_E_UNSUPPORTED, "too many frames");
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    constexpr int G = 64 / (NC / kPts);
    constexpr int RF = kWsProd * G;                                // frames per round
    const long long nrounds = (g.total_frames + RF - 1) / RF;
    const unsigned grid = (unsigned)std::min<long long>(nrounds, cus);         // 1 workgroup / CU
    const long long tickets = (g.total_frames + G - 1) / G;                    // a ticket = G frames (one wave's round)
    if (int e = status_word_ready()) return e;                  // (the kernel's bounded waits report there)
    hipLaunchKernelGGL((k_mel_ws<NC, FROM_MAG, RES, LD8>), dim3(grid), dim3(kWsThreads), lds, st, x, g, window, tw, fbp,
                       sch, db, stats, out, (int)(tickets / grid), (int)(tickets % grid), g_debug_stamps);
    return launch_check("k_mel_ws", NC);
}

// fbp: the fragment section of the packed blob (behind its header)
template <int NC, bool FROM_MAG = false>
static int launch_mel_ws(const float* x, const Geom& g, const float* window, const float2* tw,
                         const float* fbp, const MelSched& sch, const DbDev& db, unsigned* stats,
                         float* out, hipStream_t st) {
    if constexpr (!FROM_MAG) {
        // every consumer wave's slice fits the register-resident form (mel banks: 37 chunks at 1025 x 128)?
        int slice_max = 0;
        for (int i = 0; i < 4; ++i) slice_max = std::max(slice_max, (int)sch.wave_nchunks[i]);
        if (slice_max <= kWsResident && opt(OPT_MEL_VARIANT) != 2)
            return launch_mel_ws_inst<NC, false, true>(x, g, window, tw, fbp, sch, db, stats, out, st);
    }
    if constexpr (FROM_MAG) {
        if (g.K > 512) return launch_mel_ws_inst<NC, true, false, true>(x, g, window, tw, fbp, sch, db, stats, out, st);   // wide rows: 8 loaders
    }
    return launch_mel_ws_inst<NC, FROM_MAG, false>(x, g, window, tw, fbp, sch, db, stats, out, st);
}

// ---- k_mel_ts: schedule (whole (frame tile, filter tile) items per wave, heavy filter tiles cut) + launch ----------
// Builds the per-wave chunk-entry table described at MelSchedTs (host copy in *tab).
// FT = 16-frame tiles per round, S = magnitude row stride (floats), nwaves = waves per workgroup of the kernel the table
// is for: k_mel_ts (8 waves; entry word 1 = float offset 16 ft S + k0) or k_mel_mr (4 waves, S = 0: entry word 1 =
// k0 | ft << 16, the kernel forms the row address itself).
static int build_sched_ts(int K, int M, const int32_t* kr_host, int FT, int S, int nwaves, MelSchedTs* sch,
                          std::vector<unsigned>* tab) {
    int lo[kMaxTiles], hi[kMaxTiles];
    const int ntiles = (M + 15) / 16;
    if (ntiles > kTsMaxTiles) return fail(KPR_E_UNSUPPORTED, "filterbank too wide for k_mel_ts");
    if (int e = tile_ranges(K, M, kr_host, lo, hi)) return e;
    std::memset(sch, 0, sizeof(*sch));
    sch->M = M; sch->ntiles = ntiles; sch->FT = FT;
    int total = 0, nch[kTsMaxTiles], chunk0[kTsMaxTiles + 1];
    for (int t = 0; t < ntiles; ++t) {
        chunk0[t] = total;
        nch[t] = (hi[t] - lo[t]) / kChunkRows;
        total += nch[t];
    }
    chunk0[ntiles] = total;
    if (total > 60000) return fail(KPR_E_UNSUPPORTED, "filterbank too wide for k_mel_ts");
    // parts: a filter tile with more chunks than an even share of the round's work is cut into near-equal parts
    const int share = std::max(1, (FT * total + nwaves - 1) / nwaves);
    std::vector<MelItemTs> parts;
    int nslots = 0;
    for (int ft = 0; ft < FT; ++ft)
        for (int t = 0; t < ntiles; ++t) {
            const int np = std::min(4, (nch[t] + share - 1) / share);
            int c = chunk0[t];
            const int slot0 = nslots;
            for (int pi = 0; pi < np; ++pi) {
                const int n = (nch[t] * (pi + 1)) / np - (nch[t] * pi) / np;
                if (n > 255) return fail(KPR_E_UNSUPPORTED, "filterbank too wide for k_mel_ts");
                MelItemTs im{};
                im.ft = (unsigned char)ft; im.t = (unsigned char)t; im.nch = (unsigned char)n; im.c0 = (unsigned short)c;
                if (pi == 0) { im.kind = 0; im.slot0 = (unsigned char)slot0; im.nslots = (unsigned char)(np - 1); }
                else { im.kind = 1; im.slot0 = (unsigned char)nslots++; im.nslots = 0; }
                parts.push_back(im);
                c += n;
            }
        }
    if (nslots > kTsMaxSlots) return fail(KPR_E_UNSUPPORTED, "too many cut filter tiles for k_mel_ts");
    sch->nslots = nslots;
    // longest first onto the least loaded SIMD (waves w and w + 4 share one), then onto its less loaded wave; a wave takes
    // at most one owner of a cut tile (its accumulators stay live across the second barrier)
    std::vector<int> order(parts.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return parts[a].nch > parts[b].nch; });
    int wload[kTsWaves] = {0}, wown[kTsWaves] = {0};
    std::vector<int> mine[kTsWaves];
    for (int idx : order) {
        const MelItemTs& im = parts[idx];
        const bool cut_owner = im.kind == 0 && im.nslots > 0;
        int best = -1;
        for (int w = 0; w < nwaves; ++w) {
            if (wload[w] + im.nch > kTsMaxEnt) continue;
            if (cut_owner && wown[w]) continue;
            if (best < 0) { best = w; continue; }
            const int sl = wload[w & 3] + wload[(w & 3) + 4], bl = wload[best & 3] + wload[(best & 3) + 4];   // (zero beyond nwaves)
            if (sl < bl || (sl == bl && wload[w] < wload[best])) best = w;
        }
        if (best < 0) return fail(KPR_E_UNSUPPORTED, "too many filterbank chunks for k_mel_ts");
        mine[best].push_back(idx);
        wload[best] += im.nch;
        if (cut_owner) wown[best] = 1;
    }
    tab->assign(8 + 3 * kTsWaves * kTsMaxEnt, 0u);
    for (int w = 0; w < nwaves; ++w) {
        // order per wave: parts, whole tiles, the owner of a cut tile last
        auto rank = [&](int i) { const MelItemTs& im = parts[i]; return im.kind == 1 ? 0 : (im.nslots == 0 ? 1 : 2); };
        std::stable_sort(mine[w].begin(), mine[w].end(), [&](int a, int b) { return rank(a) < rank(b); });
        int n = 0;
        for (int idx : mine[w]) {
            const MelItemTs& im = parts[idx];
            for (int i = 0; i < im.nch; ++i, ++n) {
                unsigned* e = tab->data() + 8 + 3 * (w * kTsMaxEnt + n);
                const bool last = i + 1 == im.nch;
                e[0] = (unsigned)(im.c0 + i) | (last ? 0x80000000u : 0u);
                const unsigned k0 = (unsigned)(lo[im.t] + kChunkRows * (im.c0 + i - chunk0[im.t]));
                e[1] = S ? (unsigned)(16 * im.ft * S) + k0 : (k0 | (unsigned)im.ft << 16);
                e[2] = last ? ((unsigned)im.kind | (unsigned)im.t << 1 | (unsigned)im.ft << 5 | (unsigned)im.slot0 << 8 |
                               (unsigned)im.nslots << 16) : 0u;
            }
        }
        (*tab)[w] = (unsigned)n;
    }
    return 0;
}

struct SchedTsKey {
    int dev, K, M, FT, S, nwaves; uint32_t h;
    bool operator<(const SchedTsKey& o) const {
        return std::tie(dev, K, M, FT, S, nwaves, h) < std::tie(o.dev, o.K, o.M, o.FT, o.S, o.nwaves, o.h);
    }
};
static std::map<SchedTsKey, MelSchedTs> g_sched_ts;      // entries own a small device table (kept for the process lifetime)

static int get_sched_ts(int K, int M, const int32_t* kr_host, int FT, int S, int nwaves, MelSchedTs* out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    const SchedTsKey key{dev, K, M, FT, S, nwaves, kranges_hash(K, M, kr_host)};
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_sched_ts.find(key);
    if (it == g_sched_ts.end()) {
        MelSchedTs sch;
        std::vector<unsigned> tab;
        if (int e = build_sched_ts(K, M, kr_host, FT, S, nwaves, &sch, &tab)) return e;
        unsigned* d = nullptr;
        KPR_HIP(hipMalloc(&d, tab.size() * sizeof(unsigned)));
        KPR_HIP(hipMemcpy(d, tab.data(), tab.size() * sizeof(unsigned), hipMemcpyHostToDevice));   // first use only, like the twiddles
        sch.tab = d;
        it = g_sched_ts.emplace(key, sch).first;
    }
    *out = it->second;
    return 0;
}

// true when k_mel_ts can take the call (geometry of the filterbank schedule + LDS); *sch is filled then
static bool mel_ts_ok(int n_fft, int K, int M, const int32_t* kr_host, const Geom& g, MelSchedTs* sch, int RF = 0) {
    if (n_fft != 2048 && n_fft != 1024 && n_fft != 512 && n_fft != 256) return false;
    if ((M + 15) / 16 > kTsMaxTiles || g.total_frames >= 0x7fffff00LL) return false;
    const int NC = n_fft / 2;
    if (!RF) RF = mel_ts_rf(NC);
    if (get_sched_ts(K, M, kr_host, RF / 16, mel_ws_row_stride(NC + 1), kTsWaves, sch)) return false;
    return mel_ts_lds_bytes(NC, sch->nslots, RF) <= 80 * 1024;       // two workgroups per CU
}

template <int NC, int RF_ = 0>
static int launch_mel_ts(const float* x, const Geom& g, const float* window, const float2* tw, const float* fbp,
                         const MelSchedTs& sch, const DbDev& db, unsigned* stats, float* out, hipStream_t st) {
    constexpr int G = 64 / (NC / kPts), RF = RF_ ? RF_ : mel_ts_rf(NC);
    const size_t lds = mel_ts_lds_bytes(NC, sch.nslots, RF);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_mel_ts<NC, RF_>))) return e;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const long long tickets = (g.total_frames + G - 1) / G;                     // a ticket = G frames (the unit the runs are cut at)
    const long long nrounds = (g.total_frames + RF - 1) / RF;
    const unsigned grid = (unsigned)std::min<long long>(nrounds, 2LL * cus);    // 2 workgroups / CU
    hipLaunchKernelGGL((k_mel_ts<NC, RF_>), dim3(grid), dim3(kTsWaves * 64), lds, st, x, g, window, tw, fbp, sch, db, stats, out,
                       (int)(tickets / grid), (int)(tickets % grid), g_debug_stamps);
    return launch_check("k_mel_ts", NC);
}

// ---- k_mel_pw: every wave owns its frames end to end, banded mel sums (kpr_mel_pw_kernels.h) -------------------------
template <int NC, int W>
static int launch_mel_pw(const float* x, const Geom& g, const float* window, const float2* tw, const float* blob,
                         const PackInfo& pi, int M, const DbDev& db, unsigned* stats, float* out, hipStream_t st) {
    constexpr int L = NC / kPts, G = 64 / L;
    PwPlan pl{(int)pi.L, (int)pi.NR, (int)pi.CMQ, (int)pi.nlist, M, reinterpret_cast<const unsigned*>(blob) + pi.band_off,
              reinterpret_cast<const unsigned*>(blob), pi.band_off, 0, 0};
    const size_t lds = pw_lds_bytes(NC, W, pl.NR, pl.CMQ);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_mel_pw<NC, W>))) return e;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const long long tickets = (g.total_frames + G - 1) / G;                     // a ticket = G frames of one wave
    const int per_cu = std::max(1, std::min(16 / W, (int)(160 * 1024 / lds)));  // sixteen waves per CU (128 VGPRs)
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((tickets + W - 1) / W, (long long)per_cu * cus));
    if (opt(OPT_VERBOSE))
        fprintf(stderr, "[kapre_hip] k_mel_pw<%d,%d>: grid %u, lds %zu B, NR %d CMQ %d list %d, %lld tickets\n", NC, W, grid, lds,
                pl.NR, pl.CMQ, pl.nlist, tickets);
    if (int e = status_word_ready()) return e;                  // (a stale band plan is reported there)
    hipLaunchKernelGGL((k_mel_pw<NC, W>), dim3(grid), dim3(W * 64), lds, st, x, g, window, tw, pl, db, stats, out,
                       (int)(tickets / grid), (int)(tickets % grid), g_debug_stamps);
    return launch_check("k_mel_pw", NC, W == 4 ? "w4" : W == 8 ? "w8" : "w16");
}
// the PAIR form (interleaved waveforms, even channel count: two channel-frames per fetch; three waves per SIMD)
template <int NC>
static int launch_mel_pw_pair(const float* x, const Geom& g, const float* window, const float2* tw, const float* blob,
                              const PackInfo& pi, int M, const DbDev& db, unsigned* stats, float* out, hipStream_t st) {
    constexpr int L = NC / kPts, G = 64 / L, W = 12;
    PwPlan pl{(int)pi.L, (int)pi.NR, (int)pi.CMQ, (int)pi.nlist, M, reinterpret_cast<const unsigned*>(blob) + pi.band_off,
              reinterpret_cast<const unsigned*>(blob), pi.band_off, 0, 0};
    // channels_last output with C >= 4: the M x C block of an (item, frame) through a ring of LDS slots, stored as one contiguous
    // run by the last of its C / 2 pair-waves (kpr_mel_pw_kernels.h; n_fft 2048 only).  A power of two, whatever fits the CU's LDS, and
    // at least the blocks the W waves can hold pairs of (need / 2: progress is guaranteed by the ticket order with any ring of one
    // slot or more -- tests/test_cl_ring_model.py -- but waves prefetch a ticket ahead, so with fewer slots than 2x the in-flight
    // span the bounded slot wait is taken routinely, not exceptionally); else (or C = 2, whose 8-byte pairs are contiguous
    // anyway) the 8-byte stores.
    const int CP = g.C / 2;
    if (G == 1 && g.out_cl && CP >= 2 && g.C <= 64 && opt(OPT_MEL_CL_STAGE) != 0 && (M * g.C) % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int blk = M * g.C;
        const int need = 2 * ((W * G + CP - 1) / CP) + 2;
        int slots = 32;
        while (slots >= 4 && pw_lds_bytes(NC, W, pl.NR, pl.CMQ, true, slots, blk) > 160 * 1024) slots >>= 1;
        if (slots >= 4 && slots >= need / 2) { pl.cl_slots = slots; pl.cl_blk = blk; }
    }
    const size_t lds = pw_lds_bytes(NC, W, pl.NR, pl.CMQ, true, pl.cl_slots, pl.cl_blk);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_mel_pw<NC, W, true>))) return e;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const long long pairs = g.total_frames / 2;
    long long tickets = (pairs + G - 1) / G;                                    // a ticket = G channel pairs of one wave
    const int unit = pl.cl_slots ? CP : 1;                                      // staged: workgroups own whole blocks (CP tickets = G blocks)
    tickets = (tickets + unit - 1) / unit;                                      // ... counted in units from here on
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((tickets * unit + W - 1) / W, (long long)cus));
    if (opt(OPT_VERBOSE))
        fprintf(stderr, "[kapre_hip] k_mel_pw<%d,%d,pair>: grid %u, lds %zu B, %lld ticket units of %d, %d output slots\n", NC, W, grid, lds,
                tickets, unit, pl.cl_slots);
    if (int e = status_word_ready()) return e;                  // (a stale band plan is reported there)
    hipLaunchKernelGGL((k_mel_pw<NC, W, true>), dim3(grid), dim3(W * 64), lds, st, x, g, window, tw, pl, db, stats, out,
                       (int)(tickets / grid), (int)(tickets % grid), g_debug_stamps);
    return launch_check("k_mel_pw_pair", NC);
}
template <int NC>
static int launch_mel_pw_w(int w, const float* x, const Geom& g, const float* window, const float2* tw, const float* blob,
                           const PackInfo& pi, int M, const DbDev& db, unsigned* stats, float* out, hipStream_t st) {
    switch (w) {
        case 4:  return launch_mel_pw<NC, 4>(x, g, window, tw, blob, pi, M, db, stats, out, st);
        case 16: return launch_mel_pw<NC, 16>(x, g, window, tw, blob, pi, M, db, stats, out, st);
        default: return launch_mel_pw<NC, 8>(x, g, window, tw, blob, pi, M, db, stats, out, st);
    }
}

// ---- k_fb_pw: the stand-alone ApplyFilterbank as banded row sums (kpr_fb_pw_kernels.h; round 6) -------------------------
// ST: two interleaved channels (channels_last, C = 2): `rows` counts (item, frame) blocks of K x 2 floats
template <int NC, bool ST>
static int launch_fb_pw(const float* x, long long rows, int K, const float* blob, const PackInfo& pi, int M, const float* fb, float* out,
                        hipStream_t st) {
    constexpr int L = NC / kPts, G = 64 / L;
    PwPlan pl{(int)pi.L, (int)pi.NR, (int)pi.CMQ, (int)pi.nlist, M, reinterpret_cast<const unsigned*>(blob) + pi.band_off,
              reinterpret_cast<const unsigned*>(blob), pi.band_off, 0, 0};
    const size_t lds = fb_pw_lds_bytes(NC, pl.NR, pl.CMQ, ST);
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_fb_pw<NC, ST>))) return e;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const long long tickets = (rows + G - 1) / G;                               // a ticket = G rows (ST: blocks) of one wave
    const int per_cu = std::max(1, std::min(16 / kFbW, (int)(160 * 1024 / lds)));   // sixteen waves per CU (eight: 22.5 vs 20.7 us)
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((tickets + kFbW - 1) / kFbW, (long long)per_cu * cus));
    if (opt(OPT_VERBOSE))
        fprintf(stderr, "[kapre_hip] k_fb_pw<%d%s>: grid %u, lds %zu B, NR %d CMQ %d list %d, %lld tickets\n", NC, ST ? ",st" : "", grid, lds,
                pl.NR, pl.CMQ, pl.nlist, tickets);
    if (int e = status_word_ready()) return e;                  // (a stale band plan is reported there)
    hipLaunchKernelGGL((k_fb_pw<NC, ST>), dim3(grid), dim3(kFbW * 64), lds, st, x, rows, K, M, pl, fb, out, (int)(tickets / grid),
                       (int)(tickets % grid));
    return ST ? launch_check("k_fb_pw", NC, "st") : launch_check("k_fb_pw", NC);
}
template <bool ST>
static int launch_fb_pw_l(int lanes, const float* x, long long rows, int K, const float* blob, const PackInfo& pi, int M, const float* fb,
                          float* out, hipStream_t st) {
    switch (lanes) {
        case 8:  return launch_fb_pw<128, ST>(x, rows, K, blob, pi, M, fb, out, st);
        case 16: return launch_fb_pw<256, ST>(x, rows, K, blob, pi, M, fb, out, st);
        case 32: return launch_fb_pw<512, ST>(x, rows, K, blob, pi, M, fb, out, st);
        default: return launch_fb_pw<1024, ST>(x, rows, K, blob, pi, M, fb, out, st);
    }
}

// ---- k_mel_mr: the same schedule for the mixed-radix sizes (four-wave workgroups, up to three per CU) ---------------
static bool mel_mr_nfft(int n_fft) { return mixed_radix_plan(n_fft) != 0; }
template <class FF>
static int launch_mel_mr_inst(const float* x, const Geom& g, const float* window, const float2* tw, const float* fbp,
                              const int32_t* kr_host, int M, const DbDev& db, unsigned* stats, float* out,
                              hipStream_t st, bool* taken) {
    constexpr int G = 64 / FF::L, RF = mel_mr_rf<FF>();
    *taken = false;
    MelSchedTs sch;
    if (get_sched_ts(FF::N + 1, M, kr_host, mel_mr_nt<FF>(), 0, kMrWaves, &sch)) return 0;   // no schedule: the caller's other path
    const size_t lds = mel_mr_lds_bytes<FF>(sch.nslots);
    if (lds > 160 * 1024) return 0;
    *taken = true;
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_mel_mr<FF>))) return e;
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const long long tickets = (g.total_frames + G - 1) / G;                      // runs are cut at G-frame granularity
    const long long nrounds = (g.total_frames + RF - 1) / RF;
    const int per_cu = std::max(1, std::min(12 / kMrWaves, (int)(160 * 1024 / lds)));      // 12 waves per CU at 168 VGPRs
    const unsigned grid = (unsigned)std::min<long long>(nrounds, (long long)per_cu * cus);
    hipLaunchKernelGGL((k_mel_mr<FF>), dim3(grid), dim3(kMrWaves * 64), lds, st, x, g, window, tw, fbp, sch, db, stats, out,
                       (int)(tickets / grid), (int)(tickets % grid), g_debug_stamps);
    return launch_check("k_mel_mr", FF::N);
}
static int launch_mel_mr(const float* x, const Geom& g, const float* window, const float* fbp, const int32_t* kr_host,
                         int M, const DbDev& db, unsigned* stats, float* out, hipStream_t st, bool* taken) {
    *taken = false;
    if ((M + 15) / 16 > kTsMaxTiles || g.total_frames >= 0x7fffff00LL) return 0;
    const float2* tw = nullptr;
    if (int e = get_twiddles(g.n_fft, &tw)) return e;
    switch (g.n_fft) {
        case 160:  return launch_mel_mr_inst<Fft160>(x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken);
        case 200:  return launch_mel_mr_inst<Fft200>(x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken);
        case 320:  return launch_mel_mr_inst<Fft320>(x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken);
        case 400:  return launch_mel_mr_inst<Fft400>(x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken);
        case 640:  return launch_mel_mr_inst<Fft640>(x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken);
        case 800:  return launch_mel_mr_inst<Fft800>(x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken);
        case 1000: return launch_mel_mr_inst<Fft1000>(x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken);
        KPR_2P_CASES(launch_mel_mr_inst, x, g, window, tw, fbp, kr_host, M, db, stats, out, st, taken)
        default:   return 0;
    }
}

// blocks per item of the decibel passes: enough blocks to fill the GPU (about 2048), at least 4096
// floats each, and few per item (every block ends with two atomics on the item's statistics)
static int db_chunks(long long n_items, long long item_size) {
    const long long want = (2048 + n_items - 1) / std::max<long long>(1, n_items);
    return (int)std::max<long long>(1, std::min<long long>(std::min<long long>(256, want), item_size / 4096));
}

static int db_clamp(float* out, long long n_items, long long item_size, float dyn,
                    const unsigned* stats, hipStream_t st, int slots = 1) {
    if (n_items <= 0 || item_size <= 0) return 0;
    const int chunks = db_chunks(n_items, item_size);
    const int stride = (int)(2 * n_items);
    if ((((uintptr_t)out) & 15) == 0)            // 16-byte accesses on the aligned middle of every chunk
        hipLaunchKernelGGL(k_db_clamp<4>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, out,
                           item_size, chunks, dyn, stats, slots, stride);
    else
        hipLaunchKernelGGL(k_db_clamp<1>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, out,
                           item_size, chunks, dyn, stats, slots, stride);
    return launch_check("k_db_clamp");
}

// banded filterbank product on contiguous |X| rows (k_band_mel) + the decibel clamp pass
static int run_band_mel(const float* mag, const Geom& g, const float* fb, const MelSched& sch, const DbDev& dbd,
                        unsigned* stats, float* out, long long batch, long long item_size, hipStream_t st) {
    // (k_band_mel's own atomics use slot 0; the other slots keep their initial values and drop out of the reduction)
    const size_t lds = sizeof(float) * (size_t)kBandRows * g.K;
    static LdsOptIn lds_opt_in;
    if (int e = allow_big_lds(lds_opt_in, reinterpret_cast<const void*>(&k_band_mel))) return e;
    const long long nsteps = (g.total_frames + kBandRows - 1) / kBandRows;
    hipLaunchKernelGGL(k_band_mel, dim3(grid_1d(nsteps, 1, 256 * 8)), dim3(256), lds, st, mag, g, fb, sch, dbd, stats, out);
    if (int e = launch_check("k_band_mel")) return e;
    return dbd.enabled ? db_clamp(out, batch, item_size, dbd.dyn, stats, st, dbd.slot_mask + 1) : 0;
}

}  // namespace kpr

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace kpr;

// ---- backward passes (kpr_grad_kernels.h): launch helpers of the C entry points at the end of this file ----
template <typename T>
static int run_cplx_bwd(const void* x, const T* g, int64_t n, int phase, void* gx, kpr_stream_t stream) {
    if (n < 0) return fail(KPR_E_BADARG, "negative element count");
    if (n == 0) return 0;
    if (!x || !g || !gx) return fail(KPR_E_BADARG, "x / g / gx must not be NULL");
    hipLaunchKernelGGL(k_cplx_to_real_bwd<T>, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const GCplx<T>*)x, g, (long long)n, phase, (GCplx<T>*)gx);
    return launch_check("k_cplx_to_real_bwd");
}
template <typename T>
static int run_edge_scale(const void* in, int64_t n, int n_freq, int inner, int n_fft, T s_edge, T s_mid, void* out,
                          kpr_stream_t stream) {
    if (n < 0 || n_freq <= 0 || inner <= 0 || n_fft <= 0) return fail(KPR_E_BADARG, "bad sizes");
    if (n_freq != n_fft / 2 + 1) return fail(KPR_E_BADARG, "n_freq %d is not n_fft / 2 + 1 (n_fft %d)", n_freq, n_fft);
    if (n % ((int64_t)n_freq * inner)) return fail(KPR_E_BADARG, "element count is not a multiple of n_freq * inner");
    if (n == 0) return 0;
    if (!in || !out) return fail(KPR_E_BADARG, "in / out must not be NULL");
    const int nyq = (n_fft & 1) ? -1 : n_fft / 2;
    hipLaunchKernelGGL(k_spec_edge_scale<T>, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const GCplx<T>*)in, (long long)n, n_freq, inner, nyq, s_edge, s_mid, (GCplx<T>*)out);
    return launch_check("k_spec_edge_scale");
}
template <typename T>
static int run_db_bwd(const T* x, const T* gy, int64_t n_items, int64_t item_size, double ref_value, double amin,
                      double dynamic_range, T* gx, kpr_stream_t stream) {
    if (n_items < 0 || item_size < 0) return fail(KPR_E_BADARG, "negative size");
    // same checks (and order) as backend.py:168-173
    if (!(ref_value > 0)) return fail(KPR_E_BADARG, "ref_value must be positive");
    if (!(amin > 0)) return fail(KPR_E_BADARG, "amin must be positive");
    if (!(dynamic_range > 0)) return fail(KPR_E_BADARG, "dynamic_range must be positive");
    if (n_items == 0 || item_size == 0) return 0;
    if (!x || !gy || !gx) return fail(KPR_E_BADARG, "x / gy / gx must not be NULL");
    if (n_items > 0x7fffffffLL) return fail(KPR_E_UNSUPPORTED, "decibel backward: more than 2^31 - 1 items");
    const double ref_term = 10.0 * std::log10(std::max(amin, ref_value));
    // float32: the forward (make_db / to_db) raises amin to the smallest normal float -- the backward floors at the same value
    const double amin_k = sizeof(T) == 4 ? std::max(amin, 1.17549435e-38) : amin;
    hipLaunchKernelGGL(k_db_bwd<T>, dim3((unsigned)n_items), dim3(1024), 0, (hipStream_t)stream, x, gy,
                       (long long)item_size, (T)amin_k, (T)ref_term, (T)dynamic_range, gx);
    return launch_check("k_db_bwd");
}
extern "C" {

int kpr_version(void) { return KPR_VERSION; }

const char* kpr_last_launches(void) { return g_launches.c_str(); }

int kpr_device_status(unsigned* flags_out) {
    unsigned bits = 0;
    if (unsigned* hostp = g_status.host.load(std::memory_order_acquire)) bits = __atomic_exchange_n(hostp, 0u, __ATOMIC_ACQ_REL);
    if (flags_out) *flags_out = bits;
    if (bits & kStStalePlan) {                               // whatever was cached about packed blobs is re-read from the device
        std::lock_guard<std::mutex> lk(g_mu);
        g_pack_ok.clear();
    }
    if (bits) return fail(KPR_E_DEVICE, "kernels raised the device status word (%s)", status_text(bits));
    return 0;
}

// development / tests: a kernel whose wait can never end, with a limit of 64 polls -- the whole reporting chain without a protocol bug
__global__ void k_spin_selftest() {
    __shared__ int flag;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    int spin = 0;
    for (; spin < 64 && __hip_atomic_load(&flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 1; ++spin) __builtin_amdgcn_s_sleep(2);
    if (__builtin_expect(spin >= 64, 0)) status_raise(kStSelfTest);
}
int kpr_debug_spin_timeout(kpr_stream_t stream) {
    if (int e = status_word_ready()) return e;
    hipLaunchKernelGGL(k_spin_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream);
    return launch_check("k_spin_selftest");
}

static int option_id(const char* name) {
    static const char* const names[OPT_COUNT] = {"mel_variant", "istft_path", "mixed_radix", "db_chunks", "verbose", "stft_variant", "db_slots",
                                                  "mel_cl_stage", "fb_variant"};
    if (name)
        for (int i = 0; i < OPT_COUNT; ++i)
            if (std::strcmp(name, names[i]) == 0) return i;
    return -1;
}

int kpr_set_option(const char* name, int value) {
    const int id = option_id(name);
    if (id < 0) return fail(KPR_E_BADARG, "unknown option '%s'", name ? name : "(null)");
    static const int lo[OPT_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, hi[OPT_COUNT] = {8, 4, 1, 4096, 1, 3, 32, 1, 1};
    if (value < lo[id] || value > hi[id])
        return fail(KPR_E_BADARG, "option '%s': value %d outside [%d, %d]", name, value, lo[id], hi[id]);
    // kernels removed in round 5 (dominated on every shape of tools/sweep_dispatch.py): the 4-wave ring kernel k_mel_fused
    // (mel_variant 1) and the static-run STFT kernel k_stft2 (stft_variant 2)
    if ((id == OPT_MEL_VARIANT && value == 1) || (id == OPT_STFT_VARIANT && value == 2))
        return fail(KPR_E_BADARG, "option '%s': value %d names a kernel that was removed (KPR_VERSION >= 110)", name, value);
    g_opt[id].store(value, std::memory_order_relaxed);
    return 0;
}

int kpr_get_option(const char* name, int* value) {
    const int id = option_id(name);
    if (id < 0 || !value) return fail(KPR_E_BADARG, "unknown option '%s'", name ? name : "(null)");
    *value = opt(id);
    return 0;
}

/* diagnostics (declared in include/kapre_hip.h): device buffer of 12*32 + 1 int64: cycle stamps written by
 * the waves of one workgroup of k_mel_ws (the one whose index is stored in the last element; k_mel_fused
 * and k_stft: workgroup 0, 4*32 entries); NULL disables */
int kpr_debug_stamps(void* dev_buf) { g_debug_stamps = (long long*)dev_buf; return 0; }

/* development aid: known-traffic kernel for calibrating the FETCH_SIZE counter (reads n*8 bytes) */
int kpr_debug_calib_read8(const void* x, int64_t n_float2, float* out, kpr_stream_t stream) {
    hipLaunchKernelGGL(k_calib_read8, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (long long)n_float2, out);
    return launch_check("k_calib_read8");
}

/* development aid: shader clock (MHz) under a dense packed-f32 load: mean over the waves of 2 workgroups per CU running
 * ~100 us; out_mhz_host receives one float.  Blocking. */
int kpr_debug_sclk_mhz(float* out_mhz_host) {
    if (!out_mhz_host) return fail(KPR_E_BADARG, "out_mhz_host is NULL");
    int cus = 256;
    if (int e = device_cus(&cus)) return e;
    const int blocks = 2 * cus, n = blocks * 4;
    float* d = nullptr;
    KPR_HIP(hipMalloc(&d, n * sizeof(float)));
    hipLaunchKernelGGL(k_sclk, dim3(blocks), dim3(256), 0, 0, 600, d);          // warm-up (clock ramp)
    hipLaunchKernelGGL(k_sclk, dim3(blocks), dim3(256), 0, 0, 600, d);
    std::vector<float> h(n);
    hipError_t e1 = hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e1 != hipSuccess) return fail(KPR_E_HIP, "kpr_debug_sclk_mhz: %s", hipGetErrorString(e1));
    double s = 0;
    for (float v : h) s += v;
    *out_mhz_host = (float)(s / n);
    return 0;
}

/* ---- size-generic FFT engine (kpr_generic_kernels.h): plan + launch helpers --------------------------- */
// run-time radices of n: 4s first, then 2, then the odd primes; false when a prime factor exceeds 64 (a pass costs
// R multiply-adds per point: beyond that the DFT-as-GEMM path is the better fallback) or n is out of range
static bool gen_plan(int n, GenPlan* p) {
    if (n < 2) return false;
    p->n = n;
    p->npass = 0;
    int m = n;
    auto push = [&](int r) { if (p->npass < kGenMaxPasses) p->radix[p->npass] = r; ++p->npass; };
    while (m % 4 == 0) { push(4); m /= 4; }
    if (m % 2 == 0) { push(2); m /= 2; }
    for (int f = 3; f <= 64 && m > 1; f += 2)
        while (m % f == 0) { push(f); m /= f; }
    return m == 1 && p->npass <= kGenMaxPasses;
}

// the FFT length a frame is transformed with: n_fft / 2 for even sizes (real-FFT packing), n_fft for odd ones
static int gen_fft_len(int n_fft) { return (n_fft % 2 == 0 && n_fft >= 4) ? n_fft / 2 : n_fft; }

// LDS of one workgroup: two frame buffers of the FFT length, plus the n_fft-entry twiddle table when it fits as well
static bool gen_lds(size_t elem_bytes, int n_fft, size_t* lds, int* tw_lds) {
    const size_t buf = elem_bytes * (size_t)gen_fft_len(n_fft), tab = elem_bytes * (size_t)n_fft;
    if (2 * buf > 160 * 1024) return false;
    *tw_lds = 2 * buf + tab <= 160 * 1024;
    *lds = 2 * buf + (*tw_lds ? tab : 0);
    return true;
}

static bool gen_ok_f32(const kpr_stft_geom* s) {
    GenPlan p;
    size_t lds;
    int tl;
    return s->win_length <= s->n_fft && gen_plan(gen_fft_len(s->n_fft), &p) && gen_lds(sizeof(float2), s->n_fft, &lds, &tl);
}

static int gen_grid(const Geom& g, size_t lds) {
    const long long per_cu = std::max<long long>(1, std::min<long long>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
    return (int)std::max<long long>(1, std::min<long long>(g.total_frames, 256 * per_cu));
}

static int launch_stft_gen_f32(const float* x, const Geom& g, const float* window, int mode, void* out, hipStream_t st) {
    GenPlan p;
    size_t lds;
    int tl;
    if (!gen_plan(gen_fft_len(g.n_fft), &p) || !gen_lds(sizeof(float2), g.n_fft, &lds, &tl))
        return fail(KPR_E_UNSUPPORTED, "no generic FFT plan for n_fft %d", g.n_fft);
    const float2* tw = nullptr;
    if (int e = get_twiddles(g.n_fft, &tw)) return e;
    auto kern = tl ? &k_stft_gen<float, true> : &k_stft_gen<float, false>;
    static LdsOptIn opt_in_1[2];
    if (int e = allow_big_lds(opt_in_1[tl ? 1 : 0], reinterpret_cast<const void*>(kern))) return e;
    hipLaunchKernelGGL(kern, dim3(gen_grid(g, lds)), dim3(kF64Threads), lds, st, x, g, window, tw, p, mode, out);
    return launch_check("k_stft_gen<float>");
}

static int launch_irfft_gen_f32(const float2* spec, const Geom& g, const float* synth_window, float* frames,
                                hipStream_t st) {
    GenPlan p;
    size_t lds;
    int tl;
    if (!gen_plan(gen_fft_len(g.n_fft), &p) || !gen_lds(sizeof(float2), g.n_fft, &lds, &tl))
        return fail(KPR_E_UNSUPPORTED, "no generic FFT plan for n_fft %d", g.n_fft);
    const float2* tw = nullptr;
    if (int e = get_twiddles(g.n_fft, &tw)) return e;
    auto kern = tl ? &k_irfft_gen<float, true> : &k_irfft_gen<float, false>;
    static LdsOptIn opt_in_2[2];
    if (int e = allow_big_lds(opt_in_2[tl ? 1 : 0], reinterpret_cast<const void*>(kern))) return e;
    hipLaunchKernelGGL(kern, dim3(gen_grid(g, lds)), dim3(kF64Threads), lds, st, spec, g, synth_window, tw, p, frames);
    return launch_check("k_irfft_gen<float>");
}

const char* kpr_last_error(void) { return g_err.c_str(); }

int kpr_fft_fast_path(int n_fft) { return fast_nfft(n_fft) ? 1 : 0; }

// same order as the dispatch in kpr_stft_f32 / kpr_istft_f32
int kpr_fft_plan(int n_fft, int win_length) {
    if (n_fft < 2 || win_length < 1) return -1;
    if (fast_nfft(n_fft)) return KPR_FFT_POW2;
    kpr_stft_geom s{};
    s.batch = 1; s.channels = 1; s.time = n_fft; s.n_fft = n_fft; s.win_length = std::min(win_length, n_fft); s.hop_length = 1;
    if (bluestein_ok(&s)) {
        const int mr = mixed_radix_plan(n_fft);
        if (mr && opt(OPT_MIXED_RADIX)) return mr == 1 ? KPR_FFT_MIXED_RADIX : KPR_FFT_TWO_PASS;
        return KPR_FFT_BLUESTEIN;
    }
    if (big_nfft(n_fft) && win_length <= n_fft) return KPR_FFT_SUB_FFT;
    if (gen_ok_f32(&s)) return KPR_FFT_GENERIC;
    return KPR_FFT_DFT_GEMM;
}

int64_t kpr_num_frames(const kpr_stft_geom* s) {
    if (check_geom(s)) return -1;
    return frames_of(s);
}

int64_t kpr_stft_workspace_bytes(const kpr_stft_geom* s, int mode) {
    if (check_geom(s)) return -1;
    const long long F = frames_of(s);
    const kpr_stft_geom se = forward_geom(s);
    s = &se;
    if (fast_nfft(s->n_fft) || bluestein_ok(s) || mode == KPR_OUT_COMPLEX) return 0;
    if (big_nfft(s->n_fft) && s->win_length <= s->n_fft) return 0;     // k_stft_big writes |X| / phase itself
    if (gen_ok_f32(s)) return 0;                                        // so does the size-generic FFT kernel
    // DFT-GEMM path with a real-valued epilogue: complex spectrum staged in the workspace
    return (int64_t)sizeof(float) * 2 * s->batch * s->channels * F * (s->n_fft / 2 + 1);
}

int kpr_stft_f32(const float* x, const kpr_stft_geom* s, const float* window, void* out, int mode,
                 void* workspace, int64_t workspace_bytes, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (int e = check_geom(s)) return e;
    if (mode < 0 || mode > 2) return fail(KPR_E_BADARG, "bad output mode %d", mode);
    const long long F = frames_of(s);
    const kpr_stft_geom* s_call = s;
    const kpr_stft_geom se = forward_geom(s);                    // win_length > n_fft: cropped frames (see forward_geom)
    s = &se;
    Geom g = make_geom(s, F);
    if (g.total_frames == 0) return 0;
    if (!x || !window || !out) return fail(KPR_E_BADARG, "x / window / out must not be NULL");
    hipStream_t st = (hipStream_t)stream;
    if (fast_nfft(s->n_fft)) {
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        // channel-fastest frame numbering whenever either side is interleaved: the frames that share the waveform's cache
        // lines / the spectrogram's channel runs sit in one wave
        g.cfast = ((g.in_cl || g.out_cl) && g.C > 1) ? 1 : 0;
        switch (s->n_fft) {
            case 256:  return launch_stft_fast<128>(x, g, window, tw, mode, out, st);
            case 512:  return launch_stft_fast<256>(x, g, window, tw, mode, out, st);
            case 1024: return launch_stft_fast<512>(x, g, window, tw, mode, out, st);
            default:   return launch_stft_fast<1024>(x, g, window, tw, mode, out, st);
        }
    }
    if (bluestein_ok(s)) return launch_stft_bs(x, g, window, mode, out, st);
    if (big_nfft(s->n_fft) && s->win_length <= s->n_fft) return launch_stft_big(x, g, window, mode, out, st);
    // every other size with small prime factors (odd sizes, 1200, 1536, 2000 ...): run-time mixed-radix FFT
    if (gen_ok_f32(s)) return launch_stft_gen_f32(x, g, window, mode, out, st);
    if (mode == KPR_OUT_COMPLEX) return stft_gemm(x, s, g, window, (float*)out, false, st);
    const int64_t need = kpr_stft_workspace_bytes(s_call, mode);
    if (!workspace || workspace_bytes < need)
        return fail(KPR_E_WORKSPACE, "stft workspace: need %lld bytes", (long long)need);
    if (int e = stft_gemm(x, s, g, window, (float*)workspace, false, st)) return e;
    const long long n = g.total_frames * g.K;
    hipLaunchKernelGGL(k_cplx_to_real, dim3(grid_1d(n, 256)), dim3(256), 0, st,
                       (const float2*)workspace, n, mode == KPR_OUT_PHASE ? 1 : 0, (float*)out);
    return launch_check("k_cplx_to_real");
}

static bool fused_nfft(int n_fft) { return n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

static int64_t stats_region_bytes(int64_t batch) {
    int64_t b = 256 + (int64_t)sizeof(unsigned) * 2 * std::max<int64_t>(1, batch) * db_slots_cap(batch);
    return (b + 255) & ~(int64_t)255;
}

int64_t kpr_mel_workspace_bytes(const kpr_stft_geom* s, int n_filt, const kpr_db_params* db) {
    if (check_geom(s) || n_filt <= 0) return -1;
    (void)db;
    int64_t bytes = stats_region_bytes(s->batch);
    // two-kernel path (stages the complex spectrum): every n_fft without a fused kernel, and filterbanks with more
    // 16-filter tiles than the tile-synchronous kernel's schedule holds (banks without a band plan of more than 256 filters are
    // beyond k_mel_ts, and beyond k_mel_ws whenever a consumer's slice exceeds 64 chunks -- always at n_fft 512; ADVICE r05: such a
    // call used to fail with KPR_E_WORKSPACE once k_mel_fused, the catch-all of round 1, was gone).  A bank of at most 256
    // filters that neither kernel's schedule holds (dense matrices) still needs kpr_mel_workspace_bytes_unpacked(): the call then
    // fails with KPR_E_WORKSPACE and names that size; the Python layer retries with it.
    if (!fused_nfft(s->n_fft) || (n_filt + 15) / 16 > kTsMaxTiles)
        bytes += (int64_t)sizeof(float) * 2 * s->batch * s->channels * frames_of(s) *
                 (s->n_fft / 2 + 1);
    return bytes;
}

int kpr_filterbank_forget(const float* fb_packed) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!fb_packed) { g_pack_ok.clear(); return 0; }
    for (auto it = g_pack_ok.begin(); it != g_pack_ok.end();)
        it = (it->first.first == (const void*)fb_packed) ? g_pack_ok.erase(it) : std::next(it);
    return 0;
}

int64_t kpr_filterbank_pack_floats(int n_freq, int n_filt, const int32_t* fb_kranges_host) {
    if (n_freq <= 0 || n_filt <= 0) return -1;
    MelSched sch;
    if (build_sched(n_freq, n_filt, fb_kranges_host, &sch)) return -1;
    int64_t chunks = 0;
    for (int t = 0; t < sch.ntiles; ++t) chunks += (sch.khi[t] - sch.klo[t]) / kChunkRows;
    return chunks * 512 + kPackHeaderFloats + pw_section_cap(n_freq);   // header | fp32 MFMA fragments | band plan
}

int kpr_filterbank_pack(const float* fb_host, int n_freq, int n_filt, const int32_t* fb_kranges_host,
                        float* out_host) {
    if (!fb_host || !out_host || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad arguments to kpr_filterbank_pack");
    MelSched sch;
    if (int e = build_sched(n_freq, n_filt, fb_kranges_host, &sch)) return e;
    {
        uint32_t hdr[kPackHeaderFloats] = {0};
        int chunks = 0;
        for (int t = 0; t < sch.ntiles; ++t) chunks += (sch.khi[t] - sch.klo[t]) / kChunkRows;
        hdr[0] = kPackMagic; hdr[1] = (uint32_t)n_freq; hdr[2] = (uint32_t)n_filt; hdr[3] = (uint32_t)sch.ntiles;
        hdr[4] = (uint32_t)chunks; hdr[5] = kranges_hash(n_freq, n_filt, fb_kranges_host);
        if (const int cap = pw_section_cap(n_freq)) {                 // band plan for k_mel_pw (after the fragments)
            uint32_t* sec = reinterpret_cast<uint32_t*>(out_host) + kPackHeaderFloats + (size_t)chunks * 512;
            std::memset(sec, 0, sizeof(uint32_t) * cap);
            if (build_band_plan(fb_host, n_freq, n_filt, sec, &hdr[7])) hdr[6] = (uint32_t)(kPackHeaderFloats + chunks * 512);
        }
        std::memcpy(out_host, hdr, sizeof(hdr));
        out_host += kPackHeaderFloats;
    }
    for (int t = 0; t < sch.ntiles; ++t) {
        size_t pos = (size_t)sch.chunk0[t] * 512;
        for (int c = 0; c < (sch.khi[t] - sch.klo[t]) / kChunkRows; ++c)
            for (int g = 0; g < 2; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int k = sch.klo[t] + kChunkRows * c + 16 * g + 4 * s4 + (l >> 4);
                        const int m = 16 * t + (l & 15);
                        out_host[pos++] = (k < n_freq && m < n_filt) ? fb_host[(size_t)k * n_filt + m] : 0.0f;
                    }
    }
    return 0;
}

int64_t kpr_mel_workspace_bytes_unpacked(const kpr_stft_geom* s, int n_filt) {
    if (check_geom(s) || n_filt <= 0) return -1;
    return stats_region_bytes(s->batch) +
           (int64_t)sizeof(float) * 2 * s->batch * s->channels * frames_of(s) * (s->n_fft / 2 + 1);
}

int kpr_mel_f32(const float* x, const kpr_stft_geom* s, const float* window, const float* fb,
                const float* fb_packed, int n_filt, const int32_t* fb_kranges_host,
                const kpr_db_params* db, float* out, void* workspace, int64_t workspace_bytes,
                kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (int e = check_geom(s)) return e;
    if (int e = check_db(db)) return e;
    if (n_filt <= 0) return fail(KPR_E_BADARG, "n_filt must be positive");
    const long long F = frames_of(s);
    const int64_t need = kpr_mel_workspace_bytes(s, n_filt, db);
    const kpr_stft_geom se = forward_geom(s);                    // win_length > n_fft: cropped frames (see forward_geom)
    s = &se;
    Geom g = make_geom(s, F);
    if (g.total_frames == 0) return 0;
    if (!x || !window || !fb || !out)
        return fail(KPR_E_BADARG, "x / window / fb / out must not be NULL");
    if (!workspace || workspace_bytes < need)
        return fail(KPR_E_WORKSPACE, "mel workspace: need %lld bytes", (long long)need);
    hipStream_t st = (hipStream_t)stream;
    DbDev dbd = make_db(db);
    unsigned* stats = reinterpret_cast<unsigned*>(workspace);
    const int slots = dbd.enabled ? db_slots(s->batch) : 1;       // statistics slots per item (small batches, see DbDev)
    if (dbd.enabled) {
        dbd.slot_mask = slots - 1;
        dbd.slot_stride = (int)(2 * s->batch);
        hipLaunchKernelGGL(k_stats_init, dim3(grid_1d(s->batch * slots, 256)), dim3(256), 0, st, stats,
                           (long long)s->batch * slots);
        if (int e = launch_check("k_stats_init")) return e;
    }
    MelSched sch{};
    const int sched_rc = get_sched(g.K, n_filt, fb_kranges_host, &sch);
    if (sched_rc != 0 && sched_rc != KPR_E_UNSUPPORTED) return sched_rc;          // malformed k-ranges etc.: report
    const bool have_sched = sched_rc == 0;                                         // UNSUPPORTED: more tiles than the
    if (!have_sched) { fb_packed = nullptr; fb_kranges_host = nullptr; }           // schedule holds -> dense GEMM
    PackInfo pinfo{0, 0, 0, 0, 0};
    const float* blob = fb_packed;
    if (fb_packed) {
        if (int e = verify_packed(fb_packed, g.K, n_filt, fb_kranges_host, sch, st, &pinfo)) return e;
        fb_packed += kPackHeaderFloats;
    }
    const long long item_size = (long long)s->channels * F * n_filt;
    // k_mel_pw (round 4): n_fft 256 ... 2048 with a band plan in the packed filterbank (mel / triangular banks) -- the
    // default since round 4 (same-box sweeps against k_mel_ws / k_mel_ts / the ring kernel: tools/sweep_dispatch.py mel).
    // mel_variant 5 / 6 / 7 = k_mel_pw with 8 / 4 / 16 waves per workgroup, 8 = its PAIR form where it applies (A/B runs, tests).
    if (fb_packed && pinfo.band_off && (fused_nfft(s->n_fft) || s->n_fft == 256) && s->win_length <= s->n_fft &&
        (int)pinfo.L * kPts == g.K - 1 &&
        g.total_frames < 0x7fffff00LL && (opt(OPT_MEL_VARIANT) >= 5 || opt(OPT_MEL_VARIANT) == 0)) {
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        g.cfast = (g.in_cl && g.C > 1) ? 1 : 0;
        // waves per workgroup: sixteen (one workgroup per CU, one copy of the tables, tickets shared by the whole CU) once
        // there are sixteen tickets per CU; smaller launches are spread over more, smaller workgroups
        int cus_w = 256;
        if (int e = device_cus(&cus_w)) return e;
        const long long tickets_w = (g.total_frames + (64 / (s->n_fft / 32)) - 1) / (64 / (s->n_fft / 32));
        // (tools/sweep_dispatch.py mel, gpurun_out/r04i: 8-wave workgroups win from ~4 tickets per CU up -- 8.2 vs 9.1 us at
        //  1328 tickets -- and 16-wave ones from 16 per CU: 14.8 vs 15.9 vs 17.6 us at 5312)
        const int w_auto = tickets_w >= 16LL * cus_w ? 16 : tickets_w >= 4LL * cus_w ? 8 : 4;
        const int w = opt(OPT_MEL_VARIANT) == 6 ? 4 : opt(OPT_MEL_VARIANT) == 7 ? 16 : opt(OPT_MEL_VARIANT) == 5 ? 8 : w_auto;
        int rc;
        // interleaved waveforms with an even channel count, launches that fill the chip: the PAIR form (kpr_mel_pw_kernels.h).
        // n_fft 1024 stereo stays on the plain kernel (its stereo pair fetch does the same at four waves per SIMD).
        // mel_variant 8 forces it wherever it applies (tests), 5 / 6 / 7 never take it.
        const bool pair_ok = g.cfast && (g.C % 2) == 0 && (s->n_fft == 2048 || s->n_fft == 1024);
        const bool pair_auto = pair_ok && (s->n_fft == 2048 || g.C >= 4) && tickets_w >= 24LL * cus_w;
        if (pair_ok && (opt(OPT_MEL_VARIANT) == 8 || (opt(OPT_MEL_VARIANT) == 0 && pair_auto))) {
            rc = s->n_fft == 2048 ? launch_mel_pw_pair<1024>(x, g, window, tw, blob, pinfo, n_filt, dbd, stats, out, st)
                                  : launch_mel_pw_pair<512>(x, g, window, tw, blob, pinfo, n_filt, dbd, stats, out, st);
            if (rc) return rc;
            return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
        }
        switch (s->n_fft) {
            case 256:  rc = launch_mel_pw_w<128>(w, x, g, window, tw, blob, pinfo, n_filt, dbd, stats, out, st); break;
            case 512:  rc = launch_mel_pw_w<256>(w, x, g, window, tw, blob, pinfo, n_filt, dbd, stats, out, st); break;
            case 1024: rc = launch_mel_pw_w<512>(w, x, g, window, tw, blob, pinfo, n_filt, dbd, stats, out, st); break;
            default:   rc = launch_mel_pw_w<1024>(w, x, g, window, tw, blob, pinfo, n_filt, dbd, stats, out, st); break;
        }
        if (rc) return rc;
        return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
    }
    if (fused_nfft(s->n_fft) && fb_packed) {
        // Banks WITHOUT a band plan (log-frequency banks, dense matrices) or a forced variant: the MFMA kernels.
        // mel_variant: 0 = automatic, 2 = k_mel_ws with a streamed filterbank slice, 3 = k_mel_ws wherever it applies,
        // 4 = the tile-synchronous kernel k_mel_ts wherever it applies (A/B runs, tests).  The round-1 4-wave ring kernel
        // (k_mel_fused, mel_variant 1) lost on every shape of tools/sweep_dispatch.py (profiles/r05_sweep_before_prune.log:
        // within 4 % of k_mel_ts on launches of a few thousand frames at n_fft 512, 25-95 % behind elsewhere) and was
        // removed in round 5.
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        int rc;
        const int cfast_in = g.cfast;
        g.cfast = (g.in_cl && g.C > 1) ? 1 : 0;
        // k_mel_ts: n_fft 512 (sweep, log-frequency bank: 38.6 vs 58.8 us on 256 x 1 s @22 kHz; 14.0 vs 13.4 on 16 clips),
        // interleaved stereo at n_fft 1024 (pair fetch: 126 vs 180 us on 64 x 2 x 10 s @16 kHz), n_fft 1024 from ~12 k frames
        // up (32-frame rounds: 6-9 % faster than k_mel_ws on 256 x 10 s @16 kHz); k_mel_ws for the rest of n_fft 1024 / 2048
        const bool stereo_cl = g.in_cl && g.C == 2 && s->n_fft == 1024;
        const bool long_1024 = s->n_fft == 1024 && g.total_frames >= 12288;
        auto run_ts = [&](bool* taken) -> int {
            *taken = false;
            MelSchedTs sts;
            // n_fft 512, long runs (>= 64 k frames, two 64-frame rounds per workgroup): 64-frame rounds, two tickets per wave
            // (256 x 2 x 1 s @22 kHz, dB: 61 vs 65 us; 43 k frames mono: 29.7 vs 27.6, hence the threshold)
            if (s->n_fft == 512 && g.total_frames >= 65536 && mel_ts_ok(s->n_fft, g.K, n_filt, fb_kranges_host, g, &sts, 64)) {
                *taken = true;
                return launch_mel_ts<256, 64>(x, g, window, tw, fb_packed, sts, dbd, stats, out, st);
            }
            if (!mel_ts_ok(s->n_fft, g.K, n_filt, fb_kranges_host, g, &sts)) return 0;
            *taken = true;
            switch (s->n_fft) {
                case 512:  return launch_mel_ts<256>(x, g, window, tw, fb_packed, sts, dbd, stats, out, st);
                case 1024: return launch_mel_ts<512>(x, g, window, tw, fb_packed, sts, dbd, stats, out, st);
                default:   return launch_mel_ts<1024>(x, g, window, tw, fb_packed, sts, dbd, stats, out, st);
            }
        };
        bool taken = false;
        if (opt(OPT_MEL_VARIANT) == 4 || (opt(OPT_MEL_VARIANT) == 0 && (s->n_fft == 512 || stereo_cl || long_1024))) {
            if ((rc = run_ts(&taken))) return rc;
            if (taken) return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
        }
        int slice_max = 0;      // the consumers keep one lane of schedule per chunk of their slice
        for (int i = 0; i < 4; ++i) slice_max = std::max(slice_max, (int)sch.wave_nchunks[i]);
        if ((s->n_fft == 2048 || s->n_fft == 1024) && slice_max <= 64 &&
            g.total_frames < 0x7fffff00LL && mel_ws_lds_bytes(s->n_fft / 2, sch.nseg) <= 160 * 1024) {
            rc = (s->n_fft == 2048)
                     ? launch_mel_ws<1024>(x, g, window, tw, fb_packed, sch, dbd, stats, out, st)
                     : launch_mel_ws<512>(x, g, window, tw, fb_packed, sch, dbd, stats, out, st);
            if (rc) return rc;
            return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
        }
        // what k_mel_ws cannot take (very wide banks): k_mel_ts if its schedule holds the bank, else the two-launch path below
        if ((rc = run_ts(&taken))) return rc;
        if (taken) return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
        g.cfast = cfast_in;
    }
    // n_fft 256 (round 3): the tile-synchronous kernel takes it too (eight lanes per frame, 64-frame rounds); mel_variant 3
    // keeps the two-launch path
    if (fb_packed && s->n_fft == 256 && s->win_length <= s->n_fft && opt(OPT_MEL_VARIANT) != 3) {
        MelSchedTs sts;
        Geom gt = g;
        gt.cfast = (g.in_cl && g.C > 1) ? 1 : 0;
        if (mel_ts_ok(s->n_fft, g.K, n_filt, fb_kranges_host, gt, &sts)) {
            const float2* tw = nullptr;
            if (int e = get_twiddles(s->n_fft, &tw)) return e;
            if (int e = launch_mel_ts<128>(x, gt, window, tw, fb_packed, sts, dbd, stats, out, st)) return e;
            return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
        }
    }
    // mixed-radix sizes (n_fft 400, 320, 640 ...: speech front ends): one launch as well (k_mel_mr);
    // kpr_set_option("mel_variant", 3) keeps the two-launch path of round 2 (A/B runs, tests)
    if (fb_packed && mel_mr_nfft(s->n_fft) && opt(OPT_MIXED_RADIX) && s->win_length <= s->n_fft &&
        opt(OPT_MEL_VARIANT) != 3) {
        bool taken = false;
        Geom gm = g;
        gm.cfast = (g.in_cl && g.C > 1) ? 1 : 0;  // channel-fastest frame numbering: the C frames that share cache lines sit in one wave
        if (int e = launch_mel_mr(x, gm, window, fb_packed, fb_kranges_host, n_filt, dbd, stats, out, st, &taken)) return e;
        if (taken) return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
    }
    // two-kernel path: STFT (complex, frame-contiguous) -> (|.| x filterbank) GEMM [+ dB]
    {
        const int64_t need2 = stats_region_bytes(s->batch) +
                              (int64_t)sizeof(float) * 2 * g.total_frames * g.K;
        if (workspace_bytes < need2)
            return fail(KPR_E_WORKSPACE, "mel workspace (unpacked filterbank path): need %lld bytes",
                        (long long)need2);
    }
    float* spec = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) +
                                           stats_region_bytes(s->batch));
    if (fast_nfft(s->n_fft)) {   // Stockham STFT (n_fft = 256, or no packed filterbank given)
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        Geom gc = g;
        gc.out_cl = 0;
        int rc;
        switch (s->n_fft) {
            case 256:  rc = launch_stft_fast<128>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
            case 512:  rc = launch_stft_fast<256>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
            case 1024: rc = launch_stft_fast<512>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
            default:   rc = launch_stft_fast<1024>(x, gc, window, tw, KPR_OUT_COMPLEX, spec, st); break;
        }
        if (rc) return rc;
    } else if (bluestein_ok(s)) {   // even non-power-of-two n_fft: chirp-z STFT, frame-contiguous
        Geom gc = g;
        gc.out_cl = 0;
        // wide packed filterbank: |X| rows straight into the fused kernel's MFMA consumers
        // (loader producers, FROM_MAG) instead of the complex spectrum + generic GEMM
        int slice_max = 0;
        for (int i = 0; i < 4; ++i) slice_max = std::max(slice_max, (int)sch.wave_nchunks[i]);
        if (fb_packed && g.K <= 1025 && slice_max <= 64 && !g.out_cl &&
            g.total_frames < 0x7fffff00LL && mel_ws_lds_bytes(1024, sch.nseg, 2) <= 160 * 1024) {
            if (int e = launch_stft_bs(x, gc, window, KPR_OUT_MAGNITUDE, spec, st)) return e;
            if (int e = launch_mel_ws<1024, true>(spec, g, nullptr, nullptr, fb_packed, sch, dbd, stats, out, st))
                return e;
            return dbd.enabled ? db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots) : 0;
        }
        if (fb_kranges_host) {           // e.g. channels_last output with several channels: |X| rows + banded product
            if (int e = launch_stft_bs(x, gc, window, KPR_OUT_MAGNITUDE, spec, st)) return e;
            return run_band_mel(spec, g, fb, sch, dbd, stats, out, s->batch, item_size, st);
        }
        if (int e = launch_stft_bs(x, gc, window, KPR_OUT_COMPLEX, spec, st)) return e;
    } else if (big_nfft(s->n_fft) && s->win_length <= s->n_fft) {   // n_fft 4096 / 8192: FFT kernel, frame-contiguous
        Geom gc = g;
        gc.out_cl = 0;
        if (fb_kranges_host) {
            // |X| rows, then the banded product (a mel / log bank has ~2 K non-zeros: bandwidth work)
            if (int e = launch_stft_big(x, gc, window, KPR_OUT_MAGNITUDE, spec, st)) return e;
            return run_band_mel(spec, g, fb, sch, dbd, stats, out, s->batch, item_size, st);
        }
        if (int e = launch_stft_big(x, gc, window, KPR_OUT_COMPLEX, spec, st)) return e;
    } else if (gen_ok_f32(s)) {
        Geom gc = g;
        gc.out_cl = 0;
        if (int e = launch_stft_gen_f32(x, gc, window, KPR_OUT_COMPLEX, spec, st)) return e;
    } else {
        if (int e = stft_gemm(x, s, g, window, spec, true, st)) return e;
    }
    GemmArgs ga{};
    ga.in = frames_contig_map(g, g.K);
    ga.out = frames_out_map(g, n_filt);
    ga.Kdim = g.K; ga.N = n_filt; ga.ldb = n_filt;
    ga.db = dbd; ga.stats = stats;
    if (fb_kranges_host) {
        ga.has_kr = 1;
        for (int t = 0; t < sch.ntiles; ++t) { ga.klo[t] = sch.klo[t]; ga.khi[t] = sch.khi[t]; }
    }
    if (dbd.enabled) {
        if (int e = run_gemm<A_CABS, E_DB>(spec, fb, ga, out, st)) return e;
        return db_clamp(out, s->batch, item_size, dbd.dyn, stats, st, slots);
    }
    return run_gemm<A_CABS, E_PLAIN>(spec, fb, ga, out, st);
}

int kpr_filterbank_kranges(const float* fb_host, int n_freq, int n_filt, int32_t* out_host) {
    if (!fb_host || !out_host || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad arguments to kpr_filterbank_kranges");
    const int ntiles = (n_filt + 15) / 16;
    for (int t = 0; t < ntiles; ++t) {
        int lo = n_freq, hi = 0;
        for (int k = 0; k < n_freq; ++k)
            for (int m = t * 16; m < std::min(n_filt, t * 16 + 16); ++m) {
                float v = fb_host[(size_t)k * n_filt + m];
                if (v != 0.0f || v != v) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
            }
        if (lo >= hi) { lo = 0; hi = 0; }
        out_host[2 * t] = lo & ~3;
        out_host[2 * t + 1] = (hi + 3) & ~3;
    }
    return 0;
}

int kpr_abs_c64(const void* x, int64_t n, float* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (n < 0) return fail(KPR_E_BADARG, "negative size");
    if (n == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    hipLaunchKernelGGL(k_cplx_to_real, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (long long)n, 0, out);
    return launch_check("k_cplx_to_real");
}

int kpr_angle_c64(const void* x, int64_t n, float* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (n < 0) return fail(KPR_E_BADARG, "negative size");
    if (n == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    hipLaunchKernelGGL(k_cplx_to_real, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (long long)n, 1, out);
    return launch_check("k_cplx_to_real");
}

int kpr_apply_filterbank_f32(const float* x, int64_t batch, int channels, int64_t frames,
                             int n_freq, int layout, const float* fb, int n_filt,
                             const int32_t* fb_kranges_host, float* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad sizes");
    if ((unsigned)layout > 1u) return fail(KPR_E_BADARG, "bad layout enum");
    const long long rows = batch * channels * frames;
    if (rows == 0) return 0;
    if (!x || !fb || !out) return fail(KPR_E_BADARG, "x / fb / out must not be NULL");
    const int ntiles = (n_filt + 15) / 16;
    if (ntiles > kMaxTiles || n_freq > 32000) fb_kranges_host = nullptr;   // GemmArgs holds 16-bit k-ranges for 64 tiles: dense product
    // narrow matrices on contiguous rows (LogmelToMFCC's DCT, small filterbanks): the thin GEMM
    const bool contiguous = layout == KPR_CHANNELS_FIRST || channels == 1;
    if (contiguous && ntiles <= 4 && n_freq <= 512 && (n_freq & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        const size_t lds = sizeof(float) * 4 * 64 * (size_t)ntiles * ((n_freq + 15) / 16);
        const unsigned grid = (unsigned)std::min<long long>((rows + 63) / 64, 256 * 8);
        hipStream_t st = (hipStream_t)stream;
        switch (ntiles) {
            case 1: hipLaunchKernelGGL(k_thin_gemm<1>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
            case 2: hipLaunchKernelGGL(k_thin_gemm<2>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
            case 3: hipLaunchKernelGGL(k_thin_gemm<3>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
            default: hipLaunchKernelGGL(k_thin_gemm<4>, dim3(grid), dim3(256), lds, st, x, rows, n_freq, fb, n_filt, out); break;
        }
        return launch_check("k_thin_gemm");
    }
    GemmArgs ga{};
    ga.in.rows = rows; ga.out.rows = rows;
    if (layout == KPR_CHANNELS_LAST) {
        // rows r = (b*F + f)*C + c
        ga.in.D0 = channels; ga.in.D1 = 1;
        ga.in.s2 = (long long)n_freq * channels; ga.in.s1 = 0; ga.in.s0 = 1; ga.in.es = channels;
        ga.out.D0 = channels; ga.out.D1 = 1;
        ga.out.s2 = (long long)n_filt * channels; ga.out.s1 = 0; ga.out.s0 = 1; ga.out.es = channels;
    } else {
        ga.in.D0 = 1; ga.in.D1 = 1; ga.in.s2 = n_freq; ga.in.s1 = 0; ga.in.s0 = 0; ga.in.es = 1;
        ga.out.D0 = 1; ga.out.D1 = 1; ga.out.s2 = n_filt; ga.out.s1 = 0; ga.out.s0 = 0; ga.out.es = 1;
    }
    ga.Kdim = n_freq; ga.N = n_filt; ga.ldb = n_filt;
    if (fb_kranges_host) {
        const int kp = (n_freq + 3) & ~3;
        ga.has_kr = 1;
        for (int t = 0; t < ntiles; ++t) {
            int lo = fb_kranges_host[2 * t], hi = fb_kranges_host[2 * t + 1];
            if (lo < 0 || hi > kp || lo > hi) return fail(KPR_E_BADARG, "bad k-range");
            ga.klo[t] = (short)lo; ga.khi[t] = (short)hi;
        }
    }
    return run_gemm<A_PLAIN, E_PLAIN>(x, fb, ga, out, (hipStream_t)stream);
}

int kpr_apply_filterbank_packed_f32(const float* x, int64_t batch, int channels, int64_t frames,
                                    int n_freq, int layout, const float* fb, const float* fb_packed,
                                    int n_filt, const int32_t* fb_kranges_host, float* out,
                                    kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad sizes");
    if ((unsigned)layout > 1u) return fail(KPR_E_BADARG, "bad layout enum");
    const long long rows = batch * channels * frames;
    const bool contiguous = layout == KPR_CHANNELS_FIRST || channels == 1;
    MelSched sch{};
    // (channels_last with C > 1: the loader waves read rows strided by C, channel-fastest row order)
    if (fb_packed && x && out && rows > 0 && rows < 0x7fffff00LL && n_freq <= 1025 &&
        // (narrow matrices on rows of a multiple of four floats: the thin GEMM of kpr_apply_filterbank_f32)
        (n_filt > 64 || n_freq > 512 || (n_freq & 3)) && get_sched(n_freq, n_filt, fb_kranges_host, &sch) == 0) {
        PackInfo pinfo{0, 0, 0, 0, 0};
        if (int e = verify_packed(fb_packed, n_freq, n_filt, fb_kranges_host, sch, (hipStream_t)stream, &pinfo)) return e;
        // a bank with a band plan (mel / triangular banks; n_freq - 1 a multiple of four up to 1024: every even n_fft / 4): the banded
        // row kernel (round 6; 21 248 x 1025 -> 128: k_mel_ws 35 / 41 us, this one 19 / 25 -- same buffers / rotating) on contiguous rows
        // and on rows of TWO interleaved channels (channels_last stereo: the ST instances; 10 624 blocks of 1025 bins: 38 / 44 ->
        // 25 / 30 us, a single-block launch 8.1 -> 6.9).  More channels stay on the MFMA kernels (kpr_fb_pw_kernels.h).
        // fb_variant 1 = never (A/B runs, tests).
        if (pinfo.band_off && fb && (contiguous || channels == 2) && opt(OPT_FB_VARIANT) != 1 && pinfo.L >= 8 && pinfo.L <= 64) {
            hipStream_t st = (hipStream_t)stream;
            if (contiguous) return launch_fb_pw_l<false>((int)pinfo.L, x, rows, n_freq, fb_packed, pinfo, n_filt, fb, out, st);
            return launch_fb_pw_l<true>((int)pinfo.L, x, rows / 2, n_freq, fb_packed, pinfo, n_filt, fb, out, st);
        }
        fb_packed += kPackHeaderFloats;
        int slice_max = 0;
        for (int i = 0; i < 4; ++i) slice_max = std::max(slice_max, (int)sch.wave_nchunks[i]);
        if (slice_max <= 64 && mel_ws_lds_bytes(1024, sch.nseg, 2) <= 160 * 1024) {
            Geom g{};
            g.total_frames = rows; g.T = 0; g.F = (int)frames; g.C = channels;
            g.n_fft = 2 * (n_freq - 1); g.win = 0; g.hop = 0; g.pad_left = 0; g.K = n_freq;
            g.in_cl = 0; g.out_cl = contiguous ? 0 : 1; g.cfast = contiguous ? 0 : 1;
            geom_set_magic(g);
            DbDev dbd = make_db(nullptr);
            return launch_mel_ws<1024, true>(x, g, nullptr, nullptr, fb_packed, sch, dbd, nullptr, out,
                                             (hipStream_t)stream);
        }
    }
    // long contiguous rows of a banded matrix (K = 2049 / 4097 bins after an n_fft 4096 / 8192 STFT: beyond the
    // MFMA consumers' tile): the banded product instead of the generic GEMM
    if (x && out && fb && contiguous && rows > 0 && rows < 0x7fffff00LL && n_freq > 1025 && fb_kranges_host &&
        sizeof(float) * (size_t)kBandRows * n_freq <= 160 * 1024 && get_sched(n_freq, n_filt, fb_kranges_host, &sch) == 0) {
        Geom g{};
        g.total_frames = rows; g.T = 0; g.F = (int)frames; g.C = channels;
        g.n_fft = 2 * (n_freq - 1); g.win = 0; g.hop = 0; g.pad_left = 0; g.K = n_freq;
        g.in_cl = 0; g.out_cl = 0; g.cfast = 0;
        geom_set_magic(g);
        return run_band_mel(x, g, fb, sch, make_db(nullptr), nullptr, out, batch, 0, (hipStream_t)stream);
    }
    return kpr_apply_filterbank_f32(x, batch, channels, frames, n_freq, layout, fb, n_filt, fb_kranges_host,
                                    out, stream);
}

int64_t kpr_db_workspace_bytes(int64_t n_items) {
    if (n_items < 0) return -1;
    return 256 + (int64_t)sizeof(unsigned) * 2 * std::max<int64_t>(1, n_items);
}

int kpr_mag_to_db_f32(const float* x, int64_t n_items, int64_t item_size, const kpr_db_params* db,
                      float* out, void* workspace, int64_t workspace_bytes, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (!db) return fail(KPR_E_BADARG, "db params are NULL");
    kpr_db_params p = *db;
    p.enabled = 1;
    if (int e = check_db(&p)) return e;
    if (n_items < 0 || item_size < 0) return fail(KPR_E_BADARG, "negative size");
    if (n_items == 0 || item_size == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    if (!workspace || workspace_bytes < kpr_db_workspace_bytes(n_items))
        return fail(KPR_E_WORKSPACE, "db workspace: need %lld bytes",
                    (long long)kpr_db_workspace_bytes(n_items));
    hipStream_t st = (hipStream_t)stream;
    DbDev dbd = make_db(&p);
    unsigned* stats = reinterpret_cast<unsigned*>(workspace);
    hipLaunchKernelGGL(k_stats_init, dim3(grid_1d(n_items, 256)), dim3(256), 0, st, stats,
                       (long long)n_items);
    if (int e = launch_check("k_stats_init")) return e;
    int chunks = db_chunks(n_items, item_size);
    if (opt(OPT_DB_CHUNKS) > 0) chunks = opt(OPT_DB_CHUNKS);
    if (((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0)   // 16-byte accesses on the aligned middle of every chunk
        hipLaunchKernelGGL(k_db_log<4>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, x,
                           (long long)item_size, chunks, dbd, stats, out);
    else
        hipLaunchKernelGGL(k_db_log<1>, dim3((unsigned)(n_items * chunks)), dim3(256), 0, st, x,
                           (long long)item_size, chunks, dbd, stats, out);
    if (int e = launch_check("k_db_log")) return e;
    return db_clamp(out, n_items, item_size, dbd.dyn, stats, st);
}

int64_t kpr_istft_workspace_bytes(const kpr_stft_geom* s, int64_t n_frames) {
    if (check_geom(s) || n_frames < 0) return -1;
    return 256 + (int64_t)sizeof(float) * s->batch * s->channels * n_frames * s->win_length;
}

int kpr_istft_f32(const void* spec, const kpr_stft_geom* s, int64_t n_frames,
                  const float* synth_window, float* out, void* workspace, int64_t workspace_bytes,
                  kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (int e = check_geom(s)) return e;
    if (n_frames < 0) return fail(KPR_E_BADARG, "negative frame count");
    Geom g = make_geom(s, n_frames);
    g.pad_left = 0;
    if (g.total_frames == 0) return 0;
    if (!spec || !synth_window || !out) return fail(KPR_E_BADARG, "spec / window / out must not be NULL");
    const int64_t need = kpr_istft_workspace_bytes(s, n_frames);
    if (!workspace || workspace_bytes < need)
        return fail(KPR_E_WORKSPACE, "istft workspace: need %lld bytes", (long long)need);
    hipStream_t st = (hipStream_t)stream;
    float* frames = reinterpret_cast<float*>(workspace);
    if (fast_nfft(s->n_fft) && opt(OPT_ISTFT_PATH) != 2) {
        // fused irFFT + window + overlap-add (no workspace traffic) whenever the frames overlap
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        bool launched = false;
        int rc = 0;
        // Launches of a few thousand frames -- batch 1 ... 8 of clips, the serving case -- are latency bound and the
        // barrier kernel (many small workgroups, one barrier) beats the ring kernel's serial walk over short segments:
        // 4 x 434 frames at n_fft 1024 12.0 vs 15.2 us, 16 x 83 at n_fft 2048 16.0 vs 20.3, 16 x 61 at n_fft 512 8.0 vs
        // 9.1; from ~3.5 k frames up the ring kernel wins (16 x 434: 20.0 vs 24.4; cfg4: 77 vs 122)
        // (tools/kbench_istft_variants.py; kpr_set_option("istft_path", 3) = the ring kernel whatever the size).
        const bool small = opt(OPT_ISTFT_PATH) == 0 && g.total_frames <= 3072;
        auto ring = [&]() {      // wave-specialised ring kernel when its preconditions hold
            switch (s->n_fft) {
                case 256:  return launch_istft_ws<128>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched);
                case 512:  return launch_istft_ws<256>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched);
                case 1024: return launch_istft_ws<512>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched);
                default:   return launch_istft_ws<1024>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched);
            }
        };
        // k_istft_pw (round 4): hop = n_fft / 8, / 4, / 2, launches that fill the chip (istft_path 4 = wherever it applies)
        if ((opt(OPT_ISTFT_PATH) == 0 && !small) || opt(OPT_ISTFT_PATH) == 4) {
            switch (s->n_fft) {
                case 512:  rc = launch_istft_pw<256>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
                case 1024: rc = launch_istft_pw<512>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
                case 2048: rc = launch_istft_pw<1024>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
                default:   rc = 0; break;
            }
            if (rc) return rc;
            if (launched) return 0;
        }
        if (!small) {
            rc = ring();
            if (rc) return rc;
            if (launched) return 0;
        }
        switch (s->n_fft) {
            case 256:  rc = launch_istft_fused<128, 4>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            case 512:  rc = launch_istft_fused<256, 4>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            case 1024: rc = launch_istft_fused<512, 4>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
            default:   rc = launch_istft_fused<1024, 8>((const float2*)spec, s, n_frames, synth_window, tw, out, st, &launched); break;
        }
        if (rc) return rc;
        if (launched) return 0;
        if (small) {             // the barrier kernel did not apply (no overlap ...): the ring kernel may
            rc = ring();
            if (rc) return rc;
            if (launched) return 0;
        }
    }
    if (!fast_nfft(s->n_fft) && opt(OPT_ISTFT_PATH) != 2) {
        // n_fft = 2^a 5^b: the ring kernel with mixed-radix producers
        bool launched = false;
        if (int e = launch_istft_ws_mr((const float2*)spec, s, n_frames, synth_window, out, st, &launched)) return e;
        if (launched) return 0;
    }
    if (fast_nfft(s->n_fft)) {
        const float2* tw = nullptr;
        if (int e = get_twiddles(s->n_fft, &tw)) return e;
        int rc;
        switch (s->n_fft) {
            case 256:  rc = launch_irfft_fast<128>((const float2*)spec, g, synth_window, tw, frames, st); break;
            case 512:  rc = launch_irfft_fast<256>((const float2*)spec, g, synth_window, tw, frames, st); break;
            case 1024: rc = launch_irfft_fast<512>((const float2*)spec, g, synth_window, tw, frames, st); break;
            default:   rc = launch_irfft_fast<1024>((const float2*)spec, g, synth_window, tw, frames, st); break;
        }
        if (rc) return rc;
    } else if (bluestein_ok(s)) {     // even non-power-of-two n_fft: inverse chirp-z, then the gather
        if (int e = launch_irfft_bs((const float2*)spec, g, synth_window, frames, st)) return e;
    } else if (big_nfft(s->n_fft)) {  // 4096 / 8192: sub-FFT kernel, then the gather
        if (int e = launch_irfft_big((const float2*)spec, g, synth_window, frames, st)) return e;
    } else if (gen_ok_f32(s)) {       // small prime factors: run-time mixed-radix inverse FFT, then the gather
        if (int e = launch_irfft_gen_f32((const float2*)spec, g, synth_window, frames, st)) return e;
    } else {
        const float* idft = nullptr;
        if (int e = get_dft_inv(s->n_fft, &idft)) return e;
        GemmArgs ga{};
        ga.in = frames_out_map(g, g.K);          // spectrum in the caller's layout (complex units)
        ga.out = frames_contig_map(g, g.win);
        ga.Kdim = 2 * g.K; ga.N = std::min(g.n_fft, g.win); ga.ldb = g.n_fft;
        ga.window = synth_window; ga.win = g.win;
        if (int e = run_gemm<A_CPLX, E_WINDOW>((const float*)spec, idft, ga, frames, st)) return e;
        if (g.win > g.n_fft) {
            hipLaunchKernelGGL(k_fill_cols, dim3(grid_1d(g.total_frames * (g.win - g.n_fft), 256)),
                               dim3(256), 0, st, frames, g.total_frames, (long long)g.win, g.n_fft,
                               g.win);
            if (int e = launch_check("k_fill_cols")) return e;
        }
    }
    const long long t_out = (n_frames - 1) * (long long)s->hop_length + s->win_length;
    const long long n_sig = (long long)s->batch * s->channels;
    hipLaunchKernelGGL(k_ola<float>, dim3(grid_1d(n_sig * t_out, 256)), dim3(256), 0, st, frames, n_sig,
                       (int)n_frames, s->channels, s->win_length, s->hop_length, t_out,
                       s->in_layout == KPR_CHANNELS_LAST ? 1 : 0, out);
    return launch_check("k_ola");
}

/* ---- float64 / complex128 variants -------------------------------------------------------------- */
static std::map<std::pair<int, int>, double2*> g_tw64;

static int get_twiddles64(int n_fft, const double2** out) {
    int dev;
    if (int e = cur_device(&dev)) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tw64.find({dev, n_fft});
    if (it == g_tw64.end()) {
        std::vector<double2> h(n_fft);
        for (int j = 0; j < n_fft; ++j) {
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)n_fft;
            h[j] = make_double2((double)cosl(a), (double)sinl(a));
        }
        double2* d = nullptr;
        KPR_HIP(hipMalloc(&d, sizeof(double2) * n_fft));
        KPR_HIP(hipMemcpy(d, h.data(), sizeof(double2) * n_fft, hipMemcpyHostToDevice));
        it = g_tw64.emplace(std::make_pair(dev, n_fft), d).first;
    }
    *out = it->second;
    return 0;
}

// launch shape of the float64 kernels; n_fft with a prime factor above 64 gets ONE pass of radix n_fft (a direct DFT)
static int f64_plan(const kpr_stft_geom* s, GenPlan* p, size_t* lds, int* tw_lds) {
    const int m = gen_fft_len(s->n_fft);
    if (!gen_plan(m, p)) { p->n = m; p->npass = 1; p->radix[0] = m; }
    if (!gen_lds(sizeof(double2), s->n_fft, lds, tw_lds))
        return fail(KPR_E_UNSUPPORTED, "float64 path: n_fft = %d does not fit in LDS (limit 10240 even / 5120 odd)", s->n_fft);
    return 0;
}

int kpr_stft_f64(const double* x, const kpr_stft_geom* s, const double* window, void* out, int mode,
                 kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (int e = check_geom(s)) return e;
    if (mode < KPR_OUT_COMPLEX || mode > KPR_OUT_PHASE) return fail(KPR_E_BADARG, "bad mode %d", mode);
    const long long F = frames_of(s);
    Geom g = make_geom(s, F);
    if (g.total_frames == 0) return 0;
    if (!x || !window || !out) return fail(KPR_E_BADARG, "x / window / out must not be NULL");
    g.win = std::min(g.win, g.n_fft);                            // win_length > n_fft: cropped frames (see forward_geom)
    size_t lds;
    int tl;
    GenPlan p;
    if (int e = f64_plan(s, &p, &lds, &tl)) return e;
    const double2* tw = nullptr;
    if (int e = get_twiddles64(s->n_fft, &tw)) return e;
    auto kern = tl ? &k_stft_gen<double, true> : &k_stft_gen<double, false>;
    static LdsOptIn opt_in_3[2];
    if (int e = allow_big_lds(opt_in_3[tl ? 1 : 0], reinterpret_cast<const void*>(kern))) return e;
    hipLaunchKernelGGL(kern, dim3(gen_grid(g, lds)), dim3(kF64Threads), lds, (hipStream_t)stream, x, g, window, tw, p,
                       mode, out);
    return launch_check("k_stft_gen<double>");
}

int64_t kpr_istft_f64_workspace_bytes(const kpr_stft_geom* s, int64_t n_frames) {
    if (check_geom(s) || n_frames < 0) return -1;
    return 256 + (int64_t)sizeof(double) * s->batch * s->channels * n_frames * s->win_length;
}

int kpr_istft_f64(const void* spec, const kpr_stft_geom* s, int64_t n_frames, const double* synth_window,
                  double* out, void* workspace, int64_t workspace_bytes, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (int e = check_geom(s)) return e;
    if (n_frames < 0) return fail(KPR_E_BADARG, "negative frame count");
    Geom g = make_geom(s, n_frames);
    g.pad_left = 0;
    if (g.total_frames == 0) return 0;
    if (!spec || !synth_window || !out) return fail(KPR_E_BADARG, "spec / window / out must not be NULL");
    const int64_t need = kpr_istft_f64_workspace_bytes(s, n_frames);
    if (!workspace || workspace_bytes < need)
        return fail(KPR_E_WORKSPACE, "istft (float64) workspace: need %lld bytes", (long long)need);
    size_t lds;
    int tl;
    GenPlan p;
    if (int e = f64_plan(s, &p, &lds, &tl)) return e;
    const double2* tw = nullptr;
    if (int e = get_twiddles64(s->n_fft, &tw)) return e;
    hipStream_t st = (hipStream_t)stream;
    double* frames = reinterpret_cast<double*>(workspace);
    auto kern = tl ? &k_irfft_gen<double, true> : &k_irfft_gen<double, false>;
    static LdsOptIn opt_in_4[2];
    if (int e = allow_big_lds(opt_in_4[tl ? 1 : 0], reinterpret_cast<const void*>(kern))) return e;
    hipLaunchKernelGGL(kern, dim3(gen_grid(g, lds)), dim3(kF64Threads), lds, st, (const double2*)spec, g, synth_window, tw,
                       p, frames);
    if (int e = launch_check("k_irfft_gen<double>")) return e;
    const long long t_out = (n_frames - 1) * (long long)s->hop_length + s->win_length;
    const long long n_sig = (long long)s->batch * s->channels;
    hipLaunchKernelGGL(k_ola<double>, dim3(grid_1d(n_sig * t_out, 256)), dim3(256), 0, st, frames, n_sig,
                       (int)n_frames, s->channels, s->win_length, s->hop_length, t_out,
                       (s->in_layout == KPR_CHANNELS_LAST && s->channels > 1) ? 1 : 0, out);
    return launch_check("k_ola");
}

int kpr_abs_c128(const void* x, int64_t n, double* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (n < 0) return fail(KPR_E_BADARG, "negative element count");
    if (n == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    hipLaunchKernelGGL(k_cplx_to_real_f64, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const double2*)x, (long long)n, 0, out);
    return launch_check("k_cplx_to_real_f64");
}

int kpr_angle_c128(const void* x, int64_t n, double* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (n < 0) return fail(KPR_E_BADARG, "negative element count");
    if (n == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    hipLaunchKernelGGL(k_cplx_to_real_f64, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const double2*)x, (long long)n, 1, out);
    return launch_check("k_cplx_to_real_f64");
}

int kpr_apply_filterbank_f64(const double* x, int64_t batch, int channels, int64_t frames, int n_freq, int layout,
                             const double* fb, int n_filt, double* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || n_filt <= 0)
        return fail(KPR_E_BADARG, "bad filterbank shape");
    if (layout != KPR_CHANNELS_FIRST && layout != KPR_CHANNELS_LAST) return fail(KPR_E_BADARG, "bad layout %d", layout);
    const long long total = (long long)batch * channels * frames * n_filt;
    if (total == 0) return 0;
    if (!x || !fb || !out) return fail(KPR_E_BADARG, "x / fb / out must not be NULL");
    hipLaunchKernelGGL(k_filterbank_f64, dim3(grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)batch, channels, (long long)frames, n_freq,
                       (layout == KPR_CHANNELS_LAST && channels > 1) ? 1 : 0, fb, n_filt, out);
    return launch_check("k_filterbank_f64");
}

int kpr_mag_to_db_f64(const double* x, int64_t n_items, int64_t item_size, double ref_value, double amin,
                      double dynamic_range, double* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (n_items < 0 || item_size < 0) return fail(KPR_E_BADARG, "negative size");
    // same checks (and order) as backend.py:168-173
    if (!(ref_value > 0)) return fail(KPR_E_BADARG, "ref_value must be positive");
    if (!(amin > 0)) return fail(KPR_E_BADARG, "amin must be positive");
    if (!(dynamic_range > 0)) return fail(KPR_E_BADARG, "dynamic_range must be positive");
    if (n_items == 0 || item_size == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    if (n_items > 0x7fffffffLL) return fail(KPR_E_UNSUPPORTED, "float64 decibel: more than 2^31 - 1 items");
    const double ref_term = 10.0 * std::log10(std::max(amin, ref_value));
    hipLaunchKernelGGL(k_db_f64, dim3((unsigned)n_items), dim3(1024), 0, (hipStream_t)stream, x,
                       (long long)item_size, amin, ref_term, dynamic_range, out);
    return launch_check("k_db_f64");
}

/* ---- backward passes (kpr_grad_kernels.h) ---------------------------------------------------- */
int kpr_abs_c64_bwd(const void* x, const float* g, int64_t n, void* gx, kpr_stream_t stream) {
    return run_cplx_bwd<float>(x, g, n, 0, gx, stream);
}
int kpr_angle_c64_bwd(const void* x, const float* g, int64_t n, void* gx, kpr_stream_t stream) {
    return run_cplx_bwd<float>(x, g, n, 1, gx, stream);
}
int kpr_abs_c128_bwd(const void* x, const double* g, int64_t n, void* gx, kpr_stream_t stream) {
    return run_cplx_bwd<double>(x, g, n, 0, gx, stream);
}
int kpr_angle_c128_bwd(const void* x, const double* g, int64_t n, void* gx, kpr_stream_t stream) {
    return run_cplx_bwd<double>(x, g, n, 1, gx, stream);
}

int kpr_spec_edge_scale_c64(const void* in, int64_t n, int n_freq, int inner, int n_fft, float s_edge, float s_mid,
                            void* out, kpr_stream_t stream) {
    return run_edge_scale<float>(in, n, n_freq, inner, n_fft, s_edge, s_mid, out, stream);
}
int kpr_spec_edge_scale_c128(const void* in, int64_t n, int n_freq, int inner, int n_fft, double s_edge, double s_mid,
                             void* out, kpr_stream_t stream) {
    return run_edge_scale<double>(in, n, n_freq, inner, n_fft, s_edge, s_mid, out, stream);
}

int kpr_mag_to_db_bwd_f32(const float* x, const float* gy, int64_t n_items, int64_t item_size,
                          const kpr_db_params* db, float* gx, kpr_stream_t stream) {
    if (!db) return fail(KPR_E_BADARG, "db params are NULL");
    return run_db_bwd<float>(x, gy, n_items, item_size, db->ref_value, db->amin, db->dynamic_range, gx, stream);
}
int kpr_mag_to_db_bwd_f64(const double* x, const double* gy, int64_t n_items, int64_t item_size, double ref_value,
                          double amin, double dynamic_range, double* gx, kpr_stream_t stream) {
    return run_db_bwd<double>(x, gy, n_items, item_size, ref_value, amin, dynamic_range, gx, stream);
}

/* ---- Frame / Energy / Delta ------------------------------------------------------------------ */
int64_t kpr_frame_count(int64_t time, int frame_length, int hop_length, int pad_end) {
    if (time < 0 || frame_length <= 0 || hop_length <= 0) {
        fail(KPR_E_BADARG, "bad time/frame_length/hop_length (%lld, %d, %d)", (long long)time, frame_length,
             hop_length);
        return -1;
    }
    if (pad_end) return (time + hop_length - 1) / hop_length;
    return time < frame_length ? 0 : 1 + (time - frame_length) / hop_length;
}

static int frame_args(int64_t batch, int channels, int64_t time, int layout, int frame_length,
                      int hop_length, int pad_end, float pad_value, FrameArgs* a) {
    if (batch < 0 || channels <= 0 || (unsigned)layout > 1u)
        return fail(KPR_E_BADARG, "bad batch/channels/layout (%lld, %d, %d)", (long long)batch, channels, layout);
    const int64_t f = kpr_frame_count(time, frame_length, hop_length, pad_end);
    if (f < 0) return KPR_E_BADARG;
    if (f > 0x7fffffffLL) return fail(KPR_E_UNSUPPORTED, "too many frames per signal");
    a->n_sig = batch * channels; a->T = time; a->C = channels; a->F = (int)f; a->L = frame_length;
    a->hop = hop_length; a->cl = layout == KPR_CHANNELS_LAST && channels > 1; a->pad_value = pad_value;
    return 0;
}

int kpr_frame_f32(const float* x, int64_t batch, int channels, int64_t time, int layout,
                  int frame_length, int hop_length, int pad_end, float pad_value, float* out,
                  kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    FrameArgs a;
    if (int e = frame_args(batch, channels, time, layout, frame_length, hop_length, pad_end, pad_value, &a))
        return e;
    const long long nrows = (a.cl ? (long long)batch : a.n_sig) * a.F;
    if (nrows == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    const int rowlen = a.cl ? a.L * a.C : a.L;
    const bool vec = (rowlen & 3) == 0 && (((uintptr_t)out) & 15) == 0;
    const long long work = nrows * (vec ? rowlen / 4 : rowlen);
    if (vec)
        hipLaunchKernelGGL(k_frame<4>, dim3(grid_1d(work, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x,
                           a, out, nrows);
    else
        hipLaunchKernelGGL(k_frame<1>, dim3(grid_1d(work, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x,
                           a, out, nrows);
    return launch_check("k_frame");
}

int kpr_energy_f32(const float* x, int64_t batch, int channels, int64_t time, int layout,
                   int frame_length, int hop_length, int pad_end, float pad_value, float scale,
                   float* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    FrameArgs a;
    if (int e = frame_args(batch, channels, time, layout, frame_length, hop_length, pad_end, pad_value, &a))
        return e;
    a.cl = layout == KPR_CHANNELS_LAST;            // output order depends on it even for one channel
    const long long nout = a.n_sig * a.F;
    if (nout == 0) return 0;
    if (!x || !out) return fail(KPR_E_BADARG, "x / out must not be NULL");
    if (time == 0) return fail(KPR_E_UNSUPPORTED, "energy of an empty signal with pad_end");
    const int chunks = (a.F + kEnFrames - 1) / kEnFrames;
    const int q = frame_length / hop_length;
    size_t lds = sizeof(float) * 2 * (size_t)(kEnFrames + q + 1);
    if (lds > 64 * 1024) return fail(KPR_E_UNSUPPORTED, "frame_length / hop_length = %d is too large", q);
    // room for one partial sum per float4 of a workgroup's range (streaming path), when it fits 64 KiB
    int part_words = 0;
    if (hop_length % 4 == 0) {
        const size_t pw = (size_t)(kEnFrames + q + 1) * (hop_length / 4);
        if (lds + sizeof(float) * pw <= 64 * 1024) { part_words = (int)pw; lds += sizeof(float) * pw; }
    }
    hipLaunchKernelGGL(k_energy, dim3((unsigned)std::min<long long>(a.n_sig * chunks, 1 << 16)), dim3(256), lds,
                       (hipStream_t)stream, x, a, scale, out, chunks, part_words);
    return launch_check("k_energy");
}

int kpr_delta_f32(const float* x, int64_t batch, int channels, int64_t frames, int n_freq, int layout,
                  int win_length, int pad_mode, float* out, kpr_stream_t stream) {
    if (int e = api_enter()) return e;
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || (unsigned)layout > 1u)
        return fail(KPR_E_BADARG, "bad batch/channels/frames/n_freq/layout");
    if (win_length < 3 || (win_length & 1) == 0)
        return fail(KPR_E_BADARG, "win_length must be odd and >= 3, got %d", win_length);
    if (pad_mode < 0 || pad_mode > 2) return fail(KPR_E_BADARG, "bad pad mode %d", pad_mode);
    const long long total = (long long)batch * channels * frames * n_freq;
    if (total == 0) return 0;
    if (!x || !out || x == out) return fail(KPR_E_BADARG, "x / out must not be NULL or aliased");
    const int n = (win_length - 1) / 2;
    double denom = 0;
    for (int i = 1; i <= n; ++i) denom += 2.0 * i * i;
    const long long outer = layout == KPR_CHANNELS_LAST ? batch : batch * channels;
    const long long inner = layout == KPR_CHANNELS_LAST ? (long long)n_freq * channels : n_freq;
    const bool vec = (inner & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(k_delta<4>, dim3(grid_1d(total / 4, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream,
                           x, outer, (long long)frames, inner, n, pad_mode, (float)(1.0 / denom), out);
    else
        hipLaunchKernelGGL(k_delta<1>, dim3(grid_1d(total, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x,
                           outer, (long long)frames, inner, n, pad_mode, (float)(1.0 / denom), out);
    return launch_check("k_delta");
}

/* ---- Frame / Energy / Delta backward (kpr_grad_kernels.h) ---------------------------------- */
int kpr_frame_bwd_f32(const float* g, int64_t batch, int channels, int64_t time, int layout, int frame_length,
                      int hop_length, int pad_end, float* gx, kpr_stream_t stream) {
    FrameArgs a;
    if (int e = frame_args(batch, channels, time, layout, frame_length, hop_length, pad_end, 0.0f, &a)) return e;
    a.cl = layout == KPR_CHANNELS_LAST;
    const long long total = a.n_sig * a.T;
    if (total == 0) return 0;
    if (!gx || (!g && a.F > 0)) return fail(KPR_E_BADARG, "g / gx must not be NULL");
    hipLaunchKernelGGL(k_frame_bwd, dim3(grid_1d(total, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, g, a, gx);
    return launch_check("k_frame_bwd");
}

int kpr_energy_bwd_f32(const float* x, const float* g, int64_t batch, int channels, int64_t time, int layout,
                       int frame_length, int hop_length, int pad_end, float scale, float* gx, kpr_stream_t stream) {
    FrameArgs a;
    if (int e = frame_args(batch, channels, time, layout, frame_length, hop_length, pad_end, 0.0f, &a)) return e;
    a.cl = layout == KPR_CHANNELS_LAST;
    const long long total = a.n_sig * a.T;
    if (total == 0) return 0;
    if (!x || !gx || (!g && a.F > 0)) return fail(KPR_E_BADARG, "x / g / gx must not be NULL");
    hipLaunchKernelGGL(k_energy_bwd, dim3(grid_1d(total, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, x, g, a,
                       scale, gx);
    return launch_check("k_energy_bwd");
}

int kpr_delta_bwd_f32(const float* g, int64_t batch, int channels, int64_t frames, int n_freq, int layout,
                      int win_length, int pad_mode, float* gx, kpr_stream_t stream) {
    if (batch < 0 || channels <= 0 || frames < 0 || n_freq <= 0 || (unsigned)layout > 1u)
        return fail(KPR_E_BADARG, "bad batch/channels/frames/n_freq/layout");
    if (win_length < 3 || (win_length & 1) == 0)
        return fail(KPR_E_BADARG, "win_length must be odd and >= 3, got %d", win_length);
    if (pad_mode < 0 || pad_mode > 2) return fail(KPR_E_BADARG, "bad pad mode %d", pad_mode);
    const long long total = (long long)batch * channels * frames * n_freq;
    if (total == 0) return 0;
    if (!g || !gx || g == gx) return fail(KPR_E_BADARG, "g / gx must not be NULL or aliased");
    const int n = (win_length - 1) / 2;
    double denom = 0;
    for (int i = 1; i <= n; ++i) denom += 2.0 * i * i;
    const long long outer = layout == KPR_CHANNELS_LAST ? batch : batch * channels;
    const long long inner = layout == KPR_CHANNELS_LAST ? (long long)n_freq * channels : n_freq;
    hipLaunchKernelGGL(k_delta_bwd, dim3(grid_1d(total, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, g, outer,
                       (long long)frames, inner, n, pad_mode, (float)(1.0 / denom), gx);
    return launch_check("k_delta_bwd");
}

}  // extern "C"
