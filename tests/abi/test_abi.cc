// test_abi.cc -- torch-free use of the C ABI (include/kapre_hip.h): hipMalloc + a raw hipStream_t, nothing else.
// Proves the header alone is sufficient for a non-Python binder.  Built by kapre_amd/build.py (hipcc, host code
// only + HIP runtime) into tests/abi/test_abi; run by tests/test_abi_native.py (-m gpu) on case files it dumps:
//
//   test_abi mel   <case.bin> <rel_tol>      kpr_filterbank_kranges -> kpr_filterbank_pack -> kpr_mel_f32
//   test_abi istft <case.bin> <rel_tol>      kpr_istft_f32
//   test_abi fb    <case.bin> <rel_tol>      kpr_filterbank_kranges -> kpr_filterbank_pack -> kpr_apply_filterbank_packed_f32 (round 6:
//                                            the banded row kernel k_fb_pw, named by kpr_last_launches(); the case's last row
//                                            carries a NaN bin: every filter of that row must come back non-finite)
//
// case.bin = int64 meta[16], then float32 arrays in the order read below.  Exit code 0 = parity within tolerance.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/kapre_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define KPR_OK_(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, kpr_last_error()); return 3; } } while (0)

static bool read_floats(FILE* f, std::vector<float>& v, size_t n) {
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(float), n, f) == n;
}

template <typename T>
static int upload(const std::vector<T>& h, T** d) {
    HIP_OK(hipMalloc((void**)d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIP_OK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int compare(const std::vector<float>& got, const std::vector<float>& want, double rel, bool absolute) {
    double scale = 0, err = 0;
    for (size_t i = 0; i < want.size(); ++i) {
        scale = std::fmax(scale, std::fabs((double)want[i]));
        const double e = std::fabs((double)got[i] - (double)want[i]);
        if (!(e == e)) { std::fprintf(stderr, "NaN at %zu\n", i); return 4; }
        err = std::fmax(err, e);
    }
    const double bound = absolute ? rel : rel * scale;
    std::printf("max |err| %.3g, bound %.3g (scale %.3g, %zu values)\n", err, bound, scale, want.size());
    return err <= bound ? 0 : 4;
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: test_abi mel|istft|fb case.bin rel_tol\n"); return 1; }
    const bool mel = std::strcmp(argv[1], "mel") == 0;
    const bool fbm = std::strcmp(argv[1], "fb") == 0;
    const double rel = std::atof(argv[3]);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f) { std::perror(argv[2]); return 1; }
    int64_t m[16];
    if (std::fread(m, sizeof(int64_t), 16, f) != 16) return 1;
    kpr_stft_geom g;
    g.batch = m[0]; g.channels = (int32_t)m[1]; g.time = m[2]; g.n_fft = (int32_t)m[3]; g.win_length = (int32_t)m[4];
    g.hop_length = (int32_t)m[5]; g.pad_begin = (int32_t)m[6]; g.pad_end = (int32_t)m[7];
    g.in_layout = (int32_t)m[8]; g.out_layout = (int32_t)m[9];
    const int n_filt = (int)m[10];
    const int64_t n_frames = m[12];
    const int K = g.n_fft / 2 + 1;
    if (kpr_version() < 100) return 1;
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    if (fbm) {
        // meta: [0] batch [1] channels [3] n_freq (in the n_fft slot) [8] layout [10] n_filt [12] frames; arrays: x, fb, want
        const int n_freq = g.n_fft, layout = g.in_layout;
        const size_t rows = (size_t)(g.batch * g.channels * n_frames);
        std::vector<float> x, fb, want;
        if (!read_floats(f, x, rows * n_freq) || !read_floats(f, fb, (size_t)n_freq * n_filt) || !read_floats(f, want, rows * n_filt)) return 1;
        std::vector<int32_t> kr(2 * ((n_filt + 15) / 16));
        KPR_OK_(kpr_filterbank_kranges(fb.data(), n_freq, n_filt, kr.data()));
        const int64_t pf = kpr_filterbank_pack_floats(n_freq, n_filt, kr.data());
        if (pf <= 0) { std::fprintf(stderr, "pack size: %s\n", kpr_last_error()); return 3; }
        std::vector<float> packed((size_t)pf);
        KPR_OK_(kpr_filterbank_pack(fb.data(), n_freq, n_filt, kr.data(), packed.data()));
        float *dx, *dfb, *dpk, *dout;
        if (upload(x, &dx) || upload(fb, &dfb) || upload(packed, &dpk)) return 2;
        HIP_OK(hipMalloc((void**)&dout, rows * n_filt * sizeof(float)));
        for (int rep = 0; rep < 2; ++rep)
            KPR_OK_(kpr_apply_filterbank_packed_f32(dx, g.batch, g.channels, n_frames, n_freq, layout, dfb, dpk, n_filt, kr.data(), dout,
                                                    (kpr_stream_t)stream));
        if (!std::strstr(kpr_last_launches(), "k_fb_pw")) { std::fprintf(stderr, "expected k_fb_pw, launched [%s]\n", kpr_last_launches()); return 4; }
        HIP_OK(hipStreamSynchronize(stream));
        unsigned flags = 0;
        KPR_OK_(kpr_device_status(&flags));                   // (what a binder checks after its batch: INTEGRATION.md section 3)
        std::vector<float> got(rows * n_filt);
        HIP_OK(hipMemcpy(got.data(), dout, got.size() * sizeof(float), hipMemcpyDeviceToHost));
        // the last row holds a NaN bin: the reference's dense tensordot leaves no finite value in it
        for (int mfil = 0; mfil < n_filt; ++mfil)
            if (std::isfinite(got[(rows - 1) * n_filt + mfil])) { std::fprintf(stderr, "filter %d of the NaN row is finite\n", mfil); return 4; }
        got.resize((rows - 1) * n_filt);
        want.resize((rows - 1) * n_filt);
        return compare(got, want, rel, false);
    }
    if (mel) {
        kpr_db_params db;
        db.enabled = (int32_t)m[11];
        std::vector<float> dbp, x, window, fb, want;
        if (!read_floats(f, dbp, 4) || !read_floats(f, x, (size_t)(g.batch * g.channels * g.time)) ||
            !read_floats(f, window, g.win_length) || !read_floats(f, fb, (size_t)K * n_filt))
            return 1;
        db.ref_value = dbp[0]; db.amin = dbp[1]; db.dynamic_range = dbp[2];
        if (kpr_num_frames(&g) != n_frames) { std::fprintf(stderr, "frame count %lld != %lld\n", (long long)kpr_num_frames(&g), (long long)n_frames); return 4; }
        const size_t n_out = (size_t)(g.batch * g.channels * n_frames) * n_filt;
        if (!read_floats(f, want, n_out)) return 1;
        // host-side constants: k-ranges, packed blob
        std::vector<int32_t> kr(2 * ((n_filt + 15) / 16));
        KPR_OK_(kpr_filterbank_kranges(fb.data(), K, n_filt, kr.data()));
        const int64_t pf = kpr_filterbank_pack_floats(K, n_filt, kr.data());
        if (pf <= 0) { std::fprintf(stderr, "pack size: %s\n", kpr_last_error()); return 3; }
        std::vector<float> packed((size_t)pf);
        KPR_OK_(kpr_filterbank_pack(fb.data(), K, n_filt, kr.data(), packed.data()));
        float *dx, *dw, *dfb, *dpk, *dout;
        if (upload(x, &dx) || upload(window, &dw) || upload(fb, &dfb) || upload(packed, &dpk)) return 2;
        HIP_OK(hipMalloc((void**)&dout, n_out * sizeof(float)));
        const int64_t ws_bytes = kpr_mel_workspace_bytes(&g, n_filt, &db);
        void* ws;
        HIP_OK(hipMalloc(&ws, (size_t)ws_bytes));
        for (int rep = 0; rep < 2; ++rep)        // twice: first use verifies the blob header, second is steady state
            KPR_OK_(kpr_mel_f32(dx, &g, dw, dfb, dpk, n_filt, kr.data(), &db, dout, ws, ws_bytes, (kpr_stream_t)stream));
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<float> got(n_out);
        HIP_OK(hipMemcpy(got.data(), dout, n_out * sizeof(float), hipMemcpyDeviceToHost));
        // a blob that belongs to other k-ranges must be refused, not trusted
        std::vector<int32_t> kr2(kr);
        kr2[1] = std::min(kr2[1] + 32, (K + 3) & ~3);
        const int rc = kpr_mel_f32(dx, &g, dw, dfb, dpk, n_filt, kr2.data(), &db, dout, ws, ws_bytes, (kpr_stream_t)stream);
        if (rc != KPR_E_BADARG) { std::fprintf(stderr, "mismatched k-ranges were accepted (rc %d)\n", rc); return 4; }
        // a caller that releases the blob says so; the next use of the address is checked from scratch (and accepted again)
        KPR_OK_(kpr_filterbank_forget(dpk));
        KPR_OK_(kpr_mel_f32(dx, &g, dw, dfb, dpk, n_filt, kr.data(), &db, dout, ws, ws_bytes, (kpr_stream_t)stream));
        HIP_OK(hipStreamSynchronize(stream));
        return compare(got, want, rel, db.enabled != 0);
    }
    // istft: spec (complex64 interleaved), synthesis window, expected waveform
    std::vector<float> spec, window, want;
    const size_t n_spec = (size_t)(g.batch * g.channels * n_frames) * K * 2;
    const size_t t_out = (size_t)((n_frames - 1) * g.hop_length + g.win_length);
    if (!read_floats(f, spec, n_spec) || !read_floats(f, window, g.win_length) ||
        !read_floats(f, want, (size_t)(g.batch * g.channels) * t_out))
        return 1;
    float *dspec, *dw, *dout;
    if (upload(spec, &dspec) || upload(window, &dw)) return 2;
    HIP_OK(hipMalloc((void**)&dout, want.size() * sizeof(float)));
    const int64_t ws_bytes = kpr_istft_workspace_bytes(&g, n_frames);
    void* ws;
    HIP_OK(hipMalloc(&ws, (size_t)ws_bytes));
    KPR_OK_(kpr_istft_f32(dspec, &g, n_frames, dw, dout, ws, ws_bytes, (kpr_stream_t)stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> got(want.size());
    HIP_OK(hipMemcpy(got.data(), dout, got.size() * sizeof(float), hipMemcpyDeviceToHost));
    return compare(got, want, rel, false);
}
