// valu_micro.hip -- issue cost of single gfx950 instructions and short sequences, by waves per SIMD.
//
// Every test is a block of 64 instructions on 8 independent register chains, repeated `iters` times by 1..8 waves per
// SIMD (one workgroup per CU, 256 workgroups); reported: cycles per wave-instruction on one SIMD at the clock measured
// in the loop (s_memtime / s_memrealtime).  Also: the cross-lane primitives an LDS-free FFT exchange would be built of
// (v_permlane32_swap / v_permlane16_swap, DPP moves with bank masks, v_cndmask), packed-f32 operands on the same /
// different VGPR banks, and MFMA issued by the same wave between packed adds (do the matrix and vector pipes overlap
// inside ONE wave).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Regs {
    float a[8], b[8];
    f2 p[8], q[8];
};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

struct T_add   { static constexpr const char* name = "v_add_f32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_fma   { static constexpr const char* name = "v_fma_f32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_fmac  { static constexpr const char* name = "v_fmac_f32 (VOP2)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_mul   { static constexpr const char* name = "v_mul_f32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_mov   { static constexpr const char* name = "v_mov_b32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_cnd   { static constexpr const char* name = "v_cndmask_b32 (vcc)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r.a[i]) : "v"(r.b[i]) : );
        BLOCK(X)
#undef X
    } };
struct T_pkadd { static constexpr const char* name = "v_pk_add_f32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r.p[i]) : "v"(r.q[i]));
        BLOCK(X)
#undef X
    } };
struct T_pkaddm { static constexpr const char* name = "v_pk_add_f32 op_sel/neg"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(r.p[i]) : "v"(r.q[i]));
        BLOCK(X)
#undef X
    } };
struct T_pkmul { static constexpr const char* name = "v_pk_mul_f32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r.p[i]) : "v"(r.q[i]));
        BLOCK(X)
#undef X
    } };
struct T_pkfma { static constexpr const char* name = "v_pk_fma_f32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r.p[i]) : "v"(r.q[i]));
        BLOCK(X)
#undef X
    } };
struct T_pkmov { static constexpr const char* name = "v_pk_mov_b32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[0,1]" : "+v"(r.p[i]) : "v"(r.q[i]));
        BLOCK(X)
#undef X
    } };
struct T_sqrt  { static constexpr const char* name = "v_sqrt_f32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r.a[i]));
        BLOCK(X)
#undef X
    } };
// packed add on FIXED registers: both sources on the same VGPR bank pair (register numbers equal mod 4) vs not
struct T_pkbank_same { static constexpr const char* name = "v_pk_add_f32, both sources on the same VGPR banks (regs equal mod 4)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs&) {
#define X(i) asm volatile("v_pk_add_f32 v[16:17], v[20:21], v[24:25]\n\tv_pk_add_f32 v[28:29], v[32:33], v[36:37]\n\tv_pk_add_f32 v[16:17], v[20:21], v[24:25]\n\tv_pk_add_f32 v[28:29], v[32:33], v[36:37]\n\tv_pk_add_f32 v[16:17], v[20:21], v[24:25]\n\tv_pk_add_f32 v[28:29], v[32:33], v[36:37]\n\tv_pk_add_f32 v[16:17], v[20:21], v[24:25]\n\tv_pk_add_f32 v[28:29], v[32:33], v[36:37]" ::: "v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37");
        REP8(X)
#undef X
    } };
struct T_pkbank_diff { static constexpr const char* name = "v_pk_add_f32 (src regs differ mod 4)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs&) {
#define X(i) asm volatile("v_pk_add_f32 v[16:17], v[20:21], v[26:27]\n\tv_pk_add_f32 v[28:29], v[32:33], v[38:39]\n\tv_pk_add_f32 v[16:17], v[20:21], v[26:27]\n\tv_pk_add_f32 v[28:29], v[32:33], v[38:39]\n\tv_pk_add_f32 v[16:17], v[20:21], v[26:27]\n\tv_pk_add_f32 v[28:29], v[32:33], v[38:39]\n\tv_pk_add_f32 v[16:17], v[20:21], v[26:27]\n\tv_pk_add_f32 v[28:29], v[32:33], v[38:39]" ::: "v16","v17","v20","v21","v26","v27","v28","v29","v32","v33","v38","v39");
        REP8(X)
#undef X
    } };
struct T_dppq  { static constexpr const char* name = "v_mov_b32_dpp quad_perm:[1,0,3,2]"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_dppm  { static constexpr const char* name = "v_mov_b32_dpp row_half_mirror bank_mask:0x5"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_dppadd { static constexpr const char* name = "v_add_f32_dpp row_mirror"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_dppchain { static constexpr const char* name = "v_mov_b32_dpp chain (dst feeds next dpp src)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r.a[i]));
        BLOCK(X)
#undef X
    } };
struct T_swap32 { static constexpr const char* name = "v_permlane32_swap_b32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r.a[i]), "+v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_swap16 { static constexpr const char* name = "v_permlane16_swap_b32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r.a[i]), "+v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
// swap followed by a packed add that consumes both results (the hazard the compiler pads with s_nop 1)
struct T_swapuse { static constexpr const char* name = "v_permlane32_swap + s_nop 1 + v_pk_add on the pair (per 2 instr)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(r.a[i]), "+v"(r.b[i]));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } };
struct T_bperm { static constexpr const char* name = "ds_bpermute_b32 (8 in flight, then wait)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(r.a[i]) : "v"(r.b[i]));
#define W asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        REP8(X) W REP8(X) W REP8(X) W REP8(X) W REP8(X) W REP8(X) W REP8(X) W REP8(X) W
#undef X
#undef W
    } };
// 1 MFMA (16x16x4 f32, 32 matrix-pipe cycles) + 6 packed adds (24 vector cycles) per group, 8 groups: overlap inside one wave?
struct T_mfma { static constexpr const char* name = "v_mfma_f32_16x16x4_f32 only (per MFMA)"; static constexpr int n = 16;
    static __device__ __forceinline__ void run(Regs& r) {
        f4 c0 = {r.a[0], r.a[1], r.a[2], r.a[3]}, c1 = {r.a[4], r.a[5], r.a[6], r.a[7]};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.b[i], r.b[(i + 1) & 7], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.b[(i + 2) & 7], r.b[(i + 3) & 7], c1, 0, 0, 0);
        }
        r.a[0] = c0.x; r.a[1] = c0.y; r.a[2] = c0.z; r.a[3] = c0.w; r.a[4] = c1.x; r.a[5] = c1.y; r.a[6] = c1.z; r.a[7] = c1.w;
    } };
struct T_mfma_pk { static constexpr const char* name = "1 MFMA + 6 v_pk_add_f32 per group (per group of 7)"; static constexpr int n = 16;
    static __device__ __forceinline__ void run(Regs& r) {
        f4 c0 = {r.a[0], r.a[1], r.a[2], r.a[3]}, c1 = {r.a[4], r.a[5], r.a[6], r.a[7]};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.b[i], r.b[(i + 1) & 7], c0, 0, 0, 0);
#define X(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r.p[j]) : "v"(r.q[j]));
            X(0) X(1) X(2) X(3) X(4) X(5)
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.b[(i + 2) & 7], r.b[(i + 3) & 7], c1, 0, 0, 0);
            X(6) X(7) X(0) X(1) X(2) X(3)
#undef X
        }
        r.a[0] = c0.x; r.a[1] = c0.y; r.a[2] = c0.z; r.a[3] = c0.w; r.a[4] = c1.x; r.a[5] = c1.y; r.a[6] = c1.z; r.a[7] = c1.w;
    } };
// 4 dependent MFMAs into one accumulator (the "transpose through the matrix pipe" pattern: D = sum_q A_q x P_q)
struct T_mfma_chain { static constexpr const char* name = "4-deep dependent MFMA chains x4 accumulators (per MFMA)"; static constexpr int n = 16;
    static __device__ __forceinline__ void run(Regs& r) {
        f4 c[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) c[g] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) c[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[(g + q) & 7], r.b[q], c[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) { r.a[2 * g] += c[g].x + c[g].z; r.a[2 * g + 1] += c[g].y + c[g].w; }
    } };

struct T_bfi   { static constexpr const char* name = "v_bfi_b32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r.a[i]) : "v"(r.b[i]), "v"(r.b[(i + 1) & 7]));
        BLOCK(X)
#undef X
    } };
struct T_cnd64 { static constexpr const char* name = "v_cndmask_b32_e64 (mask in an SGPR pair)"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
        unsigned long long msk = 0x5555555555555555ull;
        asm volatile("" : "+s"(msk));
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r.a[i]) : "v"(r.b[i]), "s"(msk));
        BLOCK(X)
#undef X
    } };
struct T_xor   { static constexpr const char* name = "v_xor_b32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
struct T_lshladd { static constexpr const char* name = "v_lshl_add_u32"; static constexpr int n = 64;
    static __device__ __forceinline__ void run(Regs& r) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r.a[i]) : "v"(r.b[i]));
        BLOCK(X)
#undef X
    } };
// 8 INDEPENDENT accumulators from one wave: is a lone wave's MFMA rate bound by the pipe or by its own issue?
struct T_mfma8 { static constexpr const char* name = "v_mfma_f32_16x16x4_f32, 8 independent accumulators (per MFMA)"; static constexpr int n = 16;
    static __device__ __forceinline__ void run(Regs& r) {
        f4 c[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) c[g] = f4{r.a[g], 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 8; ++g) c[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.b[g], r.b[(g + q + 1) & 7], c[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) r.a[g] = c[g].x + c[g].y + c[g].z + c[g].w;
    } };
struct T_mfma_bf16 { static constexpr const char* name = "v_mfma_f32_16x16x32_bf16, 8 independent accumulators (per MFMA)"; static constexpr int n = 16;
    static __device__ __forceinline__ void run(Regs& r) {
        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
        f4 c[8];
        bf8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)r.b[i]; b[i] = (__bf16)r.a[i]; }
#pragma unroll
        for (int g = 0; g < 8; ++g) c[g] = f4{r.a[g], 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 8; ++g) c[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) r.a[g] = c[g].x + c[g].y + c[g].z + c[g].w;
    } };

// WPS <= 4: one workgroup of 4 WPS waves per CU; WPS == 8: two 16-wave workgroups per CU (512 workgroups)
template <class T, int WPS>
__global__ __launch_bounds__((WPS > 4 ? 4 : WPS) * 256, WPS) void k_micro(float* __restrict__ sink, int iters, unsigned long long* __restrict__ stamps) {
    Regs r;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.a[i] = 1.0f + lane * 0.001f + i;
        r.b[i] = (float)((lane * 4 + i * 8) & 255);        // also a valid bpermute byte address
        r.p[i] = f2{r.a[i], 0.5f};
        r.q[i] = f2{1e-3f, 2e-3f};
    }
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        T::run(r);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(r.a[i]), "+v"(r.b[i]), "+v"(r.p[i]), "+v"(r.q[i]));
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        unsigned long long* s = stamps + 4ull * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        s[0] = t0; s[1] = r0; s[2] = t1; s[3] = r1;
    }
    float v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v += r.a[i] + r.b[i] + r.p[i].x + r.p[i].y + r.q[i].x;
    if (v == 1.2345e-30f) sink[blockIdx.x] = v;
}

template <class T, int WPS>
static void run1(float* d_sink, unsigned long long* d_st, int iters, double* cyc_per_instr, double* mhz_out, double* wall_ns = nullptr, double* span = nullptr) {
    const int threads = (WPS > 4 ? 4 : WPS) * 256, grid = 256 * WPS * 256 / threads;
    k_micro<T, WPS><<<grid, threads>>>(d_sink, iters, d_st);
    HIP_OK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0));
    k_micro<T, WPS><<<grid, threads>>>(d_sink, iters, d_st);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipDeviceSynchronize());
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (wall_ns) *wall_ns = (double)ms * 1e6 / ((double)iters * T::n * WPS);      // wall ns per wave-instruction on one SIMD
    HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
    std::vector<unsigned long long> st(4ull * 256 * WPS * 4);
    HIP_OK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mhz = 0, cyc = 0; int n = 0;
    unsigned long long tmin = ~0ull, tmax = 0, rmin = ~0ull, rmax = 0;
    for (size_t w = 0; w < st.size() / 4; ++w) {
        const double dc = (double)(st[4 * w + 2] - st[4 * w]), dr = (double)(st[4 * w + 3] - st[4 * w + 1]);
        if (dr > 0) { mhz += dc / dr * 100.0; ++n; }
        cyc += dc;
        tmin = std::min(tmin, st[4 * w]); tmax = std::max(tmax, st[4 * w + 2]);
        rmin = std::min(rmin, st[4 * w + 1]); rmax = std::max(rmax, st[4 * w + 3]);
    }
    cyc /= (double)(st.size() / 4);
    *mhz_out = mhz / (n ? n : 1);
    if (span) { span[0] = (double)(tmax - tmin); span[1] = (double)(rmax - rmin); span[2] = cyc; span[3] = (double)ms * 1e3; }
    // WPS waves share the SIMD: per wave-instruction on the SIMD = loop cycles / (instructions per wave * WPS)
    *cyc_per_instr = cyc / ((double)iters * T::n * WPS);
}

template <class T>
static void row(float* d_sink, unsigned long long* d_st, int iters) {
    double c[4], m[4], w[4];
    run1<T, 1>(d_sink, d_st, iters, &c[0], &m[0], &w[0]);
    run1<T, 2>(d_sink, d_st, iters, &c[1], &m[1], &w[1]);
    run1<T, 4>(d_sink, d_st, iters, &c[2], &m[2], &w[2]);
    run1<T, 8>(d_sink, d_st, iters, &c[3], &m[3], &w[3]);
    // cycles per wave-instruction on one SIMD = WALL ns x clock (the in-loop mean per wave is biased: oldest-first
    // arbitration lets the waves of a SIMD finish one after another -- the calibration rows above show both)
    std::printf("| %s | %.2f | %.2f | %.2f | %.2f | %.0f |\n", T::name, w[0] * m[0] * 1e-3, w[1] * m[1] * 1e-3, w[2] * m[2] * 1e-3, w[3] * m[3] * 1e-3, m[2]);
    (void)c;
}

template <class T, int WPS>
static void calib(float* d_sink, unsigned long long* d_st, int iters) {
    double c, m, w, sp[4];
    run1<T, WPS>(d_sink, d_st, iters, &c, &m, &w, sp);
    std::printf("calibration %s, %d waves/SIMD: first-start..last-end s_memtime ticks %.0f, s_memrealtime ticks %.0f, mean per-wave loop ticks %.0f, kernel wall %.1f us"
                " -> s_memtime %.1f MHz, s_memrealtime %.2f MHz (if the span were the whole kernel)\n",
                T::name, WPS, sp[0], sp[1], sp[2], sp[3], sp[0] / sp[3], sp[1] / sp[3]);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    float* d_sink; unsigned long long* d_st;
    HIP_OK(hipMalloc(&d_sink, 4096 * sizeof(float)));
    HIP_OK(hipMalloc(&d_st, 4ull * 256 * 32 * sizeof(unsigned long long)));
    calib<T_pkadd, 1>(d_sink, d_st, 20000);
    calib<T_pkadd, 4>(d_sink, d_st, 20000);
    calib<T_add, 4>(d_sink, d_st, 20000);
    calib<T_mfma, 4>(d_sink, d_st, 5000);
    std::printf("\ncycles per wave-instruction on one SIMD = wall time (HIP events) x shader clock measured in the loop, by waves per SIMD\n\n");
    std::printf("| instruction | 1 wave | 2 waves | 4 waves | 8 waves | sclk MHz (4 waves) |\n|---|---|---|---|---|---|\n");
    row<T_add>(d_sink, d_st, iters);
    row<T_mul>(d_sink, d_st, iters);
    row<T_mov>(d_sink, d_st, iters);
    row<T_cnd>(d_sink, d_st, iters);
    row<T_cnd64>(d_sink, d_st, iters);
    row<T_bfi>(d_sink, d_st, iters);
    row<T_xor>(d_sink, d_st, iters);
    row<T_lshladd>(d_sink, d_st, iters);
    row<T_fma>(d_sink, d_st, iters);
    row<T_fmac>(d_sink, d_st, iters);
    row<T_pkadd>(d_sink, d_st, iters);
    row<T_pkaddm>(d_sink, d_st, iters);
    row<T_pkmul>(d_sink, d_st, iters);
    row<T_pkfma>(d_sink, d_st, iters);
    row<T_pkmov>(d_sink, d_st, iters);
    row<T_pkbank_same>(d_sink, d_st, iters);
    row<T_pkbank_diff>(d_sink, d_st, iters);
    row<T_sqrt>(d_sink, d_st, iters);
    row<T_dppq>(d_sink, d_st, iters);
    row<T_dppm>(d_sink, d_st, iters);
    row<T_dppadd>(d_sink, d_st, iters);
    row<T_dppchain>(d_sink, d_st, iters);
    row<T_swap32>(d_sink, d_st, iters);
    row<T_swap16>(d_sink, d_st, iters);
    row<T_swapuse>(d_sink, d_st, iters);
    row<T_bperm>(d_sink, d_st, iters);
    row<T_mfma>(d_sink, d_st, iters);
    row<T_mfma8>(d_sink, d_st, iters);
    row<T_mfma_bf16>(d_sink, d_st, iters);
    row<T_mfma_pk>(d_sink, d_st, iters);
    row<T_mfma_chain>(d_sink, d_st, iters);
    return 0;
}
