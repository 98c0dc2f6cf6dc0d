// Does an fp32 MFMA share the SIMD's execution resources with VALU work of OTHER waves?
// Workgroup = 8 waves (2 per SIMD): waves 0..3 run `role_a`, waves 4..7 run `role_b`.
// Wall-clock per configuration (HIP events), one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float run_valu(int iters, float seed) {
    f2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = f2{seed + i, seed - i};
    f2 k = f2{1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
    return acc;
}
template <int OP>
__device__ __forceinline__ float run_valu2(int iters, float seed) {
    f2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = f2{seed + i, seed - i};
    f2 k = f2{1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(k.x));
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i].x) : "v"(k.x));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i].x) : "v"(k.x));
            }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
    return acc;
}
template <int CHAINS>
__device__ __forceinline__ float run_mfma(int iters, float seed) {
    f32x4 acc[CHAINS];
    for (int i = 0; i < CHAINS; ++i) acc[i] = f32x4{seed, 0, 0, 0};
    float a = seed, b = 1.0f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 32 / CHAINS; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    float s = 0;
    for (int i = 0; i < CHAINS; ++i) s += acc[i][0];
    return s;
}
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
template <int KIND>
__device__ __forceinline__ float run_mfma_bf16(int iters, float seed) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, 0, 0, 0};
    s4 a4 = {1, 2, 3, 4}, b4 = {1, 1, 1, 1};
    s8 a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
                else           asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
            }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    return s;
}
// mode: 0 = A only valu, 1 = A only mfma(2 chains), 2 = A valu + B mfma, 3 = A valu + B valu,
//       4 = A only mfma (8 chains), 5 = A mfma8 + B mfma8, 6 = A valu + B mfma(8 chains)
__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode, float seed) {
    const int wave = threadIdx.x >> 6;
    float r = 0;
    const bool A = wave < 4;
    if (mode == 0) { if (A) r = run_valu(iters, seed); }
    if (mode == 1) { if (A) r = run_mfma<2>(iters, seed); }
    if (mode == 2) { r = A ? run_valu(iters, seed) : run_mfma<2>(iters, seed); }
    if (mode == 3) { r = run_valu(iters, seed); }
    if (mode == 4) { if (A) r = run_mfma<8>(iters, seed); }
    if (mode == 5) { r = run_mfma<8>(iters, seed); }
    if (mode == 6) { r = A ? run_valu(iters, seed) : run_mfma<8>(iters, seed); }
    if (mode == 7) { if (A) r = run_mfma_bf16<0>(iters, seed); }
    if (mode == 8) { r = A ? run_valu(iters, seed) : run_mfma_bf16<0>(iters, seed); }
    if (mode == 9) { if (A) r = run_mfma_bf16<1>(iters, seed); }
    if (mode == 10) { r = A ? run_valu(iters, seed) : run_mfma_bf16<1>(iters, seed); }
    if (mode == 11) { if (A) r = run_valu2<0>(iters, seed); }
    if (mode == 12) { r = A ? run_valu2<0>(iters, seed) : run_mfma<8>(iters, seed); }
    if (mode == 13) { if (A) r = run_valu2<1>(iters, seed); }
    if (mode == 14) { r = A ? run_valu2<1>(iters, seed) : run_mfma<8>(iters, seed); }
    if (mode == 15) { if (A) r = run_valu2<2>(iters, seed); }
    if (mode == 16) { r = A ? run_valu2<2>(iters, seed) : run_mfma<8>(iters, seed); }
    if (mode == 17) { if (A) r = run_valu2<3>(iters, seed); }
    if (mode == 18) { r = A ? run_valu2<3>(iters, seed) : run_mfma<8>(iters, seed); }
    if (mode == 19) { r = A ? run_valu2<0>(iters, seed) : run_mfma_bf16<1>(iters, seed); }
    if (r == 12345.678f) out[0] = r;
}
int main() {
    float* d; hipMalloc(&d, 16);
    const int iters = 20000;
    const char* names[] = {"A: pk_fma alone (1 wave/SIMD)", "A: mfma 2 chains alone", "A: pk_fma + B: mfma 2 chains",
                           "A+B: pk_fma both (2 waves/SIMD)", "A: mfma 8 chains alone", "A+B: mfma 8 chains both", "A: pk_fma + B: mfma 8 chains", "A: mfma 16x16x16 bf16 alone", "A: pk_fma + B: mfma 16x16x16 bf16", "A: mfma 16x16x32 bf16 alone", "A: pk_fma + B: mfma 16x16x32 bf16", "A: v_add_f32 alone", "A: v_add_f32 + B: mfma f32", "A: v_fma_f32 alone", "A: v_fma_f32 + B: mfma f32", "A: v_pk_add_f32 alone", "A: v_pk_add_f32 + B: mfma f32", "A: v_xor alone", "A: v_xor + B: mfma f32", "A: v_add_f32 + B: mfma bf16x32"};
    for (int mode = 0; mode < 20; ++mode) {
        probe<<<256, 512>>>(d, 100, mode, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<<<256, 512>>>(d, iters, mode, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-36s %8.3f ms  -> %.2f ns per instruction of the slower role (32*iters per wave)\n", names[mode], ms, ms * 1e6 / (iters * 32.0));
    }
    return 0;
}
