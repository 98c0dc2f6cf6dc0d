// VALU issue-rate probe for gfx950: cycles per wave-instruction for scalar and packed f32 ops,
// as a function of waves per SIMD.  Development aid (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(long long* out, int iters, float seed) {
    f2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = f2{seed + i, seed - i};
    f2 k = f2{1.0001f, 0.9999f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(k.x));
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i].x) : "v"(k.x));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
                if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(r[i]) : "v"(k));
                if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i].x) : "v"(k.x));
                if (OP == 7) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i].x));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)acc; }
}

template <int OP>
void run(const char* name) {
    long long* d; hipMalloc(&d, 16);
    const int iters = 20000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int threads = 64 * 4 * waves_per_simd;       // one block per CU: 4 SIMDs
        if (threads > 1024) threads = 1024;
        int blocks = 256 * (64 * 4 * waves_per_simd / threads);
        probe<OP><<<blocks, threads>>>(d, iters, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<OP><<<blocks, threads>>>(d, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double per = (double)h[0] / (iters * 32.0);
        double ns_per = ms * 1e6 / (iters * 32.0);
        printf("%-28s waves/SIMD=%d  %.2f ticks per wave-instr (%.2f per SIMD) | wall %.2f ns per wave-instr, %.3f ns per SIMD-instr, tick=%.3f ns\n",
               name, waves_per_simd, per, per / waves_per_simd, ns_per, ns_per / waves_per_simd, ms * 1e6 / (double)h[0]);
    }
    hipFree(d);
}

int main() {
    run<0>("v_add_f32");
    run<1>("v_fma_f32");
    run<2>("v_pk_add_f32");
    run<3>("v_pk_fma_f32");
    run<4>("v_pk_mul_f32");
    run<5>("v_pk_add_f32 op_sel+neg");
    run<6>("v_xor_b32");
    run<7>("v_sqrt_f32");
    return 0;
}
