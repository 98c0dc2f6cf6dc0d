// VALU dependent-issue latency on gfx950: a stream of v_pk_fma_f32 (or v_pk_add_f32) where each
// instruction depends on the one DIST instructions earlier, with 1 / 2 / 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int DIST, int OP>
__global__ void probe(float* out, int iters, float seed) {
    f2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = f2{seed + i, seed - i};
    f2 k = f2{1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int i = u % DIST;
            if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
            if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
            if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(k.x));
        }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
    if (acc == 12345.678f) out[0] = acc;
}
template <int DIST, int OP>
void run(const char* name) {
    float* d; hipMalloc(&d, 16);
    const int iters = 20000;
    for (int wps : {1, 2, 3}) {
        const int threads = 64 * 4 * wps;          // one block per CU
        probe<DIST, OP><<<256, threads>>>(d, 100, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<DIST, OP><<<256, threads>>>(d, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-14s dist %d  waves/SIMD %d: %.2f ns per wave-instr, %.2f ns per SIMD-instr\n", name, DIST, wps,
               ms * 1e6 / (iters * 32.0), ms * 1e6 / (iters * 32.0) / wps);
    }
    hipFree(d);
}
int main() {
    run<1, 0>("v_pk_fma_f32"); run<2, 0>("v_pk_fma_f32"); run<4, 0>("v_pk_fma_f32"); run<8, 0>("v_pk_fma_f32");
    run<1, 1>("v_pk_add_f32"); run<2, 1>("v_pk_add_f32"); run<4, 1>("v_pk_add_f32");
    run<1, 2>("v_add_f32"); run<2, 2>("v_add_f32"); run<4, 2>("v_add_f32");
    return 0;
}
