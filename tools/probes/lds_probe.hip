// LDS instruction throughput per CU on gfx950 for the ops the FFT kernels use:
// ds_bpermute_b32, ds_write_b32, ds_write_b64, ds_read_b32, ds_read_b64 (conflict-free addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void probe(float* out, int iters) {
    __shared__ float lds[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (wave * 1024 + lane) * 4;              // b32: conflict-free
    unsigned addr8 = (wave * 1024 + lane * 2) * 4;         // b64
    unsigned perm = ((64 - lane) & 63) * 4;                // bpermute byte index (lane reversal)
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = lane + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w[8]; for (int i = 0; i < 8; ++i) w[i] = f2{(float)lane, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(v[i]) : "v"(perm));
                if (OP == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v[i]), "i"(i * 256) : "memory");
                if (OP == 2) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr8), "v"(w[i]), "i"(i * 512) : "memory");
                if (OP == 3) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "i"(i * 256));
                if (OP == 4) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[i]) : "v"(addr8), "i"(i * 512));
                if (OP == 5) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
                if (OP == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float acc = 0; for (int i = 0; i < 8; ++i) acc += v[i] + w[i].x + w[i].y;
    if (acc == 12345.678f) out[0] = acc + lds[0];
}
template <int OP> void run(const char* name) {
    float* d; hipMalloc(&d, 16);
    const int iters = 20000;
    for (int wps : {1, 2, 3}) {
        const int threads = 64 * 4 * wps;
        probe<OP><<<256, threads>>>(d, 100); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); probe<OP><<<256, threads>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_cu = ms * 1e6 / (iters * 32.0) / (4 * wps);   // ns per instruction per CU
        printf("%-22s waves/SIMD %d: %.2f ns per wave-instr, %.2f ns per instruction per CU\n", name, wps, ms * 1e6 / (iters * 32.0), per_cu);
    }
}
int main() {
    run<0>("ds_bpermute_b32"); run<1>("ds_write_b32"); run<2>("ds_write_b64"); run<3>("ds_read_b32"); run<4>("ds_read_b64");
    run<5>("v_mov_b32_dpp row_mirror"); run<6>("v_permlane32_swap_b32");
    return 0;
}
