// kpr_signal_kernels.h -- k_thin_gemm (LogmelToMFCC), k_frame, k_energy, k_delta (kapre/signal.py, time_frequency.py:563-644).
// Part of the single translation unit kapre_hip.hip (included there, in this order; not stand-alone).
#pragma once

namespace kpr {

// ------------------------------------------------------------------------------------------
// Thin GEMM: out[rows][N] = A[rows][K] x B[K][N] for small K and N (LogmelToMFCC: 80 x 13,
// 128 x 20, ...; any narrow ApplyFilterbank matrix on contiguous rows).  HBM-bound: A is read once
// with 16-byte loads (lane (m, kq) takes A[row m][16j + 4kq .. +3]; those four values feed four
// MFMA k-steps, the B fragments in LDS are stored in the matching order), every wave owns 16 rows
// per step and keeps all N-tiles' accumulators in registers.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_thin_gemm(const float* __restrict__ a, long long rows, int K,
                                                   const float* __restrict__ bm, int N,
                                                   float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int J = (K + 15) / 16;
    f32x4* bfrag = reinterpret_cast<f32x4*>(smem);               // [NT][J][64]
    for (int idx = threadIdx.x; idx < NT * J * 64; idx += blockDim.x) {
        const int l = idx & 63, j = (idx >> 6) % J, nt = (idx >> 6) / J;
        const int n = nt * 16 + (l & 15), kq = l >> 4;
        f32x4 v;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = 16 * j + 4 * kq + s4;
            v[s4] = (k < K && n < N) ? bm[(long long)k * N + n] : 0.0f;
        }
        bfrag[idx] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const long long nblk = (rows + 15) / 16;
    for (long long rb = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); rb < nblk; rb += (long long)gridDim.x * 4) {
        const long long row = rb * 16 + m;
        const float* ar = a + min(row, rows - 1) * K;            // rows past the end: clamped, never stored
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < J; ++j) {
            const int k0 = 16 * j + 4 * kq;
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (k0 + 3 < K) av = *reinterpret_cast<const f32x4*>(ar + k0);      // K % 4 == 0: all or nothing
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 bv = bfrag[(nt * J + j) * 64 + lane];
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc[nt], 0, 0, 0);
            }
        }
        // lane holds D[row 4*kq + r][col m]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long orow = rb * 16 + 4 * kq + r;
                if (orow < rows && n < N) out[orow * N + n] = acc[nt][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Frame / Energy / Delta (kapre/signal.py:22-213, time_frequency.py:563-644): bandwidth kernels
// ------------------------------------------------------------------------------------------
struct FrameArgs {
    long long n_sig;        // batch * channels
    long long T;
    int C, F, L, hop;
    int cl;                 // waveform (b, t, c) and frames (b, f, l, c) if 1; (b, c, t) / (b, c, f, l) if 0
    float pad_value;
};

// Output-stationary copy: every thread produces VEC consecutive output floats (one 16-byte store when
// VEC = 4) of one row; a row = one frame of one batch item (all channels, channels_last: out[b][f]
// is L*C contiguous floats and so is its source) or of one signal (channels_first).  The source of
// a float4 is only 4-byte aligned in general (hop is arbitrary), so it is read as four dwords --
// still fully coalesced across the wave.
template <int VEC>
__global__ __launch_bounds__(256) void k_frame(const float* __restrict__ x, FrameArgs a,
                                               float* __restrict__ out, long long nrows) {
    const int rowlen = a.cl ? a.L * a.C : a.L;
    const int per_row = rowlen / VEC;                           // VEC == 4 only when rowlen % 4 == 0
    const long long total = nrows * per_row;
    const bool small = total < (1LL << 31);                      // 32-bit index arithmetic (the usual case):
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;     // two 64-bit divisions
         e += (long long)gridDim.x * blockDim.x) {                                        // per 16 bytes cost more
        long long row, bq;                                                                // than the copy itself
        int i, f;
        if (small) {
            const unsigned e32 = (unsigned)e, row32 = e32 / (unsigned)per_row, bq32 = row32 / (unsigned)a.F;
            row = row32; bq = bq32;
            i = (int)(e32 - row32 * (unsigned)per_row) * VEC;
            f = (int)(row32 - bq32 * (unsigned)a.F);
        } else {
            row = e / per_row;
            i = (int)(e - row * per_row) * VEC;
            bq = row / a.F;                                      // batch item (cl) or signal b*C + c (cf)
            f = (int)(row - bq * a.F);
        }
        const long long t0 = (long long)f * a.hop;
        const float* src = a.cl ? x + (bq * a.T + t0) * a.C : x + bq * a.T + t0;
        const long long avail = (a.T - t0) * (a.cl ? a.C : 1);   // valid elements from src on
        float v[VEC];
        if (VEC == 4 && i + 3 < avail && (((unsigned long long)(src + i)) & 15ull) == 0) {
            const float4 q4 = *reinterpret_cast<const float4*>(src + i);    // hop, T multiples of 4: one 16-byte load
            v[0] = q4.x; v[1 % VEC] = q4.y; v[2 % VEC] = q4.z; v[3 % VEC] = q4.w;
        } else {
#pragma unroll
            for (int u = 0; u < VEC; ++u) v[u] = (i + u < avail) ? src[i + u] : a.pad_value;
        }
        float* dst = out + row * rowlen + i;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else dst[0] = v[0];
    }
}

// Energy: every sample is read ONCE.  With L = q*hop + r a frame is q whole hop-blocks plus the first
// r samples of the next one, so a workgroup (4 waves) that owns kEnFrames consecutive frames of one
// signal first reduces each of its kEnFrames + q hop-blocks to two numbers in LDS -- the block's
// sum of squares and the sum of its first r squares -- and then adds q + 1 of them per output.
// Samples beyond the end of the signal count as pad_value (tf.signal.frame pad_end semantics).
constexpr int kEnFrames = 64;

__global__ __launch_bounds__(256) void k_energy(const float* __restrict__ x, FrameArgs a, float scale,
                                                float* __restrict__ out, int chunks, int part_words) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = a.L / a.hop, r = a.L - q * a.hop;
    const int nblk = kEnFrames + q + (r ? 1 : 0);               // hop-blocks this workgroup needs
    float* full = smem;                                          // [nblk]
    float* pre = smem + nblk;                                    // [nblk]
    float* part = part_words ? smem + 2 * nblk : nullptr;        // [nblk * hop / 4] (streaming path)
    const long long total = a.n_sig * chunks;
    for (long long wg = blockIdx.x; wg < total; wg += gridDim.x) {
        const long long sig = wg / chunks;                       // b*C + c  (cf)  /  b, c from it (cl)
        const int f0 = (int)(wg - sig * chunks) * kEnFrames;
        const long long b = sig / a.C;
        const int c = (int)(sig - b * a.C);
        const float* src = a.cl ? x + b * a.T * a.C + c : x + sig * a.T;
        const int es = a.cl ? a.C : 1;
        // Streaming path (contiguous signal, hop and r multiples of 4, 16-byte aligned, no frame
        // reaches beyond the signal): every thread turns float4s of the workgroup's contiguous range
        // into partial sums -- coalesced, up to twelve loads per thread in flight, ONE round trip to
        // HBM --, parks them in LDS, and one thread per hop block adds the block's hop/4 partials in
        // a fixed order.  Samples past the end of the signal are not read: without pad_end no frame
        // uses them (they would only enter block sums that no output needs).
        // The general path below walks the hop blocks one per wave, one load in flight, three
        // dependent round trips per block: the ONE workgroup per signal that touched the end of
        // the signal took ~65 us there and set the kernel's duration.
        const long long t_lo = (long long)f0 * a.hop;
        const int n4 = nblk * (a.hop >> 2);                       // float4s in the range
        if (part && es == 1 && (a.hop & 3) == 0 && (r & 3) == 0 && (a.T & 3) == 0 &&
            (long long)(a.F - 1) * a.hop + a.L <= a.T && (((unsigned long long)(src + t_lo)) & 15ull) == 0) {
            const f32x4* p4 = reinterpret_cast<const f32x4*>(src + t_lo);
            const int n4v = (int)min((long long)n4, (a.T - t_lo) >> 2);       // float4s that exist
            for (int i0 = threadIdx.x; i0 < n4; i0 += 12 * 256) {
                f32x4 v[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) v[k] = p4[min(i0 + 256 * k, n4v - 1)];
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const int i = i0 + 256 * k;
                    if (i < n4)
                        part[i] = (i < n4v) ? (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3])
                                            : 0.0f;
                }
            }
            __syncthreads();
            const int h4 = a.hop >> 2, r4 = r >> 2;
            for (int bl = threadIdx.x; bl < nblk; bl += 256) {
                const float* pb = part + bl * h4;
                float s_pre = 0.0f;
                for (int k = 0; k < r4; ++k) s_pre += pb[k];
                float s_all = s_pre;
                for (int k = r4; k < h4; ++k) s_all += pb[k];
                full[bl] = s_all;
                pre[bl] = s_pre;
            }
        } else
        for (int i = wave; i < nblk; i += 4) {
            const long long t0 = (long long)(f0 + i) * a.hop;
            float s_all = 0.0f, s_pre = 0.0f;
            for (int l = lane; l < a.hop; l += 64) {
                const long long t = t0 + l;
                const float v = src[min(t, a.T - 1) * es];       // unconditional load, then select
                const float w = (t < a.T) ? v : a.pad_value;
                const float w2 = w * w;
                s_all += w2;
                s_pre += (l < r) ? w2 : 0.0f;
            }
            for (int sft = 32; sft > 0; sft >>= 1) {
                s_all += __shfl_xor(s_all, sft, 64);
                s_pre += __shfl_xor(s_pre, sft, 64);
            }
            if (lane == 0) { full[i] = s_all; pre[i] = s_pre; }
        }
        __syncthreads();
        if (threadIdx.x < kEnFrames && f0 + (int)threadIdx.x < a.F) {
            const int f = threadIdx.x;
            float acc = 0.0f;
            for (int k = 0; k < q; ++k) acc += full[f + k];
            if (r) acc += pre[f + q];
            const long long fo = f0 + f;
            out[a.cl ? (b * a.F + fo) * a.C + c : sig * a.F + fo] = scale * acc;
        }
        __syncthreads();
    }
}

// x viewed as (outer, T, inner): channels_last (b, t, f, c): outer = b, inner = f*c;
// channels_first (b, c, t, f): outer = b*c, inner = f
__device__ __forceinline__ long long delta_src_index(long long t, long long T, int mode) {
    if (t >= 0 && t < T) return t;
    if (mode == KPR_PAD_CONSTANT) return -1;
    if (T == 1) return 0;
    if (mode == KPR_PAD_SYMMETRIC) {             // ... 1 0 | 0 1 2 ... T-1 | T-1 T-2 ...
        const long long p = 2 * T;
        long long m = t % p; if (m < 0) m += p;
        return m < T ? m : p - 1 - m;
    }
    const long long p = 2 * T - 2;               // reflect: ... 2 1 | 0 1 ... T-1 | T-2 ...
    long long m = t % p; if (m < 0) m += p;
    return m < T ? m : p - m;
}

// every thread produces VEC consecutive outputs along `inner` of one (o, t) row; the 2n neighbour
// rows are read with the same vector width (they are L1/L2 hits for all but the first reader)
template <int VEC>
__global__ __launch_bounds__(256) void k_delta(const float* __restrict__ x, long long outer, long long T,
                                               long long inner, int n, int mode, float inv_denom,
                                               float* __restrict__ out) {
    typedef float vf __attribute__((ext_vector_type(VEC)));
    const long long per_row = inner / VEC;
    const long long total = outer * T * per_row;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long q = e / per_row;
        const long long i = (e - q * per_row) * VEC;
        const long long o = q / T;
        const long long t = q - o * T;
        const float* base = x + o * T * inner + i;
        vf acc = {};
        const bool interior = t - n >= 0 && t + n < T;
        if (__all(interior)) {
            // rows t-n .. t+n all exist for the whole wave: eight unconditional loads in flight per step
            // (a load under a per-lane condition is waited for on the spot: 2n dependent round trips)
            for (int j0 = 1; j0 <= n; j0 += 4) {
                vf vp[4], vm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = min(j0 + u, n);
                    vp[u] = *reinterpret_cast<const vf*>(base + (t + j) * inner);
                    vm[u] = *reinterpret_cast<const vf*>(base + (t - j) * inner);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + u <= n) acc += (float)(j0 + u) * (vp[u] - vm[u]);    // same order as below: j ascending
            }
        } else
        for (int j = 1; j <= n; ++j) {            // pairs (+j, -j): j * (x[t+j] - x[t-j])
            long long ip = t + j, im = t - j;
            if (!interior) { ip = delta_src_index(ip, T, mode); im = delta_src_index(im, T, mode); }
            vf vp = {}, vm = {};
            if (ip >= 0) vp = *reinterpret_cast<const vf*>(base + ip * inner);
            if (im >= 0) vm = *reinterpret_cast<const vf*>(base + im * inner);
            acc += (float)j * (vp - vm);
        }
        *reinterpret_cast<vf*>(out + q * inner + i) = acc * inv_denom;
    }
}

}  // namespace kpr
