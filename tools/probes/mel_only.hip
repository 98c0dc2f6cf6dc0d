// compile-only probe: just the fused mel kernels (fast iteration on register allocation / ISA)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/kapre_hip.h"
#include "../../kapre_amd/csrc/kpr_fft.h"
#include "../../kapre_amd/csrc/kpr_fft_mr.h"
#include "../../kapre_amd/csrc/kpr_common.h"
#include "../../kapre_amd/csrc/kpr_mel_kernels.h"
template __global__ void kpr::k_mel_ws<1024, false, RESV>(const float*, kpr::Geom, const float*, const float2*, const float*,
                                                    kpr::MelSched, kpr::DbDev, unsigned*, float*, int, int, long long*);
#ifdef ALSO_512
template __global__ void kpr::k_mel_ws<512, false, true>(const float*, kpr::Geom, const float*, const float2*, const float*,
                                                   kpr::MelSched, kpr::DbDev, unsigned*, float*, int, int, long long*);
#endif
