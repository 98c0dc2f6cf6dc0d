#!/bin/bash
# A/B two (or more) builds of the library ON THE SAME BOX (box-to-box variance is ~5 %): tools/ab_libs.sh lib1.so lib2.so ...
# For each: the north-star workload at batch 64 / 256 / 1024 and cfg5 / reftest under mel_variant $ABV (default 7), two rounds.
T=target_mel_b256x1x44100_nfft2048_hop512_mel128
V=${ABV:-7}
for round in 1 2; do
  for lib in "$@"; do
    export KAPRE_AMD_LIB=$lib
    echo "== round $round $lib"
    for b in 64 256 1024; do timeout 100 python tools/kbench_custom.py $T batch=$b option:mel_variant=$V 2>&1 | grep -v amdgpu.ids; done
    timeout 100 python tools/kbench_custom.py cfg5_mel_b256x1x160000_nfft1024_hop160_mel80 option:mel_variant=$V 2>&1 | grep -v amdgpu.ids
    timeout 100 python tools/kbench_custom.py reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40 option:mel_variant=$V 2>&1 | grep -v amdgpu.ids
  done
done
