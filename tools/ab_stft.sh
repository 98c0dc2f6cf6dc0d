#!/bin/bash
# A/B builds of the library on the cfg4 STFT ON THE SAME BOX: tools/ab_stft.sh lib1.so lib2.so ...   (development aid)
I=cfg4_stft_b128x1x110250_nfft1024_hop256_pad
for round in 1 2; do
  for lib in "$@"; do
    export KAPRE_AMD_LIB=$lib
    echo "== round $round $lib"
    for a in "" "batch=512" "n_fft=2048 hop=512"; do timeout 100 python tools/kbench_custom.py $I $a 2>&1 | grep -v amdgpu.ids | tail -1; done
  done
done
