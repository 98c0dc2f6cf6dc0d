#!/bin/bash
# A/B builds of the library on the cfg4 ISTFT ON THE SAME BOX: tools/ab_istft.sh lib1.so lib2.so ...   (development aid)
I=cfg4_istft_b128x1x434f_nfft1024_hop256
for round in 1 2; do
  for lib in "$@"; do
    export KAPRE_AMD_LIB=$lib
    echo "== round $round $lib"
    for a in "" "batch=512" "batch=64"; do timeout 100 python tools/kbench_custom.py $I $a option:istft_path=4 2>&1 | grep -v amdgpu.ids | tail -1; done
  done
done
