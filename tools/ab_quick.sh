#!/bin/bash
# same-box A/B of library builds over a list of bench workloads (cold clocks, hipGraph of 100 launches):
#   tools/ab_quick.sh "lib1.so lib2.so ..." "workload ..." [rounds]
run() { KAPRE_AMD_LIB=$1 python tools/kbench.py $2 2>&1 | grep -v "Warn\|amdgpu.ids" | sed "s#^#$(basename $1)  #"; }
for r in $(seq 1 ${3:-2}); do for w in $2; do for l in $1; do run $l $w; done; done; done
