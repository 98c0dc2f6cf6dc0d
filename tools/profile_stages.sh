#!/bin/bash
# rocprofv3 evidence for the FFT stage and the filterbank stage (run on the GPU box via gpurun):
#   tools/profile_stages.sh r01   ->  gpurun_out/prof_<round>_stages/ ; tools/profile_collect_stages.py distils it
R=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_${R}_stages
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/tools/pmc_stages.py > $OUT/workload.txt 2> $OUT/stats.log
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  N=$(echo $G | tr ' ' '+')
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_$N -- python $REPO/tools/pmc_stages.py > /dev/null 2> $OUT/pmc_$N.log
done
ls $OUT
