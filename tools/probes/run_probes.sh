#!/bin/bash
# Run the probes ON THE GPU BOX (through gpurun, from the repo root):  tools/probes/run_probes.sh [tag]
# Writes gpurun_out/probes_<tag>/: the tables, and one rocprofv3 --pmc pass per SQ counter group over fft_core
# (counters only -- never combined with tracing).
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/probes_$TAG
mkdir -p $OUT
B=$REPO/tools/probes/bin
cd /tmp && export TMPDIR=/tmp
timeout 300 $B/valu_micro 2000 > $OUT/valu_micro.md 2> $OUT/valu_micro.err
timeout 300 $B/fft_core 200 5 > $OUT/fft_core.md 2> $OUT/fft_core.err
cp $B/*.isa.md $OUT/ 2>/dev/null
for G in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  N=$(echo $G | tr ' ' '+')
  timeout 600 rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_$N -- $B/fft_core 100 1 > /dev/null 2> $OUT/pmc_$N.log
done
python $REPO/tools/probes/pmc_table.py $OUT > $OUT/fft_core_pmc.md 2> $OUT/pmc_table.err
ls $OUT
