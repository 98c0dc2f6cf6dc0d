#!/bin/bash
# Collect the rocprofv3 evidence for one round ON THE GPU BOX (run from the repo root through gpurun):
#   tools/profile_round.sh r02          (every pass)      tools/profile_round.sh r03 stats   (step 1 only)
# Writes gpurun_out/prof_<round>/ ; tools/profile_collect.py then distils it into profiles/.
# Kernel-trace statistics and every --pmc counter group are SEPARATE passes (never combined).
R=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$R
rm -rf $OUT            # (stale runs of the same round would be picked up by profile_collect.py)
mkdir -p $OUT
sha256sum $REPO/kapre_amd/lib/libkapre_hip.so | cut -c1-16 > $OUT/lib_sha16.txt     # which binary the passes were taken on
cd /tmp && export TMPDIR=/tmp
# gpurun copies back at most 64 MiB: per pass only kernel_stats / counter_collection are kept (the raw kernel trace of a
# 100-step run is ~10 MB per workload)
prune() { find $OUT \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*.db" -o -name "*.json.gz" \) -delete 2>/dev/null; }
TGT=target_mel_b256x1x44100_nfft2048_hop512_mel128
WL=$(python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(' '.join(bench.WORKLOADS))")
# 1. kernel statistics, one bench.py workload per run (--no-also / --workload: the kernel's AverageNs is then
#    comparable with the kernel_us bench.py prints for that workload)
for W in $WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python $REPO/bench.py --steps 100 --warmup 10 --workload $W --no-also --no-cpu-baseline --sustain 0 > $OUT/bench_under_rocprof_$W.json 2> $OUT/stats_$W.log
  prune
done
if [ "$2" = "stats" ]; then ls $OUT | head -40; exit 0; fi     # tools/profile_round.sh r03 stats: kernel statistics only
# 2. HBM traffic counters, one pass each
for C in FETCH_SIZE WRITE_SIZE; do
  for W in $WL; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_$W -- python $REPO/tools/pmc_run.py $W > /dev/null 2> $OUT/pmc_${C}_$W.log
    prune
  done
done
# 3. where the cycles go (target workload), small groups per pass
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $G | tr ' ' '+')
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_sq_$N -- python $REPO/tools/pmc_run.py $TGT > /dev/null 2> $OUT/pmc_sq_$N.log
done
# 4. the issue-slot picture of EVERY workload (bench.py: roofline_compute.issue_util): one small group, one pass each
for W in $WL; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_issue_$W -- python $REPO/tools/pmc_run.py $W > /dev/null 2> $OUT/pmc_issue_$W.log
done
prune
du -sh $OUT
ls $OUT | head -80
