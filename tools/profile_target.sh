#!/bin/bash
# The headline workload only: kernel statistics + the SQ counter groups (separate passes) -> gpurun_out/prof_<round>_target/
#   tools/profile_target.sh r04a
R=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_${R}_target
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TGT=${2:-target_mel_b256x1x44100_nfft2048_hop512_mel128}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --steps 100 --warmup 10 --workload $TGT --no-also --no-cpu-baseline --sustain 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  N=$(echo $G | tr ' ' '+')
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc_sq_$N -- python $REPO/tools/pmc_run.py $TGT > /dev/null 2> $OUT/pmc_sq_$N.log
done
python - <<PY
import csv, glob, os, json
out = "$OUT"
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_sq_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_mel" in r["Kernel_Name"] or "k_stft" in r["Kernel_Name"] or "k_istft" in r["Kernel_Name"]:
                acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            res.setdefault(k, {})[c] = sum(v) / len(v)
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("void kpr", "kpr")):
            print("stats", r["Name"][:60], r["Calls"], r["AverageNs"])
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "sq_counters.json"), "w"), indent=1)
PY
