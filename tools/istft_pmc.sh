#!/bin/bash
# where the cycles of the cfg4 ISTFT kernel go (rocprofv3 --pmc, one small group per pass): tools/istft_pmc.sh [istft_path]
REPO=${GRAFT_REPO_ROOT:-$PWD}
P=${1:-0}
OUT=$REPO/gpurun_out/istft_pmc_$P
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=cfg4_istft_b128x1x434f_nfft1024_hop256
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '+')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python $REPO/tools/pmc_run.py $W istft_path=$P > /dev/null 2> $OUT/pmc_$N.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/tools/pmc_run.py $W istft_path=$P > /dev/null 2> $OUT/stats.log
python - <<PY > $OUT/summary.txt
import csv, glob, os
out = "$OUT"
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_istft" in r["Kernel_Name"]:
                acc.setdefault((r["Kernel_Name"][:30], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            print(k, c, "%.4g" % (sum(v) / len(v)), len(v))
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_istft" in r["Name"]:
            print(r["Name"][:50], r["Calls"], r["AverageNs"])
PY
find $OUT -name "*.csv" -size +200k -delete; find $OUT \( -name "*.db" -o -name "*.json.gz" \) -delete
cat $OUT/summary.txt
