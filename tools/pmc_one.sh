#!/bin/bash
# one rocprofv3 --pmc pass on one bench workload: tools/pmc_one.sh WORKLOAD "COUNTER ..." [option=value ...]   (development aid)
REPO=${GRAFT_REPO_ROOT:-$PWD}
W=$1; C=$2; shift 2
OUT=$REPO/gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $C --output-format csv -d $OUT -- python $REPO/tools/pmc_run.py $W "$@" > /dev/null 2> $OUT/log.txt
python - <<PY
import csv, glob
acc = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kpr::k_" in r["Kernel_Name"] and "calib" not in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"][10:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(k, c, "%.4g" % (sum(v) / len(v)), len(v))
PY
rm -rf $OUT
