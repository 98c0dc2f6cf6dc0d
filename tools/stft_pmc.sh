#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r04k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=cfg4_stft_b128x1x110250_nfft1024_hop256_pad
for V in 2 3; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | tr ' ' '+')
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_v${V}_$N -- python $REPO/tools/pmc_run.py $W stft_variant=$V > /dev/null 2> $OUT/pmc_v${V}_$N.log
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v$V -- python $REPO/tools/pmc_run.py $W stft_variant=$V > /dev/null 2> $OUT/stats_v$V.log
done
python - <<PY
import csv, glob, os
out = "$OUT"
for d in sorted(glob.glob(os.path.join(out, "pmc_v*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_stft" in r["Kernel_Name"] or "k_calib" in r["Kernel_Name"]:
                acc.setdefault((r["Kernel_Name"][:28], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            print(os.path.basename(d), k, c, sum(v) / len(v), len(v))
for d in sorted(glob.glob(os.path.join(out, "stats_v*"))):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_stft" in r["Name"]:
                print(os.path.basename(d), r["Name"][:40], r["Calls"], r["AverageNs"])
PY
