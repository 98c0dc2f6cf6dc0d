REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/small_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
 i=$((i+1))
 for V in "1 1 channels_first" "1 2 channels_last"; do
   set -- $V
   rocprofv3 --pmc $G --output-format csv -d $OUT/g${i}_c$2 -- python $REPO/tools/probes/small_fb.py $V > /dev/null 2> $OUT/g${i}_c$2.log
 done
done
python - <<PY > $OUT/summary.txt
import csv, glob, os
res = {}
for d in sorted(glob.glob("$OUT/g*_c*")):
    if not os.path.isdir(d): continue
    v = d.rsplit("_", 1)[1]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_fb_pw" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for c, x in acc.items():
            res.setdefault(c, {})[v] = sum(x) / len(x)
for c in sorted(res):
    print("%-28s %14.1f %14.1f" % (c, res[c].get("c1", -1), res[c].get("c2", -1)))
PY
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -- python tools/probes/small_fb.py 1 1 channels_first > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt2 -- python tools/probes/small_fb.py 1 2 channels_last > /dev/null 2>&1
for d in kt1 kt2; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); grep k_fb_pw $f | cut -c1-200 >> $OUT/summary.txt; done
