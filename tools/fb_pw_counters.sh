#!/bin/bash
# Why MORE rows in flight make k_fb_pw slower (profiles/r06_fb_pw.md section 4): L1 / L2 / fabric counters of the kernel with two
# (product) and three (tools/probes/bin/lib_depth3.so, -DKPR_FB_DEPTH=3) rows in flight per wave on the bench row k2_filterbank,
# one small --pmc group per pass.   bash tools/fb_pw_counters.sh  -> gpurun_out/fb_pw_counters/summary.txt
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/fb_pw_counters
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=k2_filterbank_b256x83x1025_mel128
i=0
for G in "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  for V in depth2 depth3; do
    LIB=$REPO/kapre_amd/lib/libkapre_hip.so
    [ $V = depth3 ] && LIB=$REPO/tools/probes/bin/lib_depth3.so
    KAPRE_AMD_LIB=$LIB rocprofv3 --pmc $G --output-format csv -d $OUT/g${i}_$V -- python $REPO/tools/pmc_run.py $W > /dev/null 2> $OUT/g${i}_$V.log
  done
done
python - <<PY > $OUT/summary.txt
import csv, glob, os
out = "$OUT"
res = {}
for d in sorted(glob.glob(os.path.join(out, "g*_depth*"))):
    if not os.path.isdir(d): continue
    v = os.path.basename(d).split("_", 1)[1]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_fb_pw" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for c, x in acc.items():
            res.setdefault(c, {})[v] = sum(x) / len(x)
print("%-44s %16s %16s   (mean per launch of k_fb_pw<1024>, 21 248 x 1025 -> 128)" % ("counter", "two rows", "three rows"))
for c in sorted(res):
    print("%-44s %16.0f %16.0f" % (c, res[c].get("depth2", float("nan")), res[c].get("depth3", float("nan"))))
PY
cat $OUT/summary.txt
