#!/bin/bash
# The stereo instances of k_fb_pw (profiles/r06_fb_pw.md section 7): L1 / address-unit counters of the committed request layout (four
# lanes per 64 contiguous bytes + transposition through LDS) against the first one (every lane requests its own 128 bytes:
# tools/probes/bin/lib_st_direct.so, -DKPR_FB_ST_DIRECT=1) on the bench row k2_filterbank_cl2, one small --pmc group per pass.
#   bash tools/fb_st_l1_counters.sh  -> gpurun_out/fb_st_l1_counters/summary.txt
# (every pass under `timeout`: a counter name this rocprofv3 does not know aborts the pass and then hangs in its finaliser)
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/fb_st_l1_counters
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=k2_filterbank_cl2_b128x83x1025x2_mel128
i=0
for G in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_sum GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  for V in halves direct; do
    LIB=$REPO/kapre_amd/lib/libkapre_hip.so
    [ $V = direct ] && LIB=$REPO/tools/probes/bin/lib_st_direct.so
    KAPRE_AMD_LIB=$LIB timeout 180 rocprofv3 --pmc $G --output-format csv -d $OUT/g${i}_$V -- python $REPO/tools/pmc_run.py $W > /dev/null 2> $OUT/g${i}_$V.log
  done
done
python - <<PY > $OUT/summary.txt
import csv, glob, os
out = "$OUT"
res = {}
for d in sorted(glob.glob(os.path.join(out, "g*_*"))):
    if not os.path.isdir(d): continue
    v = os.path.basename(d).split("_", 1)[1]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "k_fb_pw" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for c, x in acc.items():
            res.setdefault(c, {})[v] = sum(x) / len(x)
print("%-40s %18s %18s   (mean per launch of k_fb_pw<1024,st>, 10 624 blocks x 1025 x 2 -> 128 x 2)" % ("counter", "own 128 bytes", "by halves"))
for c in sorted(res):
    a, b = res[c].get("direct"), res[c].get("halves")
    print("%-40s %18s %18s   %s" % (c, "%.0f" % a if a is not None else "-", "%.0f" % b if b is not None else "-", "x %.2f" % (b / a) if a and b else ""))
PY
cd $REPO
for V in halves direct; do
  LIB=$REPO/kapre_amd/lib/libkapre_hip.so
  [ $V = direct ] && LIB=$REPO/tools/probes/bin/lib_st_direct.so
  KAPRE_AMD_LIB=$LIB python tools/kbench_fb.py 0 shape=1025,83,128,2,channels_last,128 | tail -1 | sed "s/^/$V: /" >> $OUT/summary.txt
done
cat $OUT/summary.txt
