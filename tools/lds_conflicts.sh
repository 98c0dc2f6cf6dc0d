#!/bin/bash
# where the LDS bank conflicts of the fused mel kernel come from.  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of
# k_mel_pw<1024,w16> on the north-star workload for the product build and three development builds (tools/probes/bin/lib_*.so,
# built by the caller with -DKPR_PW_GATHER64 / -DKPR_PW_KO_APPEND / -DKPR_PW_KO_STAGE2), one --pmc pass each, plus the kernel time of
# each build.  Run on the GPU box from the repo root:   bash tools/lds_conflicts.sh   -> gpurun_out/lds_conflicts/summary.txt
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/lds_conflicts
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=target_mel_b256x1x44100_nfft2048_hop512_mel128
for V in product gather64 ko_append ko_stage2; do
  LIB=$REPO/tools/probes/bin/lib_$V.so
  [ $V = product ] && LIB=$REPO/kapre_amd/lib/libkapre_hip.so
  [ -f $LIB ] || continue
  KAPRE_AMD_LIB=$LIB rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/$V -- python $REPO/tools/pmc_run.py $W > /dev/null 2> $OUT/$V.log
  KAPRE_AMD_LIB=$LIB python $REPO/tools/kbench.py settle=1.0 $W 2>/dev/null | grep ' us ' > $OUT/$V.time
done
python - <<PY > $OUT/summary.txt
import csv, glob, os
out = "$OUT"
print("build       kernel us   SQ_INSTS_LDS  SQ_LDS_IDX_ACTIVE  SQ_LDS_BANK_CONFLICT  conflict / active   (per launch of k_mel_pw, north-star workload)")
for v in ("product", "gather64", "ko_append", "ko_stage2"):
    acc = {}
    for f in glob.glob(os.path.join(out, v, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mel_pw" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if not acc:
        continue
    m = {k: sum(x) / len(x) for k, x in acc.items()}
    t = open(os.path.join(out, v + ".time")).read().split()
    us = t[1] if len(t) > 1 else "?"
    print("%-10s %9s %14.0f %18.0f %21.0f %18.3f" % (v, us, m.get("SQ_INSTS_LDS", 0), m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0),
                                                   m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
cat $OUT/summary.txt
