#!/bin/bash
# L2 <-> fabric and L1 stall counters of the HBM-side kernels (cfg4 STFT / InverseSTFT) next to the fused mel
# kernel, one small --pmc group per pass (never combined with a trace).  Run on the GPU box from the repo root:
#   bash tools/stall_counters.sh        -> gpurun_out/stalls/summary.txt   (copy to profiles/r05_stall_counters.txt)
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/stalls
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL="cfg4_stft_b128x1x110250_nfft1024_hop256_pad cfg4_istft_b128x1x434f_nfft1024_hop256 target_mel_b256x1x44100_nfft2048_hop512_mel128"
i=0
for G in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for W in $WL; do
    rocprofv3 --pmc $G --output-format csv -d $OUT/g${i}_$W -- python $REPO/tools/pmc_run.py $W > /dev/null 2> $OUT/g${i}_$W.log
  done
done
python - <<PY > $OUT/summary.txt
import csv, glob, os
out = "$OUT"
ksub = {"cfg4_stft": "k_stft3", "cfg4_istft": "k_istft_pw", "target_mel": "k_mel_pw"}
res = {}
for d in sorted(glob.glob(os.path.join(out, "g*_*"))):
    if not os.path.isdir(d): continue
    w = os.path.basename(d).split("_", 1)[1]
    sub = [v for k, v in ksub.items() if w.startswith(k)][0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for c, v in acc.items():
            res.setdefault(w, {})[c] = sum(v) / len(v)
print("mean per launch of the hot kernel (rocprofv3 --pmc, one group of three per pass; *_sum = over all 16 L2 channels x 8 XCDs / all CUs)")
for w, cs in res.items():
    print()
    print(w)
    for c in sorted(cs): print("  %-44s %16.0f" % (c, cs[c]))
    g = cs.get("GRBM_GUI_ACTIVE"); b = cs.get("TCC_BUSY_sum")
    if b and cs.get("TCC_EA0_WRREQ_STALL_sum") is not None:
        print("  -> EA write-request stall cycles / L2-busy cycles: %.3f" % (cs["TCC_EA0_WRREQ_STALL_sum"] / b))
    if cs.get("TCP_GATE_EN1_sum") and cs.get("TCP_PENDING_STALL_CYCLES_sum") is not None:
        print("  -> L1 pending-stall cycles / L1 active cycles:      %.3f" % (cs["TCP_PENDING_STALL_CYCLES_sum"] / cs["TCP_GATE_EN1_sum"]))
        print("  -> L1 stalled-by-L2-return cycles / L1 active:      %.3f" % (cs["TCP_TCR_TCP_STALL_CYCLES_sum"] / cs["TCP_GATE_EN1_sum"]))
PY
cat $OUT/summary.txt
find $OUT -name "*.db" -delete 2>/dev/null
