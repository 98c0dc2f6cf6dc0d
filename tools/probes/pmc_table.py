#!/usr/bin/env python3
"""Per-kernel SQ counter table from the rocprofv3 --pmc passes tools/probes/run_probes.sh wrote.

    python tools/probes/pmc_table.py gpurun_out/probes_<tag>

One row per fft_core kernel instantiation (the LAST dispatch of each: the timed repetition), one column per counter,
plus derived issue figures.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_BUSY_CYCLES counts per-SE busy cycles (MI355X_MICROARCH.md, per-instruction constants table)."""
import csv
import glob
import os
import re
import sys

FLAGS = [(1, "exch"), (2, "pair"), (4, "magw"), (8, "win"), (16, "sqrt"), (32, "narrow"), (64, "passes"), (128, "paira")]


def describe(name):
    m = re.search(r"k_coreILi(\d+)ELj(\d+)E", name)
    if not m:
        return name[:40], 0
    wps, fl = int(m.group(1)), int(m.group(2))
    return "wps=%d %s" % (wps, "+".join(n for b, n in FLAGS if fl & b)), wps


def main():
    root = sys.argv[1]
    data, order = {}, []
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not hits:
            continue
        for r in csv.DictReader(open(max(hits, key=os.path.getmtime))):
            k = r["Kernel_Name"]
            if "k_core" not in k:
                continue
            if k not in data:
                data[k] = {}
                order.append(k)
            data[k][r["Counter_Name"]] = float(r["Counter_Value"])      # later dispatches overwrite earlier ones
    counters = sorted({c for v in data.values() for c in v})
    print("| kernel | " + " | ".join(counters) + " | VALU insts / wave | active VALU quad-cycles / wave-cycles | wait-any / wave-cycles | wait-inst-any / wave-cycles |")
    print("|---|" + "---|" * (len(counters) + 4))
    for k in order:
        v = data[k]
        name, wps = describe(k)
        waves = 256.0 * 4 * max(wps, 1)
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        cells = ["%.4g" % v.get(c, float("nan")) for c in counters]
        der = ["%.0f" % (v.get("SQ_INSTS_VALU", 0.0) / waves),
               "%.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc) if wc else "-",
               "%.3f" % (v.get("SQ_WAIT_ANY", 0.0) / wc) if wc else "-",
               "%.3f" % (v.get("SQ_WAIT_INST_ANY", 0.0) / wc) if wc else "-"]
        print("| " + name + " | " + " | ".join(cells + der) + " |")


if __name__ == "__main__":
    main()
