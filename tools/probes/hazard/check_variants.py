#!/usr/bin/env python3
"""Run assembly variants of k_istft_pw<512, 2, 16> through the stand-alone harness of pad_bisect.py and report the wrong samples.
python tools/probes/hazard/check_variants.py A.s.txt B.s.txt ..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pad_bisect as hz

r = hz.Runner()
os.makedirs("gpurun_out/hazard", exist_ok=True)
for path in sys.argv[1:]:
    lines, idx = hz.split_asm(open(path).read())
    co = hz.make_variant(lines, idx, set(), "gpurun_out/hazard/c")
    n, where = r.run(co)
    print("%-50s instructions %d  wrong samples %6d  (t mod 64 in %s)" % (os.path.basename(path), len(idx), n, where), flush=True)
