#!/usr/bin/env python3
"""Kernel time of every fused-mel bench workload under each mel_variant (development aid: which kernel should be the default?).
    python tools/kbench_variants.py [variant ...]          default variants: 0 (auto) 3 (k_mel_ws) 4 (k_mel_ts)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

from kapre_amd import _ffi  # noqa: E402

variants = [int(a) for a in sys.argv[1:]] or [0, 3, 4]
print("%-62s %s" % ("workload", " ".join("v%-7d" % v for v in variants)))
for name, w in bench.WORKLOADS.items():
    if w["kind"] != "mel" or w["n_fft"] not in (512, 1024, 2048):
        continue
    row = []
    for v in variants:
        _ffi.set_option("mel_variant", v)
        model = bench.build_model(w)
        x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
        us, how = bench.kernel_time_us(model, x, launches=60)
        row.append(us)
    print("%-62s %s" % (name, " ".join("%8.2f" % u for u in row)), flush=True)
_ffi.set_option("mel_variant", 0)
