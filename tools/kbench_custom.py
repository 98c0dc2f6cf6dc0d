#!/usr/bin/env python3
"""Kernel time of a bench.py workload with fields overridden on the command line (development aid).
    python tools/kbench_custom.py WORKLOAD [field=value ...] [option:name=value ...]
e.g.  tools/kbench_custom.py reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40 fmt=channels_first db=0 option:mel_variant=4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from kapre_amd import _ffi

name = sys.argv[1]
w = dict(bench.WORKLOADS[name])
for a in sys.argv[2:]:
    k, v = a.split("=", 1)
    if k.startswith("option:"):
        _ffi.set_option(k[7:], int(v))
    elif k in ("fmt",):
        w[k] = v
    elif k in ("db",):
        w[k] = bool(int(v))
    else:
        w[k] = type(w.get(k, 0))(float(v)) if not isinstance(w.get(k), int) else int(v)
model = bench.build_model(w)
x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
us, how = bench.kernel_time_us(model, x, launches=100)
frames = w["batch"] * w["ch"] * bench.frames_of(w)
print("%s %s -> %.2f us  %.3e frames/s" % (name, " ".join(sys.argv[2:]), us, frames / (us * 1e-6)))
