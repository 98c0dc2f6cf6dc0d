#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few launches of the fused mel kernel on one
BASELINE workload plus a known-traffic calibration kernel (reads exactly N bytes with the same
8-byte-per-lane access width as the frame loads)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import torch
from kapre_amd import _ffi

for a in [a for a in sys.argv[1:] if "=" in a]:          # name=value -> kpr_set_option (A/B of dispatch variants under the profiler)
    _ffi.set_option(a.split("=")[0], int(a.split("=")[1]))
sys.argv = [a for a in sys.argv if "=" not in a]
name = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT
w = bench.WORKLOADS[name]
model = bench.build_model(w)
x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
for _ in range(5):
    y = model(x)
torch.cuda.synchronize()
L = _ffi.lib()
nbytes = 1 << 30                                   # 1 GiB >> 256 MiB Infinity Cache
buf = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
out = torch.zeros(1, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    L.kpr_debug_calib_read8(ctypes.c_void_p(buf.data_ptr()), nbytes // 8, ctypes.c_void_p(out.data_ptr()),
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("workload", name, "in_bytes", x.numel() * 4, "out_bytes", y.numel() * 4, "calib_bytes", nbytes)
