#!/usr/bin/env python3
"""Print in-kernel cycle stamps of workgroup 0 of k_mel_ws / k_mel_fused (development aid)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import torch
from kapre_amd import _ffi
name = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT
w = bench.WORKLOADS[name]
model = bench.build_model(w)
x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
model(x); torch.cuda.synchronize()
NW = int(os.environ.get("KPR_STAMP_WAVES", "16"))
buf = torch.zeros(NW * 32 + 1, dtype=torch.int64, device="cuda")
buf[NW * 32] = int(os.environ.get("KPR_STAMP_BLOCK", "0"))        # workgroup to observe (k_mel_ws)
L = _ffi.lib()
L.kpr_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
model(x); torch.cuda.synchronize()
L.kpr_debug_stamps(ctypes.c_void_p(0))
b = buf.cpu().numpy()[:NW * 32].reshape(NW, 32)
t0 = b[:, 0][b[:, 0] != 0].min()
for wv in range(NW):
    row = b[wv]; n = int((row != 0).sum())
    if n == 0: continue
    print("wave", wv, " ".join("%7d" % (v - t0) for v in row[:n]))
    print("  delta", " ".join("%7d" % d for d in (row[1:n] - row[:n-1])))
