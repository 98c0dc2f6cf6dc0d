#!/usr/bin/env python3
"""Print in-kernel cycle stamps of workgroup 0 of k_mel_ws (development aid; k_mel_fused was removed in round 5)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import torch
from kapre_amd import _ffi
for a in [a for a in sys.argv[1:] if "=" in a]:          # name=value -> kpr_set_option
    _ffi.set_option(a.split("=")[0], int(a.split("=")[1]))
sys.argv = [a for a in sys.argv if "=" not in a]
name = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT
w = bench.WORKLOADS[name]
model = bench.build_model(w)
x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
model(x); torch.cuda.synchronize()
NW = int(os.environ.get("KPR_STAMP_WAVES", "16"))
buf = torch.zeros(1024 + 4 * 4096, dtype=torch.int64, device="cuda")
buf[NW * 32] = int(os.environ.get("KPR_STAMP_BLOCK", "0"))        # workgroup to observe (k_mel_ws)
L = _ffi.lib()
L.kpr_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
model(x); torch.cuda.synchronize()
L.kpr_debug_stamps(ctypes.c_void_p(0))
full = buf.cpu().numpy()
wg = full[1024:].reshape(4096, 4)
wg = wg[wg[:, 0] != 0]
if len(wg):          # k_mel_ts: per-workgroup start / end (100 MHz ticks -> us) and shader cycles
    import numpy as np
    t0r = wg[:, 0].min()
    st, en = (wg[:, 0] - t0r) / 100.0, (wg[:, 1] - t0r) / 100.0
    cyc = wg[:, 3] - wg[:, 2]
    print("workgroups %d: start us min/median/max %.1f %.1f %.1f | end us min/median/max %.1f %.1f %.1f | duration us median %.1f | shader cycles median %d -> %.0f MHz"
          % (len(wg), st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), np.median(en - st), np.median(cyc), np.median(cyc / np.maximum(en - st, 1e-9))))
    print("end-time histogram (us):", np.histogram(en, bins=8)[0].tolist(), [round(v, 1) for v in np.histogram(en, bins=8)[1].tolist()])
    print("start-time histogram (us):", np.histogram(st, bins=8)[0].tolist(), [round(v, 1) for v in np.histogram(st, bins=8)[1].tolist()])
b = full[:NW * 32].reshape(NW, 32)
t0 = b[:, 0][b[:, 0] != 0].min()
for wv in range(NW):
    row = b[wv]; n = int((row != 0).sum())
    if n == 0: continue
    print("wave", wv, " ".join("%7d" % (v - t0) for v in row[:n]))
    print("  delta", " ".join("%7d" % d for d in (row[1:n] - row[:n-1])))
