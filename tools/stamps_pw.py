#!/usr/bin/env python3
"""In-kernel cycle stamps of k_mel_pw (development aid; needs the stamps build: tools/build_variant.py stamps
-DKPR_DEV_STAMPS, KAPRE_AMD_LIB=kapre_amd/lib/libkapre_hip_stamps.so).
    python tools/stamps_pw.py [mel_variant=7] [workload]      KPR_STAMP_BLOCK=<workgroup to observe>
Per wave of the observed workgroup: kernel start, after the prologue barrier, then per frame (after |X| row, after the
band sums + stores).  Per workgroup of the grid: start / end on the 100 MHz clock."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import bench
import torch
from kapre_amd import _ffi
for a in [a for a in sys.argv[1:] if "=" in a]:
    _ffi.set_option(a.split("=")[0], int(a.split("=")[1]))
args = [a for a in sys.argv[1:] if "=" not in a]
name = args[0] if args else bench.DEFAULT
w = bench.WORKLOADS[name]
model = bench.build_model(w)
x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
for _ in range(3):
    model(x)
torch.cuda.synchronize()
print("kernels:", _ffi.last_launches())
buf = torch.zeros(1024 + 4 * 4096, dtype=torch.int64, device="cuda")
buf[16 * 32] = int(os.environ.get("KPR_STAMP_BLOCK", "0"))
L = _ffi.lib()
L.kpr_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
model(x); torch.cuda.synchronize()
L.kpr_debug_stamps(ctypes.c_void_p(0))
full = buf.cpu().numpy()
wg = full[1024:].reshape(4096, 4)
wg = wg[wg[:, 0] != 0]
if len(wg):
    t0r = wg[:, 0].min()
    st, en = (wg[:, 0] - t0r) / 100.0, (wg[:, 1] - t0r) / 100.0
    cyc = wg[:, 3] - wg[:, 2]
    print("workgroups %d: start us min/median/max %.1f %.1f %.1f | end us min/median/max %.1f %.1f %.1f | duration us median %.1f | clock %.0f MHz"
          % (len(wg), st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), np.median(en - st),
             np.median(cyc / np.maximum(en - st, 1e-9))))
    print("end-time histogram (us):", np.histogram(en, bins=8)[0].tolist(), [round(v, 1) for v in np.histogram(en, bins=8)[1].tolist()])
    # who is slow?  workgroup b runs on XCD b mod 8 (round-robin dispatch); duration = end - start of the workgroup
    dur = en - st
    idx = np.nonzero(full[1024:].reshape(4096, 4)[:, 0] != 0)[0]
    print("duration by XCD (us, mean / max):", " ".join("%d: %.1f/%.1f" % (x, dur[idx % 8 == x].mean(), dur[idx % 8 == x].max()) for x in range(8)))
    print("duration by workgroup octile (us, mean):", " ".join("%.1f" % dur[(idx * 8 // max(1, idx.max() + 1)) == o].mean() for o in range(8)))
    order = np.argsort(-dur)[:12]
    print("slowest workgroups (index: start, duration):", " ".join("%d: %.1f, %.1f" % (idx[i], st[i], dur[i]) for i in order))
    cyc_per_us = cyc / np.maximum(dur, 1e-9)
    print("shader clock by XCD (MHz, mean):", " ".join("%d: %.0f" % (x, cyc_per_us[idx % 8 == x].mean()) for x in range(8)))
b = full[:16 * 32].reshape(16, 32)
nzv = b[:, 0][b[:, 0] != 0]
if len(nzv):
    t0 = nzv.min()
    for wv in range(16):
        row = b[wv]; n = int((row != 0).sum())
        if n == 0: continue
        # stamps 0..4: entry, first fetch issued, twiddles issued, tables + window written, after the barrier
        print("wave %2d start %6d prologue (fetch, twiddles, tables, barrier) %s | frames (fft+row, sums+store):"
              % (wv, row[0] - t0, "+".join(str(row[i] - row[i - 1]) for i in range(1, 5))),
              " ".join("%d+%d" % (row[i] - row[i - 1], row[i + 1] - row[i]) for i in range(5, n - 1, 2)), "| end", row[n - 1] - t0)
