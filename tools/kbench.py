#!/usr/bin/env python3
"""Quick kernel-time table for bench.py workloads (development aid; uses bench.py helpers).
    python tools/kbench.py [option=value ...] [settle=SECONDS] [workload ...]      default: the headline + cfg2
(settle=1.0: the figure after a second of continuous replay -- settled clocks, as bench.py reports it; KAPRE_AMD_LIB=<path>
selects another build of the library for same-box A/B runs)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    import torch

    from kapre_amd import _ffi
    settle = 0.0
    for a in [a for a in sys.argv[1:] if "=" in a]:          # name=value -> kpr_set_option
        if a.startswith("settle="):
            settle = float(a.split("=")[1])
            continue
        _ffi.set_option(a.split("=")[0], int(a.split("=")[1]))
        print("option", a)
    sys.argv = [a for a in sys.argv if "=" not in a]
    names = sys.argv[1:] or [bench.DEFAULT, "cfg2_mel_b64x1x44100_nfft2048_hop512_mel128"]
    for name in names:
        w = bench.WORKLOADS[name]
        model = bench.build_model(w)
        x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
        us, how = bench.kernel_time_us(model, x, launches=100, settle_s=settle)
        frames = w["batch"] * w["ch"] * bench.frames_of(w)
        kernel = _ffi.last_launches()
        hbm, comp = bench.rooflines(name, w, w["batch"], us, kernel)
        print("%-62s %8.2f us  %.3e frames/s  hbm %.3f%s  [%s]" % (
            name, us, frames / (us * 1e-6), hbm["frac"], "  valu+mfma %.3f" % comp["frac"] if comp else "", kernel))
