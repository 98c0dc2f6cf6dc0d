#!/usr/bin/env python3
"""Quick kernel-time table for the fused mel kernel (development aid; uses bench.py helpers)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    import torch

    names = sys.argv[1:] or [bench.DEFAULT, bench.ALSO]
    for name in names:
        w = bench.WORKLOADS[name]
        model = bench.build_model(w)
        x = bench.make_input(w, 0, torch.device("cuda", 0))
        us, how = bench.kernel_time_us(model, x, launches=100)
        frames = w["batch"] * w["ch"] * bench.frames_of(w)
        hbm, mfma = bench.roofline(w, us)
        print("%-62s %8.2f us  %8.1f Mframes/s  hbm %.3f  mfma-dense-eq %.3f" %
              (name, us, frames / us, hbm["frac"], mfma["frac"]))
        del x


if __name__ == "__main__":
    main()
