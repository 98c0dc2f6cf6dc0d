#!/usr/bin/env python3
"""Small launches of the fused mel chain (batch 1 ... 64: the serving case) under each mel_variant: which kernel should
take them?  Kernel time per call (hipGraph of launches, HIP events), min of three repetitions.
    python tools/kbench_mel_small.py [variant ...]        default variants: 0 (auto) 1 (ring kernel) 3 (k_mel_ws) 4 (k_mel_ts)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402
from kapre_amd import _ffi  # noqa: E402

variants = [int(a) for a in sys.argv[1:]] or [0, 3, 4]
shapes = [("target_mel_b256x1x44100_nfft2048_hop512_mel128", {}), ("cfg5_mel_b256x1x160000_nfft1024_hop160_mel80", {}),
          ("reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40", {}), ("cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf", {})]
print("%-58s %5s  %s" % ("workload", "batch", " ".join("v%-7d" % v for v in variants)))
for name, over in shapes:
    for batch in (1, 2, 4, 8, 16, 32, 64):
        w = dict(bench.WORKLOADS[name]); w.update(over); w["batch"] = batch
        row = []
        for v in variants:
            _ffi.set_option("mel_variant", v)
            model = bench.build_model(w)
            x = bench.make_input(w, 0, torch.device("cuda", 0), batch)
            row.append(min(bench.kernel_time_us(model, x, launches=100)[0] for _ in range(3)))
        best = min(row)
        print("%-58s %5d  %s%s" % (name[:58], batch, " ".join("%8.2f" % u for u in row),
                                  "" if row[0] <= 1.03 * best else "   <-- auto is %.0f %% off" % (100 * (row[0] / best - 1))), flush=True)
_ffi.set_option("mel_variant", 0)
