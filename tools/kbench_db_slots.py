#!/usr/bin/env python3
"""Decibel mel calls at small batches with one statistics slot per item (`db_slots` 1, rounds 1-2) and with the automatic
number of slots: kernel time per call (hipGraph of launches, HIP events), min of three repetitions, and a bit-identity
check of the two outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402
from kapre_amd import _ffi  # noqa: E402

shapes = [("cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf", {}), ("cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf", {"ch": 1}),
          ("reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40", {}), ("speech_mel_b256x1x160000_nfft400_hop160_mel80", {"db": True})]
for name, over in shapes:
    for batch in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        w = dict(bench.WORKLOADS[name]); w.update(over); w["batch"] = batch
        x = bench.make_input(w, 0, torch.device("cuda", 0), batch)
        row, outs = [], []
        for slots in (1, 0, 1, 0):
            _ffi.set_option("db_slots", slots)
            model = bench.build_model(w)
            outs.append(model(x).clone())
            row.append(min(bench.kernel_time_us(model, x, launches=100)[0] for _ in range(3)))
        one, auto = min(row[0], row[2]), min(row[1], row[3])
        print("%-44s ch %d batch %4d   one slot %7.2f us   automatic %7.2f us   %+5.1f %%   identical %s" % (
            name[:44], w["ch"], batch, one, auto, 100 * (auto / one - 1), bool(torch.equal(outs[0], outs[1]))), flush=True)
_ffi.set_option("db_slots", 0)
