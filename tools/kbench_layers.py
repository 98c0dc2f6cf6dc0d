#!/usr/bin/env python3
"""Kernel times of the stand-alone layers (Magnitude, ApplyFilterbank, MagnitudeToDecibel) on the
target workload's shapes (development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre
from tools.kbench_row4 import timeit  # noqa

rng = np.random.default_rng(0)
b, f, k, m = 256, 83, 1025, 128
spec = torch.from_numpy((rng.standard_normal((b, f, k, 1)) + 1j * rng.standard_normal((b, f, k, 1))).astype(np.complex64)).cuda()
mag = spec.abs().contiguous()
mel = torch.from_numpy(rng.uniform(0, 5, (b, f, m, 1)).astype(np.float32)).cuda()
rows = []
lay = kapre.Magnitude(); us = timeit(lambda: lay(spec)); rows.append(("Magnitude (c64 -> f32)", us, 12 * spec.numel()))
lay = kapre.Phase(); us = timeit(lambda: lay(spec)); rows.append(("Phase (c64 -> f32)", us, 12 * spec.numel()))
fb = kapre.ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=k, n_mels=m))
us = timeit(lambda: fb(mag)); rows.append(("ApplyFilterbank mel 1025x128", us, 4 * mag.numel() + 4 * mel.numel()))
db = kapre.MagnitudeToDecibel(); us = timeit(lambda: db(mel)); rows.append(("MagnitudeToDecibel (mel)", us, 8 * mel.numel()))
us = timeit(lambda: db(mag)); rows.append(("MagnitudeToDecibel (spec)", us, 8 * mag.numel()))
for name, us, byts in rows:
    print("%-32s %8.1f us   %6.0f GB/s algorithmic (%.2f of 8 TB/s)" % (name, us, byts / us / 1e3, byts / us / 1e3 / 8000))
