#!/usr/bin/env python3
"""MagnitudeToDecibel on a spectrogram-sized input, a few launches (run under rocprofv3 --kernel-trace --stats
to see k_db_log / k_db_clamp separately; development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre
rng = np.random.default_rng(0)
mag = torch.from_numpy(np.abs(rng.standard_normal((256, 83, 1025, 1)) + 1j * rng.standard_normal((256, 83, 1025, 1))).astype(np.float32)).cuda()
quiet = mag.clone(); quiet[:, :40] *= 1e-6                      # > 80 dB of range: the clamp pass runs
db = kapre.MagnitudeToDecibel()
for _ in range(10):
    a = db(mag); b = db(quiet)
torch.cuda.synchronize()
print(float(a.max()), float(b.min()))
