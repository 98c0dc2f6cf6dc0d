#!/usr/bin/env python3
"""In-kernel cycle stamps of one workgroup of k_mel_ws<.., FROM_MAG> (stand-alone ApplyFilterbank; development aid; needs a
library built with -DKPR_DEV_STAMPS: tools/build_variant.py stamps -DKPR_DEV_STAMPS, KAPRE_AMD_LIB=...)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi
rng = np.random.default_rng(0)
b, c, f, k, m = 256, 6, 44, 1025, 128
mag = torch.from_numpy(rng.uniform(0, 1, (b, c, f, k)).astype(np.float32)).cuda()
fb = kapre.ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=k, n_mels=m), data_format="channels_first")
fb(mag); torch.cuda.synchronize()
buf = torch.zeros(1024 + 4 * 4096, dtype=torch.int64, device="cuda")
buf[12 * 32] = int(os.environ.get("KPR_STAMP_BLOCK", "100"))
L = _ffi.lib()
L.kpr_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
fb(mag); torch.cuda.synchronize()
L.kpr_debug_stamps(ctypes.c_void_p(0))
bb = buf.cpu().numpy()[:12 * 32].reshape(12, 32)
t0 = bb[:, 0][bb[:, 0] != 0].min()
for wv in range(12):
    row = bb[wv]; n = int((row != 0).sum())
    if n == 0: continue
    print("wave", wv, " ".join("%7d" % (v - t0) for v in row[:n]))
    print("  delta", " ".join("%7d" % d for d in (row[1:n] - row[:n - 1])))
