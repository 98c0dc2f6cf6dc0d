import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kapre_amd import ApplyFilterbank, _ffi
k, rows, batch, ch, fmt = 1025, 83, int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=k, n_mels=128), data_format=fmt)
shp = (batch, rows, k, ch) if fmt == "channels_last" else (batch, ch, rows, k)
x = torch.rand(shp, device="cuda")
for _ in range(60):
    y = layer(x)
torch.cuda.synchronize()
print(_ffi.last_launches())
