# cycle stamps of one wave of k_fb_pw on single-row / single-block launches (profiles/r06_fb_pw.md section 7).  Needs a library built
# with -DKPR_FB_STAMPS:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -DKPR_RING_DEPTH=3 -DKPR_FB_STAMPS \
#     -o tools/probes/bin/lib_stamps.so kapre_amd/csrc/kapre_hip.hip
export KAPRE_AMD_LIB=tools/probes/bin/lib_stamps.so
for a in "1 1 channels_first" "1 2 channels_last"; do
python - $a <<'PY' 2>&1 | grep -v "^$" | tail -8
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from kapre_amd import ApplyFilterbank, _ffi
k, rows, batch, ch, fmt = 1025, 83, int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=k, n_mels=128), data_format=fmt)
for nrows in (1, 83):
    shp = (batch, nrows, k, ch) if fmt == "channels_last" else (batch, ch, nrows, k)
    x = torch.rand(shp, device="cuda")
    for _ in range(4):
        y = layer(x)
        torch.cuda.synchronize()
    print(_ffi.last_launches(), nrows, flush=True)
PY
done
