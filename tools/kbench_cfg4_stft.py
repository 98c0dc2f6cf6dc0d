import os, sys
sys.path.insert(0, os.getcwd())
import bench, torch
from kapre_amd import _ffi
for name in ["cfg4_stft_b128x1x110250_nfft1024_hop256_pad"]:
    w = bench.WORKLOADS[name]
    model = bench.build_model(w)
    x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
    for rep in range(3):
        us, _ = bench.kernel_time_us(model, x, launches=100)
        print(name[:12], "%.2f us" % us, _ffi.last_launches(), flush=True)
