import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from kapre_amd import ApplyFilterbank
def time_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n
kw = dict(sample_rate=44100, n_freq=1025, n_mels=128, f_min=0.0, f_max=22050.0)
for c in (1, 2, 6):
    xl = torch.rand(256, 42, 1025, c, device="cuda")
    xf = xl.permute(0, 3, 1, 2).contiguous()
    fl = ApplyFilterbank('mel', kw, data_format='channels_last'); ff = ApplyFilterbank('mel', kw, data_format='channels_first')
    yl = fl(xl); yf = ff(xf)
    err = (yl.permute(0,3,1,2) - yf).abs().max().item() / yf.abs().max().item()
    print("C=%d channels_last %8.1f us  channels_first %8.1f us  rel diff %.1e" % (c, time_us(lambda: fl(xl)), time_us(lambda: ff(xf)), err))
