import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import kapre_oracle as o
from kapre_amd import STFT, Magnitude, Sequential, _ffi
rng = np.random.default_rng(0)
for n_fft, hop, ch, batch, t in ((1024, 256, 4, 6, 110250), (512, 128, 2, 8, 60000), (2048, 512, 2, 4, 120000)):
    x = rng.standard_normal((batch, t, ch)).astype(np.float32)
    xt = np.ascontiguousarray(x.transpose(0, 2, 1))
    for fi, fo, xin in (("channels_last", "channels_first", x), ("channels_first", "channels_last", xt), ("channels_last", "channels_last", x)):
        for var in (0, 3):
            _ffi.set_option("stft_variant", var)
            kw = dict(n_fft=n_fft, hop_length=hop, input_data_format=fi, output_data_format=fo)
            got = STFT(**kw)(xin).cpu().numpy()
            lab = _ffi.last_launches()
            want = o.kapre_stft(xin, **kw)
            err = np.abs(got - want).max() / np.abs(want).max()
            gm = Sequential([STFT(**kw), Magnitude()])(xin).cpu().numpy()
            errm = np.abs(gm - np.abs(want)).max() / np.abs(want).max()
            print(n_fft, ch, fi[9:], "->", fo[9:], "variant", var, lab, "rel err %.2e  mag %.2e" % (err, errm), "BAD" if err > 1e-4 or errm > 1e-4 else "")
_ffi.set_option("stft_variant", 0)
