import sys, os, ctypes
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi, STFT, Magnitude, ApplyFilterbank, Sequential
B, T, n_fft, hop, M, sr = 100, 44100, 2048, 512, 128, 44100
K = n_fft // 2 + 1
x = np.random.default_rng(1).uniform(-1, 1, (B, T, 1)).astype(np.float32)
layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=sr, n_freq=K, n_mels=M))
fb = np.array(layer.filterbank)
st = STFT(n_fft=n_fft, hop_length=hop)
mag = Sequential([st, Magnitude()])(x).cpu().numpy().astype(np.float64)
want = np.einsum("bfkc,km->bfmc", mag, fb.astype(np.float64))
model = Sequential([st, Magnitude(), layer])
model(x); torch.cuda.synchronize()
NB = 1024 + 256 * 128 + 256 * 64 + (100 * 83 * 1056) // 2 + 1024
buf = torch.zeros(NB, dtype=torch.int64, device="cuda"); buf[12 * 32] = -1
L = _ffi.lib()
L.kpr_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
got = model(x).cpu().numpy(); torch.cuda.synchronize()
L.kpr_debug_stamps(ctypes.c_void_p(0))
b = buf.cpu().numpy()
F = want.shape[1]; total = B * F; grid = 256
err = (np.abs(got - want) / np.maximum(np.abs(want), 1e-30)).reshape(total, M)
badf = np.nonzero((err > 1e-3).any(axis=1))[0]
print("bad frames", len(badf))
rows = b[1024 + 256 * 128 + 256 * 64:].view(np.float32)[: total * 1056].reshape(total, 1056)[:, :K]
magf = mag.reshape(total, K)
rerr = np.abs(rows - magf).max(axis=1) / magf.max(axis=1)
badrows = np.nonzero(rerr > 1e-4)[0]
print("rows differing from |STFT| as seen by the consumer:", len(badrows), "first", badrows[:10].tolist(), "same set as bad frames:", set(badrows.tolist()) == set(badf.tolist()))
for q in badrows[:3]:
    d = np.abs(rows[q] - magf[q]) / magf[q].max()
    bb = np.nonzero(d > 1e-4)[0]
    print("   row", int(q), "bad bins", len(bb), "range", int(bb.min()), int(bb.max()), "max", float(d.max()))
seen = set()
for g in badf:
    wg = min(int(g * grid // total), grid - 1)
    while (total * wg // grid) > g: wg -= 1
    while (total * (wg + 1) // grid) <= g: wg += 1
    if wg in seen: continue
    seen.add(wg)
    if len(seen) > 3: break
    pos = [int(q - total * wg // grid) for q in badf if total * wg // grid <= q < total * (wg + 1) // grid]
    cons = b[1024 + wg * 128: 1024 + (wg + 1) * 128].reshape(4, 8, 4)
    prod = b[1024 + 256 * 128 + wg * 64: 1024 + 256 * 128 + (wg + 1) * 64]
    t0 = min(int(v & ((1 << 56) - 1)) for v in prod if v) 
    print("WG", wg, "frames", total * (wg + 1) // grid - total * wg // grid, "bad positions", pos)
    for q in [q for q in badf if total * wg // grid <= q < total * (wg + 1) // grid][:3]:
        bm = np.nonzero(err[q] > 1e-3)[0]
        print("   frame", int(q - total * wg // grid), "bad mels:", bm.min(), "..", bm.max(), "count", len(bm), "errs", np.round(err[q][bm][:6], 3))
    for w in range(4):
        print("  consumer wave", w, [(int(c[0]), int(c[1]), int(c[2]), int(c[3]) - t0) for c in cons[w][1:4]])
    print("  publish (pos: wave, clock):", [(i, int(v >> 56), int(v & ((1 << 56) - 1)) - t0) for i, v in enumerate(prod) if v][:40])
