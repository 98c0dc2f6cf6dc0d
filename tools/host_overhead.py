#!/usr/bin/env python3
"""Host side of one fused mel call at batch 1 (the serving case): wall-clock per call over 2000 back-to-back calls and a
cProfile of where it goes (plan lookup, torch.empty, the ctypes call).  Measured: 11.3 us without dB, 17.7 us with (three
launches) -- about the kernel time of such a launch, so neither side waits long for the other."""
import sys, time; sys.path.insert(0, ".")
import torch, bench
import cProfile, pstats
name = "speech_mel_b256x1x160000_nfft400_hop160_mel80"
for db in (False, True):
    w = dict(bench.WORKLOADS[name]); w["batch"] = 1; w["db"] = db
    model = bench.build_model(w); x = bench.make_input(w, 0, torch.device("cuda", 0), 1)
    for _ in range(200): model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): model(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("db", db, "host per call %.1f us, incl. drain %.1f us" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): model(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
