#!/usr/bin/env python3
"""A/B of the staged channels_last store of k_mel_pw's PAIR form ("mel_cl_stage" 1 / 0) on cfg3-shaped launches, same process,
same box (development aid).  python tools/cl_stage_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kapre_amd as kapre
from kapre_amd import _ffi
from tools.sweep_dispatch import time_graph

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(1)
for ch, batch, hop in ((6, 256, 1024), (4, 256, 1024), (8, 192, 1024), (6, 256, 512)):
    x = torch.from_numpy(rng.uniform(-1, 1, (batch, 44100, ch)).astype(np.float32)).cuda()
    for db in (True, False):
        m = kapre.get_melspectrogram_layer(n_fft=2048, hop_length=hop, sample_rate=44100, n_mels=128, return_decibel=db,
                                           input_data_format="channels_last", output_data_format="channels_last")
        res = {0: [], 1: []}
        for r in range(rounds):
            for stage in (0, 1):
                _ffi.set_option("mel_cl_stage", stage)
                res[stage].append(time_graph(lambda: m(x)))
        _ffi.set_option("mel_cl_stage", 1)
        print("C %d batch %d hop %d dB %d   8-byte pairs %s   staged blocks %s   [%s]" % (
            ch, batch, hop, db, " ".join("%.1f" % v for v in res[0]), " ".join("%.1f" % v for v in res[1]), _ffi.last_launches()), flush=True)
