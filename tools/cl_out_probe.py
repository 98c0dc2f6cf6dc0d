#!/usr/bin/env python3
"""cfg3-shaped mel launches (256 x 6 x 44100, n_fft 2048, hop 1024, 128 mels) for every (input, output) layout pair, decibels on
and off: what the channels_last OUTPUT costs on its own (development aid).  python tools/cl_out_probe.py [ch] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kapre_amd as kapre
from kapre_amd import _ffi
from tools.sweep_dispatch import time_graph

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 6
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(1)
xl = torch.from_numpy(rng.uniform(-1, 1, (batch, 44100, ch)).astype(np.float32)).cuda()
xf = xl.permute(0, 2, 1).contiguous()
for db in (True, False):
    for fi in ("channels_last", "channels_first"):
        for fo in ("channels_last", "channels_first"):
            m = kapre.get_melspectrogram_layer(n_fft=2048, hop_length=1024, sample_rate=44100, n_mels=128, return_decibel=db,
                                               input_data_format=fi, output_data_format=fo)
            x = xl if fi == "channels_last" else xf
            us = time_graph(lambda: m(x))
            print("dB %d  in %-14s out %-14s %8.2f us  [%s]" % (db, fi, fo, us, _ffi.last_launches()), flush=True)
