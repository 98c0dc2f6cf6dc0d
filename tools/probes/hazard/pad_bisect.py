#!/usr/bin/env python3
"""ISA-level bisection of the wrong-lane schedule of k_istft_pw<512, 2, 16> (round 4: commit 56d7f24 without its two
`__builtin_amdgcn_sched_barrier(0)` lines; `ipw512_2_failing.s.txt` is `hipcc -S` of exactly that source, instruction for
instruction the stream that produced wrong samples in the lanes `fl mod 16 < 2`).

Runs ON the GPU box (needs /opt/rocm/lib/llvm/bin/clang + ld.lld and a gfx950 device):
  * patches the assembly text -- `s_nop 7` after chosen instructions, nothing else moves --, assembles it to a code object,
    loads it with hipModuleLoad and launches the kernel on one signal of 230 frames (n_fft 1024, hop 128),
  * compares with a numpy inverse STFT (irfft + synthesis window + overlap-add; float64),
  * delta-debugs the set of insertion points down to a minimal set that makes the result right.
No library code is involved: the kernel's inputs (twiddle table, plan, window) are rebuilt here from their definitions
(kapre_hip.hip get_twiddles / launch_istft_pw of that commit).

python tools/probes/hazard/pad_bisect.py [--asm FILE] [--out DIR]
"""
import argparse
import ctypes
import os
import re
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KERNEL = "_ZN3kpr10k_istft_pwILi512ELi2ELi16EEEvPK15HIP_vector_typeIfLj2EENS_11IstftPwPlanEPKfS4_Pf"
LLVM = "/opt/rocm/lib/llvm/bin"
N_FFT, HOP, F, NC, W = 1024, 128, 230, 512, 16
LDS_BYTES = 4 * (W * 2 * 552 + 2 * NC + 2 * 64 * 10) + 4 * (W * 2 + 4)       # ipw_lds_bytes(512, 16) of that commit = 80016


def split_asm(text):
    """-> (lines, [line index of every instruction of the kernel body])"""
    lines = text.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    idx = [i for i in range(start + 1, end) if re.match(r"^\s+[a-z]", lines[i]) and not lines[i].strip().startswith((";", "."))]
    return lines, idx


def make_variant(lines, idx, pads, path, nop="s_nop 7"):
    """pads: set of positions (into idx) after which `nop` is inserted"""
    out = []
    where = {idx[p] for p in pads}
    for i, l in enumerate(lines):
        if ".amdhsa_group_segment_fixed_size" in l:
            l = "\t\t.amdhsa_group_segment_fixed_size %d" % LDS_BYTES
        elif ".group_segment_fixed_size:" in l:
            l = "    .group_segment_fixed_size: %d" % LDS_BYTES
        out.append(l)
        if i in where:
            out.append("\t" + nop)
    with open(path + ".s", "w") as f:
        f.write("\n".join(out) + "\n")
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", path + ".s", "-o",
                    path + ".o"], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", path + ".o", "-o", path + ".co"], check=True)
    return path + ".co"


class Runner:
    def __init__(self):
        import torch
        self.torch = torch
        torch.zeros(1, device="cuda")
        libpath = None
        for l in open("/proc/self/maps"):
            if "libamdhip64" in l:
                libpath = l.split()[-1]
                break
        self.hip = ctypes.CDLL(libpath)
        rng = np.random.default_rng(1)
        K = NC + 1
        spec = (rng.standard_normal((F, K)) + 1j * rng.standard_normal((F, K))).astype(np.complex64)
        n = np.arange(N_FFT)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * n / N_FFT)
        den = np.zeros(N_FFT)
        for r in range(N_FFT // HOP):                                        # tf.signal.inverse_stft_window_fn
            den += np.roll(w * w, r * HOP)
        synth = (w / den).astype(np.float32)
        t_out = (F - 1) * HOP + N_FFT
        ref = np.zeros(t_out)
        frames = np.fft.irfft(spec.astype(np.complex128), N_FFT, axis=-1) * synth.astype(np.float64)
        for f in range(F):
            ref[f * HOP:f * HOP + N_FFT] += frames[f]
        self.ref, self.t_out = ref, t_out
        j = np.arange(N_FFT)
        tw = np.stack([np.cos(-2 * np.pi * j / N_FFT), np.sin(-2 * np.pi * j / N_FFT)], -1).astype(np.float32)
        self.d_spec = torch.from_numpy(spec.view(np.float32).reshape(F, K, 2).copy()).cuda()
        self.d_synth = torch.from_numpy(synth).cuda()
        self.d_tw = torch.from_numpy(tw).cuda()
        self.d_out = torch.zeros(t_out + 4096, device="cuda")
        self.plan = struct.pack("<qiiiii", t_out, F, N_FFT, HOP, 1, 1) + b"\0" * 4      # IstftPwPlan of that commit (32 bytes)

    def run(self, co):
        hip, torch = self.hip, self.torch
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), co.encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL.encode()) == 0
        self.d_out.zero_()
        a0 = ctypes.c_void_p(self.d_spec.data_ptr())
        a1 = ctypes.create_string_buffer(self.plan, 32)
        a2 = ctypes.c_void_p(self.d_synth.data_ptr())
        a3 = ctypes.c_void_p(self.d_tw.data_ptr())
        a4 = ctypes.c_void_p(self.d_out.data_ptr())
        params = (ctypes.c_void_p * 5)(ctypes.addressof(a0), ctypes.addressof(a1), ctypes.addressof(a2), ctypes.addressof(a3),
                                       ctypes.addressof(a4))
        torch.cuda.synchronize()
        rc = hip.hipModuleLaunchKernel(fn, 1, 1, 1, W * 64, 1, 1, 0, None, params, None)
        assert rc == 0, rc
        assert hip.hipDeviceSynchronize() == 0
        got = self.d_out[:self.t_out].cpu().numpy().astype(np.float64)
        hip.hipModuleUnload(mod)
        bad = np.nonzero(np.abs(got - self.ref) > 1e-5 * np.abs(self.ref).max())[0]
        return len(bad), sorted(set((bad % 64).tolist()))[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=os.path.join(HERE, "ipw512_2_failing.s.txt"))
    ap.add_argument("--out", default="gpurun_out/hazard")
    ap.add_argument("--nop", default="s_nop 7")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    lines, idx = split_asm(open(a.asm).read())
    n = len(idx)
    r = Runner()
    log = open(os.path.join(a.out, "bisect.log"), "w")

    def say(*x):
        s = " ".join(str(v) for v in x)
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()
    tested = {}

    def wrong(pads, tag):
        key = frozenset(pads)
        if key not in tested:
            co = make_variant(lines, idx, key, os.path.join(a.out, "v"), a.nop)
            tested[key] = r.run(co)
            say("%-28s pads %5d -> wrong samples %6d  (t mod 64 in %s)" % (tag, len(key), tested[key][0], tested[key][1]))
        return tested[key][0]

    say("kernel instructions:", n)
    base = wrong(set(), "unpatched")
    full = wrong(set(range(n)), "pad after every instruction")
    if base == 0:
        say("RESULT: the unpatched stream is RIGHT on this box: nothing to bisect")
        return
    if full != 0:
        say("RESULT: padding every instruction does NOT repair it: not a wait-state hazard (dataflow / logic)")
        # which single-eighths change anything?
        for k in range(8):
            wrong(set(range(k * n // 8, (k + 1) * n // 8)), "eighth %d" % k)
        return
    # ddmin on the set of pad positions: find a minimal subset that still repairs the result
    cur = list(range(n))
    gran = 2
    while len(cur) >= 2:
        chunk = max(1, len(cur) // gran)
        subsets = [cur[i:i + chunk] for i in range(0, len(cur), chunk)]
        reduced = False
        for s in subsets:                                                    # a subset alone repairs it
            if wrong(set(s), "subset %d/%d" % (len(s), len(cur))) == 0:
                cur, gran, reduced = s, 2, True
                break
        if not reduced:
            for s in subsets:                                                # the complement repairs it
                comp = [p for p in cur if p not in set(s)]
                if comp and wrong(set(comp), "complement %d/%d" % (len(comp), len(cur))) == 0:
                    cur, gran, reduced = comp, max(gran - 1, 2), True
                    break
        if not reduced:
            if gran >= len(cur):
                break
            gran = min(len(cur), gran * 2)
    say("RESULT: minimal repairing pad set (%d positions):" % len(cur))
    for p in cur:
        lo = max(0, p - 6)
        say("  --- pad after instruction %d (asm line %d)" % (p, idx[p] + 1))
        for q in range(lo, min(n, p + 5)):
            say("    %s%s" % (">>" if q == p else "  ", lines[idx[q]].strip()))
    # how many states does it take?
    for nop in ("s_nop 0", "s_nop 1", "s_nop 3"):
        co = make_variant(lines, idx, set(cur), os.path.join(a.out, "w"), nop)
        say("with `%s` at the minimal set: wrong samples %d" % (nop, r.run(co)[0]))


if __name__ == "__main__":
    main()
