#!/usr/bin/env python3
"""Build kapre_amd/lib/libkapre_hip_<name>.so with extra -D flags (development probes; select with
KAPRE_AMD_LIB=...).  usage: tools/build_variant.py name [-DFLAG ...]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(REPO, "kapre_amd", "lib", "libkapre_hip_%s.so" % name)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
       "-DKPR_RING_DEPTH=3"] + flags + ["-o", out, os.path.join(REPO, "kapre_amd", "csrc", "kapre_hip.hip")]
subprocess.run(cmd, check=True)
print(out)
