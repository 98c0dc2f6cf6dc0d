#!/bin/bash
# Build the stand-alone probes (gfx950 cross-compile; no GPU needed).  Binaries + assembly land in tools/probes/bin/
# (git-ignored, but they travel to the GPU box with gpurun).
set -e
cd "$(dirname "$0")"
mkdir -p bin
for p in fft_core valu_micro mel_epilogue "$@"; do
  [ -f $p.hip ] || continue
  EXTRA=""
  # mel_epilogue takes its band plan from kpr_filterbank_pack of the product library
  [ $p = mel_epilogue ] && EXTRA="-L../../kapre_amd/lib -lkapre_hip -Wl,-rpath,\$ORIGIN/../../../kapre_amd/lib"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -Wno-unused-command-line-argument -save-temps=obj ${PROBE_FLAGS} -o bin/$p $p.hip $EXTRA
  python isa_count.py bin/$p-hip-amdgcn-amd-amdhsa-gfx950.s > bin/$p.isa.md
  rm -f bin/$p-*.bc bin/$p-*.hipi bin/$p-*.o bin/$p-*.out bin/$p-*.txt bin/$p-*.hipfb
done
ls -la bin
