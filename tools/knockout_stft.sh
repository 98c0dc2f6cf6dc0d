#!/bin/bash
# cfg4 STFT / InverseSTFT with parts of the kernels knocked out: builds the variants HERE (CPU: hipcc
# cross-compiles), runs them on the GPU box:   tools/knockout_stft.sh build   |   tools/knockout_stft.sh run  (on the box)
REPO=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $REPO
if [ "$1" = build ]; then
  for v in 1 2 3; do python tools/build_variant.py stft_ko$v -DKPR_STFT3_KO=$v & done
  for v in 1 2 3; do python tools/build_variant.py ipw_ko$v -DKPR_IPW_KO=$v & done
  wait; exit 0
fi
run() { KAPRE_AMD_LIB=$1 python tools/kbench.py $2 2>&1 | grep -v "Warn\|amdgpu.ids" | sed "s#^#$3  #"; }
for r in 1 2; do
  run kapre_amd/lib/libkapre_hip.so cfg4_stft_b128x1x110250_nfft1024_hop256_pad "stft full                 "
  run kapre_amd/lib/libkapre_hip_stft_ko1.so cfg4_stft_b128x1x110250_nfft1024_hop256_pad "stft stores only          "
  run kapre_amd/lib/libkapre_hip_stft_ko2.so cfg4_stft_b128x1x110250_nfft1024_hop256_pad "stft loads + transform    "
  run kapre_amd/lib/libkapre_hip_stft_ko3.so cfg4_stft_b128x1x110250_nfft1024_hop256_pad "stft loads only           "
  run kapre_amd/lib/libkapre_hip.so cfg4_istft_b128x1x434f_nfft1024_hop256 "istft full                "
  run kapre_amd/lib/libkapre_hip_ipw_ko1.so cfg4_istft_b128x1x434f_nfft1024_hop256 "istft loads only          "
  run kapre_amd/lib/libkapre_hip_ipw_ko2.so cfg4_istft_b128x1x434f_nfft1024_hop256 "istft loads + transform   "
  run kapre_amd/lib/libkapre_hip_ipw_ko3.so cfg4_istft_b128x1x434f_nfft1024_hop256 "istft stores only         "
done
