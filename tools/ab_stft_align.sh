run() { KAPRE_AMD_LIB=$1 python tools/kbench.py $2 2>&1 | grep -v "Warn\|amdgpu.ids" | sed "s#^#$3  #"; }
W=cfg4_stft_b128x1x110250_nfft1024_hop256_pad
for r in 1 2; do
run kapre_amd/lib/libkapre_hip.so $W "aligned nontemporal    "
run kapre_amd/lib/libkapre_hip_plain.so $W "aligned plain          "
run kapre_amd/lib/libkapre_hip_st_al.so $W "stores only nontemporal"
run kapre_amd/lib/libkapre_hip_st_plain.so $W "stores only plain      "
done
