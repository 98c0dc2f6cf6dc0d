# wave priorities at the end of a short run of k_mel_pw (profiles/r06_mel_tail.md): product against the build before the change
for lib in tools/probes/bin/lib_before.so "" tools/probes/bin/lib_before.so ""; do
echo "== lib $lib"
KAPRE_AMD_LIB=$lib python tools/kbench.py settle=1.0 target_mel_b256x1x44100_nfft2048_hop512_mel128 cfg2_mel_b64x1x44100_nfft2048_hop512_mel128 cfg5_mel_b256x1x160000_nfft1024_hop160_mel80 cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cl cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40 2>&1 | grep -v amdgpu | cut -c1-110
done
