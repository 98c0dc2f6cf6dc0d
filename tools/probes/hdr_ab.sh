for lib in tools/probes/bin/lib_before.so "" tools/probes/bin/lib_before.so ""; do
echo "== lib $lib"
KAPRE_AMD_LIB=$lib python tools/kbench.py settle=1.0 target_mel_b256x1x44100_nfft2048_hop512_mel128 cfg2_mel_b64x1x44100_nfft2048_hop512_mel128 cfg5_mel_b256x1x160000_nfft1024_hop160_mel80 2>&1 | tail -4
done
