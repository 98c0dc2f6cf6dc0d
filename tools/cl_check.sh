#!/bin/bash
# channels_last (interleaved) waveforms against channels_first on one box: parity tests, then kernel times (development aid)
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "interleaved or mel_kernel_variants or cfg3 or channels" 2>&1 | tail -5
T=target_mel_b256x1x44100_nfft2048_hop512_mel128
C3=cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128
C5=cfg5_mel_b256x1x160000_nfft1024_hop160_mel80
python tools/kbench.py $T ${C3}_cl ${C3}_cf 2>&1 | grep -v Warn
python tools/kbench.py mel_variant=7 ${C3}_cl 2>&1 | grep -v Warn
for a in "ch=2" "ch=2 option:mel_variant=7" "ch=2 fmt=channels_first" "ch=4 batch=64" "ch=4 batch=64 fmt=channels_first"; do python tools/kbench_custom.py $T $a 2>&1 | tail -1; done
for a in "ch=4 batch=64" "ch=4 batch=64 option:mel_variant=7" "ch=4 batch=64 fmt=channels_first" "ch=2 batch=128" "ch=2 batch=128 option:mel_variant=8" "ch=2 batch=128 fmt=channels_first"; do python tools/kbench_custom.py $C5 $a 2>&1 | tail -1; done
