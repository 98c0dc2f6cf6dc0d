#!/bin/bash
# k_istft_pw against the ring kernel on one box: parity tests, then kernel times (development aid)
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "istft" 2>&1 | tail -8
I=cfg4_istft_b128x1x434f_nfft1024_hop256
for a in "" "option:istft_path=3" "option:istft_path=4" "batch=32" "batch=32 option:istft_path=3" "batch=512" "batch=512 option:istft_path=3"; do python tools/kbench_custom.py $I $a 2>&1 | tail -1; done
