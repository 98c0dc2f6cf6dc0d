#!/bin/bash
# Rehearse the multi-process launch of bench.py on a box with ONE GPU: the driver's command line for N = 2
# (torch.distributed.run, one process per rank), but every rank on device 0 and the process group on gloo
# (RCCL refuses two ranks on one device).  Exercises env parsing, the constants broadcast, the barriers, the
# max-over-ranks reduction and the single JSON line; the number it prints is NOT a scaling measurement.
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
KAPRE_AMD_DIST_BACKEND=gloo KAPRE_AMD_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 \
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port ${1:-29517} \
    bench.py --gpus 2 --steps 50 --warmup 5
