#!/bin/bash
# End-of-round battery on the GPU box (run from the repo root through gpurun): the GPU suite, the stand-alone fuzz on three seeds,
# the rocprofv3 passes of tools/profile_round.sh on the same binary, a bench.py run, and the multi-process rehearsal.
#   bash tools/end_of_round.sh r06 [fuzz seconds per seed]
R=${1:-r06}
FZ=${2:-150}
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/${R}_pytest_gpu_final.log
tail -2 gpurun_out/${R}_pytest_gpu_final.log
for S in 0 17 23; do
  python tools/fuzz_parity.py $FZ $S 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_fuzz_seed$S.log
  tail -1 gpurun_out/${R}_fuzz_seed$S.log
done
bash tools/profile_round.sh $R > gpurun_out/${R}_profile_round.log 2>&1
tail -3 gpurun_out/${R}_profile_round.log
cd $REPO
python bench.py > gpurun_out/${R}_bench_stdout.log 2> gpurun_out/${R}_bench_stderr.log
tail -1 gpurun_out/${R}_bench_stdout.log > gpurun_out/${R}_bench_line.json
cp gpurun_out/bench_full.json gpurun_out/${R}_bench_full.json 2>/dev/null
bash tools/rehearse_multi_gpu.sh > gpurun_out/${R}_rehearsal_n2_one_gpu.log 2>&1
tail -1 gpurun_out/${R}_rehearsal_n2_one_gpu.log | cut -c1-400
