#!/bin/bash
# Kernel trace of the one-rank RCCL rehearsal of the data-parallel step (process group from the environment, no launcher).
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LOCATE_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --step-only $2 $3 > $O/trace.log 2>&1 || exit 1
python tools/prof_summary.py $O/trace 14 > $O/by_category.txt
python tools/step_sequence.py $O/trace 2 > $O/step_sequence.txt
rm -rf $O/trace
