#!/bin/bash
# One rocprofv3 kernel trace of the benchmarked step plus the three summaries (category, kernel x grid, sequence) into gpurun_out/$1/.
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 2 --step-only > $O/trace.log 2>&1 || exit 1
python tools/prof_summary.py $O/trace 14 > $O/by_category.txt
python tools/trace_by_grid.py $O/trace 14 "" 60 > $O/by_kernel_and_grid.txt
python tools/step_sequence.py $O/trace 2 > $O/step_sequence.txt
rm -rf $O/trace
