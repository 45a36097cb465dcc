#!/bin/bash
# Same-box A/B of the whole step: the working tree against the snapshot under .ab/base (a git worktree of an earlier commit with
# its own built library; `git worktree add -f .ab/base <commit>` + `python -m locate_amd.build` inside it).  MI355X devices differ
# by several per cent between gpurun boxes, so only runs inside ONE call compare.  usage: tools/ab_bench.sh [rounds] [extra bench args]
R=${1:-2}; shift
for i in $(seq 1 $R); do
  echo -n "base: "; (cd .ab/base && python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['losses'])")
  echo -n "new:  "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['losses'])"
done
