#!/bin/bash
# Same-box A/B of bench.py under different values of ONE environment variable: tools/ab_env.sh OUTDIR VAR v1 v2 ... (two rounds, alternating)
O=gpurun_out/$1; VAR=$2; shift 2; mkdir -p $O
for round in 1 2; do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --step-only > $O/b_${v}_$round.json 2> $O/b_${v}_$round.err || exit 1
    echo "$VAR=$v round $round: $(python -c "import json,sys; print(json.load(open('$O/b_${v}_$round.json'))['ms_per_step'])")"
  done
done
