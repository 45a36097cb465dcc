#!/bin/bash
# A/B: production library vs debug library (different compile-time settings) - alternating, two rounds
O=gpurun_out/$1; mkdir -p $O
for round in 1 2 3; do
  for v in 0 1; do
    LOCATE_HIP_DEBUG_LIBRARY=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --step-only > $O/b_${v}_$round.json 2> $O/b_${v}_$round.err || exit 1
    echo "debug_library=$v round $round: $(python -c "import json; print(json.load(open('$O/b_${v}_$round.json'))['ms_per_step'])")"
  done
done
