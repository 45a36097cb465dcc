#!/bin/bash
# Split-K sweep of one mid-sized contraction with the debug library (LOCATE_KS_MAX): tools/ks_sweep.sh OUTDIR "kind,Cin,Cout,k,stride,pad,H,B" ...
O=gpurun_out/$1; shift; mkdir -p $O
export LOCATE_HIP_DEBUG_LIBRARY=1
for shape in "$@"; do
  for ks in 1 2 4 8 16 32 64; do
    echo "== $shape LOCATE_KS_MAX=$ks"
    LOCATE_KS_MAX=$ks python tools/bench_conv.py --f16 --shape $shape --reps 20 2>/dev/null | tail -2
  done
  echo "== $shape no counters (separate reduction)"
  python tools/bench_conv.py --f16 --no-counters --shape $shape --reps 20 2>/dev/null | tail -2
done
