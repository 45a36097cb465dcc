#!/bin/bash
# debug-library sweeps of the eight-wave threshold (LOCATE_W8_MAX) and the split cap (LOCATE_KS_MAX) on chosen shapes
S=${SHAPES:-"--shape convT,768,768,4,2,1,4,64 --shape convT,384,384,4,2,1,8,64 --shape convT,192,192,4,2,1,16,64 --shape convT,96,96,4,2,1,32,64 --shape conv,48,48,3,1,1,64,64"}
for m in ${W8:-320 4096}; do for k in ${KS:-64}; do echo "W8_MAX=$m KS_MAX=$k"; LOCATE_HIP_DEBUG_LIBRARY=1 LOCATE_W8_MAX=$m LOCATE_KS_MAX=$k python tools/bench_conv.py --reps 30 $S 2>/dev/null | grep -v "^stage"; done; done
