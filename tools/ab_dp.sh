#!/bin/bash
# Same-box A/B of the one-rank RCCL rehearsal (LOCATE_DP_FORCE=1) with different bench.py flags: tools/ab_dp.sh OUTDIR "flags A" "flags B" ...
O=gpurun_out/$1; shift; mkdir -p $O
export LOCATE_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
port=29540
for round in 1 2; do
  i=0
  for flags in "$@"; do
    port=$((port+1)); i=$((i+1))
    MASTER_PORT=$port timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 5 --step-only $flags > $O/b_${i}_$round.json 2> $O/b_${i}_$round.err || { tail -5 $O/b_${i}_$round.err; exit 1; }
    echo "[$flags] round $round: $(tail -1 $O/b_${i}_$round.json | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")"
  done
done
unset RANK LOCAL_RANK WORLD_SIZE LOCATE_DP_FORCE
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --step-only > $O/b_nodp.json 2> $O/b_nodp.err
echo "no DP: $(python -c "import json; print(json.load(open('$O/b_nodp.json'))['ms_per_step'])")"
