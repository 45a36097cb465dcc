#!/bin/bash
# SQ counters of the contraction kernels on ONE shape of tools/bench_conv.py, gather form against window form (one rocprofv3 --pmc
# pass per counter group and form).  usage (GPU box): bash tools/pmc_shape.sh <outdir> <bench_conv --only filter> [kernel filter]
O=$1; ONLY=$2; FLT=${3:-conv_}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  for form in gather win; do
    extra=""; [ $form = win ] && extra="--win"
    rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/p$i$form -- python3 tools/bench_conv.py --f16 $extra --reps 5 --only "$ONLY" > $O/p$i$form.log 2>&1 || { echo "pass $i $form failed"; tail -3 $O/p$i$form.log; continue; }
    echo "== $form" >> $O/summary.txt
    python tools/pmc_summary.py $O/p$i$form $FLT >> $O/summary.txt
    rm -rf $O/p$i$form
  done
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES
GROUPS
cat $O/summary.txt
