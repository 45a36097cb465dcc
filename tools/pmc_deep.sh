#!/bin/bash
# Instruction mix / pipe activity / memory-path counters of the dominant kernel (tools/roofline_stages.py), one rocprofv3 --pmc pass
# per counter group (--kernel-trace only).  usage (GPU box): bash tools/pmc_deep.sh [outdir] [extra roofline_stages args]
set -o pipefail
O=${1:-gpurun_out/pmc_deep}; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/p$i -- python3 tools/roofline_stages.py --reps 10 "$@" > $O/p$i.log 2>&1 || { echo "pass $i failed: $group"; tail -5 $O/p$i.log; continue; }
  python tools/pmc_summary.py $O/p$i conv_igemm >> $O/summary.txt
  rm -rf $O/p$i
done <<'GROUPS'
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_CVT
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL
TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr
SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_CYCLES
GROUPS
cat $O/summary.txt
