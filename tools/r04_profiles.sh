#!/bin/bash
# Round-4 evidence, one gpurun call: bench lines, kernel trace + summaries, PMC passes (each counter set in a pass of its own,
# --kernel-trace only, as MI355X_MICROARCH.md prescribes), micro-benchmarks.  Everything lands in gpurun_out/r04/; the
# summaries are copied into profiles/ by hand afterwards.
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
python bench.py --steps 30 --warmup 5 --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err || exit 1
python bench.py --steps 30 --warmup 5 --dtype fp8 --no-cpu-baseline > $O/bench_fp8.json 2> $O/bench_fp8.err || exit 1
LOCATE_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/dp_world1_rccl.json 2> $O/dp_world1_rccl.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 2 --step-only > $O/trace.log 2>&1 || exit 1
python tools/prof_summary.py $O/trace 14 > $O/by_category.txt
python tools/trace_by_grid.py $O/trace 14 "" 60 > $O/by_kernel_and_grid.txt
python tools/step_sequence.py $O/trace 2 > $O/step_sequence.txt
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 tools/roofline_stages.py --reps 5 > $O/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 tools/roofline_stages.py --reps 5 > $O/pmc_w.log 2>&1 || exit 1
python tools/pmc_summary.py --json $O/conv_pmc_mem.json $O/pmc_f $O/pmc_w > $O/conv_pmc_mem.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/roofline_stages.py --reps 40 > $O/pmc_sq.log 2>&1 || exit 1
python tools/pmc_summary.py $O/pmc_sq conv_ > $O/conv_pmc_sq.txt
for b in 64 256; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ew_f$b -- python3 tools/bench_elementwise.py --reps 2 --batch $b > $O/ew_f$b.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/ew_w$b -- python3 tools/bench_elementwise.py --reps 2 --batch $b > $O/ew_w$b.log 2>&1 || exit 1
  python tools/pmc_summary.py $O/ew_f$b > $O/ew_fetch_b$b.txt; python tools/pmc_summary.py $O/ew_w$b > $O/ew_write_b$b.txt
  python tools/bench_elementwise.py --batch $b > $O/elementwise_microbench_b$b.txt 2>&1
done
python tools/bench_conv.py --f16 --check > $O/conv_microbench_f16.txt 2>&1
python tools/bench_conv.py --f16 --win --check > $O/conv_microbench_f16_window.txt 2>&1
python tools/bench_conv.py --check > $O/conv_microbench_bf16x6.txt 2>&1
python tools/layer_table.py > $O/layer_table.txt 2>&1
python tools/bench_elementwise.py --all-shapes > $O/elementwise_all_shapes.txt 2>&1
python tools/amax_overhead.py > $O/amax_overhead.txt 2>&1
rm -rf $O/trace $O/pmc_f $O/pmc_w $O/pmc_sq $O/ew_f64 $O/ew_f256 $O/ew_w64 $O/ew_w256 2>/dev/null
ls $O
