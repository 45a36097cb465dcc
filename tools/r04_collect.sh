#!/bin/bash
# Copies the summaries of gpurun_out/r04 (tools/r04_profiles.sh) into profiles/ under their committed names.
O=gpurun_out/r04
cp $O/by_kernel_and_grid.txt profiles/r04_bench_default_by_kernel_and_grid.txt
cp $O/kernel_stats.csv profiles/r04_bench_default_kernel_stats.csv
cp $O/step_sequence.txt profiles/r04_step_sequence.txt
cp $O/conv_pmc_sq.txt profiles/r04_conv_pmc_sq_summary.txt
cp $O/conv_pmc_mem.json profiles/r04_conv_pmc_mem.json
cp $O/bench_bf16.json profiles/r04_bf16_bench.json
cp $O/bench_fp8.json profiles/r04_fp8_bench.json
cp $O/conv_microbench_f16.txt profiles/r04_conv_microbench_f16_pieces.txt
cp $O/conv_microbench_f16_window.txt profiles/r04_conv_microbench_f16_pieces_window_everywhere.txt
cp $O/conv_microbench_bf16x6.txt profiles/r04_conv_microbench_bf16_pieces.txt
grep -v "window form" $O/layer_table.txt > profiles/r04_layer_table.txt
cp $O/amax_overhead.txt profiles/r04_amax_overhead.txt
cp $O/elementwise_all_shapes.txt profiles/r04_elementwise_all_shapes.txt
cp $O/dp_world1_rccl.json profiles/r04_dp_world1_rccl.json
N=$(head -1 $O/step_sequence.txt | sed -E 's/.*one step: ([0-9]+) launches.*/\1/')
(echo "# rocprofv3 --kernel-trace of \`bench.py --steps 10 --warmup 2 --step-only\`, all launches / 14: the eager warm-up iterations before the capture are"
 echo "# inside the trace, so launches/step reads high here - one REPLAYED step is $N launches (profiles/r04_step_sequence.txt)"
 cat $O/by_category.txt) > profiles/r04_bench_default_by_category.txt
for b in 64 256; do
  (for n in ew_fetch_b$b.txt ew_write_b$b.txt elementwise_microbench_b$b.txt; do [ -f $O/$n ] && { echo "## $n"; cat $O/$n; }; done) > profiles/r04_elementwise_pmc_mem_b$b.txt
done
