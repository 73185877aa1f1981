#!/bin/bash
# copies the judged files of one final measurement set (scripts/gpu_r6_final.sh) from gpurun_out/<name>/ into profiles/r06_*
# usage: scripts/copy_final_profiles.sh <name> <build fingerprint> <git head>
set -e
R=gpurun_out/$1
cp $R/rocprof_kernel_stats.csv profiles/r06_rocprof_kernel_stats.csv
cp $R/rocprof_kernel_stats_single_stream.csv profiles/r06_rocprof_kernel_stats_single_stream.csv
cp $R/rocprof_kernel_stats_bf16.csv profiles/r06_rocprof_kernel_stats_bf16.csv
cp $R/rocprof_kernel_stats_bf16_single_stream.csv profiles/r06_rocprof_kernel_stats_bf16_single_stream.csv
cp $R/pmc_fetch_write_summary.json profiles/r06_pmc_fetch_write_summary.json
cp $R/sq_step.json profiles/r06_sq_step.json
cp $R/sq_classes.txt profiles/r06_sq_classes.txt
tail -1 $R/bench_default.json | grep -q '"metric"' || { echo "no bench line"; exit 1; }
grep '"metric"' $R/bench_default.json | tail -1 > profiles/r06_bench.json
for f in infer_b1024 infer_b1024_bf16r infer_b1024_bf16s infer_b1_t400 infer_ragged_32pairs train_b4 train_b4_bf16 train_bf16_b256 train_f32x3_b256 train_m512_b128 train_t1024_b64; do
  grep '"metric"' $R/$f.json | tail -1 > profiles/r06_$f.json
done
cp $R/configs.txt profiles/r06_configs.txt
cp $R/timeline_f32.txt profiles/r06_step_timeline_f32.txt
cp $R/timeline_bf16.txt profiles/r06_step_timeline_bf16.txt
{ echo "# -m gpu (the driver's run), build $2, git $3"; tail -45 $R/pytest_gpu.txt; echo "# -m \"gpu and slow\""; tail -16 $R/pytest_gpu_slow.txt; echo "# smoke"; cat $R/smoke.txt; echo "# default bench wall clock"; tail -4 $R/bench_default.err; } > profiles/r06_pytest_gpu_tail.txt
