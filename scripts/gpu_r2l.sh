#!/bin/bash
# small-batch sweep: which opt-in kernels pay when the step is launch-latency-bound?
OUT=gpurun_out/${1:-r2l}; mkdir -p $OUT; export TMPDIR=/tmp
for B in 4 16 64; do
  echo "== batch $B"
  EXTRA="--batch $B --steps 60 --warmup 10 --no-profile" bash scripts/gpu_tune.sh ${1:-r2l}_b$B default "in_fusion=1" "conv_small=3" "in_fusion=1 conv_small=3" "wgrad_batch=100" default
done
