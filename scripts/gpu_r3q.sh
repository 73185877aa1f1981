#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3q
mkdir -p $O
EXTRA="--no-profile" bash scripts/gpu_tune.sh r3q/f32 default "conv_min_lds=42000" "conv_min_lds=56000" "conv_min_lds=84000" > $O/tune_f32.log 2>&1
EXTRA="--no-profile --dtype bf16" bash scripts/gpu_tune.sh r3q/bf16 default "conv_min_lds=42000" "conv_min_lds=56000" "conv_min_lds=84000" > $O/tune_bf16.log 2>&1
cat $O/tune_f32.log $O/tune_bf16.log
