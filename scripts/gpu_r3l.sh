#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3l
mkdir -p $O
timeout 600 python scripts/conv_ablate_bh.py in > $O/in_nv.log 2>&1
EXTRA="--dtype bf16s --no-profile" bash scripts/gpu_tune.sh r3l/tune default "bh_ck5=8" default "bh_ck5=8" "kg_wgs=0" "bh_ck5=8 kg_wgs=0" "in_pairs_nv=2" "in_pairs_nv=4" "bh_ck5=8 in_pairs_nv=2" > $O/tune.log 2>&1
cat $O/in_nv.log $O/tune.log
