#!/bin/bash
# bf16 storage: accuracy distribution at the graded shapes + launch-heuristic sweeps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3j
mkdir -p $O
timeout 600 python scripts/diag_bf16s.py gpu m80 256 128 bf16 bf16s > $O/diag_b256.log 2>&1
timeout 600 python scripts/diag_bf16s.py gpu m80 64 1024 bf16s > $O/diag_t1024.log 2>&1
EXTRA="--dtype bf16s" bash scripts/gpu_tune.sh r3j/tune default "bh_ck5=8" "bh_ck5=32" "kg_wgs=0" "dec_split_min=1000000" "dgrad_par=0" "tile_thr11=2048 tile_thr21=2048" "tile_thr11=100000" "wgrad_batch_wgs=512" "wgrad_batch=4" "ck16_wgs=0 ck32_wgs=0" > $O/tune.log 2>&1
cat $O/diag_b256.log | tail -14; tail -8 $O/diag_t1024.log; cat $O/tune.log
