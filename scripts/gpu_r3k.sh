#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3k
mkdir -p $O
timeout 900 python scripts/conv_ablate_bh.py > $O/ablate_bh.log 2>&1
cat $O/ablate_bh.log
