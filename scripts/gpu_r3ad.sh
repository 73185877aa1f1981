#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3ad; mkdir -p $O
timeout 55 python scripts/conv_ablate_small.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_ablate_small.log
