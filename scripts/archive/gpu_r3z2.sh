#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3z; mkdir -p $O
timeout 300 python scripts/halves_probe2.py 2>&1 | grep -v amdgpu.ids | tee $O/halves_probe2.log
