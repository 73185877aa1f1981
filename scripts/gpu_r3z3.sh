#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3z; mkdir -p $O
{ timeout 200 python scripts/streams_probe.py 256; timeout 200 python scripts/streams_probe.py 4; } 2>&1 | grep -v amdgpu.ids | tee $O/streams_probe.log
