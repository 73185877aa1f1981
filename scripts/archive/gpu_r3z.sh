#!/bin/bash
# round 3, call z: why is the half-batch pipeline slow and nondeterministic?  HW queue oversubscription (GPU_MAX_HW_QUEUES) / plan-internal streams
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3z; mkdir -p $O
{
timeout 120 python scripts/halves_probe.py 256 128 10
GPU_MAX_HW_QUEUES=8 timeout 120 python scripts/halves_probe.py 256 128 10
GPU_MAX_HW_QUEUES=16 timeout 120 python scripts/halves_probe.py 256 128 10
timeout 120 python scripts/halves_probe.py 256 128 10 single_stream=1
GPU_MAX_HW_QUEUES=16 timeout 120 python scripts/halves_probe.py 128 64 4
} 2>&1 | grep -v amdgpu.ids | tee $O/halves_probe.log
