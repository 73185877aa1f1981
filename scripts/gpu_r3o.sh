#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3o
timeout 600 python scripts/host_bound_probe.py > gpurun_out/r3o/host_bound.log 2>&1
cat gpurun_out/r3o/host_bound.log
