#!/bin/bash
# round 3, call v: multi-stream timeline of the fp32 step at B = 256 (span / union-busy / sum, per-stream occupancy)
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_trace.sh r3v 0
cat gpurun_out/r3v/trace_summary.txt | head -50
head -60 gpurun_out/r3v/trace_timeline.txt
