#!/bin/bash
# round 3, call af (last seconds): scripts/probe/instnorm_clamped_loads.patch as an alternative library (AVC_HIP_LIB) -- T = 1024 rows (NV = 4 instances) and the headline shape
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3af; mkdir -p $O
C=$GRAFT_REPO_ROOT/adaptive_voice_conversion_amd/csrc
b() { lib=$1; shift; AVC_HIP_LIB=$lib timeout 20 python bench.py --no-cpu-baseline --no-profile "$@" 2>> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $lib) $*', '->', round(d['ms_per_step'], 4), 'ms', d['config']['final_losses'])" | tee -a $O/ab.log; }
b $C/libavc_hip.so --batch 64 --frames 1024 --steps 12 --warmup 3
b $C/libavc_hip_in.so --batch 64 --frames 1024 --steps 12 --warmup 3
b $C/libavc_hip_in.so --steps 30 --warmup 8
b $C/libavc_hip.so --batch 64 --frames 1024 --steps 12 --warmup 3
b $C/libavc_hip_in.so --batch 64 --frames 1024 --steps 12 --warmup 3
