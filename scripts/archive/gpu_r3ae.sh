#!/bin/bash
# round 3, call ae (the last seconds of the GPU budget): the staged weight-gradient probe (scripts/probe/conv_wgrad_stages.hip) as ALTERNATIVE libraries
# (AVC_HIP_LIB; the product library and its sources are untouched) -- A/B of the step; the final losses must equal the product's digit for digit
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3ae; mkdir -p $O
C=$GRAFT_REPO_ROOT/adaptive_voice_conversion_amd/csrc
b() { lib=$1; shift; AVC_HIP_LIB=$lib timeout 30 python bench.py --no-cpu-baseline --no-profile --steps 30 --warmup 8 "$@" 2>> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $lib) $*', '->', round(d['ms_per_step'], 4), 'ms', d['config']['final_losses'])" | tee -a $O/ab.log; }
b $C/libavc_hip.so
b $C/libavc_hip_wg3.so
b $C/libavc_hip_wg4.so
b $C/libavc_hip.so
b $C/libavc_hip_wg3.so
b $C/libavc_hip_wg3.so --dtype bf16
b $C/libavc_hip.so --dtype bf16
