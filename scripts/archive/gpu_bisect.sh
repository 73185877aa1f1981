# Same-box A/B of several commits in ONE gpurun call (boxes differ by ~4 %): for each commit C beforehand
#   git worktree add _w_C C && (cd _w_C && ./adaptive_voice_conversion_amd/csrc/build.sh)      (_w_*/ is git-ignored; the built trees travel to the box)
# then list the trees below.  Round 4 used it to find a 4 % regression ("a code-generation trap").
run() { (cd $1 && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config2 $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes']; print('$2: step', round(d['ms_per_step'],4), 'wgrad', k['conv_wgrad']['ms_per_step'], 'in_bwd', k['instnorm_bwd']['ms_per_step'], 'pack', k.get('pack_weights',{}).get('ms_per_step'), 'adam', k.get('clip_adam',{}).get('ms_per_step'))"); }
for i in 1 2 3; do
run _w_0d72758 "0d72758               "
run . "HEAD                  " ""
done
