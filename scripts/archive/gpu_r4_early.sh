# round 4: decoder weight gradients flushed early under the decoder's backward chain, as launches that occupy only part of the chip
# (avc_tuning.dec_wgrad_flush / dec_wgrad_wgs) -- A/B against the held schedule on ONE box
OUT=gpurun_out/${1:-r4early}; mkdir -p $OUT
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config2 --no-profile $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-52s step %.4f ms' % ('$1', d['ms_per_step']))"; }
for i in 1 2; do
run "held (default)" ""
for n in 2 4 6; do for w in 64 96 128 160 192; do
run "dec_wgrad_flush=$n dec_wgrad_wgs=$w" "--tune dec_wgrad_flush=$n --tune dec_wgrad_wgs=$w"
done; done
run "held (default)" ""
done 2>&1 | tee $OUT/early.log
