# round 4: weight-gradient launches with more than one workgroup per CU (avc_tuning.wgrad_batch_wgs): two 8-wave workgroups per CU = two consumer waves per SIMD from different workgroups
for w in 256 512 384 256 512; do
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config2 --tune wgrad_batch_wgs=$w | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes']; print('wgs $w: step', round(d['ms_per_step'],4), 'wgrad', k['conv_wgrad']['ms_per_step'], 'reduce', k['slab_reduce']['ms_per_step'])"
done
