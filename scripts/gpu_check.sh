mkdir -p gpurun_out && rm -rf gpurun_out/r1a && mkdir -p gpurun_out/r1a
export TMPDIR=/tmp
for f in test_ops_rowops test_ops_conv test_engine test_model; do
  timeout 900 python -m pytest tests/$f.py -m gpu -q --timeout 300 -s 2>&1 | tail -60 > gpurun_out/r1a/$f.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r1a/smoke.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --profile-json gpurun_out/r1a/prof_classes.json > gpurun_out/r1a/bench.log 2>&1
tail -3 gpurun_out/r1a/*.log
