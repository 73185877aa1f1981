# the GPU test suite as the driver runs it (-m gpu) + its slow half (-m "gpu and slow": the full-count property examples), with durations.
# $1 = output name
OUT=gpurun_out/${1:-suite}; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu --durations=40 -p no:cacheprovider ) > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt
if [ "$2" = "slow" ]; then
( time timeout 1500 python -m pytest tests -q -m "gpu and slow" --durations=10 -p no:cacheprovider ) > $OUT/pytest_gpu_slow.txt 2>&1; tail -n 4 $OUT/pytest_gpu_slow.txt
fi
