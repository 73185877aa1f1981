mkdir -p gpurun_out/r1d; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine.py -m gpu -q --timeout 300 -s 2>&1 | grep -E "grad rel-L2|passed|failed|FAILED|Error|assert" > gpurun_out/r1d/test_engine.log
cat gpurun_out/r1d/test_engine.log
