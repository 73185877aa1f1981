mkdir -p gpurun_out/r1d; export TMPDIR=/tmp
timeout 600 python scripts/diag_parity.py 2>&1 | grep -v amdgpu > gpurun_out/r1d/diag.log; cat gpurun_out/r1d/diag.log
