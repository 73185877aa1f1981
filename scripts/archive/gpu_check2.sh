mkdir -p gpurun_out/r1b
export TMPDIR=/tmp
timeout 600 python scripts/diag_parity.py > gpurun_out/r1b/diag.log 2>&1
