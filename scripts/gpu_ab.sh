export TMPDIR=/tmp
for v in 0 1; do echo "AVC_IN_VARIANT=$v"; AVC_IN_VARIANT=$v python scripts/in_micro.py 2>&1 | grep -v amdgpu; done
AVC_IN_VARIANT=1 timeout 300 python -m pytest tests/test_ops_rowops.py -m gpu -q --timeout 300 2>&1 | tail -1
