# refresh of the build-locked summaries after a host-side one-liner in conv_gemm.hip (the fused epilogues keep their tile under tile12_wgs):
# targeted GPU tests, SQ + PMC passes, the default bench line
OUT=gpurun_out/${1:-r5final3}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_in_fuse.py tests/test_conv_walk.py tests/test_engine.py tests/test_ops_conv.py -q -m gpu 2>&1 | tail -2 | tee $OUT/tests.txt
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-config2 --single-stream"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sqA -o pmc -- $B > $GRAFT_REPO_ROOT/$OUT/sqA.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/sqB -o pmc -- $B > $GRAFT_REPO_ROOT/$OUT/sqB.log 2>&1)
AVC_GIT_HEAD=${2:-unknown} python scripts/sq_step_summary.py $OUT/sq_step.json /tmp/sqA /tmp/sqB | tee $OUT/sq_classes.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-config2 > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1)
done
AVC_GIT_HEAD=${2:-unknown} python scripts/pmc_summary.py $OUT/pmc_fetch_write_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
cp $OUT/pmc_fetch_write_summary.json profiles/r05_pmc_fetch_write_summary.json; cp $OUT/sq_step.json profiles/r05_sq_step.json
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 200 $OUT/bench.json
