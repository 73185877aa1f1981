# round 5, call A: SQ counters over the WHOLE single-stream bench step of the shipped build (every conv / weight-gradient instance of
# the step, not one layer) + the default bench line of this box.  $1 = output name, $2 = git head
OUT=gpurun_out/${1:-r5a}; mkdir -p $OUT; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-config2 --single-stream"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sqA -o pmc -- $B > $GRAFT_REPO_ROOT/$OUT/sqA.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/sqB -o pmc -- $B > $GRAFT_REPO_ROOT/$OUT/sqB.log 2>&1)
AVC_GIT_HEAD=${2:-unknown} python scripts/sq_step_summary.py $OUT/sq_step.json /tmp/sqA /tmp/sqB
for d in sqA sqB; do f=$(find /tmp/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $OUT/${d}_counter_collection.csv.gz; done
tail -3 $OUT/sqA.log | cut -c1-300
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
