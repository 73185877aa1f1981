#!/bin/bash
# (the conv_stagger part is the record of a removed experiment: profiles/r03_stagger.log)
# round 3, call s: workgroup phase stagger (conv_stagger) on the conv micro-benchmark and the step; effective-clock probe (GRBM_GUI_ACTIVE)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $O; export TMPDIR=/tmp
for s in 0 12 28 -12 -28; do
  echo "## conv_stagger=$s" >> $O/stagger_micro.log
  timeout 120 python scripts/stagger_micro.py $s >> $O/stagger_micro.log 2>&1
done
cat $O/stagger_micro.log
EXTRA="--no-profile" bash scripts/gpu_tune.sh r3s/step default "conv_stagger=28" "conv_stagger=-28" "conv_stagger=-12" default > $O/stagger_step.log 2>&1
cat $O/stagger_step.log
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk -o clk -- python $GRAFT_REPO_ROOT/scripts/clock_probe.py > $O/clock_probe.out 2>&1)
python scripts/clock_probe_parse.py /tmp/clk $O/clock_probe.out > $O/clock_probe.log 2>&1; cat $O/clock_probe.log
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clkb -o clk -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $O/clock_bench.out 2>&1)
python scripts/clock_probe_parse.py /tmp/clkb > $O/clock_bench.log 2>&1; cat $O/clock_bench.log
head -3 $(find /tmp/clk -name "*counter_collection.csv" | head -1) > $O/cc_head.txt
