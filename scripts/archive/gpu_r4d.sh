# round 4: per-launch view of the weight-gradient class (single-stream step under rocprofv3 --kernel-trace)
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4d}; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-config2 --single-stream > $OUT/rocprof_single.log 2>&1)
python - $OUT/rocprof_single/trace_kernel_trace.csv > $OUT/wgrad_launches.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step only: find the last clip_adam and the one before it
idx = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
lo, hi = idx[-2], idx[-1]
for r in rows[lo + 1:hi + 1]:
    n = r["Kernel_Name"]
    if "wgrad" in n or "reduce" in n:
        print("%8.1f us  grid %7s  lds %6s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"), n[:90]))
PY
cat $OUT/wgrad_launches.txt
cp $OUT/rocprof_single/trace_kernel_stats.csv $OUT/kernel_stats_single_stream.csv 2>/dev/null
