# per-kernel-instance rocprofv3 --stats of two builds (a worktree and the main tree) in ONE gpurun call; see gpu_bisect.sh
export TMPDIR=/tmp
prof() { (cd /tmp && rm -rf /tmp/rp_$2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$2 -o trace -- python $GRAFT_REPO_ROOT/$1/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --single-stream $3 > /dev/null 2>&1); python - /tmp/rp_$2/trace_kernel_stats.csv "$2" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("==", sys.argv[2])
for r in rows:
    if 'wgrad' in r['Name'] or 'reduce' in r['Name']:
        print(f"  {r['Name'][5:75]:72s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  total/step {float(r['TotalDurationNs'])/1e3/14:8.1f}")
PY
}
prof _w_7903551 old ""
prof . head_cw8_0 "--tune wgrad_cw8=0"
