"""Aggregates a rocprofv3 kernel trace by (kernel, grid, workgroup, LDS): count, mean / min duration.  usage: trace_by_shape.py <dir> [filter]"""
import csv, glob, sys, collections, re
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
f = [p for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)][0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    short = re.sub(r"\(.*", "", n)[:70]
    key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print(f"{k[0]:70s} grid=({k[1]},{k[2]}) wg={k[3]} lds={k[4]:>7s} n={len(v):4d} mean={sum(v)/len(v):8.1f}us min={min(v):8.1f}us")
print("total us", round(tot, 1))
