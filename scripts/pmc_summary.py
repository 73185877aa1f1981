"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: mean counter value per launch."""
import csv, sys, collections, glob, json
out = {}
for path in sys.argv[2:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            key = (name, r["Counter_Name"], r.get("Grid_Size", ""))
            a = agg[key]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for (name, cnt, grid), (n, tot) in agg.items():
            out.setdefault(name, {}).setdefault(grid, {})[cnt] = {"launches": n, "mean_per_launch": tot / n}
import os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import build_fingerprint
    out["_meta"] = {"build_fingerprint": build_fingerprint(), "git_head": os.environ.get("AVC_GIT_HEAD", "unknown")}
except Exception as e:  # pragma: no cover
    out["_meta"] = {"error": str(e)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("kernels:", len(out))
