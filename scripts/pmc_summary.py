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
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("kernels:", len(out))
