"""usage: clock_probe_parse.py <rocprofv3 output dir> [<stdout log of clock_probe.py>]
Per group of dispatches between two marker kernels: mean duration, mean GRBM_GUI_ACTIVE, effective clock (MHz).
Without a log (a pmc pass over bench.py): per kernel name."""
import csv, glob, sys, collections
d = sys.argv[1]
cc = [f for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)]
kt = [f for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)]
rows = []
for f in cc:
    rows += list(csv.DictReader(open(f)))
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r.get("Dispatch_Id")] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = [r for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
def ns(r):
    if r.get("Start_Timestamp") and r.get("End_Timestamp"):
        return int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    s, e = dur[r["Dispatch_Id"]]
    return e - s
names = []
if len(sys.argv) > 2:
    names = [l.split("group:", 1)[1].strip() for l in open(sys.argv[2]) if l.startswith("group:")]
    groups, cur = [], None
    for r in rows:
        if "elementwise" in r["Kernel_Name"]:
            cur = []
            groups.append(cur)
        elif cur is not None and "conv_gemm" in r["Kernel_Name"]:
            cur.append(r)
    groups = [g for g in groups if g]
    for i, g in enumerate(groups):
        g = g[5:]   # (warm-up)
        t = sum(ns(r) for r in g) / len(g)
        c = sum(float(r["Counter_Value"]) for r in g) / len(g)
        print(f"{names[i] if i < len(names) else i:60s} n={len(g):3d}  {t/1e3:7.1f} us  {c:10.0f} cycles  {c/t*1e3:6.0f} MHz (counter / duration; / 8 if the counter sums the XCDs: {c/t*1e3/8:6.0f})")
else:
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows:
        a = agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]]
        a[0] += 1; a[1] += ns(r); a[2] += float(r["Counter_Value"])
    tt = sum(a[1] for a in agg.values()); tc = sum(a[2] for a in agg.values())
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{a[1]/1e3:10.1f} us  n={a[0]:4d}  {a[2]/a[1]*1e3:6.0f} MHz  {k}")
    print(f"all kernels: {tc/tt*1e3:6.0f} MHz (time-weighted; / 8 if summed over XCDs: {tc/tt*1e3/8:6.0f})")
