"""Per-stream timeline of one bench step from a rocprofv3 kernel trace: concurrency histogram + chronological list."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one step = everything behind the optimizer kernel of the step before it, up to and including its own (the weight images of a step are
# packed right behind the previous optimizer step since round 4)
idx = [i for i, r in enumerate(rows) if 'clip_adam_kernel' in r['Kernel_Name']]
step = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
streams = sorted(set(r['Stream_Id'] for r in step))
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']) - t0, 1)); ev.append((int(r['End_Timestamp']) - t0, -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = 0
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("concurrency (kernels in flight): " + "  ".join(f"{k}:{v/1e3:.0f}us" for k, v in sorted(hist.items())))
for s in streams:
    rs = [r for r in step if r['Stream_Id'] == s]
    b = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print(f"stream {s}: n={len(rs)} first {(int(rs[0]['Start_Timestamp'])-t0)/1e3:.0f} last_end {(max(int(r['End_Timestamp']) for r in rs)-t0)/1e3:.0f} busy {b/1e3:.0f} us")
for r in step:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  s{streams.index(r['Stream_Id'])}  {name} {g}")
