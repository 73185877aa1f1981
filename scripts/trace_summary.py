"""Summarise a rocprofv3 kernel trace of bench.py: per-step span, union-busy time, per-kernel totals."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one step = everything behind the optimizer kernel of the step before it, up to and including its own (the weight images of a step are
# packed right behind the previous optimizer step since round 4)
idx = [i for i, r in enumerate(rows) if 'clip_adam_kernel' in r['Kernel_Name']]
step = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e in iv) - t0
tot = sum(e - s for s, e in iv)
print(f"kernels {len(step)}  span {span/1e3:.0f} us  union-busy {busy/1e3:.0f} us  sum {tot/1e3:.0f} us  streams {len(set(r['Stream_Id'] for r in step))}")
agg = collections.OrderedDict()
for r in step:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    a = agg.setdefault((name, g), [0, 0.0]); a[0] += 1; a[1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{v[1]:9.1f} us  n={v[0]:3d} avg={v[1]/v[0]:8.1f}  {k}")
