"""Launches of one training step from a rocprofv3 kernel trace: python scratch/launch_inventory.py trace.csv"""
import csv, collections, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'multi_tensor_apply' in r['Kernel_Name'] or 'adam_kernel' in r['Kernel_Name']]
groups = []
for i in idx:
    if not groups or i - groups[-1][-1] > 50:
        groups.append([i])
    else:
        groups[-1].append(i)
a, b = groups[6][-1] + 1, groups[7][-1] + 1
seg = tr[a:b]
print('launches in step', len(seg), 'span ms', (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6)
c, d = collections.Counter(), collections.Counter()
for r in seg:
    n = r['Kernel_Name'].split('(')[0][-60:]
    c[n] += 1
    d[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, k in c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    print(f"{k:4d} {d[n]/1e3:8.1f} us  {n}")
print('sum of durations ms', sum(d.values()) / 1e6)
