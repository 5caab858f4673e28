"""Prints the kernels of the LAST call of a rocprofv3 --kernel-trace run (split at a marker kernel) with durations.
   python profiles/scratch/trace_one_call.py <trace dir> <marker substring>"""
import csv, glob, sys
t = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
agg = {}
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][-44:]
    key = (name, r.get('Grid_Size_X') or r.get('Grid_Size'))
    agg.setdefault(key, []).append((e - s) / 1e3)
for (name, grid), v in agg.items():
    print(f"{name:46s} grid {grid:>8s} x{len(v):3d}  avg {sum(v)/len(v):7.2f} us  min {min(v):7.2f}  total {sum(v):8.1f}")
print('call span us', (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3, 'kernels', b - a)
