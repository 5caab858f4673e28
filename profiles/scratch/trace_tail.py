"""Aggregates the last N kernels of a rocprofv3 --kernel-trace csv by name.  python profiles/scratch/trace_tail.py <dir> [N]"""
import csv, glob, sys
t = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))[-n:]
agg = {}
for r in rows:
    name = r['Kernel_Name'].split('(')[0][-48:]
    agg.setdefault((name, r.get('Grid_Size_X') or r.get('Grid_Size')), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:50s} grid {grid:>8s} x{len(v):4d}  avg {sum(v)/len(v):7.2f} us  min {min(v):7.2f}  total {sum(v):8.1f}")
print('span us', (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3, 'kernels', len(rows))
