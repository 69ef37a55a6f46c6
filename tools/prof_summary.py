"""Summarise a rocprofv3 kernel trace: median duration per kernel name (optionally per consecutive run)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:60]
if len(sys.argv) > 2 and sys.argv[2] == 'runs':
    i = 0
    while i < len(rows):
        j = i
        while j < len(rows) and short(rows[j]['Kernel_Name']) == short(rows[i]['Kernel_Name']): j += 1
        v = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[i:j])
        if 'mlp' in rows[i]['Kernel_Name'] or 'sk_' in rows[i]['Kernel_Name']:
            print(f"{short(rows[i]['Kernel_Name']):50s} n={len(v):4d} median={v[len(v)//2]:8.2f} us")
        i = j
else:
    g = collections.defaultdict(list)
    for r in rows: g[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = sum(sum(v) for v in g.values())
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:25]:
        v.sort(); print(f"{k:60s} n={len(v):5d} median={v[len(v)//2]:8.2f} us  total%={100*sum(v)/tot:5.1f}")
