"""Per-queue busy time, span and largest gaps of the last optimisation step in a rocprofv3 kernel trace."""
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
names = [r['Kernel_Name'] for r in rows]
sgd = [i for i, n in enumerate(names) if 'sgd_multi' in n]
step = rows[sgd[-2] + 1:sgd[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
by = collections.defaultdict(list)
for r in step: by[r['Queue_Id']].append(r)
for q, rs in by.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
    print(f'queue {q}: {len(rs)} kernels, busy {busy:.2f} ms, span {(int(rs[0]["Start_Timestamp"])-t0)/1e6:.2f}..{(int(rs[-1]["End_Timestamp"])-t0)/1e6:.2f} ms')
mq = max(by, key=lambda q: sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in by[q]))
rs = by[mq]
gaps = [((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3, a['Kernel_Name'][:36], b['Kernel_Name'][:36]) for a, b in zip(rs, rs[1:])]
print(f'main queue {mq}: total gap {sum(g for g, _, _ in gaps)/1e3:.2f} ms; gaps > 3 us: {sum(1 for g,_,_ in gaps if g > 3)}')
for g in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]: print('  %.1f us  %s -> %s' % g)
