"""Summarise a rocprofv3 kernel trace of bench.py: per-kernel totals for the last step + big conv launches."""
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
names = [r['Kernel_Name'] for r in rows]
sgd = [i for i, n in enumerate(names) if 'sgd_multi' in n]
step = rows[sgd[-2] + 1:sgd[-1] + 1]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = r['Kernel_Name'].split('(')[0][-60:]
    tot[k][0] += 1; tot[k][1] += dur(r)
wall = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3
print(f'step wall {wall/1e3:.2f} ms, kernels {len(step)}, sum {sum(v[1] for v in tot.values())/1e3:.2f} ms')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    print(f'{k:62s} n={v[0]:3d} {v[1]/1e3:7.3f} ms')
if len(sys.argv) > 3:
    for r in step:
        if sys.argv[3] in r['Kernel_Name'] and dur(r) > 30:
            print(f"{r['Kernel_Name'][:56]:56s} grid {r['Grid_Size_X']:>9s} wg {r['Workgroup_Size_X']:>4s} {dur(r):8.1f} us")
