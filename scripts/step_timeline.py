"""Timeline of one traced step (rocprofv3 --kernel-trace CSV of scripts/prof_step.sh): per phase of the step, how long each
queue is busy, and the intervals in which no big kernel (>= 256 workgroups) runs anywhere.
usage: python scripts/step_timeline.py <tag>"""
import csv, glob, os, sys
tag = sys.argv[1]
# (the merged gpurun_out/ keeps the traces of earlier runs: take the newest)
rows = list(csv.DictReader(open(max(glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_trace.csv'), key=os.path.getmtime))))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'sgd_multi' in r['Kernel_Name']]
seg = rows[sgd[-2] + 1:sgd[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
ev = []
for r in seg:
    wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
    ev.append(((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r['Queue_Id'], wgs, r['Kernel_Name'][:48]))
end = max(e[1] for e in ev)
print(f'step span {end:.0f} us, {len(ev)} kernels')
# busy time per queue
for q in sorted({e[2] for e in ev}):
    iv = sorted((e[0], e[1]) for e in ev if e[2] == q)
    busy, cur_s, cur_e = 0.0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f'queue {q}: {sum(1 for e in ev if e[2] == q)} kernels, busy {busy:.0f} us')
# intervals without a big kernel
big = sorted((e[0], e[1]) for e in ev if e[3] >= 256)
gaps, cur = [], 0.0
for s, e in big:
    if s > cur + 1.0: gaps.append((cur, s))
    cur = max(cur, e)
if cur < end: gaps.append((cur, end))
tot = sum(b - a for a, b in gaps)
print(f'no kernel of >= 256 workgroups running: {tot:.0f} us in {len(gaps)} intervals; the longest:')
for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:14]:
    inside = [e[4] for e in ev if e[0] < b and e[1] > a]
    print(f'  {a:8.0f} .. {b:8.0f}  ({b - a:5.0f} us)  running: {", ".join(sorted(set(inside)))[:150]}')
