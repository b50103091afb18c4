"""The forward / backward junction of one traced C3 step (rocprofv3 --kernel-trace CSV of scripts/prof_step.sh): kernels of
the image chain's queue between the global average pool and the first data gradient of layer4, and the span they cover.
usage: python scripts/junction_span.py <tag>"""
import csv, glob, os, sys
f = max(glob.glob(f'gpurun_out/prof_{sys.argv[1]}/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'sgd_multi' in r['Kernel_Name']]
seg = rows[sgd[-2] + 1:sgd[-1] + 1]
a = [i for i, r in enumerate(seg) if 'global_avgpool_fwd' in r['Kernel_Name']][0]
q = seg[a]['Queue_Id']
chain = [r for r in seg[a:] if r['Queue_Id'] == q]
b = [i for i, r in enumerate(chain) if 'conv_win_kernel' in r['Kernel_Name']][0]
t0 = int(chain[0]['Start_Timestamp'])
prev = t0
for r in chain[:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f'{(s - t0) / 1e3:8.1f} +{(s - prev) / 1e3:5.1f} gap  {(e - s) / 1e3:6.1f} us  {r["Kernel_Name"][:80]}')
    prev = e
print(f'junction span {(int(chain[b]["Start_Timestamp"]) - t0) / 1e3:.1f} us, {b} kernels')
