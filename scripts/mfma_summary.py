"""MFMA utilisation per kernel of the last optimisation step of a `scripts/prof_mfma.sh` run.

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the matrix pipes' busy cycles (32 per
32x32x16 bf16 MFMA, 64 FLOP... i.e. 32768 FLOP per 32 busy cycles per SIMD) over the cycles the chip was active during
the dispatch (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; MI355X_MICROARCH.md, DVFS note).  `tflops_from_counter` =
busy cycles x 1024 FLOP / duration: the bf16 MFMA rate the counter implies, to be compared with the algorithmic rate.
Writes profiles/<tag>_mfma_util.{json,md}.   usage: python scripts/mfma_summary.py <tag>"""
import collections, csv, glob, json, re, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/mfma_{tag}/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
trace = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'gpurun_out/mfma_{tag}/*/*kernel_trace.csv')[0]))}
by = collections.OrderedDict()
for r in rows:
    d = by.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name']})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
disp = list(by.items())
sgd = [i for i, (_, d) in enumerate(disp) if 'sgd_multi' in d['name']]
step = disp[sgd[-2] + 1:sgd[-1] + 1]


def short(name):
    m = re.match(r'(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)', name)
    return m.group(1) if m else name[:50]


agg = collections.defaultdict(lambda: {'launches': 0, 'busy': 0.0, 'active': 0.0, 'us': 0.0})
for did, d in step:
    a = agg[short(d['name'])]
    a['launches'] += 1
    a['busy'] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    a['active'] += d.get('GRBM_GUI_ACTIVE', 0.0)
    t = trace.get(did)
    if t:
        a['us'] += (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3
out = {}
for k, a in agg.items():
    if a['busy'] <= 0:
        continue
    cyc = a['active'] / 8.0
    out[k] = {'launches_per_step': a['launches'], 'us_per_step': round(a['us'], 1),
              'mfma_busy_cycles': a['busy'], 'chip_cycles': cyc,
              'mfma_utilisation': round(a['busy'] / (1024.0 * cyc), 4) if cyc else None,
              # (GRBM_GUI_ACTIVE / duration reads high on short dispatches -- MI355X_MICROARCH.md, DVFS note: valid from ~0.3 ms;
              #  no clock is reported for kernels whose launches average under 100 us)
              'clock_ghz': round(cyc / (a['us'] * 1e3), 3) if a['us'] and a['us'] / a['launches'] >= 100.0 else None,
              'tflops_from_counter': round(a['busy'] * 1024.0 / (a['us'] * 1e-6) / 1e12, 1) if a['us'] else None}
json.dump(out, open(f'profiles/{tag}_mfma_util.json', 'w'), indent=1)
with open(f'profiles/{tag}_mfma_util.md', 'w') as fh:
    fh.write('| kernel | launches/step | us/step | MFMA utilisation | clock GHz | bf16 TFLOP/s implied |\n|---|---|---|---|---|---|\n')
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['us_per_step']):
        fh.write(f"| `{k}` | {v['launches_per_step']} | {v['us_per_step']} | {v['mfma_utilisation']} | {v['clock_ghz'] if v['clock_ghz'] is not None else '--'} | {v['tflops_from_counter']} |\n")
print(open(f'profiles/{tag}_mfma_util.md').read())
