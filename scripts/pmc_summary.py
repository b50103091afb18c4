"""HBM traffic per kernel from the two rocprofv3 PMC passes of scripts/prof_pmc.sh (FETCH_SIZE, WRITE_SIZE).

Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are in KiB;
FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) streaming reads at 64 B, so read bytes = 2 x FETCH_SIZE x 1024;
WRITE_SIZE x 1024 is exact for 16-B/lane stores and float atomics.  Writes profiles/<tag>_pmc_traffic.{json,md};
bench.py reports the per-launch figure of the dominant kernel as roofline.traffic.
usage: python scripts/pmc_summary.py <tag>"""
import collections, csv, glob, json, re, sys
tag = sys.argv[1]


def last_step(counter):
    f = glob.glob(f'gpurun_out/pmc_{tag}_{counter}/*/*counter_collection.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    sgd = [i for i, r in enumerate(rows) if 'sgd_multi' in r['Kernel_Name']]
    return rows[sgd[-2] + 1:sgd[-1] + 1]


def short(name):
    m = re.match(r'(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)', name)
    return m.group(1) if m else name[:50]


agg = collections.defaultdict(lambda: {'launches': 0, 'fetch_kib': 0.0, 'write_kib': 0.0})
for r in last_step('FETCH_SIZE'):
    a = agg[short(r['Kernel_Name'])]
    a['launches'] += 1
    a['fetch_kib'] += float(r['Counter_Value'])
for r in last_step('WRITE_SIZE'):
    agg[short(r['Kernel_Name'])]['write_kib'] += float(r['Counter_Value'])
out = {}
for k, a in agg.items():
    rd, wr = 2.0 * a['fetch_kib'] * 1024, a['write_kib'] * 1024
    out[k] = {'launches_per_step': a['launches'], 'read_bytes_per_step': rd, 'write_bytes_per_step': wr,
              'hbm_bytes_per_launch': (rd + wr) / a['launches']}


def family(pred):
    ks = [k for k in out if pred(k)]
    n = sum(out[k]['launches_per_step'] for k in ks)
    b = sum(out[k]['read_bytes_per_step'] + out[k]['write_bytes_per_step'] for k in ks)
    return {'launches_per_step': n, 'hbm_bytes_per_launch': b / max(n, 1), 'hbm_bytes_per_step': b, 'kernels': ks}


def win(k, dgrad):
    """conv_win_kernel<WM, WN, TM, TN, STAGES, DGRAD, ...> / conv_win_persist_kernel<DGRAD> / conv_win_l1_kernel<DGRAD, ...>"""
    m = re.match(r'conv_win_kernel<([^>]*)>', k)
    if m:
        return m.group(1).split(',')[5].strip() == ('true' if dgrad else 'false')
    m = re.match(r'conv_win_persist_kernel<([^>]*)>', k)
    if m:
        return m.group(1).strip() == ('true' if dgrad else 'false')
    m = re.match(r'conv_win_l1_kernel<([^>]*)>', k)      # <DGRAD, PROBE, ADD, BNB>
    return bool(m) and m.group(1).split(',')[0].strip() == ('true' if dgrad else 'false')


families = {'win_fwd': family(lambda k: win(k, False)), 'win_dgrad': family(lambda k: win(k, True)),
            'win_fwd_dgrad': family(lambda k: win(k, False) or win(k, True)),
            'dma_fwd_dgrad': family(lambda k: k.startswith('conv_igemm_dma_kernel')),
            'wgrad_win': family(lambda k: k.startswith('conv_wgrad_win_kernel')),
            'wgrad_dma': family(lambda k: k.startswith('conv_wgrad_dma_kernel')),
            'batchnorm_pool': family(lambda k: k.startswith(('bn_', '_Z', 'pool_', 'global_')) or 'bn_' in k),
            'stem_fused': family(lambda k: k.startswith('stemf_'))}
summary = {'tag': tag, 'workload': (sys.argv[2] if len(sys.argv) > 2 else 'bench.py C3 step, batch 512, one MI355X'), 'correction': 'read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB',
           'families': families,
           'launches_per_step': sum(v['launches_per_step'] for v in out.values()),
           'whole_step_hbm_bytes': sum(v['read_bytes_per_step'] + v['write_bytes_per_step'] for v in out.values()),
           'kernels': out}
json.dump(summary, open(f'profiles/{tag}_pmc_traffic.json', 'w'), indent=1)
with open(f'profiles/{tag}_pmc_traffic.md', 'w') as f:
    f.write(f'# HBM traffic per kernel, one step of {summary["workload"]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), {tag}\n\n')
    f.write('read = 2 x FETCH_SIZE KiB (gfx950 wide-read correction), write = WRITE_SIZE KiB\n\n')
    f.write('| kernel | launches/step | read MB/step | write MB/step | MB/launch |\n|---|---|---|---|---|\n')
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]['read_bytes_per_step'] + kv[1]['write_bytes_per_step'])):
        f.write(f"| {k} | {v['launches_per_step']} | {v['read_bytes_per_step']/1e6:.1f} | {v['write_bytes_per_step']/1e6:.1f} | "
                f"{v['hbm_bytes_per_launch']/1e6:.2f} |\n")
    f.write(f"\nwhole step: {summary['whole_step_hbm_bytes']/1e9:.2f} GB in {summary['launches_per_step']} launches\n\n")
    f.write('| family | launches/step | MB/launch | GB/step |\n|---|---|---|---|\n')
    for k, v in families.items():
        f.write(f"| {k} | {v['launches_per_step']} | {v['hbm_bytes_per_launch']/1e6:.1f} | {v['hbm_bytes_per_step']/1e9:.2f} |\n")
print(open(f'profiles/{tag}_pmc_traffic.md').read())
