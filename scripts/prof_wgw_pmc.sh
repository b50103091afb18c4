#!/bin/bash
# usage: scripts/prof_wgw_pmc.sh <tag> H C [target] [mode] -- LDS / MFMA counters of the sliding-window weight gradient
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/wgwpmc_$tag --output-format csv -- python3 scripts/wgw_one.py "$@" > gpurun_out/wgwpmc_$tag.log 2>&1
tail -2 gpurun_out/wgwpmc_$tag.log
python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/wgwpmc_$tag/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:60]
        if 'wgrad' not in k and 'wgw' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f'   {c:28s} {v / n[(k, c)]:16.0f} per launch')
PY
