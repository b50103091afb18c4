#!/bin/bash
# usage: scripts/prof_pmc.sh <tag> [bench.py arguments, e.g. --card model_cards/vit_base_transformer_clip.yaml --batch 128]   (on the GPU box via gpurun)
# HBM traffic of every kernel of a C3 step: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (3 + 2 of the 4 TCC slots),
# kernel trace only beside the counters, the program itself after "--".
export TMPDIR=/tmp
set -e
tag=$1; shift
set -- "$tag" "$@"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$1_$c --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "${@:2}" > gpurun_out/pmc_$1_$c.log 2>&1
  tail -1 gpurun_out/pmc_$1_$c.log | cut -c1-120
  ls gpurun_out/pmc_$1_$c/*/ | head
done
