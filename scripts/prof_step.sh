#!/bin/bash
# usage: scripts/prof_step.sh <tag>   (run on the GPU box via gpurun)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$1 --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$1.log 2>&1
tail -1 gpurun_out/bench_$1.log | cut -c1-160
