#!/bin/bash
# usage: scripts/prof_card.sh <tag> <card> <batch> [precision]   (run on the GPU box via gpurun)
# rocprofv3 kernel trace of scripts/step_card.py; per-step summary by scripts/prof_summary.py <tag>
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$1 --output-format csv -- python3 scripts/step_card.py $2 $3 ${4:-bf16-mixed} 3 > gpurun_out/step_$1.log 2>&1
tail -1 gpurun_out/step_$1.log | cut -c1-200
python3 scripts/prof_summary.py $1 45 > gpurun_out/${1}_step_kernel_summary.txt
cat gpurun_out/${1}_step_kernel_summary.txt
