#!/bin/bash
# usage: scripts/prof_mfma.sh <tag> [bench.py arguments]   (on the GPU box via gpurun)
# MFMA utilisation per kernel from hardware counters: SQ_VALU_MFMA_BUSY_CYCLES (32 per v_mfma_f32_32x32x16_bf16, summed
# over the chip's 1024 SIMDs) against GRBM_GUI_ACTIVE (summed over the 8 XCDs); kernel trace only beside the counters.
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/mfma_$tag --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/mfma_$tag.log 2>&1
tail -1 gpurun_out/mfma_$tag.log | cut -c1-120
python3 scripts/mfma_summary.py $tag
