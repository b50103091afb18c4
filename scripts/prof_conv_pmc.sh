#!/bin/bash
# usage: scripts/prof_conv_pmc.sh <tag>  -- SQ counters of the conv micro-benchmark (one pass, 8 SQ slots)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d gpurun_out/convpmc_$1 --output-format csv -- python3 scripts/bench_conv.py 0,1 > gpurun_out/convpmc_$1.log 2>&1
tail -5 gpurun_out/convpmc_$1.log
