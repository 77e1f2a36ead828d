#!/bin/bash
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie $@ > $OUT/pmc1.log 2>&1
python3 tools/pmc_summary.py $OUT | grep -A9 "rt_tile_kernel<" | grep -v "^--"
