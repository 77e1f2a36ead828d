#!/bin/bash
# busy/issue counters for the dominant kernel: tools/prof_pmc3.sh TAG [bench args]
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie $@ > $OUT/p.log 2>&1
python3 tools/pmc_summary.py $OUT | grep -A9 "rt_tile_kernel<" | grep -v "^--"
