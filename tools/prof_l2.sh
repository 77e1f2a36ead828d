#!/bin/bash
# L2 (TCC) hit / miss counters of the tile kernel: tools/prof_l2.sh TAG [bench args]
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/l2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie $@ > $OUT/l2.log 2>&1
python3 tools/pmc_summary.py $OUT | grep -A4 "rt_tile_kernel<" | grep -v "^--"
