#!/bin/bash
# HBM traffic counters only (two separate PMC passes), bench args passed through
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie $@"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 bench.py $ARGS > $OUT/pmc4.log 2>&1
python3 tools/pmc_summary.py $OUT | grep -A2 "rt_tile_kernel" | grep "SIZE"
