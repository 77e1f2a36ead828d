#!/bin/bash
# L2 (TCC) requests, hits and misses of one workload (gpurun): tools/pmc_l2.sh <workload>
set -e
W=$1; shift || true
OUT=gpurun_out/pmcl2_$W; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --no-others --no-frame --workload $W"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1 || echo "TCC pass failed"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1 || echo "TCP pass failed"
python3 tools/pmc_summary.py $OUT 2>&1 | grep -A6 "rt_tile_kernel<[0-9], false, [0-9]*, false" | grep -v "^--"
