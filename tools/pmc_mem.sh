#!/bin/bash
# L2-miss traffic and wait shares of one workload (gpurun): tools/pmc_mem.sh <workload> [env assignments...]
set -e
W=$1; shift || true
OUT=gpurun_out/pmcmem_$W; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp "$@"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --no-others --no-frame --workload $W"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 bench.py $ARGS > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 bench.py $ARGS > $OUT/w.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $OUT/s -- python3 bench.py $ARGS > $OUT/s.log 2>&1
python3 tools/pmc_summary.py $OUT 2>&1 | grep -A8 "rt_tile_kernel<[0-9], false, [0-9]*, false" | grep -v "^--"
