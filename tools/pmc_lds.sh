#!/bin/bash
# LDS / VALU counters of one library variant (gpurun): tools/pmc_lds.sh <variant|default> [bench args]
set -e
N=$1; shift || true
V=$N; [ "$N" = default ] && V=""
OUT=gpurun_out/pmc_$N; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RT_LIB_VARIANT=$V
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --no-others --no-frame $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1
python3 tools/pmc_summary.py $OUT 2>&1 | grep -A9 "rt_tile_kernel<[0-9], false, [0-9]*, false" | grep -v "^--" > $OUT/summary.txt
cat $OUT/summary.txt
