#!/bin/bash
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie $@"
rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE --output-format csv -d $OUT/m1 -- python3 bench.py $A > $OUT/m1.log 2>&1
# (a pass with TA_*_STALLED_* / TA_FLAT_READ_WAVEFRONTS counters aborted inside rocprofv3 and hung the run on this pool: not collected)
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/m3 -- python3 bench.py $A > $OUT/m3.log 2>&1
python3 tools/pmc_summary.py $OUT | grep -A8 "rt_tile_kernel<" | grep -v "^--"
