#!/bin/bash
# Run on the GPU box (via gpurun): round-4 profile of a bench workload.
#   * kernel trace with the DRIVER's exact command line (python3 bench.py --gpus 1 --steps 20 --warmup 5 [args]): the
#     trace's average launch of the timed kernel must reproduce the line's ms_per_step
#   * PMC passes (separate runs, never combined with tracing) with a short command: counters are per launch
# usage: tools/prof4.sh <tag> [bench args...]
set -e
TAG=${1:-c3}; shift || true
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
DRV="--gpus 1 --steps 20 --warmup 5 $@"
echo "python3 bench.py $DRV" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $DRV > $OUT/trace.log 2>&1
echo "trace done"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --no-others --no-frame $@"
echo "python3 bench.py $ARGS" > $OUT/cmd_pmc.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -- python3 bench.py $ARGS > $OUT/pmc1.log 2>&1
echo "pmc1 done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc2 -- python3 bench.py $ARGS > $OUT/pmc2.log 2>&1
echo "pmc2 done"
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc5 -- python3 bench.py $ARGS > $OUT/pmc5.log 2>&1 || echo "pmc5 failed"
echo "pmc5 done"
# the profiler's DERIVED busy / utilisation metrics (round-3 verdict, item 6): what fraction of the time the vector pipe is processing
# instructions, how many of its lanes are active, LDS bank-conflict and memory-unit stall shares
rocprofv3 --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $OUT/pmc6 -- python3 bench.py $ARGS > $OUT/pmc6.log 2>&1 || echo "pmc6 failed"
rocprofv3 --pmc LDSBankConflict MemUnitStalled --output-format csv -d $OUT/pmc7 -- python3 bench.py $ARGS > $OUT/pmc7.log 2>&1 || echo "pmc7 failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc8 -- python3 bench.py $ARGS > $OUT/pmc8.log 2>&1 || echo "pmc8 failed"
echo "pmc6-8 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 bench.py $ARGS > $OUT/pmc4.log 2>&1
echo "pmc3/4 done"
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
