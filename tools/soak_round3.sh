#!/bin/bash
# Long soaks of the round-3 build (gpurun): fuzz beyond the suite's counts, 10^6 spheres, the four culled-walk soaks.
mkdir -p gpurun_out/r3
( RT_FUZZ_CASES=600 RT_FUZZ_BIG=40 RT_FUZZ_MIXED=24 timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r3/soak_fuzz.log 2>&1; tail -2 gpurun_out/r3/soak_fuzz.log ) 
( timeout -k 10 200 python3 tools/cull_soak.py > gpurun_out/r3/soak_cull.log 2>&1; tail -1 gpurun_out/r3/soak_cull.log )
( timeout -k 10 200 python3 tools/tricull_soak.py > gpurun_out/r3/soak_tricull.log 2>&1; tail -1 gpurun_out/r3/soak_tricull.log )
( timeout -k 10 120 python3 tools/grazing_soak.py 60 > gpurun_out/r3/soak_grazing.log 2>&1; tail -1 gpurun_out/r3/soak_grazing.log )
( timeout -k 10 120 python3 tools/sliver_soak.py > gpurun_out/r3/soak_sliver.log 2>&1; tail -1 gpurun_out/r3/soak_sliver.log )
( timeout -k 10 200 python3 tools/million.py > gpurun_out/r3/soak_million.log 2>&1; tail -1 gpurun_out/r3/soak_million.log )
