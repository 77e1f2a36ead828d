#!/bin/bash
# bench c5 (default engine) for each build variant, printing the launch configuration
for v in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$v" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  r=$(RT_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload c5 --steps 8 --warmup 1 --no-cpu-baseline --no-pcie 2>gpurun_out/verbose.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "[$v] c5: $r ($(grep -m1 '^\[rt\] engine' gpurun_out/verbose.err | sed 's/.*lds/lds/; s/  bvh.*//'))"
done
