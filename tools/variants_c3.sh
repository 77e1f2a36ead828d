#!/bin/bash
# bench c3 / c4 (default engine) for each build variant, printing the launch configuration once
for v in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$v" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  out="[$v]"
  for w in c3 c4; do
    r=$(RT_VERBOSE=1 timeout -k 10 300 python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/verbose.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    out="$out $w: $r ($(grep -m1 '^\[rt\] engine' gpurun_out/verbose.err | sed 's/.*lds/lds/; s/  bvh.*//'))"
  done
  echo "$out"
done
