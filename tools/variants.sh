#!/bin/bash
# Build kernel variants (on the GPU box) and bench each in turn: usage tools/variants.sh "<flags A>" "<flags B>" ...
for v in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$v" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for i in 1 2; do
    r=$(timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    echo "[$v] run$i: $r Mrays/s"
  done
done
