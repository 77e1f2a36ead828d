#!/bin/bash
for v in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$v" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  r5=$(timeout -k 10 300 python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  r3=$(timeout -k 10 300 python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --flags 16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "[$v] c5: $r5  c3(traverse): $r3 Mrays/s"
done
