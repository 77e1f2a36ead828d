#!/bin/bash
# bench c2..c5 (default engines) for each build variant
for v in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$v" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  out="[$v]"
  for w in c2 c3 c4 c5; do
    r=$(timeout -k 10 300 python3 bench.py --workload $w --steps ${RT_VARIANT_STEPS:-3} --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    out="$out $w: $r"
  done
  echo "$out"
done
