#!/bin/bash
# GPU box: c5 / mesh / c3-through-L2 for compile-time variants: tools/ab3.sh "<flags>" ...
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-linear"
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] c5 $($B --workload c5 2>/dev/null | grep -o '"value": [0-9.]*')  mesh $($B --workload mesh 2>/dev/null | grep -o '"value": [0-9.]*')  c3/L2 $($B --flags 256 2>/dev/null | grep -o '"value": [0-9.]*')"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
