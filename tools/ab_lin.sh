#!/bin/bash
# GPU box: compile-time variants of the linear kernels: c2 (16 spheres) and c3 through the linear engine: tools/ab_lin.sh "<flags>" ...
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] c2 $(python3 bench.py --workload c2 --steps 6 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | grep -o '"value": [0-9.]*')  c3 linear $(python3 bench.py --flags 32 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | grep -o '"value": [0-9.]*')"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
