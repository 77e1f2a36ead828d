#!/bin/bash
# GPU box: A/B of compile-time variants on a workload: tools/ab.sh <workload> "<flags A>" "<flags B>" ...
W=$1; shift
B="python3 bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --no-linear"
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] $($B 2>/dev/null | grep -o '"value": [0-9.]*')  nostage: $(RT_NO_STAGE=1 $B 2>/dev/null | grep -o '"value": [0-9.]*')"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
