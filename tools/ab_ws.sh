#!/bin/bash
# GPU box: perf + WRITE_SIZE for compile-time variants: tools/ab_ws.sh <workload> "<flags>" ...
W=$1; shift
B="python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-linear"
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] $($B 2>/dev/null | grep -o '"value": [0-9.]*')  $(tools/wsize.sh $W 2>&1 | tail -1)"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
