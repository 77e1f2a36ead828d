#!/bin/bash
# GPU box: culled-walk knobs on c5 (and a dense field): tools/sweep_cull.sh "<hipcc flags>" ...
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --workload c5"
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] $($B 2>/dev/null | grep -o '"value": [0-9.]*')"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
