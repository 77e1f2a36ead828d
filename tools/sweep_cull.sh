#!/bin/bash
# GPU box: compile-time knobs of the quantised walks on c5: tools/sweep_cull.sh "<hipcc flags>" ...  (RT_SWEEP_ENV="VAR=val" adds an env)
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --workload c5"
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] culled $($B 2>/dev/null | grep -o '"value": [0-9.]*')  plain $($B --flags 2048 2>/dev/null | grep -o '"value": [0-9.]*')  refill3 $(RT_REFILL_EIGHTHS=3 $B 2>/dev/null | grep -o '"value": [0-9.]*') refill5 $(RT_REFILL_EIGHTHS=5 $B 2>/dev/null | grep -o '"value": [0-9.]*')"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
