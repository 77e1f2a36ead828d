#!/bin/bash
# GPU box: compile-time knobs of the quantised walks on c5: tools/sweep_cull.sh "<hipcc flags>" ...
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --workload c5"
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] culled $($B 2>/dev/null | grep -o '"value": [0-9.]*')  capped12 $(RT_FORCE_CAPPED=1 RT_STACK_LDS=12 RT_VERBOSE=1 $B 2>&1 | grep -o '"value": [0-9.]*\|workgroups/CU [0-9]*' | sort -u | tr '\n' ' ')  capped8 $(RT_FORCE_CAPPED=1 RT_STACK_LDS=8 $B 2>/dev/null | grep -o '"value": [0-9.]*')"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
