#!/bin/bash
# GPU box: c3 throughput over steps-per-check x refill threshold (LDS-tree engine)
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --no-linear"
for s in 4 6 8 10; do
  RT_EXTRA_HIPCC_FLAGS="-DRT_STEPS_PER_CHECK_LTREE=$s" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  for r in 1 2 3; do
    v=$(RT_REFILL_EIGHTHS=$r $B 2>/dev/null | grep -o '"value": [0-9.]*')
    echo "steps $s refill $r/8: $v"
  done
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
