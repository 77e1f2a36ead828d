#!/bin/bash
# GPU box: tiles per queue atomic (RT_QUEUE_TAKE_MIN) on c5 / mesh / c2: speed and WRITE_SIZE
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f] c5 $(python3 bench.py --workload c5 --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --no-linear 2>/dev/null | grep -o '"value": [0-9.]*') mesh $(python3 bench.py --workload mesh --steps 4 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | grep -o '"value": [0-9.]*') c2 $(python3 bench.py --workload c2 --steps 6 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | grep -o '"value": [0-9.]*')"
  tools/wsize.sh c5
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
