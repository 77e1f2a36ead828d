#!/bin/bash
# A/B of launch-path knobs by environment (latched once at rt_init): tools/ab_env.sh "<VAR=val|none> ..." [rounds] [bench args]
set -e
SETS=$1; ROUNDS=${2:-2}; shift; shift || true
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-linear --no-others --no-frame $@"
mkdir -p gpurun_out/abv
for r in $(seq 1 $ROUNDS); do
  for s in $SETS; do
    if [ "$s" = none ]; then python3 bench.py $ARGS > gpurun_out/abv/env.json 2> gpurun_out/abv/env.err; else env $s python3 bench.py $ARGS > gpurun_out/abv/env.json 2> gpurun_out/abv/env.err; fi
    python3 - "$s" $r <<'P'
import json, sys
d = json.loads(open("gpurun_out/abv/env.json").read())
print(f"{sys.argv[1]:24s} round {sys.argv[2]}  {d['value']:9.1f} Mrays/s  {d['ms_per_step']:7.3f} ms  engine: {d['config']['engine'][:50]}", flush=True)
P
  done
done
