#!/bin/bash
# Round-4 A/B of run-time knobs (no rebuild): tools/r4_knobs.sh "<workloads>" "ENV=V,ENV2=V ..." (each word one setting; "-" = defaults)
WLS=${1:-c3}; shift
OUT=gpurun_out/r4; mkdir -p $OUT
for setting in "$@"; do
  for w in $WLS; do
    envs=""; [ "$setting" != "-" ] && envs=$(echo $setting | tr ',' ' ')
    r=$(env $envs timeout -k 10 200 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-pcie --no-linear --no-frame --no-others 2>$OUT/knob.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(f\"{d['value']:.0f} Mrays/s {d['ms_per_step']:.3f} ms\")")
    echo "$w [$setting] $r"
  done
done
