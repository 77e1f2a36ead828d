#!/bin/bash
# Round-4 quick measurement on the GPU box: the default library's c3 / c2 / c4 / c5 / mesh rates, then the phase clock and census of c3.
# usage (inside gpurun): tools/r4_measure.sh <tag> [phase configs...]
TAG=${1:-m}; shift || true
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-linear --no-frame > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || { echo bench failed; tail -5 $OUT/bench_$TAG.err; exit 1; }
python3 - $OUT/bench_$TAG.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"c3 {d['value']:.0f} Mrays/s {d['ms_per_step']:.3f} ms | " + " | ".join(f"{k} {v['value']:.0f}" for k, v in d.get('other_workloads', {}).items()), flush=True)
P
for c in ${@:-c3}; do
  [ -f ray_tracer_s8_amd/lib/librt_s8_ptime.so ] && timeout -k 10 200 python3 tools/phase_time.py $c 2>&1 | tail -12
done
[ -f ray_tracer_s8_amd/lib/librt_s8_pcensus.so ] && timeout -k 10 200 python3 tools/phase_census.py c3 2>&1 | grep -E "iterations|acquisition|traversal step|root-test|finalize|finish path|ray generation" 
exit 0
