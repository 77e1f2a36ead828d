#!/bin/bash
# A/B runs of prebuilt library variants on the GPU box (via gpurun).  Variants are built in the CPU container with
#   RT_LIB_VARIANT=<name> RT_EXTRA_HIPCC_FLAGS="-D..." python -m ray_tracer_s8_amd.build
# into lib/librt_s8_<name>.so (the product library is never touched) and travel with the snapshot.
# usage: tools/abv.sh "<name> <name> ..." [rounds] [bench args...]     ("" or `default` = the product library)
set -e
NAMES=$1; ROUNDS=${2:-2}; shift; shift || true
OUT=gpurun_out/abv; mkdir -p $OUT
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-linear --no-others --no-frame $@"
for r in $(seq 1 $ROUNDS); do
  for n in $NAMES; do
    v=$n; [ "$n" = default ] && v=""
    RT_LIB_VARIANT=$v python3 bench.py $ARGS > $OUT/$n.$r.json 2> $OUT/$n.$r.err || { echo "$n FAILED"; tail -3 $OUT/$n.$r.err; continue; }
    python3 - $OUT/$n.$r.json $n $r <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(f"{sys.argv[2]:24s} round {sys.argv[3]}  {d['value']:9.1f} Mrays/s  {d['ms_per_step']:7.3f} ms  engine: {d['config']['engine'][:60]}", flush=True)
P
  done
done
