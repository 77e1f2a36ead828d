#!/bin/bash
# bench c3/c5 with exact (--flags 64) and quantised (--flags 128) traversal nodes for each build variant
for v in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$v" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  out="[$v]"
  for w in c3 c5; do for f in ${RT_VARIANT_FLAGS:-64 128}; do
    r=$(timeout -k 10 300 python3 bench.py --workload $w --steps 8 --warmup 1 --no-cpu-baseline --no-pcie --flags $f 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    out="$out $w/f$f: $r"
  done; done
  echo "$out"
done
