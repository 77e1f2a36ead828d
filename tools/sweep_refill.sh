#!/bin/bash
# refill threshold sweep (run-time override RT_REFILL_EIGHTHS), default node format per scene
for w in c3 c4 c5; do
  out="$w:"
  for r in 1 2 3 4 5 6; do
    v=$(RT_REFILL_EIGHTHS=$r timeout -k 10 300 python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    out="$out  $r/8: $v"
  done
  echo "$out"
done
