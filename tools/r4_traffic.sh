#!/bin/bash
# WRITE_SIZE / FETCH_SIZE per launch of the timed kernel under run-time knobs: tools/r4_traffic.sh <workload> "ENV=V,..." ...
W=${1:-c3}; shift
export TMPDIR=/tmp
for setting in "$@"; do
  envs=""; [ "$setting" != "-" ] && envs=$(echo $setting | tr ',' ' ')
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/tr_$c; [ -n "$envs" ] && export $envs
    rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$c -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-linear --no-others --no-frame > /tmp/tr_$c.log 2>&1
    for e in $envs; do unset ${e%%=*}; done
    python3 - $c <<'P'
import csv, glob, sys, collections
c = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/tr_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_tile_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
            acc[r["Kernel_Name"][:44]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"   {c} {k}: {sum(v)/len(v)/1024:.1f} MiB per launch (n={len(v)})")
P
  done
  echo "[$W $setting]"
done
