#!/bin/bash
# GPU box: FETCH_SIZE of the tile kernel for a workload (KiB per launch, as counted): tools/fsize.sh <workload>
W=$1; shift
OUT=gpurun_out/fs_$W; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-linear > $OUT/log 2>&1
python3 - <<PY
import csv, glob
tot, n = 0.0, 0
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and "rt_tile_kernel" in row["Kernel_Name"] and "true>" not in row["Kernel_Name"]:
            tot += float(row["Counter_Value"]); n += 1
print("$W FETCH_SIZE KiB per launch:", tot / max(n, 1), "launches", n)
PY
