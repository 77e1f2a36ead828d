#!/bin/bash
# tools/g.sh <seconds> '<command>' : run a command on the GPU box with gpurun_out/r2 present
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $T -- "mkdir -p gpurun_out/r2; $*"
