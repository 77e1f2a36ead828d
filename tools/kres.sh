#!/bin/bash
# Kernel resource usage (VGPRs, spills, LDS, occupancy) of the traversal / linear kernels: tools/kres.sh [trav|lin] [extra flags]
U=${1:-trav}; shift || true
F=""; [ "$U" = trav ] && F="-fno-slp-vectorize"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-unroll-loops $F -Iinclude -Iray_tracer_s8_amd/csrc "$@" \
  -Rpass-analysis=kernel-resource-usage -c ray_tracer_s8_amd/csrc/rt_kernels_$U.hip -o /tmp/kres_$U.o 2>&1 | \
  grep -E "Function Name|VGPRs:|Spill|Occupancy|SGPRs:|ScratchSize" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
