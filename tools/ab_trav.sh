#!/bin/bash
# GPU box: compile-time variants of the exact-node L2 kernel on what it serves now (meshes, mid-size sphere fields):
# tools/ab_trav.sh "<hipcc flags>" ...
for f in "$@"; do
  RT_EXTRA_HIPCC_FLAGS="$f" python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
  echo "[$f]"; python3 tools/mesh_probe.py 2>/dev/null | grep -v counted
  python3 tools/cull_matrix_small.py 2>/dev/null | grep "field 2000\|field 3500"
done
python3 -c "from ray_tracer_s8_amd import build; build.build(force=True)"
