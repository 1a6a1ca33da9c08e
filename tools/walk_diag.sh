#!/bin/bash
# Diagnostic timings of the walked ray cast (wrong results, right timing): bash tools/walk_diag.sh [bench args]
export ROVER_SCENE_CACHE=/tmp/sc ROVER_RAYCAST_VARIANT=4
for d in ${DIAGS:-0 1 2 3 4 7 15 16}; do
  ROVER_WALK_DIAG=$d python bench.py --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('diag $d', round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
done
