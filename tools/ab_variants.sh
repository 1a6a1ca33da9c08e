#!/bin/bash
# culled (3) vs walked (4) ray cast on the bench workloads, one GPU box: bash tools/ab_variants.sh
export ROVER_SCENE_CACHE=/tmp/sc
run() { for v in 3 4; do echo -n "v$v $* : "; ROVER_RAYCAST_VARIANT=$v python bench.py --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d.get('walk') or d.get('cull') or {}
print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms  ray cast', round(d['roofline']['avg_launch_ms'],4), {k:(round(v,2) if isinstance(v,float) else v) for k,v in w.items() if k in ('candidates_per_ray','entries_tested_per_ray','candidate_pairs_per_ray','rays_walking_nothing','rays_off_the_all_B_path')})"; done; }
run --mesh irregular
run --ray-precision fp16_as_shipped
run --envs-per-gpu 4096 --steps 500
run --envs-per-gpu 32768
run --rays 120 --validate-goals
run --rays native --envs-per-gpu 512 --steps 300
run --rays native --envs-per-gpu 4096
