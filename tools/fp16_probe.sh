#!/bin/bash
# as-shipped arithmetic at configs[2] size: the fp16 proof's parameters (eta, split of test (A)'s cross term) x the ray-cast kernel (run on the GPU box)
for eta in ${ETAS:-0.06}; do
for split in ${SPLITS:-2 4 6 8 12}; do
for ev in "ROVER_RAYCAST_VARIANT=3" "ROVER_RAYCAST_VARIANT=4 ROVER_LANE_ROCKS=1"; do
  echo -n "fp16 65536 eta=$eta split=$split $ev: "
  env ROVER_CULLH_ETA=$eta ROVER_CULLH_SPLIT=$split $ev python bench.py --ray-precision fp16_as_shipped --passes 5 --no-torch-ref --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'], d['cull'])"
done
done
done
