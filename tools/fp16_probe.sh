#!/bin/bash
# as-shipped arithmetic at configs[2] size: the fp16 proof's free parameter eta x the ray-cast kernel (run on the GPU box)
for eta in ${ETAS:-0.04 0.06 0.08 0.10 0.14}; do
for ev in "ROVER_RAYCAST_VARIANT=3" "ROVER_RAYCAST_VARIANT=4 ROVER_LANE_ROCKS=1"; do
  echo -n "fp16 65536 eta=$eta $ev: "
  env ROVER_CULLH_ETA=$eta $ev python bench.py --ray-precision fp16_as_shipped --passes 5 --no-torch-ref --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'], d['cull'])"
done
done
