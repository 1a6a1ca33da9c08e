#!/bin/bash
# staged ray cast behind the sort: rays per wave (ROVER_RAYCAST_RUN) over mid / large batches
for args in "--envs-per-gpu 8192 --steps 300" "--envs-per-gpu 16384 --steps 200" "--envs-per-gpu 32768 --steps 100" "--envs-per-gpu 65536" "--rays 120 --validate-goals" "--mesh irregular"; do
  for r in 16 32 64; do
    echo -n "$args run $r: "
    ROVER_LANE_ENV_ORDER=0 ROVER_RAYCAST_RUN=$r timeout -k 5 90 python bench.py --passes 5 --no-torch-ref --no-cpu-baseline --no-also $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'])"
  done
done
