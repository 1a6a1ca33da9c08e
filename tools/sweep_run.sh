#!/bin/bash
# env-order staged ray cast: slots per wave (ROVER_RAYCAST_RUN) over small batches
for args in "--envs-per-gpu 512 --steps 1000 --warmup 100" "--envs-per-gpu 1024 --steps 1000 --warmup 100" "--envs-per-gpu 2048 --steps 1000 --warmup 100" "--envs-per-gpu 4096 --steps 500 --warmup 100" "--envs-per-gpu 8192 --steps 300" "--rays 120 --envs-per-gpu 4096 --steps 300"; do
  for r in 8 16 32 64; do
    echo -n "$args run $r: "
    ROVER_RAYCAST_RUN=$r timeout -k 5 90 python bench.py --passes 5 --no-torch-ref --no-cpu-baseline --no-also $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'])"
  done
done
