#!/bin/bash
# every batch size and the other workloads: env-order kernel (1), culled (3), staged behind the sort (4), staged in env order (4e): M env-steps/s, ms per step, ray-cast ms
[ "${SWEEP_F32:-1}" = 1 ] && for args in "--envs-per-gpu 512 --steps 1000 --warmup 100" "--envs-per-gpu 1024 --steps 1000 --warmup 100" "--envs-per-gpu 2048 --steps 1000 --warmup 100" "--envs-per-gpu 4096 --steps 1000 --warmup 100" "--envs-per-gpu 8192 --steps 400" "--envs-per-gpu 16384 --steps 200" "--envs-per-gpu 32768 --steps 100" "--envs-per-gpu 65536" "--rays native --envs-per-gpu 512 --steps 500 --mesh irregular" "--rays 120 --envs-per-gpu 4096 --steps 500" "--rays 120 --validate-goals" "--mesh irregular" "--ray-precision fp16_as_shipped"; do
  for v in 1 3 4 4e; do
    echo -n "$args v$v: "
    if [ $v = 4e ]; then export ROVER_LANE_ENV_ORDER=1; vv=4; else export ROVER_LANE_ENV_ORDER=0; vv=$v; fi
    ROVER_RAYCAST_VARIANT=$vv python bench.py --passes 5 --no-torch-ref --no-cpu-baseline --no-also $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'])"
  done
done
# the as-shipped arithmetic (binned 2 / culled 3 / staged 4 / staged in env order 4e): SWEEP_FP16=1 bash tools/sweep_small.sh
if [ "${SWEEP_FP16:-0}" = 1 ]; then
for args in "--envs-per-gpu 512 --steps 1000 --warmup 100" "--envs-per-gpu 1024 --steps 1000 --warmup 100" "--envs-per-gpu 1536 --steps 1000 --warmup 100" "--envs-per-gpu 2048 --steps 1000 --warmup 100" "--envs-per-gpu 4096 --steps 1000 --warmup 100" "--envs-per-gpu 16384 --steps 200" "--envs-per-gpu 65536" "--rays 120 --validate-goals" "--rays native --envs-per-gpu 4096 --steps 100" "--rays native --envs-per-gpu 512 --steps 500" "--rays native --envs-per-gpu 512 --steps 500 --mesh irregular" "--mesh irregular" "--mesh irregular --envs-per-gpu 16384" "--mesh irregular --envs-per-gpu 4096 --steps 500"; do
  for v in 2 3 4 4e; do
    echo -n "fp16 $args v$v: "
    if [ $v = 4e ]; then export ROVER_LANE_ENV_ORDER=1; vv=4; else export ROVER_LANE_ENV_ORDER=0; vv=$v; fi
    ROVER_RAYCAST_VARIANT=$vv python bench.py --ray-precision fp16_as_shipped --passes 5 --no-torch-ref --no-cpu-baseline --no-also $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'])"
  done
done
fi
