#!/bin/bash
# M env-steps/s of the bench for ray-cast variants 3 and 4 over batch sizes / ray sets (one gpurun call): bash tools/sweep_variants.sh
for args in "--envs-per-gpu 65536" "--envs-per-gpu 32768" "--envs-per-gpu 16384" "--envs-per-gpu 8192 --steps 400" "--envs-per-gpu 4096 --steps 1000 --warmup 100" "--envs-per-gpu 2048 --steps 1000 --warmup 100" "--rays 120 --validate-goals" "--rays native --envs-per-gpu 4096" "--rays native --envs-per-gpu 512 --steps 500" "--mesh irregular"; do
  for v in 3 4 3 4; do
    echo -n "$args v$v: "
    ROVER_RAYCAST_VARIANT=$v python bench.py --no-torch-ref --no-cpu-baseline --no-also $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'])"
  done
done
