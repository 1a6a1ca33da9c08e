#!/bin/bash
# In-call A/B of two builds of librover_step.so on ONE GPU box (box-to-box spread is several per cent): build the two versions here,
# copy them to ab_tmp/base.so and ab_tmp/new.so (ab_tmp/ is git-ignored but travels with gpurun), then
#   gpurun -- 'bash tools/ab_builds.sh'
# (ABARGS="--envs-per-gpu 4096 --steps 1000 --warmup 100" for another config)
# alternates them and prints M env-steps/s, ms per step, ray-cast ms; restore csrc/librover_step.so afterwards (build.sh).
for i in 1 2 3; do
  for v in base new; do
    cp ab_tmp/$v.so isaac_rover_2.0_amd/csrc/librover_step.so
    echo -n "$v "
    python bench.py --no-torch-ref --no-cpu-baseline ${ABARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
