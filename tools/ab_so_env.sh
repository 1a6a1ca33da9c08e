#!/bin/bash
# Per-kernel table of the bench for (library, environment) pairs in ONE gpurun call:
#   gpurun -- 'bash tools/ab_so_env.sh "ab_tmp/x.so ROVER_LANE_ROCKS=1" "ab_tmp/y.so ROVER_RAYCAST_RUN=8" -- [bench args]'
# (copies each library over csrc/librover_step.so on the BOX; the local tree is untouched)
set -u
export TMPDIR=/tmp
arms=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do arms+=("$1"); shift; done; [ $# -gt 0 ] && shift
for rep in 1 2; do
for arm in "${arms[@]}"; do
  so=${arm%% *}; ev=${arm#* }; [ "$ev" = "$arm" ] && ev="X_NONE=1"
  cp "$so" isaac_rover_2.0_amd/csrc/librover_step.so
  tag=$(echo "$arm" | tr -c 'A-Za-z0-9=' '_')
  P=/tmp/abs_$tag; rm -rf $P; mkdir -p $P
  ( export $ev; rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-torch-ref "$@" > $P/bench.json 2> $P/err )
  python3 - "$arm" "$P" <<'PY'
import csv, glob, json, sys
v, P = sys.argv[1], sys.argv[2]
fs = sorted(glob.glob(P + "/**/*kernel_stats.csv", recursive=True))
if not fs:
    print(v, "no stats", open(P + "/err").read()[-600:]); sys.exit(0)
tot = 0.0; parts = []
for r in csv.DictReader(open(fs[-1])):
    if int(r["Calls"]) < 30 or "rover::" not in r["Name"]: continue
    us = float(r["AverageNs"]) / 1000; tot += us
    parts.append((r["Name"].split("(")[0].replace("void rover::", "")[:24], us))
try:
    d = json.loads(open(P + "/bench.json").read().strip().splitlines()[-1]); val = d["value"] / 1e6
except Exception as e:
    val = -1
print(f"{v:44s} sum {tot:7.1f} us {val:7.2f} M  " + "  ".join(f"{n}={u:.1f}" for n, u in sorted(parts, key=lambda x: -x[1])[:7]), flush=True)
PY
done
done
