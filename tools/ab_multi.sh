#!/bin/bash
# Per-kernel table (rocprofv3 kernel trace) of the bench for every library in ab_tmp/*.so, alternating in ONE gpurun call:
#   gpurun -- 'bash tools/ab_multi.sh [bench args]'      (restore csrc/librover_step.so afterwards: csrc/build.sh)
set -u
export TMPDIR=/tmp
for rep in 1 2; do
for so in ab_tmp/*.so; do
  v=$(basename $so .so)
  cp $so isaac_rover_2.0_amd/csrc/librover_step.so
  P=/tmp/abm_$v; rm -rf $P; mkdir -p $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-torch-ref "$@" > $P/bench.json 2> $P/err
  python3 - "$v" "$P" <<'PY'
import csv, glob, json, sys
v, P = sys.argv[1], sys.argv[2]
fs = sorted(glob.glob(P + "/**/*kernel_stats.csv", recursive=True))
if not fs:
    print(v, "no stats", open(P + "/err").read()[-400:]); sys.exit(0)
tot = 0.0; parts = []
for r in csv.DictReader(open(fs[-1])):
    if int(r["Calls"]) < 30 or "rover::" not in r["Name"]: continue
    us = float(r["AverageNs"]) / 1000; tot += us
    parts.append((r["Name"].split("(")[0].replace("void rover::", "")[:28], us))
try:
    d = json.loads(open(P + "/bench.json").read().strip().splitlines()[-1]); val = d["value"] / 1e6
except Exception as e:
    val = -1
print(f"{v:14s} sum {tot:7.1f} us  {val:7.2f} M  " + "  ".join(f"{n}={u:.1f}" for n, u in sorted(parts, key=lambda x: -x[1])[:6]))
PY
done
done
