#!/bin/bash
# Per-kernel table of the bench under different environment settings of ONE library, alternating in one gpurun call:
#   gpurun -- 'bash tools/ab_env.sh "ROVER_RAYCAST_VARIANT=3" "ROVER_RAYCAST_VARIANT=4" -- [bench args]'
set -u
export TMPDIR=/tmp
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ $# -gt 0 ] && shift
for rep in 1 2; do
for ev in "${envs[@]}"; do
  tag=$(echo "$ev" | tr -c 'A-Za-z0-9=' '_')
  P=/tmp/abe_$tag; rm -rf $P; mkdir -p $P
  ( export $ev; rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-torch-ref "$@" > $P/bench.json 2> $P/err )
  python3 - "$ev" "$P" <<'PY'
import csv, glob, json, sys
v, P = sys.argv[1], sys.argv[2]
fs = sorted(glob.glob(P + "/**/*kernel_stats.csv", recursive=True))
if not fs:
    print(v, "no stats", open(P + "/err").read()[-600:]); sys.exit(0)
tot = 0.0; parts = []
for r in csv.DictReader(open(fs[-1])):
    if int(r["Calls"]) < 30 or "rover::" not in r["Name"]: continue
    us = float(r["AverageNs"]) / 1000; tot += us
    parts.append((r["Name"].split("(")[0].replace("void rover::", "")[:28], us))
try:
    d = json.loads(open(P + "/bench.json").read().strip().splitlines()[-1]); val = d["value"] / 1e6; cull = d.get("cull", {})
except Exception as e:
    val = -1; cull = {}
print(f"{v:28s} sum {tot:7.1f} us  {val:7.2f} M  " + "  ".join(f"{n}={u:.1f}" for n, u in sorted(parts, key=lambda x: -x[1])[:7]))
print("      cull:", {k: (round(x, 3) if isinstance(x, float) else x) for k, x in cull.items() if k in ("candidate_pairs_per_ray", "rays_with_both_tests", "rays_far_skipped", "rays_not_scanned", "rays_per_bin", "lane_items_per_ray", "lane_passes", "lane_flushes")})
err = open(P + "/err").read()
for l in err.splitlines():
    if "lane_scan_kernel," in l: print("     ", l)
PY
done
done
