#!/bin/bash
# Mean per launch of arbitrary PMC counters for one kernel of the default bench (run on the GPU box):
#   KERNEL=cull_scan bash tools/pmc_probe.sh "CTR_A CTR_B ..." ["CTR_C ..." ...]      (one rocprofv3 --pmc pass per argument)
# Keep to the SQ_* / TCC_* / FETCH_SIZE / WRITE_SIZE counters of tools/collect_profiles.sh: a pass with TA_* + GRBM_GUI_ACTIVE counters
# never returned on this pool (round 4: the call was killed after 7 silent minutes) — give every pass its own `timeout -k 10 120`.
set -u
export TMPDIR=/tmp ROVER_SCENE_CACHE=/tmp/sc
K=${KERNEL:-cull_scan}
i=0
for pass in "$@"; do
  i=$((i+1)); D=/tmp/pmc_probe_$i; rm -rf $D
  timeout -k 10 120 rocprofv3 --pmc $pass --output-format csv -d $D -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also ${BENCH_ARGS:-} > /dev/null 2> $D.err
  python3 - "$D" "$K" <<'PY'
import csv, glob, sys, collections
d, k = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:40s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
if not agg:
    print("no counters collected:", open(d + ".err").read()[-400:])
PY
done
