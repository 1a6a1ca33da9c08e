#!/bin/bash
# Per-kernel average durations of the default bench (run on the GPU box): bash tools/quick_stats.sh [bench args]
set -u
export TMPDIR=/tmp
P=/tmp/quick_stats; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $P/bench.json 2> $P/err
python3 - <<'PY'
import csv, glob, json
f = sorted(glob.glob("/tmp/quick_stats/**/*kernel_stats.csv", recursive=True))[-1]
tot = 0.0
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) < 50 or "rover::" not in r["Name"]: continue
    us = float(r["AverageNs"]) / 1000; tot += us
    print(f'{r["Name"].split("(")[0].replace("void ", "")[:44]:46s}{r["Calls"]:>6s}{us:9.1f}')
print("sum", round(tot, 1))
d = json.loads(open("/tmp/quick_stats/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
PY
