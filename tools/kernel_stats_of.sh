#!/bin/bash
# Per-kernel average durations of any python command (run on the GPU box): bash tools/kernel_stats_of.sh <min calls> script.py [args]
set -u
export TMPDIR=/tmp
MIN=$1; shift
rm -rf /tmp/ks; mkdir -p /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 "$@" > /tmp/ks/out.txt 2> /tmp/ks/err.txt
python3 - "$MIN" <<'PY'
import csv, glob, sys
f = sorted(glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= int(sys.argv[1]):
        print("  %-84s %6s %9.1f us" % (r["Name"].replace("void ", "")[:84], r["Calls"], float(r["AverageNs"]) / 1000))
PY
