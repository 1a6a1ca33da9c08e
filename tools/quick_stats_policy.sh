#!/bin/bash
# Per-kernel average durations of tools/policy_time.py (run on the GPU box)
set -u
export TMPDIR=/tmp
P=/tmp/quick_stats_p; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 tools/policy_time.py > $P/out.txt 2> $P/err
cat $P/out.txt
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/quick_stats_p/**/*kernel_trace.csv", recursive=True))[-1]
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not n.startswith("void rover::") and not n.startswith("rover::"): continue
    key = (n.split("(")[0][-60:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:62s} grid {k[1]:>9s} calls {c:4d} avg {t / c:8.1f} us")
PY
