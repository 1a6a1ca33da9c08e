export TMPDIR=/tmp
P=/tmp/task_stats; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 tools/task_overhead.py > $P/out 2> $P/err
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/task_stats/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) < 500: continue
    print(f'{r["Name"].split("(")[0].replace("void ", "")[:50]:52s}{r["Calls"]:>8s}{float(r["AverageNs"])/1000:9.1f}{float(r["MinNs"])/1000:9.1f}{float(r["MaxNs"])/1000:9.1f}')
PY
