#!/bin/bash
# PMC means per rover kernel of the default bench (or "$@"), five separate --pmc passes, printed as one table: bash tools/pmc_quick.sh [bench args]
set -u
export TMPDIR=/tmp
P=/tmp/pmcq; rm -rf $P; mkdir -p $P
run() { timeout -k 10 120 rocprofv3 --pmc $1 --output-format csv -d $P/$2 -- python3 bench.py --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-also "${@:3}" > /dev/null 2> $P/$2.err; }
run "FETCH_SIZE" p1 "$@"
run "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" p2 "$@"
run "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" p3 "$@"
run "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR" p4 "$@"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rover" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void rover::", "")[:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
ks = sorted({k for k, _ in agg})
for k in ks:
    d = {c: sum(v) / len(v) for (kk, c), v in agg.items() if kk == k}
    if d.get("SQ_INSTS_VALU", 0) < 1e6: continue
    print(k)
    print("   " + "  ".join(f"{c.replace('SQ_', '')}={v:.4g}" for c, v in sorted(d.items())))
    if "SQ_WAVE_CYCLES" in d:
        print(f"   stall {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.3f}  hbm_bytes {2 * d.get('FETCH_SIZE', 0) * 1024 + d.get('WRITE_SIZE', 0) * 1024:.4g}  L2 hit {d.get('TCC_HIT_sum', 0) / max(1, d.get('TCC_HIT_sum', 0) + d.get('TCC_MISS_sum', 0)):.3f}")
PY
