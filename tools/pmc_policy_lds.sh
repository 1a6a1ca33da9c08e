#!/bin/bash
# LDS counters of the policy-forward kernels (GPU box): bash tools/pmc_policy_lds.sh
set -u
export TMPDIR=/tmp
P=/tmp/pol_lds; rm -rf $P; mkdir -p $P
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $P -- python3 tools/policy_profile_run.py > /dev/null 2> $P/err
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("/tmp/pol_lds/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "chain16" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   ", c, round(sum(v) / len(v)))
PY
tail -3 $P/err
