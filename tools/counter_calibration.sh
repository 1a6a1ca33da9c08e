#!/bin/bash
# How do the SQ_INSTS_VALU_* counters tally the instruction kinds of tools/issue_rates.hip?  (run on the GPU box)
# Every micro-kernel issues a KNOWN number of instructions of one kind: 256 CUs x 8 blocks x 4 waves x 2000 x 64 per launch.
set -u
export TMPDIR=/tmp
cd /tmp && hipcc --offload-arch=gfx950 -O2 $GRAFT_REPO_ROOT/tools/issue_rates.hip -o /tmp/issue_rates 2>/dev/null
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/counter_calibration.txt; : > $OUT
for kind in v_fma_f32 v_add_f32 v_mul_f32 v_min_f32 v_and_b32 v_cndmask_b32 v_readlane v_pk_fma_f32 v_pk_mul_f32 v_pk_add_f32 v_cmp_gt_f32 v_cvt_f32_f16 s_add_u32 mix_8pk_4s; do
  for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU"; do
    rm -rf /tmp/cal; rocprofv3 --pmc $set --output-format csv -d /tmp/cal -- /tmp/issue_rates $kind > /dev/null 2>/tmp/cal.err
    python3 - "$kind" >> $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/cal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 256 * 8 * 4 * 2000 * 64
print(sys.argv[1], {k: round(v[-1] / n, 4) for k, v in sorted(agg.items())})
PY
  done
done
cat $OUT
