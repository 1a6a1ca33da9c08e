#!/bin/bash
# rocprofv3 evidence for the policy forward, run on the GPU box from the repo root:
#   bash tools/collect_policy_profile.sh <tag>     -> gpurun_out/profiles_<tag>/<tag>_policy_*.{json,csv}
# One --kernel-trace --stats run and one separate --pmc pass (never combined) of tools/policy_profile_run.py.
set -u
TAG=${1:-rXX}
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
P=/tmp/polprof_$TAG
rm -rf "$P"; mkdir -p "$P"
python3 tools/policy_profile_run.py > "$OUT/${TAG}_policy_forward.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 tools/policy_profile_run.py > "$OUT/${TAG}_policy_forward_under_rocprof.json" 2> $P/stats.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $P/pmc -- python3 tools/policy_profile_run.py > /dev/null 2> $P/pmc.err
python3 - "$P" "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, sys
P, OUT, TAG = sys.argv[1:4]
names = ["encoder0_chain_634_80_60", "encoder1_chain_1112_80_60", "mlp_chain_124_256_160_128_2"]
mfmas = {  # 16x16x4 MFMAs per launch: rows / 16 waves x (k-steps x output tiles), as the kernel issues them
    "encoder0_chain_634_80_60": 4096 * (20 * 8 * 5 + 5 * 4 * 4),
    "encoder1_chain_1112_80_60": 4096 * (35 * 8 * 5 + 5 * 4 * 4),
    "mlp_chain_124_256_160_128_2": 4096 * (2 * 4 * 8 * 8 + 16 * 4 * 10 + 10 * 4 * 8 + 8 * 4 * 1),
}
def label(rows):       # launches in order: enc0, enc1, mlp per forward (the two encoder chains are the same kernel instance)
    out, i = [], 0
    for r in rows:
        n = r["Kernel_Name"]
        if "chain16_kernel" not in n: continue
        out.append((names[2] if "<16" in n.replace(" ", "") or "ILi16" in n else names[i % 2], r))
        if not ("<16" in n.replace(" ", "") or "ILi16" in n): i += 1
    return out
res = {}
f = sorted(glob.glob(P + "/stats/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
for k, r in label(rows): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in names: res[k] = {"launches": len(dur[k]), "avg_us": sum(dur[k]) / max(1, len(dur[k])), "min_us": min(dur[k]), "mfma_16x16x4_per_launch": mfmas[k]}
f = sorted(glob.glob(P + "/pmc/**/*counter_collection.csv", recursive=True))[-1]
byd = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "chain16_kernel" in r["Kernel_Name"]: byd.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for k, r in label([byd[d] for d in sorted(byd)]):
    for c, v in r.items():
        if c != "Kernel_Name": agg[k][c].append(v)
for k in names:
    pm = {c: sum(v) / len(v) for c, v in agg[k].items()}
    res[k]["pmc_mean_per_launch"] = pm
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and "SQ_BUSY_CYCLES" in pm:
        simd_cycles = pm["SQ_BUSY_CYCLES"] / 32.0 * 1024.0          # SQ_BUSY_CYCLES sums the 32 shader engines; 1 024 SIMDs
        res[k]["mfma_busy_frac_of_simd_cycles"] = pm["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        res[k]["mfma_busy_cycles_per_mfma"] = pm["SQ_VALU_MFMA_BUSY_CYCLES"] / mfmas[k]
        res[k]["clock_GHz_in_pmc_pass"] = None
json.dump(res, open(f"{OUT}/{TAG}_policy_kernels.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
tail -c 400 $P/*.err | tail -8
cat "$OUT/${TAG}_policy_forward.json"
