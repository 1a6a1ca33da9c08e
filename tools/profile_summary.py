"""Condense rocprofv3 output directories into small JSON/CSV summaries (run on the GPU box, then delete the raw files).

    python tools/profile_summary.py <stats_dir> <pmc_dir>... --out <out_dir> --tag <tag>
"""
import argparse, collections, csv, glob, json, os

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--out", required=True)
ap.add_argument("--tag", required=True)
ap.add_argument("--workload-key", default="", help="e.g. E65536_P37_K200_C600: also write traffic.json / valu.json entries "
                                                     "(read by bench.py's roofline object) for the ray-cast kernel")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
pmc = {}
for d in a.dirs:
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if "rover" in r["Name"] or "fillBuffer" in r["Name"]]
        with open(os.path.join(a.out, f"{a.tag}_kernel_stats.csv"), "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(keep)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        tail = rows[-28:]
        with open(os.path.join(a.out, f"{a.tag}_last_steps_timeline.txt"), "w") as o:
            prev = None
            for r in tail:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                gap = (s - prev) / 1e3 if prev else 0.0
                o.write(f'{r["Kernel_Name"][:70]:70s} dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  grid {r["Grid_Size_X"]}\n')
                prev = e
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rover" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            pmc.setdefault(k, {})[c] = {"n": len(v), "mean": sum(v) / len(v)}
if pmc:
    json.dump(pmc, open(os.path.join(a.out, f"{a.tag}_pmc_rover_kernels.json"), "w"), indent=1)
if pmc and a.workload_key:
    # the ray-cast stage that ran in these passes: the binned f32 / binned fp16 / env-order kernel, or the culled ray cast's
    # two kernels (cull_scan + cull_exact: counters summed per step, bench.py times the pair with one HIP-event bracket)
    cull = [k for k in pmc if "cull_scan" in k or "cull_exact" in k]
    cands = [k for k in pmc if ("raycast" in k or "cull_scan" in k) and "SQ_INSTS_VALU" in pmc[k] and "FETCH_SIZE" in pmc[k]]
    if len(cull) == 2 and all("SQ_INSTS_VALU" in pmc[k] and "FETCH_SIZE" in pmc[k] for k in cull):
        name = "+".join(sorted(k.split("(")[0].replace("rover::", "") for k in cull))
        pmc[name] = {cn: {"n": min(pmc[k][cn]["n"] for k in cull), "mean": sum(pmc[k][cn]["mean"] for k in cull)}
                     for cn in pmc[cull[0]] if all(cn in pmc[k] for k in cull)}
        cands = [name]
    if cands:
        k = max(cands, key=lambda n: pmc[n]["SQ_INSTS_VALU"]["mean"])
        c = pmc[k]
        src = f"profiles/{a.tag}_pmc_rover_kernels.json"
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies a 128-B read request as 64 B (MI355X_MICROARCH.md, HBM): x2
        traffic = 2.0 * c["FETCH_SIZE"]["mean"] * 1024.0 + c["WRITE_SIZE"]["mean"] * 1024.0
        t = {a.workload_key: {"hbm_bytes_per_launch": traffic, "kernel": k.split("(")[0],
                              "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes); bytes = "
                                        "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction); cross-check TCC_MISS_sum*128 B = "
                                        f"{c['TCC_MISS_sum']['mean'] * 128.0:.4g}", "source": src}}
        v = {a.workload_key: {"valu_insts_per_launch": c["SQ_INSTS_VALU"]["mean"],
                              "valu_active_quadcycles_per_launch": c["SQ_ACTIVE_INST_VALU"]["mean"],
                              "wave_quadcycles_per_launch": c["SQ_WAVE_CYCLES"]["mean"], "salu_insts_per_launch": c["SQ_INSTS_SALU"]["mean"],
                              "kernel": k.split("(")[0], "source": src,
                              "method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES "
                                        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY, mean over the launches of the pass"}}
        json.dump(t, open(os.path.join(a.out, "traffic_entry.json"), "w"), indent=1)
        json.dump(v, open(os.path.join(a.out, "valu_entry.json"), "w"), indent=1)
print("wrote", os.listdir(a.out))
