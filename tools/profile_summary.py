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
ap.add_argument("--lib", default="", help="rover_version() of the profiled library (its source hash is stored with the entries)")
ap.add_argument("--merge-into", default="", help="directory holding valu.json / traffic.json to update with this key's entries")
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
    # (round 5: the staged ray cast — lane_scan_kernel for the terrain part, cull_scan_kernel for the rocks part on regular meshes: the
    #  counters of the two launches of a step are summed, as bench.py's one HIP-event bracket times both)
    cull = [k for k in pmc if "cull_scan" in k or "cull_exact" in k or "lane_scan" in k]
    cands = [k for k in pmc if ("raycast" in k or "cull_scan" in k or "lane_scan" in k) and "SQ_INSTS_VALU" in pmc[k] and "FETCH_SIZE" in pmc[k]]
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
        # VALU time per launch from MEASURED issue rates (profiles/issue_rates.json, tools/issue_rates.hip: cycles one SIMD needs per
        # wave64 instruction when 8 waves share it) and the kind counters (profiles/counter_calibration.txt: a packed instruction counts
        # once, in the counter of its kind).  In the ray-cast kernels every f32 add / mul / fma is a packed one (the ISA holds no plain
        # v_add / v_mul / v_fma_f32); conversions, compares, lane reads and selects run at the packed rate too; 32-bit integer / logic
        # instructions at the full rate.
        here = os.path.dirname(os.path.abspath(__file__))
        rates = json.load(open(os.path.join(here, "..", "profiles", "issue_rates.json")))["rates"]
        r_pk = max(rates[k + "@8"]["cycles_per_inst"] for k in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"))
        r_cvt, r_int, r_oth = rates["v_cvt_f32_f16@8"]["cycles_per_inst"], rates["v_and_b32@8"]["cycles_per_inst"], rates["v_cmp_gt_f32@8"]["cycles_per_inst"]
        kinds = {n: c.get("SQ_INSTS_VALU_" + n, {}).get("mean") for n in ("ADD_F32", "MUL_F32", "FMA_F32", "CVT", "INT32", "TRANS_F32")}
        simd_cycles = simd_cycles_lo = None
        if all(v is not None for v in kinds.values()):
            arith = kinds["ADD_F32"] + kinds["MUL_F32"] + kinds["FMA_F32"]
            other = c["SQ_INSTS_VALU"]["mean"] - arith - kinds["CVT"] - kinds["INT32"]
            simd_cycles = arith * r_pk + kinds["CVT"] * r_cvt + kinds["INT32"] * r_int + max(other, 0.0) * r_oth
            # lane_scan_kernel's tests are plain v_mul / v_fma_f32 (2.2-2.3 cycles) next to v_fma_mix_f32 and packed forms (4.1): the kind
            # counters do not tell them apart, so the figure above (every f32 instruction at the packed rate) is an UPPER bound there;
            # the lower bound prices them all at the plain rate
            r_plain = rates["v_fma_f32@8"]["cycles_per_inst"]
            simd_cycles_lo = arith * r_plain + kinds["CVT"] * r_cvt + kinds["INT32"] * r_int + max(other, 0.0) * r_oth
        stall = (c["SQ_WAIT_ANY"]["mean"] / c["SQ_WAVE_CYCLES"]["mean"]) if ("SQ_WAIT_ANY" in c and c["SQ_WAVE_CYCLES"]["mean"]) else None
        lib = a.lib
        v = {a.workload_key: {"valu_insts_per_launch": c["SQ_INSTS_VALU"]["mean"], "valu_kind_counts_per_launch": kinds,
                              "valu_simd_cycles_per_launch": simd_cycles, "valu_simd_cycles_lower_bound": simd_cycles_lo,
                              "stall_frac": stall,
                              "issue_rates_cycles": {"packed_f32": r_pk, "cvt": r_cvt, "int32": r_int, "other": r_oth},
                              "valu_active_quadcycles_per_launch": c["SQ_ACTIVE_INST_VALU"]["mean"],
                              "wave_quadcycles_per_launch": c["SQ_WAVE_CYCLES"]["mean"], "salu_insts_per_launch": c["SQ_INSTS_SALU"]["mean"],
                              "kernel": k.split("(")[0], "source": src, "lib": lib,
                              "method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU ... and SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 / _CVT / _INT32 "
                                        "(separate passes), mean over the launches of a pass; valu_simd_cycles = sum over kinds of count x the measured "
                                        "cycles per instruction of profiles/issue_rates.json (8 waves per SIMD)"}}
        t[a.workload_key]["lib"] = lib
        json.dump(t, open(os.path.join(a.out, f"traffic_entry_{a.workload_key}.json"), "w"), indent=1)
        json.dump(v, open(os.path.join(a.out, f"valu_entry_{a.workload_key}.json"), "w"), indent=1)
        if a.merge_into:
            for name, ent in (("traffic.json", t), ("valu.json", v)):
                path = os.path.join(a.merge_into, name)
                try:
                    cur = json.load(open(path))
                except (OSError, ValueError):
                    cur = {}
                cur.update(ent)
                json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
print("wrote", os.listdir(a.out))
