"""Condense rocprofv3 output directories into small JSON/CSV summaries (run on the GPU box, then delete the raw files).

    python tools/profile_summary.py <stats_dir> <pmc_dir>... --out <out_dir> --tag <tag>
"""
import argparse, collections, csv, glob, json, os

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--out", required=True)
ap.add_argument("--tag", required=True)
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
print("wrote", os.listdir(a.out))
