"""Condensed view of a bench.py JSON line (stdin or file): headline, passes, roofline head, also{}."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
r = d["roofline"]
print(f'{d["value"] / 1e6:8.2f} M  {d["ms_per_step"]:.4f} ms/step  {d["config"]["workload"][:90]}')
print("   passes:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.get("passes", {}).items() if k != "statistic"})
print(f'   roofline: bound={r["bound"]} frac={r["frac"]} kernel={r["kernel"]} raycast_ms={r["avg_launch_ms"]:.4f} stall={r.get("stall_frac")} limited_by={r.get("limited_by")} stale={r["profile_stale"]}')
for k, v in d.get("also", {}).items():
    rr = v["roofline"]
    print(f'   also.{k:16s} {v["value"] / 1e6:8.2f} M  {v["ms_per_step"]:.4f} ms  ray {v["raycast_ms"]:.4f}  v{v.get("raycast_variant")}  spread {v.get("passes", {}).get("spread", 0):.3f}  {v["workload"][:70]}')
if "cpu_baseline" in d:
    print("   cpu_baseline:", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"], "threads")
