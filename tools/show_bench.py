"""Prints the essentials of bench.py JSON lines: python tools/show_bench.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    c = d.get("cull") or {}
    r = d["roofline"]
    print(f"{f}: {d['value'] / 1e6:.2f} M env-steps/s, {d['ms_per_step']:.4f} ms/step, ray cast {r['avg_launch_ms']:.4f} ms ({r.get('kernel')}), "
          f"pairs/ray {c.get('candidate_pairs_per_ray', 0):.2f}, both tests {c.get('rays_with_both_tests', 0):.3f}, far skipped {c.get('rays_far_skipped', 0):.3f}, not scanned {c.get('rays_not_scanned', 0):.3f}, "
          f"rays/bin {c.get('rays_per_bin', 0):.2f}, max pairs/run {c.get('max_pairs_per_run')}, queue {c.get('queue_bytes', 0) / 2**20:.0f} MiB")
