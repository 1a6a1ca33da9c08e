"""Merge gpurun_out/profiles_<tag>/{traffic,valu}_entry.json into profiles/{traffic,valu}.json (run in the build container)."""
import json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("traffic", "valu"):
    src = os.path.join(root, "gpurun_out", f"profiles_{tag}", f"{name}_entry.json")
    dst = os.path.join(root, "profiles", f"{name}.json")
    cur = json.load(open(dst)) if os.path.exists(dst) else {}
    cur.update(json.load(open(src)))
    json.dump(cur, open(dst, "w"), indent=1, sort_keys=True)
    print("updated", dst, list(cur))
