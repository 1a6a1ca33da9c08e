"""Shape fuzz (run on the GPU box): random batch sizes — multiples of 64 and not —, random numbers of heightmap points (1 .. 300, any
padding of the ray slots: the fused-histogram shapes R8 = 32 / 64 and the others), random map sizes, K and precisions; the culled and the
binned ray cast against the env-order one (as shipped fp16: the culled against the binned) on the same poses, two steps per engine (the second on other poses: what a step leaves behind
— the sort's count table, the queue — must not matter).  Everything the step returns is compared bit for bit.

    python tools/fuzz_shapes.py [cases] [seed]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaac_rover_amd import _lib, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from hip_helpers import hip_step, make_engine

print("library:", _lib.version(), flush=True)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
z = -0.26878
rays = 0
for c in range(cases):
    n = int(rng.choice([1, 7, 63, 64, 65, 100, 128, 500, 1000, 1024, 2000, 4096, 5000, 8192, 20000, 33000]))
    p = int(rng.choice([1, 2, 5, 6, 7, 13, 37, 38, 39, 70, 102, 120, 230, 300]))
    cells = int(rng.choice([24, 40, 64, 96]))
    k = int(rng.choice([8, 24, 64, 200]))
    prec = int(rng.choice([0, 0, 2]))
    pts = np.stack([rng.uniform(0.1, 2.5, p).round(4), rng.uniform(-1.5, 1.5, p).round(4), np.full(p, z)], axis=1)
    ns = int(rng.integers(0, p + 1))
    distn = (pts, np.arange(ns, dtype=np.int64), np.arange(ns, p, dtype=np.int64))
    scene = synth.make_scene(n_cells=cells, k=k, n_stones=8)
    outs = {}
    base = 2 if prec == 2 else 1        # (the as-shipped fp16 arithmetic exists in the sorted kernels only)
    for variant in (base, 3, 2, 4, 40, 41):          # 4: staged; 40: staged in env order (no sort); 41: staged, rocks part too
        if variant in outs:
            continue
        eng = make_engine(scene, distn, n, variant=min(variant, 4))
        eng.set_option("ray_precision", prec)
        if variant >= 4:
            eng.set_option("lane_env_order", 1 if variant == 40 else 0)
            eng.set_option("lane_rocks", 1 if variant == 41 else 0)
        res = []
        for step in range(2):
            st = synth.make_states(n, cells * 0.1, seed=1000 * c + step)
            res.append(hip_step(eng, st))
        eng.close()
        outs[variant] = res
    for variant in (3, 2, 4, 40, 41):
        for step in range(2):
            for key in outs[base][step]:
                a, b = outs[base][step][key], outs[variant][step][key]
                if not (np.array_equal(a, b) or np.array_equal(a, b, equal_nan=True)):
                    raise SystemExit(f"MISMATCH case {c}: envs {n} points {p} cells {cells} K {k} precision {prec} variant {variant} step {step}: {key}")
    rays += 2 * n * (26 + p)
    print(f"case {c}: envs {n}, {p} + 26 rays (R8 = {(26 + p + 7) // 8 * 8}), {cells} x {cells} cells, K = {k}, precision {prec}: ok", flush=True)
print(f"fuzz ok: {cases} shapes, {rays / 1e6:.1f} M rays x 5 comparisons, all bit-identical")
