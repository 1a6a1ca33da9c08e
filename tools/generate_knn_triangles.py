#!/usr/bin/env python3
"""CLI of assets.generate_knn_triangles — the reference's offline map generator (tasks/utils/rover_utils.py:48-123) on the
MI355X: reads map.ply and big_stones.ply from the terrain directory and writes knn_terrain/ and knn_rocks/ next to them.

    python tools/generate_knn_triangles.py /path/to/omniisaacgymenvs/tasks/utils/terrain [--cells 600 --res 0.1 --k 200]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from isaac_rover_amd import _lib, assets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("terrain_dir")
ap.add_argument("--cells", type=int, default=600)
ap.add_argument("--res", type=float, default=0.1)
ap.add_argument("--k", type=int, default=200)
ap.add_argument("--ranking", default="exact_f32", choices=["exact_f32", "reference_fp16"],
                help="reference_fp16 = rank fp16 distances between fp16 centroids and fp16 cell coordinates like rover_utils.py:71-102")
a = ap.parse_args()
eng = _lib.Engine(1, device=0)
t = time.perf_counter()
maps = assets.generate_knn_triangles(eng, a.terrain_dir, a.cells, a.cells, a.res, a.k, ranking=a.ranking)
for sub, m in maps.items():
    print(f"{sub}: map_indices {tuple(m.map_indices.shape)}, {m.triangles.shape[0]} triangles, {m.vertices.shape[0]} vertices")
print(f"done in {time.perf_counter() - t:.2f} s")
eng.close()
