"""Reader / writer for the terrain assets in the reference's on-disk formats.

Layout under ``root`` (what the reference opens relative to its working directory):
  tasks/utils/terrain/knn_terrain/{map_indices,triangles,vertices}.pt   camera.py:154-161
  tasks/utils/terrain/knn_rocks/{map_indices,triangles,vertices}.pt     rock_detect.py:151-158
  tasks/utils/terrain/stone_info.npy                                    rover.py:144
  tasks/utils/terrain/heightmap_tensor.pt                               rover.py:210
``map_indices.pt`` is stored [K, X, Y] int32 (rover_utils.py:68,108,116) and swapped to [X, Y, K] on load
(camera.py:157-158); ``vertices.pt`` is float16 [V,3]; ``triangles.pt`` int32 [T,3] (rover_utils.py:113-118).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .synth import KnnMap, Scene

_BASE = os.path.join("tasks", "utils", "terrain")


def save_reference_assets(scene: Scene, root: str) -> None:
    base = os.path.join(root, _BASE)
    for sub, m in (("knn_terrain", scene.terrain), ("knn_rocks", scene.rocks)):
        d = os.path.join(base, sub)
        os.makedirs(d, exist_ok=True)
        torch.save(m.map_indices.permute(2, 0, 1).contiguous().to(torch.int32), os.path.join(d, "map_indices.pt"))
        torch.save(m.triangles.to(torch.int32), os.path.join(d, "triangles.pt"))
        torch.save(m.vertices.to(torch.float16), os.path.join(d, "vertices.pt"))
    np.save(os.path.join(base, "stone_info.npy"), scene.stone_info_raw)
    torch.save(scene.heightmap, os.path.join(base, "heightmap_tensor.pt"))


def _load_map(d: str, cell_size: float) -> KnnMap:
    idx = torch.load(os.path.join(d, "map_indices.pt"), map_location="cpu")
    idx = idx.swapaxes(0, 1).swapaxes(1, 2).contiguous()          # [K,X,Y] -> [X,Y,K], camera.py:157-158
    tris = torch.load(os.path.join(d, "triangles.pt"), map_location="cpu")
    verts = torch.load(os.path.join(d, "vertices.pt"), map_location="cpu")
    return KnnMap(idx.to(torch.int32), tris.to(torch.int32), verts.to(torch.float16), cell_size)


def load_reference_assets(root: str, cell_size: float = 0.1, horizontal_scale: float = 0.025,
                          vertical_scale: float = 1.0, shift=(0.0, 0.0, 0.0)) -> Scene:
    base = os.path.join(root, _BASE)
    return Scene(terrain=_load_map(os.path.join(base, "knn_terrain"), cell_size),
                 rocks=_load_map(os.path.join(base, "knn_rocks"), cell_size),
                 stone_info_raw=np.load(os.path.join(base, "stone_info.npy")),
                 heightmap=torch.load(os.path.join(base, "heightmap_tensor.pt"), map_location="cpu").float(),
                 horizontal_scale=horizontal_scale, vertical_scale=vertical_scale, shift=tuple(shift))


def read_stone_info(path: str, device="cpu") -> torch.Tensor:
    """utils/terrain_utils/terrain_utils.py:416-424: append radius = max(extent_x, extent_y) / 4 -> [S,7] f32."""
    from .synth import read_stone_info_array
    return torch.from_numpy(read_stone_info_array(np.load(path))).to(device)


# ---------------------------------------------------------------------------------------------------------
# mesh ingestion + KNN map building ("next" row f-3): what rover_utils.py does with open3d / pymeshlab
# ---------------------------------------------------------------------------------------------------------
_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def load_ply(path: str):
    """Minimal PLY reader (ascii / binary_little_endian, triangle faces): -> (vertices [V,3] float32, faces [T,3] int32).
    Stands in for ``o3d.io.read_triangle_mesh`` (rover_utils.py:63-66) and ``pymeshlab`` (:187-195)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                elements[-1][2].append(tok[1:])
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        verts = faces = None
        for name, count, props in elements:
            is_list = any(p[0] == "list" for p in props)
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                if name == "vertex":
                    cols = [p[-1] for p in props]
                    ix = [cols.index(c) for c in ("x", "y", "z")]
                    verts = np.asarray([[float(r[i]) for i in ix] for r in rows], dtype=np.float32)
                elif name == "face":
                    if any(int(r[0]) != 3 for r in rows):
                        raise ValueError(f"{path}: only triangle faces are supported")
                    faces = np.asarray([[int(v) for v in r[1:4]] for r in rows], dtype=np.int32)
            else:
                if not is_list:
                    dt = np.dtype([(p[-1], "<" + _PLY_TYPES[p[0]]) for p in props])
                    data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                    if name == "vertex":
                        verts = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
                else:
                    p = props[0]
                    dt = np.dtype([("n", "<" + _PLY_TYPES[p[1]]), ("v", "<" + _PLY_TYPES[p[2]], (3,))])
                    data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                    if (data["n"] != 3).any():
                        raise ValueError(f"{path}: only triangle faces are supported")
                    if name == "face":
                        faces = data["v"].astype(np.int32)
        if verts is None or faces is None:
            raise ValueError(f"{path}: needs a vertex and a face element")
        return verts, faces


def build_knn_map(engine, vertices, triangles, n_cells: int = 600, res: float = 0.1, k: int = 200, ranking: str = "exact_f32",
                  cell_x_f16=None, cell_y_f16=None) -> KnnMap:
    """``_get_knn_triangles`` (rover_utils.py:52-123) on the GPU: K nearest triangle centroids per map cell, returned in
    the in-memory form of one ``knn_*`` directory (vertices rounded to fp16 like :113).  ``ranking="reference_fp16"`` ranks
    like the reference (fp16 centroids, fp16 cell coordinates, fp16 distances, rover_utils.py:71-102) — pinned against maps
    built by the reference's own code (tests/golden/knn_*.npz); the default ranks exact f32 distances."""
    idx = engine.build_knn_map(vertices, triangles, n_cells, n_cells, res, k, ranking=ranking, cell_x_f16=cell_x_f16,
                               cell_y_f16=cell_y_f16)
    return KnnMap(idx.cpu(), torch.as_tensor(np.asarray(triangles), dtype=torch.int32),
                  torch.as_tensor(np.asarray(vertices), dtype=torch.float32).to(torch.float16), res)


def generate_knn_triangles(engine, terrain_dir: str, res_x: int = 600, res_y: int = 600, res: float = 0.1,
                           n_triangles: int = 200, files=(("map.ply", "knn_terrain"), ("big_stones.ply", "knn_rocks")),
                           ranking: str = "exact_f32"):
    """``generate_knn_triangles`` of the reference (rover_utils.py:48-50): for ``map.ply`` and ``big_stones.ply`` in
    ``terrain_dir`` build the K-nearest-triangle map on the GPU and write ``map_indices.pt`` [K,X,Y] int32,
    ``vertices.pt`` fp16 and ``triangles.pt`` int32 into ``knn_terrain/`` and ``knn_rocks/`` (rover_utils.py:113-118) —
    the files ``Camera`` / ``Rock_Detection`` (and ``load_reference_assets`` here) open.  No open3d / pymeshlab needed.
    Returns {sub-directory: KnnMap}."""
    if res_x != res_y:
        raise ValueError("the reference's maps are square (camera.py:243 clamps both axes with the dim-0 size)")
    out = {}
    for ply, sub in files:
        vertices, triangles = load_ply(os.path.join(terrain_dir, ply))
        m = build_knn_map(engine, vertices, triangles, n_cells=res_x, res=res, k=n_triangles, ranking=ranking)
        d = os.path.join(terrain_dir, sub)
        os.makedirs(d, exist_ok=True)
        torch.save(m.map_indices.permute(2, 0, 1).contiguous().to(torch.int32), os.path.join(d, "map_indices.pt"))
        torch.save(m.vertices.to(torch.float16), os.path.join(d, "vertices.pt"))
        torch.save(m.triangles.to(torch.int32), os.path.join(d, "triangles.pt"))
        out[sub] = m
    return out


def build_irregular_scene(engine, spec, k: int = 200, ranking: str = "exact_f32"):
    """A ``synth.IrregularSpec`` scene whose two KNN maps are built on the GPU by ``rover_build_knn_map`` (terrain: the whole
    mesh = ``map.ply``; rocks: its rocks-only sub-mesh = ``big_stones.ply``, rover_utils.py:48-50) -> (Scene, height function)."""
    from . import synth
    verts, tris, rock_tris, _stones = synth.irregular_mesh(spec)
    n_x, n_y = int(round(spec.extent_x / 0.1)), int(round(spec.extent_y / 0.1))
    maps = [engine.build_knn_map(verts, t, n_x, n_y, 0.1, k, ranking=ranking).cpu() for t in (tris, rock_tris)]
    return synth.make_irregular_scene(spec, k, maps[0], maps[1])
