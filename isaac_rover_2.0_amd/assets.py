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
