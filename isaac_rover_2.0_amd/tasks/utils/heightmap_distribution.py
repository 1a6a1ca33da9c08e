"""Rover-local heightmap sample points (init-time, host).

Mirrors the interface of the reference's ``Heightmap`` class
(``tasks/utils/camera/heightmap_distribution.py:11-134``): ``get_distribution()``, ``get_sparse_vector()``,
``get_dense_vector()``, ``get_num_sparse_vector()``, ``get_num_dense_vector()``, ``coarse_idx`` / ``fine_idx``.

The point set is defined by the reference's generator (``:36-115``): a coarse 0.15 m lattice inside a wedge
∩ radius 3.5 m, plus a fine 0.05 m lattice inside a box in front of the rover, both walked with
*accumulating* float steps (``x += delta``), rounded to 4 decimals and with x/y swapped.  Because the lattice
coordinates are accumulated sums, the set is only reproducible by re-running the same float recurrences —
which this module does (pure Python floats = IEEE doubles), pinned bit for bit by
``tests/golden/heightmap_native.npz`` (1634 points, 634 sparse, 1112 dense; the counts are also hard-coded
at ``tasks/utils/learning_by_cheating/teacher_loader.py:42-47``).
"""
from __future__ import annotations

import math

import numpy as np
import torch

# border lines as (p0, p1, side) — values of heightmap_distribution.py:16-21
_COARSE = (((1.220, 0.118), (4.4455, 3.150), "over"), ((-1.220, 0.118), (-4.4455, 3.150), "over"),
           ((1.220, 0.118), (-1.220, 0.118), "over"))
_FINE = (((1.0, 0.118), (1.0, 0.119), "left"), ((-1.0, 0.118), (-1.0, 0.119), "right"),
         ((1.0, 0.118), (-1.0, 0.118), "over"), ((1.0, 1.400), (-1.0, 1.400), "below"))
_COARSE_RADIUS = 3.5
_DELTA_COARSE = 0.15
_DELTA_FINE = 0.05
_Z_OFFSET = -0.26878


def _line_params(lines):
    """Slope / intercept per border line, as heightmap_distribution.py:161-167 derives them."""
    out = []
    for (x0, y0), (x1, y1), side in lines:
        dx, dy = x0 - x1, y0 - y1
        slope = math.inf if dx == 0 else dy / dx
        icpt = y0 - slope * x0 if slope != math.inf else -math.inf
        out.append((slope, icpt, x0, side))
    return out


def _inside(x, y, params):
    """heightmap_distribution.py:153-193, including its quirk that 'left' tests the same inequality as 'right'
    on sloped lines."""
    for slope, icpt, x0, side in params:
        if slope == 0:
            if (y > icpt and side == "below") or (y < icpt and side == "over"):
                return False
        elif slope == math.inf:
            if (x < x0 and side == "right") or (x > x0 and side == "left"):
                return False
        else:
            yl = slope * x + icpt
            if (y < yl and side == "over") or (y > yl and side == "below"):
                return False
            if x < (y - icpt) / slope and side in ("right", "left"):
                return False
    return True


def _lattice(delta):
    """The reference's accumulating walk: y from -10 while < 10; x restarts at -10 and is bumped BEFORE use."""
    y = -10
    while y < 10:
        x = -10
        while x < 10:
            x += delta
            yield x, y
        y += delta


def generate_native():
    """-> (distribution [P,3] float64 in the post-swap frame, coarse_idx, fine_idx) of the reference."""
    coarse, fine = _line_params(_COARSE), _line_params(_FINE)
    pts = []
    for x, y in _lattice(_DELTA_COARSE):
        if _inside(x, y, coarse) and math.sqrt(x ** 2 + y ** 2) < _COARSE_RADIUS:
            pts.append((x, y))
    coarse_idx = list(range(len(pts)))          # every coarse point passes the coarse test again (:57-59)
    seen = set(pts)
    for x, y in _lattice(_DELTA_FINE):
        if _inside(x, y, fine) and (x, y) not in seen:
            pts.append((x, y))
            seen.add((x, y))
    fine_idx = [i for i, (x, y) in enumerate(pts) if _inside(x, y, fine)]     # over the WHOLE list (:76-78)
    arr = np.round(np.asarray([(x, y, _Z_OFFSET) for x, y in pts], dtype=np.float64), 4)
    return arr[:, [1, 0, 2]].copy(), np.asarray(coarse_idx, dtype=np.int64), np.asarray(fine_idx, dtype=np.int64)


class Heightmap:
    """Same accessors as the reference class; ``distribution`` may be replaced by a synthetic set."""

    def __init__(self, device="cpu", distribution=None, coarse_idx=None, fine_idx=None):
        self.device = device
        self.z_offset = _Z_OFFSET
        if distribution is None:
            distribution, coarse_idx, fine_idx = generate_native()
        self.distribution = torch.as_tensor(np.asarray(distribution), dtype=torch.float64, device=device)
        self.coarse_idx = torch.as_tensor(np.asarray(coarse_idx), dtype=torch.int64, device=device)
        self.fine_idx = torch.as_tensor(np.asarray(fine_idx), dtype=torch.int64, device=device)
        self.beneath_idx = torch.zeros(0, dtype=torch.int64, device=device)

    def get_distribution(self):
        return self.distribution

    def get_sparse_vector(self, rays):
        return rays[:, self.coarse_idx]

    def get_dense_vector(self, rays):
        return rays[:, self.fine_idx]

    def get_beneath_vector(self, rays):
        return rays[:, self.beneath_idx]

    def get_num_sparse_vector(self):
        return self.coarse_idx.shape[0]

    def get_num_dense_vector(self):
        return self.fine_idx.shape[0]

    def get_num_beneath_vector(self):
        return self.beneath_idx.shape[0]
