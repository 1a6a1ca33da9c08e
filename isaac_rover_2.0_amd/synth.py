"""Synthetic Mars-yard scene in the reference's on-disk formats.

The reference ships none of its terrain assets (``/root/reference/.gitignore:8-22``),
so the bench, the parity fixtures and the smoke test all run on a scene built here:

* an analytic heightfield triangulated with the diagonal convention of
  ``utils/terrain_utils/terrain_utils.py:355-367`` (two triangles per grid cell),
  vertices stored fp16 like ``tasks/utils/rover_utils.py:113``;
* the "K nearest triangle centroids per 0.1 m cell" map that
  ``tasks/utils/rover_utils.py:52-118`` builds offline (``map_indices`` [X,Y,K] int32,
  ``triangles`` [T,3] int32, ``vertices`` [V,3] fp16);
* a rocks-only map built the same way from the triangles that lie inside the
  ``stone_info`` discs;
* ``stone_info`` [S,6] (centre xyz, extents xy, unused) as ``read_stone_info``
  (``utils/terrain_utils/terrain_utils.py:416-424``) expects to find on disk;
* the 0.025 m heightfield that ``rover.py:210-213`` loads for spawn / goal z.

Everything is deterministic: the nearest-centroid ranking is done in exact integer
arithmetic (centroids of a regular grid mesh sit on a 1/3-cell lattice), ties broken
by triangle id, so CPU and GPU builds of a scene are bit-identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class KnnMap:
    """One of the reference's ``knn_terrain`` / ``knn_rocks`` directories, in memory."""
    map_indices: torch.Tensor   # [X, Y, K] int32 (the layout camera.py:156-158 swaps to)
    triangles: torch.Tensor     # [T, 3] int32
    vertices: torch.Tensor      # [V, 3] float16
    cell_size: float = 0.1


@dataclass
class Scene:
    terrain: KnnMap
    rocks: KnnMap
    stone_info_raw: np.ndarray  # [S, 6] float64, what stone_info.npy holds
    heightmap: torch.Tensor     # [N0, N1] float32 at 0.025 m
    horizontal_scale: float = 0.025
    vertical_scale: float = 1.0
    shift: tuple = (0.0, 0.0, 0.0)


def surface_height(i, j, noise=None):
    """h[i,j] of SURVEY.md §8d; i, j are vertex indices on the 0.1 m grid."""
    h = 0.30 * np.sin(0.15 * i) + 0.20 * np.cos(0.11 * j)
    if noise is not None:
        h = h + noise
    return h


def grid_mesh(n_vert: int, seed: int = 0):
    """Vertices/triangles of an n_vert x n_vert heightfield at 0.1 m spacing.

    Triangle order and winding follow terrain_utils.py:355-367:
    even triangle of cell (i,j) = (ind0, ind3, ind1), odd = (ind0, ind2, ind3).
    """
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(n_vert), np.arange(n_vert), indexing="ij")
    noise = 0.05 * rng.standard_normal((n_vert, n_vert))
    hf = surface_height(ii.astype(np.float64), jj.astype(np.float64), noise)
    verts = np.zeros((n_vert * n_vert, 3), dtype=np.float32)
    verts[:, 0] = (ii * 0.1).reshape(-1)
    verts[:, 1] = (jj * 0.1).reshape(-1)
    verts[:, 2] = hf.reshape(-1)
    nc = n_vert - 1
    ci, cj = np.meshgrid(np.arange(nc), np.arange(nc), indexing="ij")
    ind0 = (ci * n_vert + cj).reshape(-1)
    ind1 = ind0 + 1
    ind2 = ind0 + n_vert
    ind3 = ind2 + 1
    tris = np.empty((2 * nc * nc, 3), dtype=np.int32)
    tris[0::2, 0] = ind0
    tris[0::2, 1] = ind3
    tris[0::2, 2] = ind1
    tris[1::2, 0] = ind0
    tris[1::2, 1] = ind2
    tris[1::2, 2] = ind3
    return verts, tris, hf


def _centroid_lattice(n_vert: int, device):
    """Integer centroid coordinates (units of 1/3 cell) of every grid-mesh triangle."""
    nc = n_vert - 1
    ci, cj = torch.meshgrid(torch.arange(nc, device=device), torch.arange(nc, device=device), indexing="ij")
    ci = ci.reshape(-1)
    cj = cj.reshape(-1)
    cx = torch.stack((3 * ci + 1, 3 * ci + 2), dim=1).reshape(-1)   # even: (i,i+1,i) ; odd: (i,i+1,i+1)
    cy = torch.stack((3 * cj + 2, 3 * cj + 1), dim=1).reshape(-1)   # even: (j,j+1,j+1) ; odd: (j,j,j+1)
    return cx.to(torch.int64), cy.to(torch.int64)


def knn_map_from_subset(n_cells: int, cx: torch.Tensor, cy: torch.Tensor, tri_ids: torch.Tensor,
                        k: int, chunk_cells: int = 2048) -> torch.Tensor:
    """Exact K nearest centroids (xy) per cell among the triangles ``tri_ids``.

    Semantics of rover_utils.py:68-108 (cell (x,y) sits at (x*res, y*res); rank by
    euclidean distance of the triangle centre), evaluated with integer keys
    ``d2 * T + local_rank`` so the order is total and device independent.
    Returns [n_cells, n_cells, k] int32 of *global* triangle ids.
    """
    device = cx.device
    t = tri_ids.numel()
    if t < k:
        raise ValueError(f"need at least K={k} triangles, got {t}")
    sx = cx[tri_ids]
    sy = cy[tri_ids]
    out = torch.empty((n_cells * n_cells, k), dtype=torch.int32, device=device)
    cells = torch.arange(n_cells * n_cells, device=device)
    for s in range(0, n_cells * n_cells, chunk_cells):
        c = cells[s:s + chunk_cells]
        px = (3 * (c // n_cells)).unsqueeze(1)
        py = (3 * (c % n_cells)).unsqueeze(1)
        d2 = (sx.unsqueeze(0) - px) ** 2 + (sy.unsqueeze(0) - py) ** 2
        key = d2 * t + torch.arange(t, device=device).unsqueeze(0)
        top = torch.topk(key, k, dim=1, largest=False, sorted=True).values
        out[s:s + chunk_cells] = tri_ids[(top % t)].to(torch.int32)
    return out.reshape(n_cells, n_cells, k)


def knn_map_grid(n_cells: int, n_vert: int, k: int, device, chunk_rows: int = 8) -> torch.Tensor:
    """Same ranking as :func:`knn_map_from_subset` over ALL triangles of the grid mesh,
    restricted to a window that provably holds the K nearest (also at the map border)."""
    nc = n_vert - 1
    w = 2 * int(math.ceil(math.sqrt(k / (2.0 * math.pi)))) + 2
    t_total = 2 * nc * nc
    offs = torch.arange(-w, w + 1, device=device)
    oi, oj, parity = torch.meshgrid(offs, offs, torch.arange(2, device=device), indexing="ij")
    oi = oi.reshape(-1)
    oj = oj.reshape(-1)
    parity = parity.reshape(-1)
    out = torch.empty((n_cells, n_cells, k), dtype=torch.int32, device=device)
    big = torch.iinfo(torch.int64).max
    ys = torch.arange(n_cells, device=device)
    for r0 in range(0, n_cells, chunk_rows):
        xs = torch.arange(r0, min(r0 + chunk_rows, n_cells), device=device)
        gx, gy = torch.meshgrid(xs, ys, indexing="ij")
        gx = gx.reshape(-1, 1)
        gy = gy.reshape(-1, 1)
        # mesh cell that contains the map cell position (clamped into the mesh)
        ti = gx.clamp(max=nc - 1) + oi.unsqueeze(0)
        tj = gy.clamp(max=nc - 1) + oj.unsqueeze(0)
        valid = (ti >= 0) & (ti < nc) & (tj >= 0) & (tj < nc)
        tid = 2 * (ti * nc + tj) + parity.unsqueeze(0)
        ccx = 3 * ti + 1 + parity.unsqueeze(0)
        ccy = 3 * tj + 2 - parity.unsqueeze(0)
        d2 = (ccx - 3 * gx) ** 2 + (ccy - 3 * gy) ** 2
        key = torch.where(valid, d2 * t_total + tid, torch.full_like(d2, big))
        top = torch.topk(key, k, dim=1, largest=False, sorted=True).values
        if bool((top[:, -1] == big).any()):
            raise ValueError("KNN window too small for this K / map size")
        out[xs[0]:xs[-1] + 1] = (top % t_total).to(torch.int32).reshape(len(xs), n_cells, k)
    return out


def make_stones(n_stones: int, extent_m: float, seed: int = 2) -> np.ndarray:
    """stone_info.npy content: [S,6] = centre xyz, extents (U(0.2,1.7)) x2, unused."""
    rng = np.random.default_rng(seed)
    info = np.zeros((n_stones, 6), dtype=np.float64)
    info[:, 0:2] = rng.uniform(0.0, extent_m, size=(n_stones, 2))
    info[:, 3:5] = rng.uniform(0.2, 1.7, size=(n_stones, 2))
    return info


def read_stone_info_array(raw: np.ndarray) -> np.ndarray:
    """Host half of read_stone_info (terrain_utils.py:416-424): append radius = max(ext)/4."""
    rs = np.maximum(raw[:, 3], raw[:, 4]) / 4.0
    return np.concatenate([raw, rs[:, None]], axis=1).astype(np.float32)


def make_scene(n_cells: int = 600, k: int = 200, n_stones: int = 1024, device="cpu",
               heightmap_cells: int | None = None, seed: int = 0) -> Scene:
    """Bench scene: n_cells x n_cells map cells at 0.1 m, n_cells+1 vertices per side."""
    device = torch.device(device)
    n_vert = n_cells + 1
    verts, tris, _ = grid_mesh(n_vert, seed=seed)
    vertices = torch.from_numpy(verts).to(torch.float16)
    triangles = torch.from_numpy(tris)
    extent = n_cells * 0.1

    terrain_idx = knn_map_grid(n_cells, n_vert, k, device).cpu()

    raw = make_stones(n_stones, extent, seed=2)
    info = read_stone_info_array(raw)
    # triangles whose centroid lies inside a stone disc form the rocks-only mesh
    cx, cy = _centroid_lattice(n_vert, device)
    px = cx.to(torch.float32) * (0.1 / 3.0)
    py = cy.to(torch.float32) * (0.1 / 3.0)
    inside = torch.zeros_like(px, dtype=torch.bool)
    st = torch.from_numpy(info).to(device)
    for s0 in range(0, n_stones, 64):
        blk = st[s0:s0 + 64]
        d = torch.sqrt((px.unsqueeze(1) - blk[:, 0]) ** 2 + (py.unsqueeze(1) - blk[:, 1]) ** 2)
        inside |= (d <= blk[:, 6]).any(dim=1)
    rock_ids = torch.nonzero(inside).squeeze(1)
    if rock_ids.numel() < k:  # tiny fixtures: top up with the lowest-id triangles
        extra = torch.arange(cx.numel(), device=device)
        extra = extra[~inside][: k - rock_ids.numel()]
        rock_ids = torch.sort(torch.cat([rock_ids, extra])).values
    rocks_idx = knn_map_from_subset(n_cells, cx, cy, rock_ids, k).cpu()

    # 0.025 m heightfield resampled from the same analytic surface (no noise term:
    # it is only used for spawn / goal z, rover.py:216-218,582-583)
    n_h = heightmap_cells if heightmap_cells is not None else n_cells * 4
    hi = np.arange(n_h, dtype=np.float64) * 0.25
    hm = surface_height(hi[:, None], hi[None, :]).astype(np.float32)

    return Scene(
        terrain=KnnMap(terrain_idx, triangles, vertices),
        rocks=KnnMap(rocks_idx, triangles.clone(), vertices.clone()),
        stone_info_raw=raw,
        heightmap=torch.from_numpy(hm),
    )


# ---------------------------------------------------------------------------------------------------------------------
# Irregular scene: a decimated-style mesh (what the reference's real assets are: a heightfield with gaussian rocks,
# utils/terrain_utils/terrain_generation.py:18-65,104-153, reduced by pymeshlab's quadric edge collapse, :217-243).
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class IrregularSpec:
    """Parameters of :func:`irregular_mesh`; everything is derived from these and ``seed`` (numpy Generator + Qhull)."""
    extent_x: float = 10.0
    extent_y: float = 10.0
    n_rocks: int = 14
    fine: float = 0.0375            # finest target vertex spacing (on the rocks)
    coarse: float = 1.2             # coarsest (far from every rock): triangles that span many 0.1 m cells
    growth: float = 0.45            # target spacing grows by this much per metre of distance from the nearest rock
    seed: int = 0
    rock_r: tuple = (0.08, 0.5)     # rock radius range [m]
    rock_h: tuple = (0.10, 0.60)    # rock height range [m]
    frac_sliver_pts: float = 0.01   # extra points 1-5 mm from an existing one: needle triangles
    n_dup: int = 24                 # duplicated vertices (same coordinates, another index) re-used by part of their fan
    n_degenerate: int = 12          # explicit zero-area triangles (two equal indices / three collinear vertices)
    flip_frac: float = 0.10         # triangles stored with the other winding


def _irregular_rocks(spec: IrregularSpec, rng):
    n = spec.n_rocks
    xy = np.stack((rng.uniform(0.3, spec.extent_x - 0.3, n), rng.uniform(0.3, spec.extent_y - 0.3, n)), axis=1)
    nc = min(4, n)                                                # the first rocks sit in the central third (fixtures park rovers on them)
    xy[:nc, 0] = rng.uniform(0.36 * spec.extent_x, 0.64 * spec.extent_x, nc)
    xy[:nc, 1] = rng.uniform(0.36 * spec.extent_y, 0.64 * spec.extent_y, nc)
    r = rng.uniform(spec.rock_r[0], spec.rock_r[1], n)
    h = rng.uniform(spec.rock_h[0], spec.rock_h[1], n)
    p = rng.choice(np.array([2.0, 4.0, 6.0]), n)                  # 2: gaussian bump; 4, 6: flat top, flanks up to ~80 degrees
    return xy, r, h, p


def irregular_height(spec: IrregularSpec):
    """-> (z(x, y) in metres for float64 arrays, rocks = (xy [n,2], radius [n], height [n], exponent [n]))."""
    rng = np.random.default_rng(10_000 + spec.seed)
    rocks = _irregular_rocks(spec, rng)
    n_hill = 6
    hill_xy = np.stack((rng.uniform(0, spec.extent_x, n_hill), rng.uniform(0, spec.extent_y, n_hill)), axis=1)
    hill_a = rng.uniform(-0.35, 0.35, n_hill)
    hill_s = rng.uniform(1.2, 3.0, n_hill)

    def z(x, y):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        out = 0.05 * np.sin(1.7 * x) * np.cos(1.3 * y)
        for (cx, cy), a, sg in zip(hill_xy, hill_a, hill_s):
            out = out + a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2.0 * sg * sg))
        rxy, rr, rh, rp = rocks
        if len(rr) <= 64:                                          # (the fixtures' scenes: every rock at every point)
            for (cx, cy), r_, h_, p_ in zip(rxy, rr, rh, rp):
                d = np.sqrt((x - cx) ** 2 + (y - cy) ** 2) / (0.75 * r_)
                out = out + h_ * np.exp(-(d ** p_))
            return out
        # many rocks (bench-size scenes): a rock only reaches 4 r (exp(-(4 / 0.75)^2) < 1e-12)
        from scipy.spatial import cKDTree
        xb, yb = np.broadcast_arrays(x, y)
        shape = xb.shape
        pts = np.stack((xb.reshape(-1), yb.reshape(-1)), axis=1)
        out = np.broadcast_to(out, shape).reshape(-1).copy()
        tree = cKDTree(pts)
        for (cx, cy), r_, h_, p_ in zip(rxy, rr, rh, rp):
            ids = np.asarray(tree.query_ball_point([cx, cy], 4.0 * r_), dtype=np.int64)
            if ids.size:
                d = np.sqrt((pts[ids, 0] - cx) ** 2 + (pts[ids, 1] - cy) ** 2) / (0.75 * r_)
                out[ids] += h_ * np.exp(-(d ** p_))
        return out.reshape(shape)

    return z, rocks


def irregular_mesh(spec: IrregularSpec):
    """Vertices [V,3] float32, triangles [T,3] int32 of an irregular, decimated-style terrain mesh, its rocks-only sub-mesh
    (own triangle list, shared vertex table) and the stone list — deterministic in ``spec``.

    * non-uniform Delaunay triangulation of jittered multi-level point sets: vertex spacing ``fine`` on the rocks, growing to
      ``coarse`` away from them (triangle edges from ~0.03 m to > 1 m: some triangles cover hundreds of 0.1 m map cells);
    * gaussian / flat-topped rock bumps with flanks up to ~80 degrees on top of broad hills;
    * needle triangles (points 1-5 mm apart), duplicated vertices, zero-area triangles, mixed windings;
    * triangle and vertex ids shuffled (no spatial order in either table).
    """
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(20_000 + spec.seed)
    zf, rocks = irregular_height(spec)
    rxy, rr, rh, rp = rocks

    rock_tree = None
    if len(rr) > 64:
        from scipy.spatial import cKDTree
        rock_tree = cKDTree(rxy)

    def rock_gap(x, y):
        """distance to the nearest rock disc (radius 1.2 r), 0 inside one"""
        if rock_tree is None:
            d = np.full(x.shape, np.inf)
            for (cx, cy), r_ in zip(rxy, rr):
                d = np.minimum(d, np.maximum(0.0, np.sqrt((x - cx) ** 2 + (y - cy) ** 2) - 1.2 * r_))
            return d
        dist, idx = rock_tree.query(np.stack((x, y), axis=1), k=8)       # many rocks: the 8 nearest centres decide
        return np.maximum(0.0, dist - 1.2 * rr[idx]).min(axis=1)

    def target_spacing(x, y):
        return np.clip(spec.fine + spec.growth * rock_gap(x, y), spec.fine, spec.coarse)

    pts = []
    s_l = spec.coarse
    first = True
    while s_l >= spec.fine * 0.999:
        nx, ny = int(np.ceil(spec.extent_x / s_l)) + 1, int(np.ceil(spec.extent_y / s_l)) + 1
        gx, gy = np.meshgrid(np.arange(nx) * s_l, np.arange(ny) * s_l, indexing="ij")
        q = np.stack((gx.reshape(-1), gy.reshape(-1)), axis=1) + rng.uniform(-0.35, 0.35, (nx * ny, 2)) * s_l
        keep = np.ones(len(q), dtype=bool) if first else target_spacing(q[:, 0], q[:, 1]) <= s_l
        pts.append(q[keep])
        first = False
        s_l *= 0.5
    # the frame: corners + edge points, so that the triangulation covers the whole map rectangle
    ex, ey = spec.extent_x, spec.extent_y
    fr = [np.stack((np.linspace(0, ex, 17), np.zeros(17)), 1), np.stack((np.linspace(0, ex, 17), np.full(17, ey)), 1),
          np.stack((np.zeros(15), np.linspace(0, ey, 17)[1:-1]), 1), np.stack((np.full(15, ex), np.linspace(0, ey, 17)[1:-1]), 1)]
    p2 = np.concatenate(pts + fr, axis=0)
    p2[:, 0] = np.clip(p2[:, 0], 0.0, ex)
    p2[:, 1] = np.clip(p2[:, 1], 0.0, ey)
    n_sl = int(spec.frac_sliver_pts * len(p2))
    if n_sl:
        src = p2[rng.integers(0, len(p2), n_sl)]
        ang = rng.uniform(0, 2 * np.pi, n_sl)
        off = rng.uniform(1e-3, 5e-3, n_sl)
        p2 = np.concatenate([p2, np.clip(src + np.stack((off * np.cos(ang), off * np.sin(ang)), 1), 0.0, [ex, ey])], axis=0)
    p2 = np.unique(np.round(p2, 9), axis=0)
    tri = Delaunay(p2)
    tris = tri.simplices.astype(np.int64)
    verts = np.concatenate([p2, zf(p2[:, 0], p2[:, 1])[:, None]], axis=1)
    # duplicated vertices: a copy takes over part of the triangle fan of the original
    if spec.n_dup:
        orig = rng.choice(len(verts), spec.n_dup, replace=False)
        base = len(verts)
        verts = np.concatenate([verts, verts[orig]], axis=0)
        for k, o in enumerate(orig):
            rows, cols = np.nonzero(tris == o)
            take = rng.random(len(rows)) < 0.5
            tris[rows[take], cols[take]] = base + k
    # explicit degenerate triangles
    if spec.n_degenerate:
        a = rng.integers(0, len(verts), (spec.n_degenerate, 2))
        deg = np.stack((a[:, 0], a[:, 0], a[:, 1]), axis=1)
        tris = np.concatenate([tris, deg], axis=0)
    flip = rng.random(len(tris)) < spec.flip_frac
    tris[flip] = tris[flip][:, ::-1]
    # shuffle both tables
    vperm = rng.permutation(len(verts))
    inv = np.empty_like(vperm)
    inv[vperm] = np.arange(len(verts))
    verts = verts[vperm]
    tris = inv[tris]
    tris = tris[rng.permutation(len(tris))]
    verts32 = verts.astype(np.float32)
    tris32 = np.ascontiguousarray(tris.astype(np.int32))
    # rocks-only sub-mesh (the reference's big_stones.ply): triangles whose centroid lies within 1.2 r of a rock centre
    c = verts[tris].mean(axis=1)
    if rock_tree is None:
        inside = np.zeros(len(tris), dtype=bool)
        for (cx, cy), r_ in zip(rxy, rr):
            inside |= (c[:, 0] - cx) ** 2 + (c[:, 1] - cy) ** 2 <= (1.2 * r_) ** 2
    else:
        inside = rock_gap(c[:, 0], c[:, 1]) <= 0.0
    rock_tris = np.ascontiguousarray(tris32[inside])
    stones = np.zeros((len(rr), 6), dtype=np.float64)             # stone_info.npy: centre xyz, extents, unused
    stones[:, 0:2] = rxy
    stones[:, 3] = 4.0 * rr                                        # read_stone_info: radius = max(extents) / 4
    stones[:, 4] = 3.0 * rr
    return verts32, tris32, rock_tris, stones


def knn_map_bruteforce(verts: np.ndarray, tris: np.ndarray, n_x: int, n_y: int, k: int, res: float = 0.1,
                       chunk_cells: int = 1024) -> torch.Tensor:
    """Exact K nearest triangle centroids (xy) per map cell: float64 distances, ties by triangle id — the definition
    rover_utils.py:68-108 ranks in fp16.  [n_x, n_y, k] int32.  For small scenes (CPU tests); the GPU builder
    (rover_build_knn_map) computes the same ranking on f32 squared distances."""
    v = np.asarray(verts, dtype=np.float64)
    t = np.asarray(tris, dtype=np.int64)
    c = torch.from_numpy((v[t[:, 0], 0:2] + v[t[:, 1], 0:2] + v[t[:, 2], 0:2]) / 3.0)
    n_t = c.shape[0]
    if n_t < k:
        raise ValueError(f"need at least K={k} triangles, got {n_t}")
    out = torch.empty((n_x * n_y, k), dtype=torch.int32)
    cells = torch.arange(n_x * n_y)
    for s0 in range(0, n_x * n_y, chunk_cells):
        cc = cells[s0:s0 + chunk_cells]
        px = (cc // n_y).double().unsqueeze(1) * res
        py = (cc % n_y).double().unsqueeze(1) * res
        d2 = (c[:, 0].unsqueeze(0) - px) ** 2 + (c[:, 1].unsqueeze(0) - py) ** 2
        # stable ranking: sort by (d2, id) — argsort of d2 with a stable sort keeps id order among equal distances
        order = torch.sort(d2, dim=1, stable=True).indices[:, :k]
        out[s0:s0 + chunk_cells] = order.to(torch.int32)
    return out.reshape(n_x, n_y, k)


def shuffle_triangle_ids(scene: Scene, seed: int = 0) -> Scene:
    """The same scene with the triangle tables of both maps in a random order (ids in ``map_indices`` follow): what a mesh
    file without any spatial order looks like to the library.  Every result of the step is unchanged."""
    g = torch.Generator().manual_seed(seed)

    def one(m: KnnMap) -> KnnMap:
        t = m.triangles.shape[0]
        perm = torch.randperm(t, generator=g)                      # new position p holds old triangle perm[p]
        inv = torch.empty(t, dtype=torch.int64)
        inv[perm] = torch.arange(t)
        return KnnMap(inv[m.map_indices.long()].to(torch.int32), m.triangles[perm].contiguous(), m.vertices, m.cell_size)

    return Scene(terrain=one(scene.terrain), rocks=one(scene.rocks), stone_info_raw=scene.stone_info_raw, heightmap=scene.heightmap,
                 horizontal_scale=scene.horizontal_scale, vertical_scale=scene.vertical_scale, shift=scene.shift)


def make_irregular_scene(spec: IrregularSpec, k: int = 200, terrain_idx=None, rocks_idx=None) -> tuple:
    """Scene on the irregular mesh of ``spec`` -> (Scene, height function).  The KNN maps are ``terrain_idx`` / ``rocks_idx``
    ([X, Y, K] int32: e.g. the reference's own ``_get_knn_triangles`` output from a fixture, or the GPU builder's) or, when None,
    :func:`knn_map_bruteforce`."""
    verts, tris, rock_tris, stones = irregular_mesh(spec)
    zf, _ = irregular_height(spec)
    n_x, n_y = int(round(spec.extent_x / 0.1)), int(round(spec.extent_y / 0.1))
    if terrain_idx is None:
        terrain_idx = knn_map_bruteforce(verts, tris, n_x, n_y, k)
    if rocks_idx is None:
        rocks_idx = knn_map_bruteforce(verts, rock_tris, n_x, n_y, k)
    vertices = torch.from_numpy(verts).to(torch.float16)
    hx = np.arange(n_x * 4, dtype=np.float64) * 0.025
    hy = np.arange(n_y * 4, dtype=np.float64) * 0.025
    hm = zf(hx[:, None], hy[None, :]).astype(np.float32)
    scene = Scene(terrain=KnnMap(torch.as_tensor(terrain_idx, dtype=torch.int32), torch.from_numpy(tris), vertices),
                  rocks=KnnMap(torch.as_tensor(rocks_idx, dtype=torch.int32), torch.from_numpy(rock_tris), vertices.clone()),
                  stone_info_raw=stones, heightmap=torch.from_numpy(hm))
    return scene, zf


# fp16(-0.1) and fp16(1.1) as exact fractions: the padded barycentric thresholds of ray_casting.py:59 and the
# values its det / n / m substitutions compare against (:46,:51,:56)
NEG_EPS_H = 819.0 / 8192.0          # 0.0999755859375
ONE_EPS_H = 1126.0 / 1024.0         # 1.099609375


def make_lattice_scene(k: int = 48, n_cells: int = 64) -> Scene:
    """Adversarial scene for exact-tie parity: every coordinate is a small dyadic rational, so rays from
    lattice-aligned poses meet vertices, edges, the padded barycentric thresholds and the reference's
    ``det ==`` guards EXACTLY (no rounding anywhere in fp32).  On top of a 0.25 m base lattice it holds:

    * S1  a unit right triangle (thresholds n, m = -fp16(0.1) and n + m = fp16(1.1) are hit exactly);
    * S2 / S3  right triangles whose doubled area is fp16(0.1) / fp16(1.1): |det| of a vertical ray equals the
      guard constants of ray_casting.py:46,:51,:56;
    * S4  zero-area and collinear triangles (det = 0); S5 a vertical wall (det = 0 for vertical rays);
    * every special in both windings, S1 also duplicated (equal distances inside one K list).

    The K nearest centroids per 0.1 m cell are ranked on integer keys, ties by triangle id (deterministic).
    """
    nv = 27                                                    # 0 .. 6.5 m at 0.25 m
    ii, jj = np.meshgrid(np.arange(nv), np.arange(nv), indexing="ij")
    base = np.zeros((nv * nv, 3), dtype=np.float64)
    base[:, 0] = (0.25 * ii).reshape(-1)
    base[:, 1] = (0.25 * jj).reshape(-1)
    base[:, 2] = np.where((ii + jj) % 3 == 0, 0.125, 0.0).reshape(-1)
    nc = nv - 1
    ci, cj = np.meshgrid(np.arange(nc), np.arange(nc), indexing="ij")
    i0 = (ci * nv + cj).reshape(-1)
    tris = [np.stack((i0, i0 + nv + 1, i0 + 1), axis=1), np.stack((i0, i0 + nv, i0 + nv + 1), axis=1)]
    verts = [base]
    n_v = base.shape[0]

    def add(points, faces):
        nonlocal n_v
        verts.append(np.asarray(points, dtype=np.float64))
        f = np.asarray(faces, dtype=np.int64) + n_v
        tris.append(f)
        tris.append(f[:, ::-1].copy())                         # the other winding
        n_v += len(points)

    z = 0.5
    add([(1, 1, z), (2, 1, z), (1, 2, z)], [(0, 1, 2), (0, 1, 2)])                                   # S1 (+ duplicate)
    add([(0.125, 1, z), (0.125 + 819.0 / 1024.0, 1, z), (0.125, 1.125, z)], [(0, 1, 2)])            # S2: 2A = fp16(0.1)
    add([(0.5, 3, z), (0.5 + ONE_EPS_H, 3, z), (0.5, 4, z)], [(0, 1, 2)])                            # S3: 2A = fp16(1.1)
    add([(1, 3, z), (1.5, 3, z), (2, 3, z)], [(0, 0, 1), (0, 1, 0), (0, 1, 2)])                       # S4: degenerate
    add([(2, 2, 0.25), (2, 2.5, 0.25), (2, 2, 1.0), (2, 2.5, 1.0)], [(0, 1, 2), (1, 3, 2)])          # S5: vertical wall
    v64 = np.concatenate(verts, axis=0)
    vertices = torch.from_numpy(v64.astype(np.float32)).to(torch.float16)
    assert bool((vertices.double() == torch.from_numpy(v64)).all()), "lattice vertices must be fp16-exact"
    t_all = np.concatenate(tris, axis=0).astype(np.int32)
    triangles = torch.from_numpy(t_all)

    # integer keys: coordinates in units of 2^-13 m, cell positions x * 0.1 m rounded to the same grid
    q = np.round(v64 * 8192.0).astype(np.int64)
    c3 = q[t_all[:, 0]] + q[t_all[:, 1]] + q[t_all[:, 2]]                      # 3 x centroid
    cell = np.round(np.arange(n_cells, dtype=np.float64) * 0.1 * 8192.0).astype(np.int64) * 3
    n_t = t_all.shape[0]
    tid = np.arange(n_t, dtype=np.int64)
    special = tid[2 * nc * nc:]
    rock_ids = np.concatenate([tid[(tid % 7 == 0) & (tid < 2 * nc * nc)], special])      # rocks map: a sparse subset

    def knn(ids):
        out = np.empty((n_cells, n_cells, k), dtype=np.int32)
        cx, cy = c3[ids, 0], c3[ids, 1]
        for x in range(n_cells):
            dx2 = (cx - cell[x]) ** 2
            for y in range(n_cells):
                key = (dx2 + (cy - cell[y]) ** 2) * n_t + ids
                out[x, y] = ids[np.argsort(key, kind="stable")[:k]]
        return torch.from_numpy(out)

    raw = make_stones(8, n_cells * 0.1, seed=5)
    hm = np.zeros((n_cells * 4, n_cells * 4), dtype=np.float32)
    return Scene(terrain=KnnMap(knn(tid), triangles, vertices),
                 rocks=KnnMap(knn(rock_ids), triangles.clone(), vertices.clone()),
                 stone_info_raw=raw, heightmap=torch.from_numpy(hm))


def lattice_distribution():
    """9 dyadic sample offsets (0, +-0.125)^2, 0.25 m below the body frame; all in the sparse block."""
    pts = [(dx, dy, -0.25) for dx in (0.0, 0.125, -0.125) for dy in (0.0, 0.125, -0.125)]
    return (np.asarray(pts, dtype=np.float64), np.arange(9, dtype=np.int64), np.zeros(0, dtype=np.int64))


def lattice_states(num_general: int = 24, seed: int = 11):
    """Identity-orientation poses whose centre ray meets the special spots of :func:`make_lattice_scene` exactly
    (plus one f32 ulp either side of each threshold), followed by ``num_general`` randomly oriented poses."""
    f32 = np.float32
    spots = []
    for a in range(4):                                          # base lattice: vertices and edge midpoints
        for b in range(3):
            spots.append((3.0 + 0.25 * a, 0.5 + 0.125 * b))
    lo, hi = 1.0 - NEG_EPS_H, 1.0 + 563.0 / 1024.0             # S1: n or m = -fp16(0.1);  n + m = fp16(1.1)
    for base in (lo, 2.0 + NEG_EPS_H):
        for v in (np.nextafter(f32(base), f32(-9)), f32(base), np.nextafter(f32(base), f32(9))):
            spots.append((float(v), 1.25))
            spots.append((1.25, float(v)))
    for v in (np.nextafter(f32(hi), f32(-9)), f32(hi), np.nextafter(f32(hi), f32(9)), f32(hi) + f32(2.0 ** -21)):
        spots.append((float(v), hi))        # + 1 ulp still rounds n + m to fp16(1.1) (tie to even); + 4 ulp does not
    spots += [(1.0, 1.0), (2.0, 1.0), (1.0, 2.0), (1.5, 1.5), (1.25, 1.25)]         # S1 corners, hypotenuse, inside
    spots += [(0.25, 1.03125), (0.5, 1.0625)]                                       # S2 (det guard fp16(0.1))
    spots += [(0.75, 3.25), (1.0, 3.5), (0.5, 3.0)]                                 # S3 (det guard fp16(1.1))
    spots += [(1.25, 3.0), (1.5, 3.0), (1.0, 3.0)]                                  # S4 degenerate
    spots += [(2.0, 2.125), (2.0, 2.0), (2.0, 2.5)]                                 # S5 in the wall's plane
    n_exact = len(spots)
    st = make_states(n_exact + num_general, 6.4, seed=seed)
    st["pos"][:n_exact, 0:2] = torch.tensor(spots, dtype=torch.float32)
    st["pos"][:n_exact, 2] = 1.5
    st["quat"][:n_exact] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    st["joints"][:n_exact] = 0.0
    st["pos"][n_exact:, 2] = 1.25
    return st, n_exact


def quat_from_euler(roll, pitch, yaw):
    """(w,x,y,z) of the ZYX rotation; inverse of tensor_quat_to_euler.py:17-29."""
    cr, sr = torch.cos(roll / 2), torch.sin(roll / 2)
    cp, sp = torch.cos(pitch / 2), torch.sin(pitch / 2)
    cy, sy = torch.cos(yaw / 2), torch.sin(yaw / 2)
    return torch.stack((cr * cp * cy + sr * sp * sy,
                        sr * cp * cy - cr * sp * sy,
                        cr * sp * cy + sr * cp * sy,
                        cr * cp * sy - sr * sp * cy), dim=1).to(torch.float32)


def make_states(num_envs: int, extent_m: float, seed: int, heightfn=None, margin_m: float | None = None):
    """Per-env sim state of SURVEY.md §8d (host tensors, float32 / int64)."""
    g = torch.Generator().manual_seed(1000 + seed)
    lo, hi = 5.0, extent_m - 5.0
    if margin_m is not None:
        lo, hi = margin_m, extent_m - margin_m
    elif hi <= lo:
        lo, hi = 0.25 * extent_m, 0.75 * extent_m
    pos = torch.empty(num_envs, 3)
    pos[:, 0:2] = lo + (hi - lo) * torch.rand(num_envs, 2, generator=g)
    if heightfn is not None:                # z of the surface under the rover from the scene's own height function (metres)
        pos[:, 2] = torch.from_numpy(np.asarray(heightfn(pos[:, 0].numpy().astype(np.float64),
                                                         pos[:, 1].numpy().astype(np.float64)))).float() + 0.5
    else:
        i = (pos[:, 0] / 0.1).numpy().astype(np.float64)
        j = (pos[:, 1] / 0.1).numpy().astype(np.float64)
        pos[:, 2] = torch.from_numpy(surface_height(i, j)).float() + 0.5
    roll = 0.1 * torch.randn(num_envs, generator=g)
    pitch = 0.1 * torch.randn(num_envs, generator=g)
    yaw = (2 * torch.rand(num_envs, generator=g) - 1) * math.pi
    quat = quat_from_euler(roll, pitch, yaw)
    joints = 0.1 * torch.randn(num_envs, 13, generator=g)
    alpha = 2 * math.pi * torch.rand(num_envs, generator=g)
    target = pos.clone()
    target[:, 0] += 8 * torch.cos(alpha)
    target[:, 1] += 8 * torch.sin(alpha)
    lin_hist = 2 * torch.rand(num_envs, 3, generator=g) - 1
    ang_hist = 2 * torch.rand(num_envs, 3, generator=g) - 1
    progress = torch.randint(0, 3001, (num_envs,), generator=g, dtype=torch.int64)
    euler_pre = torch.stack((roll, pitch, yaw), dim=1) + 0.01 * torch.randn(num_envs, 3, generator=g)
    return dict(pos=pos.float(), quat=quat, joints=joints.float(), target=target.float(),
                lin_hist=lin_hist.float(), ang_hist=ang_hist.float(), progress=progress,
                euler_pre=euler_pre.float())


def ray_distribution(name: str):
    """Rover-local sample points for the BASELINE.json configs.

    Returns (points [P,3] float64, sparse_idx, dense_idx) in the reference's
    *post-swap* frame (x forward), i.e. what Heightmap.get_distribution() returns.
    """
    z = -0.26878
    if name == "9":
        pts = [(x, y, z) for x in (0.5, 1.0, 1.5) for y in (-0.5, 0.0, 0.5)]
        sparse = list(range(9))
        dense = []
    elif name == "37":
        pts = [(0.2, 0.0, z)]
        for r in (1.0, 2.0, 3.0):
            for a in range(12):
                th = math.radians(-82.5 + 15.0 * a)
                pts.append((round(r * math.cos(th), 4), round(r * math.sin(th), 4), z))
        sparse = list(range(37))
        dense = []
    elif name == "120":
        pts = [(round(0.15 + 0.1 * a, 4), round(-0.55 + 0.1 * b, 4), z) for a in range(10) for b in range(12)]
        sparse = []
        dense = list(range(120))
    else:
        raise ValueError(name)
    return (np.asarray(pts, dtype=np.float64), np.asarray(sparse, dtype=np.int64),
            np.asarray(dense, dtype=np.int64))


def knn_test_mesh(kind: str):
    """Meshes of the KNN-builder parity fixtures (tests/golden/knn_*.npz were captured from the reference's own
    ``_get_knn_triangles`` on exactly these): -> (vertices [V,3] f32, triangles [T,3] i32, builder arguments)."""
    if kind == "grid10m":                # regular 0.1 m grid mesh over 10 m x 10 m, the terrain generator's triangulation
        verts, tris, _ = grid_mesh(101, seed=1)
        return verts.astype(np.float32), tris.astype(np.int32), dict(res_x=100, res_y=100, res=0.1, n_triangles=16)
    if kind == "soup50m":                # irregular triangle soup over 50 m: coordinates where fp16 spacing is 0.03 m
        rng = np.random.default_rng(77)
        n = 6000
        centers = rng.uniform(-1.0, 50.0, (n, 1, 2))
        verts = np.concatenate([centers + rng.normal(0, 0.3, (n, 3, 2)), rng.normal(0, 0.2, (n, 3, 1))], axis=2)
        return verts.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.int32).reshape(n, 3), \
            dict(res_x=100, res_y=100, res=0.5, n_triangles=12)
    raise ValueError(kind)
