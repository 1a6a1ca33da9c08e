"""TEST INFRASTRUCTURE — `torch_ref`: the reference's post_physics_step as a tensor program on the CPU.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this; the product path never does.

What it is: an independent PyTorch restatement of the OP SEQUENCE the reference runs — gather the K triangles of every
ray's cell into a dense ``[E, p, K, 3, 3]`` tensor, tile rays K times, one batched Möller–Trumbore, ``min`` over K
(``tasks/utils/camera/camera.py:60-145``, ``tasks/utils/rock_detection/rock_detect.py:52-149``,
``tasks/utils/camera/ray_casting.py:3-66``), then the obs / reward / done arithmetic of ``tasks/rover.py:272-336,460-531,
610-672`` — written from the survey of those lines, not copied from them.  It is the CPU baseline that shows what the
reference's *approach* costs on the GPU box's host cores (the reference itself cannot travel there); the C oracle
(``rover_oracle.c``) is the per-ray restatement that pins parity.  ``dtype`` = ``torch.float32`` is the reference's fp32
mode, ``torch.float16`` its as-shipped ``Camera.dtype`` (every elementwise op rounds to fp16; slow on x86).

Checked against the C oracle in ``tests/test_oracle_golden.py::test_torch_ref_matches_the_c_oracle``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

NEG_EPS, ONE_EPS = -0.1, 1.1            # ray_casting.py:25-26 (become fp16(0.1) / fp16(1.1) as half tensors)


def _ray_distance(src, dirs, tri, dtype):
    """ray_casting.py:31-59 on [N,3], [N,3], [N,3,3] -> [N].  Thresholds are fp16 tensors in the reference whatever the
    dtype of the rays (ray_casting.py:3 default), so they carry their fp16 rounding in fp32 mode too."""
    neg = torch.tensor(NEG_EPS, dtype=torch.float16).to(dtype)
    one = torch.tensor(ONE_EPS, dtype=torch.float16).to(dtype)
    miss = (torch.tensor(ONE_EPS, dtype=torch.float16) * 10).to(dtype)               # :27 -> 11.0
    d = -torch.nn.functional.normalize(dirs, dim=1)                                  # :31
    a = tri[:, 2]                                                                    # :34
    b = tri[:, 1] - a                                                                # :35
    c = tri[:, 0] - a                                                                # :36
    g = src - a                                                                      # :37
    bc = torch.cross(b, c, dim=1)
    det = (bc * d).sum(1)                                                            # :40-41
    n = (torch.cross(g, c, dim=1) * d).sum(1) / det                                  # :44-45
    m = (torch.cross(b, g, dim=1) * d).sum(1) / det                                  # :49-50
    k = (bc * g).sum(1) / det                                                        # :54-55
    ok = (n >= neg) & (m >= neg) & (n + m <= one)                                    # :59
    return torch.where(ok, k, miss)


def _cell_lookup(map_indices, xy, cell, shift):
    """camera.py:233-264: half-even rounding, both axes clamped to the dim-0 size."""
    x_dim = map_indices.shape[0]
    ij = torch.round(torch.clamp((xy.float() - torch.tensor(shift, dtype=torch.float32)) / cell, 0, x_dim - 1)).long()
    ij[..., 1].clamp_(max=map_indices.shape[1] - 1)             # memory safety only (the reference would index out of range)
    return map_indices[ij[..., 0], ij[..., 1]]                  # [E, p, K]


def _cast(knn, src, dirs, dtype, partitions=4, shift=(0.0, 0.0)):
    """[E,p,3] origins, [E,p,3] directions against one KNN map -> [E,p] min distance (camera.py:77-120)."""
    e, p, _ = src.shape
    verts = knn.vertices.to(dtype)
    tri_ids = _cell_lookup(knn.map_indices.long(), src[:, :, 0:2], knn.cell_size, shift)
    out = []
    step = max(1, math.ceil(p / partitions))
    for p0 in range(0, p, step):
        ids = tri_ids[:, p0:p0 + step]                                               # [E,q,K]
        tri = verts[knn.triangles.long()[ids]]                                       # [E,q,K,3,3] — the big gather (:84)
        q, k = ids.shape[1], ids.shape[2]
        s = src[:, p0:p0 + step].to(dtype).unsqueeze(2).expand(e, q, k, 3).reshape(-1, 3)       # tiled K times (:94-101)
        d = dirs[:, p0:p0 + step].to(dtype).unsqueeze(2).expand(e, q, k, 3).reshape(-1, 3)
        dist = _ray_distance(s, d, tri.reshape(-1, 3, 3), dtype).reshape(e, q, k)
        out.append(dist.min(dim=2).values)                                           # :116-117
    return torch.cat(out, dim=1)


def _quat_to_euler(q):
    """tasks/utils/math/tensor_quat_to_euler.py:6-31 (w, x, y, z)."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = torch.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    sinp = 2 * (w * y - z * x)
    pitch = torch.where(sinp - 1 >= 0, torch.copysign(torch.tensor(math.pi / 2), sinp), torch.asin(sinp))
    yaw = torch.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return torch.stack((roll, pitch, yaw), dim=1)


def _body_xf(pts, eul, pos):
    """The ZYX transform with negated angles shared by camera.py:197-199 and rock_detect.py:305-307: [E,n,3] local -> world."""
    sx, cx = torch.sin(-eul[:, 0:1]), torch.cos(-eul[:, 0:1])
    sy, cy = torch.sin(-eul[:, 1:2]), torch.cos(-eul[:, 1:2])
    sz, cz = torch.sin(-eul[:, 2:3]), torch.cos(-eul[:, 2:3])
    dt = pts.dtype
    sx, cx, sy, cy, sz, cz = (t.to(dt) for t in (sx, cx, sy, cy, sz, cz))
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    a = y * cx + z * sx
    c = z * cx - y * sx
    b = x * cy - sy * c
    p = pos.to(dt)
    return torch.stack((p[:, 0:1] + sz * a + cz * b, p[:, 1:2] + cz * a - sz * b, p[:, 2:3] + x * sy + cy * c), dim=2)


_WHEEL_RAY = torch.tensor([[0.215 / 2, 0.130 / 2, 0.1], [0.215 / 2, -0.130 / 2, 0.1], [-0.215 / 2, 0.130 / 2, 0.1],
                           [-0.215 / 2, -0.130 / 2, 0.1]], dtype=torch.float32)              # rock_detect.py:193-197
_WP0 = torch.tensor([[0.286, 0.385, -0.197], [0.286, -0.385, -0.197], [-0.146, 0.447, -0.197], [-0.146, -0.447, -0.197],
                     [-0.440, 0.385, -0.197], [-0.440, -0.385, -0.197]], dtype=torch.float32)  # :201-206
_WP1 = torch.tensor([[0.153, 0, 0.03]] * 4 + [[0, 0, 0.03]] * 2, dtype=torch.float32)         # :210-215


def _wheel_rays(pos, eul, joints):
    """rock_detect.py:160-319: 24 ray origins [E,24,3] and their 24 directions (one per wheel, repeated 4 times)."""
    e = pos.shape[0]
    zero = torch.zeros(e)
    steer = torch.stack((joints[:, 4], joints[:, 6], zero, zero, -joints[:, 7], joints[:, 8]), dim=1)      # :248
    sus_y = torch.stack((-joints[:, 0], joints[:, 1], -joints[:, 0], joints[:, 1], zero, zero), dim=1)     # :263
    sus_x = torch.stack((zero, zero, zero, zero, -joints[:, 2], -joints[:, 2]), dim=1)                     # :264

    def chain(pt, with_offsets):                       # pt [6,n,3] wheel-local -> body frame [E,6,n,3]
        sst, cst = torch.sin(-steer)[:, :, None], torch.cos(-steer)[:, :, None]
        ssx, csx = torch.sin(sus_x)[:, :, None], torch.cos(sus_x)[:, :, None]
        ssy, csy = torch.sin(sus_y)[:, :, None], torch.cos(sus_y)[:, :, None]
        t0 = _WP0[None, :, None, :] if with_offsets else torch.zeros(1, 6, 1, 3)
        t1 = _WP1[None, :, None, :] if with_offsets else torch.zeros(1, 6, 1, 3)
        x, y, z = pt[None, ..., 0], pt[None, ..., 1], pt[None, ..., 2]
        x1 = t0[..., 0] + x * cst + y * sst                                                                 # :256-258
        y1 = t0[..., 1] + y * cst - x * sst
        z1 = t0[..., 2] + z
        c1 = z1 * csx - y1 * ssx
        return torch.stack((t1[..., 0] + x1 * csy - ssy * c1, t1[..., 1] + y1 * csx + z1 * ssx,
                            t1[..., 2] + x1 * ssy + csy * c1), dim=3)                                       # :275-277

    src_b = chain(_WHEEL_RAY[None].expand(6, 4, 3), True).reshape(e, 24, 3)
    dir_b = chain(torch.tensor([[[0.0, 0.0, -1.0]]]).expand(6, 1, 3), False).reshape(e, 6, 3)
    src = _body_xf(src_b, eul, pos)
    dirs = _body_xf(dir_b, eul, torch.zeros_like(pos))                                                       # :289-302,314
    return src, dirs.repeat_interleave(4, dim=1)


def step(scene, st, distribution, sparse_idx, dense_idx, dtype=torch.float32, num_envs_global=None, curriculum_level=2,
         max_episode_length=3000, rewards=None):
    """One post_physics_step of the reference on CPU tensors; returns a dict named like ``oracle.step``'s."""
    rw = dict(pos_reward=1.0, heading_contraint_reward=0.05, motion_contraint_reward=-0.01, goal_angle_reward=0.3,
              boogie_contraint_reward=0.5)
    rw.update(rewards or {})
    pos, quat, joints, target = (st[k].float() for k in ("pos", "quat", "joints", "target"))
    e = pos.shape[0]
    eul = _quat_to_euler(quat)
    # A4: terrain rays — the transform runs in float64 (the distribution's dtype), results cast to `dtype` (camera.py:165-212)
    pts = torch.as_tensor(np.asarray(distribution), dtype=torch.float64)
    p = pts.shape[0]
    local = torch.cat((pts, torch.tensor([[0.0, 0.0, -1.0]], dtype=torch.float64)))[None].expand(e, p + 1, 3)
    world = _body_xf(local, eul, pos)
    src = world[:, :p]
    dirs = (world[:, p:] - pos.double()[:, None, :]).expand(e, p, 3)
    ray_dist = _cast(scene.terrain, src.to(dtype), dirs.to(dtype), dtype, shift=scene.shift[0:2])        # :212 casts both
    # A6: 24 wheel + 2 body rays against the rocks map (rock_detect.py:52-149,321-371), one partition
    wsrc, wdir = _wheel_rays(pos, eul, joints)
    bsrc = _body_xf(torch.tensor([[0.340, 0, -0.01], [-0.485, 0, -0.01]])[None].expand(e, 2, 3), eul, pos)
    bdir = (_body_xf(torch.tensor([[0.0, 1.0, 0.0]])[None].expand(e, 1, 3), eul, pos) - pos[:, None, :]).expand(e, 2, 3)
    rsrc, rdir = torch.cat((wsrc, bsrc), dim=1), torch.cat((wdir, bdir), dim=1)
    rock = _cast(scene.rocks, rsrc.to(dtype), rdir.to(dtype), dtype, partitions=1, shift=scene.shift[0:2])   # :319,:371
    wheel_dist, body_dist = rock[:, :24], rock[:, 24:]
    # A7 check_collision (rover.py:663-668)
    coll = torch.zeros(e, dtype=torch.int64)
    if curriculum_level >= 2:
        coll = ((wheel_dist.min(dim=1).values.abs() < 0.8) | (body_dist.min(dim=1).values.abs() < 0.45)).long()
    # A1 obs (rover.py:279-283,320-325)
    yaw = eul[:, 2]
    tx, ty = target[:, 0] - pos[:, 0], target[:, 1] - pos[:, 1]
    hx, hy = torch.cos(yaw), torch.sin(yaw)
    heading = -torch.atan2(tx * hy - ty * hx, tx * hx + ty * hy)
    td = torch.sqrt(tx * tx + ty * ty)
    lin, ang = st["lin_hist"].float(), st["ang_hist"].float()
    sp, de = torch.as_tensor(np.asarray(sparse_idx)).long(), torch.as_tensor(np.asarray(dense_idx)).long()
    obs = torch.cat((torch.stack((td / 9, heading / math.pi, lin[:, 0], ang[:, 0]), dim=1),
                     (ray_dist[:, sp] / 2).float(), (ray_dist[:, de] / 2).float()), dim=1)
    # A8 calculate_metrics (rover.py:460-531)
    progress = st["progress"].long() + 1                                                  # rl_task.py:250
    heading_pen = torch.where(lin[:, 0] < 0, -1.0, 0.0) * rw["heading_contraint_reward"]
    goal_pen = torch.where(heading.abs() > 2, -(heading * 0.3 * rw["goal_angle_reward"]).abs(), torch.zeros(e))
    dl, da = (lin[:, 0] * 3 - 3 * lin[:, 1]).abs(), (ang[:, 0] * 3 - 3 * ang[:, 1]).abs()
    p1, p2 = torch.where(dl > 0.05, dl * dl, torch.zeros(e)), torch.where(da > 0.05, da * da, torch.zeros(e))
    motion = (p1 * p1) * rw["motion_contraint_reward"] + (p2 * p2) * rw["motion_contraint_reward"]
    pos_rew = 1.0 / (1.0 + (0.33 * 0.33) * td * td) * rw["pos_reward"]
    pos_rew = torch.where(td <= 0.18, 1.03 * (max_episode_length - progress).float(), pos_rew)
    reward = pos_rew + heading_pen + motion + goal_pen
    if curriculum_level >= 2:
        reward = torch.where(coll == 1, reward - 300.0, reward)
    reward = reward / 3000.0
    # A10 is_done (rover.py:610-647)
    ep = st["euler_pre"].float()
    reset = (progress >= max_episode_length) | (ep[:, 0].abs() >= 0.78 * 1.5) | (ep[:, 1].abs() >= 0.78 * 1.5) | (td >= 11) | (td <= 0.18)
    if curriculum_level >= 2:
        reset = reset | (coll == 1)
    return dict(euler=eul.numpy(), heading_diff=heading.numpy(), ray_dist=ray_dist.float().numpy(),
                wheel_dist=wheel_dist.float().numpy(), body_dist=body_dist.float().numpy(), rock_collision=coll.numpy(),
                obs_buf=obs.numpy(), rew_buf=reward.numpy(), reset_buf=reset.long().numpy(), progress_buf=progress.numpy())
