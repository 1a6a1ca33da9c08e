"""TEST INFRASTRUCTURE — ctypes front-end of oracle/rover_oracle.c (the CPU parity oracle).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module.  The product package never does (and fails loudly when its HIP library is missing).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librover_oracle.so")
_SRC = os.path.join(_HERE, "rover_oracle.c")


def build(force: bool = False) -> str:
    """gcc -O2 -ffp-contract=off: one IEEE rounding per operation, like one ATen kernel per op."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared",
                               "-fPIC", "-fvisibility=hidden", "-o", _SO, _SRC, "-lm"])
    return _SO


class _KnnMap(C.Structure):
    _fields_ = [("X", C.c_int32), ("Y", C.c_int32), ("K", C.c_int32), ("map_idx", C.c_void_p),
                ("tris", C.c_void_p), ("verts", C.c_void_p), ("cell", C.c_float),
                ("shift_x", C.c_float), ("shift_y", C.c_float)]


class _Cfg(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_envs_global", C.c_int32), ("P", C.c_int32), ("Ns", C.c_int32),
                ("Nd", C.c_int32), ("curriculum_level", C.c_int32), ("max_episode_length", C.c_int32),
                ("precision", C.c_int32), ("pos_reward", C.c_float), ("heading_contraint_reward", C.c_float),
                ("motion_contraint_reward", C.c_float), ("goal_angle_reward", C.c_float),
                ("boogie_contraint_reward", C.c_float)]


class _In(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pos", "quat", "joints", "target", "lin_hist", "ang_hist", "euler_pre",
                                          "progress", "distribution", "sparse_idx", "dense_idx")]


_OUT_FIELDS = ("euler", "heading", "ray_src", "ray_dist", "wheel_dist", "body_dist", "rock_collision", "obs", "rew",
               "reset", "ex_pos_reward", "ex_collision", "ex_upright", "ex_heading", "ex_motion", "ex_goal_angle",
               "ex_lin", "ex_ang")


class _Out(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _OUT_FIELDS]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _np(x, dtype):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class KnnMap:
    """Holds the reference-format arrays of one map alive for the C side."""

    def __init__(self, map_indices, triangles, vertices, cell=0.1, shift=(0.0, 0.0)):
        self.idx = _np(map_indices, np.int32)
        self.tris = _np(triangles, np.int32)
        v = vertices
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        self.verts = np.ascontiguousarray(v.astype(np.float16)).view(np.uint16)
        x, y, k = self.idx.shape
        self.c = _KnnMap(x, y, k, _p(self.idx), _p(self.tris), _p(self.verts), cell, shift[0], shift[1])


DEFAULT_REWARDS = dict(pos_reward=1.0, heading_contraint_reward=0.05, motion_contraint_reward=-0.01,
                       goal_angle_reward=0.3, boogie_contraint_reward=0.5)   # cfg/task/Rover.yaml:37-46

EXTRAS = ("pos_reward", "collision_penalty", "uprightness_penalty", "heading_contraint_penalty",
          "motion_contraint_penalty", "goal_angle_penalty", "torque_penalty_driving", "torque_penalty_steering")


def step(terrain: KnnMap, rocks: KnnMap, st: dict, distribution, sparse_idx, dense_idx, num_envs_global=None,
         curriculum_level=2, max_episode_length=3000, rewards=None, source_fp16=False, precision=None):
    """One post_physics_step on sim state ``st`` (keys as synth.make_states). Returns a dict of numpy arrays."""
    rw = dict(DEFAULT_REWARDS)
    rw.update(rewards or {})
    pos = _np(st["pos"], np.float32)
    e = pos.shape[0]
    dist = _np(distribution, np.float64)
    p = dist.shape[0]
    sp = _np(sparse_idx, np.int64)
    de = _np(dense_idx, np.int64)
    ins = dict(pos=pos, quat=_np(st["quat"], np.float32), joints=_np(st["joints"], np.float32),
               target=_np(st["target"], np.float32), lin_hist=_np(st["lin_hist"], np.float32),
               ang_hist=_np(st["ang_hist"], np.float32), euler_pre=_np(st["euler_pre"], np.float32),
               progress=_np(st["progress"], np.int64).copy(), distribution=dist, sparse_idx=sp, dense_idx=de)
    w = 4 + len(sp) + len(de)
    f32, i64 = np.float32, np.int64
    outs = dict(euler=np.zeros((e, 3), f32), heading=np.zeros(e, f32), ray_src=np.zeros((e, p, 3), f32),
                ray_dist=np.zeros((e, p), f32), wheel_dist=np.zeros((e, 24), f32), body_dist=np.zeros((e, 2), f32),
                rock_collision=np.zeros(e, i64), obs=np.zeros((e, w), f32), rew=np.zeros(e, f32),
                reset=np.zeros(e, i64), ex_pos_reward=np.zeros(e, f32), ex_collision=np.zeros(e, i64),
                ex_upright=np.zeros(e, f32), ex_heading=np.zeros(e, f32), ex_motion=np.zeros(e, f32),
                ex_goal_angle=np.zeros(e, f32), ex_lin=np.zeros(e, f32), ex_ang=np.zeros(e, f32))
    prec = {"fp32": 0, "fp16_sources": 1, "fp16_as_shipped": 2}[precision] if precision is not None else (1 if source_fp16 else 0)
    cfg = _Cfg(e, num_envs_global or e, p, len(sp), len(de), curriculum_level, max_episode_length, prec,
               rw["pos_reward"], rw["heading_contraint_reward"], rw["motion_contraint_reward"],
               rw["goal_angle_reward"], rw["boogie_contraint_reward"])
    cin = _In(*[_p(ins[n]) for n, _ in _In._fields_])
    cout = _Out(*[_p(outs[n]) for n in _OUT_FIELDS])
    lib().oracle_step(C.byref(cfg), C.byref(terrain.c), C.byref(rocks.c), C.byref(cin), C.byref(cout))
    res = dict(euler=outs["euler"], heading_diff=outs["heading"], ray_sources=outs["ray_src"],
               ray_dist=outs["ray_dist"], wheel_dist=outs["wheel_dist"], body_dist=outs["body_dist"],
               rock_collision=outs["rock_collision"], obs_buf=outs["obs"], rew_buf=outs["rew"],
               reset_buf=outs["reset"], progress_buf=ins["progress"])
    for name, key in zip(EXTRAS, ("ex_pos_reward", "ex_collision", "ex_upright", "ex_heading", "ex_motion",
                                  "ex_goal_angle", "ex_lin", "ex_ang")):
        res["extras_" + name] = outs[key]
    return res


def set_cell_index_mode(mode):
    """'cpu_div' (default; x / 0.1 as ATen-CPU evaluates camera.py:241) or 'cuda_rcp' (x * (1 / 0.1) as ATen-CUDA does)."""
    lib().oracle_set_cell_index_mode({"cpu_div": 0, "cuda_rcp": 1}[mode])


def quat_to_euler(quat):
    q = _np(quat, np.float32)
    out = np.zeros((q.shape[0], 3), np.float32)
    lib().oracle_quat_to_euler(q.shape[0], _p(q), _p(out))
    return out


def raycast(m: KnnMap, src, dirs):
    s = _np(src, np.float32).reshape(-1, 3)
    d = _np(dirs, np.float32).reshape(-1, 3)
    out = np.zeros(s.shape[0], np.float32)
    lib().oracle_raycast(C.byref(m.c), s.shape[0], _p(s), _p(d), _p(out))
    return out


def raycast_unit(m: KnnMap, src, dneg, half=False):
    """Rays given as the HIP path's ray records hold them: origin and d = -normalize(direction).  ``half``: the as-shipped fp16
    arithmetic (src / dneg then hold fp16 values)."""
    s = _np(src, np.float32).reshape(-1, 3)
    d = _np(dneg, np.float32).reshape(-1, 3)
    out = np.zeros(s.shape[0], np.float32)
    lib().oracle_raycast_unit(C.byref(m.c), s.shape[0], _p(s), _p(d), 1 if half else 0, _p(out))
    return out


def clearance(info7, xy):
    info = _np(info7, np.float32)
    q = _np(xy, np.float32).reshape(-1, 2)
    out = np.zeros(q.shape[0], np.float32)
    lib().oracle_clearance(_p(info), info.shape[0], q.shape[0], _p(q), _p(out))
    return out


def shift_spawns(info7, pos3, max_iter=100000):
    info = _np(info7, np.float32)
    p = _np(pos3, np.float32).copy()
    it = lib().oracle_shift_spawns(_p(info), info.shape[0], p.shape[0], _p(p), max_iter)
    return p, it


def pos_height(heightmap, xy, hscale=0.025, vscale=1.0, shift=(0.0, 0.0)):
    hm = _np(heightmap, np.float32)
    q = _np(xy, np.float32).reshape(-1, 2)
    out = np.zeros(q.shape[0], np.float32)
    lib().oracle_pos_height(_p(hm), hm.shape[0], hm.shape[1], C.c_float(hscale), C.c_float(vscale),
                            C.c_float(shift[0]), C.c_float(shift[1]), q.shape[0], _p(q), _p(out))
    return out


def generate_goals(info7, env_ids, initial_pos3, draws, radius=8.0, target3=None):
    """draws: [n_draws, n] float32 uniforms, one row per torch.rand call of random_goals."""
    info = _np(info7, np.float32)
    ids = _np(env_ids, np.int64)
    ip = _np(initial_pos3, np.float32)
    dr = _np(draws, np.float32).reshape(-1, ids.shape[0])
    tgt = np.zeros_like(ip) if target3 is None else _np(target3, np.float32).copy()
    used = lib().oracle_generate_goals(_p(info), info.shape[0], ids.shape[0], _p(ids), _p(ip), C.c_float(radius),
                                       _p(dr), dr.shape[0], _p(tgt))
    return tgt, used


def compact(reset):
    r = _np(reset, np.int64)
    ids = np.zeros(r.shape[0], np.int64)
    n = lib().oracle_compact(r.shape[0], _p(r), _p(ids))
    return ids[:n]


def ackermann(lin, ang):
    l = _np(lin, np.float32)
    a = _np(ang, np.float32)
    steer = np.zeros((l.shape[0], 6), np.float32)
    vel = np.zeros((l.shape[0], 6), np.float32)
    lib().oracle_ackermann(l.shape[0], _p(l), _p(a), _p(steer), _p(vel))
    return steer, vel
