"""ctypes binding of librover_step.so (C ABI in include/rover_step.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this module raises.
Tensors are PyTorch-ROCm tensors; only their ``data_ptr()`` crosses the boundary, and work is
enqueued on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "librover_step.so")

MAP_TERRAIN, MAP_ROCKS = 0, 1
STEP_INCREMENT_PROGRESS, STEP_COMPACT = 1, 2

EXTRAS = ("pos_reward", "collision_penalty", "uprightness_penalty", "heading_contraint_penalty",
          "motion_contraint_penalty", "goal_angle_penalty", "torque_penalty_driving", "torque_penalty_steering")


class RoverError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_envs_global", C.c_int32), ("env_offset", C.c_int32),
                ("device", C.c_int32), ("curriculum_level", C.c_int32), ("max_episode_length", C.c_int32),
                ("pos_reward", C.c_float), ("heading_contraint_reward", C.c_float),
                ("motion_contraint_reward", C.c_float), ("goal_angle_reward", C.c_float),
                ("boogie_contraint_reward", C.c_float)]


class StepIn(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pos", "quat", "joints", "target", "lin_hist", "ang_hist", "euler_pre",
                                          "progress")]


class StepOut(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("obs_stride", C.c_int64), ("rew", C.c_void_p), ("reset", C.c_void_p),
                ("rock_collision", C.c_void_p), ("ex_pos_reward", C.c_void_p), ("ex_collision_penalty", C.c_void_p),
                ("ex_uprightness_penalty", C.c_void_p), ("ex_heading_contraint_penalty", C.c_void_p),
                ("ex_motion_contraint_penalty", C.c_void_p), ("ex_goal_angle_penalty", C.c_void_p),
                ("ex_torque_penalty_driving", C.c_void_p), ("ex_torque_penalty_steering", C.c_void_p),
                ("reset_ids", C.c_void_p), ("n_reset", C.c_void_p), ("euler", C.c_void_p),
                ("heading_diff", C.c_void_p), ("ray_dist", C.c_void_p), ("wheel_dist", C.c_void_p),
                ("body_dist", C.c_void_p), ("stone_collision", C.c_void_p), ("stone_margin", C.c_float),
                ("done_u8", C.c_void_p), ("ray_src", C.c_void_p), ("hit_pt", C.c_void_p)]


class Info(C.Structure):
    _fields_ = [("P", C.c_int32), ("Ns", C.c_int32), ("Nd", C.c_int32), ("rays_per_env_padded", C.c_int32),
                ("K", C.c_int32 * 2), ("K8", C.c_int32 * 2), ("X", C.c_int32 * 2), ("Y", C.c_int32 * 2),
                ("table_bytes", C.c_uint64 * 2), ("workspace_bytes", C.c_uint64), ("raycast_variant", C.c_int32),
                ("cell_index_mode", C.c_int32), ("ray_precision", C.c_int32), ("raycast_sorted", C.c_int32),
                ("raycast_rocks_staged", C.c_int32)]


class CullInfo(C.Structure):
    _fields_ = [("triangles", C.c_int64 * 2), ("always_candidate_triangles", C.c_int64 * 2), ("cells_without_cone", C.c_int64 * 2),
                ("rays", C.c_uint64), ("candidate_pairs", C.c_uint64), ("rays_both_tests", C.c_uint64), ("bins", C.c_uint64),
                ("max_pairs_per_run", C.c_uint64), ("queue_bytes", C.c_uint64), ("launches_per_step", C.c_uint64), ("rays_far_skipped", C.c_uint64),
                ("cells_with_far_bound", C.c_int64 * 2), ("far_records_on_demand", C.c_uint64), ("rays_not_scanned", C.c_uint64),
                ("lane_items", C.c_uint64), ("lane_flushes", C.c_uint64)]


class ChainDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_stride", C.c_int64), ("K0", C.c_int32), ("n_layers", C.c_int32), ("weights", C.c_void_p),
                ("biases", C.c_void_p), ("widths", C.c_void_p), ("activations", C.c_void_p), ("y", C.c_void_p), ("y_stride", C.c_int64)]


class ResetIO(C.Structure):
    _fields_ = [("reset_ids", C.c_void_p), ("n_reset_dev", C.c_void_p), ("n_reset_host", C.c_int32),
                ("initial_pos3", C.c_void_p), ("pos3", C.c_void_p), ("quat4", C.c_void_p), ("joint_pos13", C.c_void_p),
                ("joint_vel13", C.c_void_p), ("base_pos3", C.c_void_p), ("reset", C.c_void_p), ("progress", C.c_void_p),
                ("yaw_deg", C.c_void_p), ("target3", C.c_void_p), ("radius", C.c_float), ("draws", C.c_void_p),
                ("max_draws", C.c_int32), ("seed", C.c_uint64), ("n_draws_used", C.c_void_p), ("yaw_deg_len", C.c_int32),
                ("seed_dev", C.c_void_p)]


class Profile(C.Structure):
    _fields_ = [("raycast_ms", C.c_double), ("launches", C.c_int32), ("pairs_per_launch", C.c_uint64)]


# every symbol include/rover_step.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "rover_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(_P)]),
    "rover_destroy": (None, [_P]),
    "rover_last_error": (C.c_char_p, [_P]),
    "rover_version": (C.c_char_p, []),
    "rover_set_knn_map": (C.c_int, [_P, C.c_int, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32,
                                    C.c_float, C.c_float, C.c_float]),
    "rover_set_distribution": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32]),
    "rover_set_heightfield": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float]),
    "rover_set_stones": (C.c_int, [_P, _P, C.c_int32]),
    "rover_set_curriculum_level": (C.c_int, [_P, C.c_int32]),
    "rover_step": (C.c_int, [_P, C.POINTER(StepIn), C.POINTER(StepOut), C.c_uint32, _P]),
    "rover_get_observations": (C.c_int, [_P, C.POINTER(StepIn), C.POINTER(StepOut), _P]),
    "rover_calculate_metrics": (C.c_int, [_P, C.POINTER(StepIn), C.POINTER(StepOut), _P]),
    "rover_is_done": (C.c_int, [_P, C.POINTER(StepIn), C.POINTER(StepOut), _P]),
    "rover_compact_resets": (C.c_int, [_P, _P, _P, _P, _P]),
    "rover_get_depths": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "rover_get_collisions": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "rover_export_rays": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "rover_cast_rays": (C.c_int, [_P, _P, _P, _P, _P]),
    "rover_quat_to_euler": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "rover_clearance": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "rover_shift_spawns": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "rover_sample_height": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "rover_generate_goals": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_float, _P, C.c_int32, C.c_uint64, _P, _P]),
    "rover_reset_envs": (C.c_int, [_P, C.POINTER(ResetIO), _P]),
    "rover_pre_physics_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rover_ackermann": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P]),
    "rover_get_info": (C.c_int, [_P, C.POINTER(Info)]),
    "rover_get_cull_info": (C.c_int, [_P, C.POINTER(CullInfo)]),
    "rover_replay_raycast": (C.c_int, [_P, _P]),
    "rover_build_knn_map": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P]),
    "rover_build_knn_map_ref": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, _P, _P]),
    "rover_linear_forward": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    "rover_mlp_chain_forward": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "rover_mlp_chain_pair_forward": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P]),
    "rover_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "rover_set_profiling": (C.c_int, [_P, C.c_int32]),
    "rover_get_profile": (C.c_int, [_P, C.POINTER(Profile)]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("rover_capi.cpp", "rover_kernels.hip", "rover_cull.hip", "rover_mlp.hip", "rover_internal.h",
                                             "rover_raymath.h", "build.sh")]
    srcs.append(os.path.join(os.path.dirname(_CSRC), "..", "include", "rover_step.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["bash", os.path.join(_CSRC, "build.sh")])
    return LIB_PATH


def load():
    """dlopen librover_step.so and bind every declared symbol; raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RoverError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback for the rover step path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def version() -> str:
    """rover_version(): "rover_step <ver> (gfx950) src-<12 hex>": the hash of the sources the loaded library was built from."""
    return load().rover_version().decode()


def source_hash() -> str:
    """The same hash computed from the source files in the tree (what a fresh build.sh would embed)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rover_capi.cpp", "rover_kernels.hip", "rover_cull.hip", "rover_mlp.hip", "rover_internal.h", "rover_raymath.h"):
        h.update(open(os.path.join(_CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_CSRC), "..", "include", "rover_step.h"), "rb").read())
    return h.hexdigest()[:12]


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device_index=None):
    """The current stream of the device as the void* the C ABI takes.  torch.cuda.current_stream() costs ~3 us of Python per call —
    a fifth of a small batch's step when a step makes three calls —, the raw accessor a tenth of that."""
    if _raw_stream is not None and device_index is not None:
        return C.c_void_p(_raw_stream(device_index))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _host(a, dtype):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


class Engine:
    """One rover_ctx: owns the device tables, launches the step kernels on torch's current stream."""

    def __init__(self, num_envs, device=0, num_envs_global=0, env_offset=0, curriculum_level=2,
                 max_episode_length=3000, rewards=None):
        self.lib = load()
        rw = dict(pos_reward=1.0, heading_contraint_reward=0.05, motion_contraint_reward=-0.01,
                  goal_angle_reward=0.3, boogie_contraint_reward=0.5)
        rw.update({k: v for k, v in (rewards or {}).items() if k in rw})
        if isinstance(device, torch.device):
            device = device.index or 0
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", int(device))
        self._dev_index = int(device)
        self.cfg = Cfg(self.num_envs, int(num_envs_global), int(env_offset), int(device), int(curriculum_level),
                       int(max_episode_length), rw["pos_reward"], rw["heading_contraint_reward"],
                       rw["motion_contraint_reward"], rw["goal_angle_reward"], rw["boogie_contraint_reward"])
        h = C.c_void_p()
        rc = self.lib.rover_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise RoverError(f"rover_create failed ({rc}): {self.lib.rover_last_error(None).decode()}")
        self._h = h
        self.P = self.Ns = self.Nd = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rover_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RoverError(f"{what} failed ({rc}): {self.lib.rover_last_error(self._h).decode()}")

    # ---- tables -------------------------------------------------------------------------------
    def set_knn_map(self, which, map_indices, triangles, vertices, cell_size=0.1, shift=(0.0, 0.0)):
        idx = _host(map_indices, np.int32)
        tris = _host(triangles, np.int32)
        v = vertices.detach().cpu().numpy() if isinstance(vertices, torch.Tensor) else np.asarray(vertices)
        verts = np.ascontiguousarray(v.astype(np.float16)).view(np.uint16)
        if idx.ndim != 3 or tris.ndim != 2 or tris.shape[1] != 3 or verts.ndim != 2 or verts.shape[1] != 3:
            raise RoverError("set_knn_map: expected map_indices [X,Y,K], triangles [T,3], vertices [V,3]")
        x, y, k = idx.shape
        self._check(self.lib.rover_set_knn_map(self._h, which, idx.ctypes.data, x, y, k, tris.ctypes.data,
                                               tris.shape[0], verts.ctypes.data, verts.shape[0], cell_size,
                                               float(shift[0]), float(shift[1])), "rover_set_knn_map")

    def set_distribution(self, points, sparse_idx, dense_idx):
        pts = _host(points, np.float64)
        sp = _host(sparse_idx, np.int64)
        de = _host(dense_idx, np.int64)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise RoverError("set_distribution: points must be [P,3]")
        self._check(self.lib.rover_set_distribution(self._h, pts.ctypes.data, pts.shape[0], sp.ctypes.data, len(sp),
                                                    de.ctypes.data, len(de)), "rover_set_distribution")
        self.P, self.Ns, self.Nd = pts.shape[0], len(sp), len(de)

    def set_heightfield(self, heightmap, horizontal_scale=0.025, vertical_scale=1.0, shift=(0.0, 0.0)):
        hm = _host(heightmap, np.float32)
        self._check(self.lib.rover_set_heightfield(self._h, hm.ctypes.data, hm.shape[0], hm.shape[1],
                                                   horizontal_scale, vertical_scale, float(shift[0]), float(shift[1])),
                    "rover_set_heightfield")

    def set_stones(self, info7):
        info = _host(info7, np.float32)
        if info.ndim != 2 or info.shape[1] != 7:
            raise RoverError("set_stones: expected [S,7] (read_stone_info output)")
        self._check(self.lib.rover_set_stones(self._h, info.ctypes.data, info.shape[0]), "rover_set_stones")

    def set_curriculum_level(self, level):
        self._check(self.lib.rover_set_curriculum_level(self._h, int(level)), "rover_set_curriculum_level")

    def set_scene(self, scene, distribution):
        """Convenience: load a synth.Scene + (points, sparse_idx, dense_idx)."""
        from . import synth
        sh = scene.shift[0:2]
        self.set_knn_map(MAP_TERRAIN, scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices,
                         scene.terrain.cell_size, sh)
        self.set_knn_map(MAP_ROCKS, scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices,
                         scene.rocks.cell_size, sh)
        self.set_distribution(*distribution)
        self.set_heightfield(scene.heightmap, scene.horizontal_scale, scene.vertical_scale, sh)
        self.set_stones(synth.read_stone_info_array(scene.stone_info_raw))

    @property
    def num_observations(self):
        return 4 + self.Ns + self.Nd

    def info(self):
        i = Info()
        self._check(self.lib.rover_get_info(self._h, C.byref(i)), "rover_get_info")
        return i

    def cull_info(self):
        """Diagnostics of the culled ray cast (rover_get_cull_info) as a dict; synchronises the device."""
        i = CullInfo()
        self._check(self.lib.rover_get_cull_info(self._h, C.byref(i)), "rover_get_cull_info")
        d = {k: (list(getattr(i, k)) if k in ("triangles", "always_candidate_triangles", "cells_without_cone", "cells_with_far_bound") else int(getattr(i, k)))
             for k, _ in CullInfo._fields_}
        d["pairs_per_ray"] = d["candidate_pairs"] / d["rays"] if d["rays"] else 0.0
        return d

    # ---- step ---------------------------------------------------------------------------------
    def _chk(self, t, shape, dtype, name):
        if t is None:
            return
        if not t.is_cuda or t.device != self.device:
            raise RoverError(f"{name}: expected a tensor on {self.device}, got {t.device}")
        if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise RoverError(f"{name}: expected contiguous {dtype} {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")

    def make_in(self, pos, quat, joints, target, lin_hist, ang_hist, euler_pre, progress):
        e, f = self.num_envs, torch.float32
        for t, s, n in ((pos, (e, 3), "pos"), (quat, (e, 4), "quat"), (joints, (e, 13), "joints"),
                        (target, (e, 3), "target"), (lin_hist, (e, 3), "lin_hist"), (ang_hist, (e, 3), "ang_hist"),
                        (euler_pre, (e, 3), "euler_pre")):
            self._chk(t, s, f, n)
        self._chk(progress, (e,), torch.int64, "progress")
        tensors = (pos, quat, joints, target, lin_hist, ang_hist, euler_pre, progress)
        sin = StepIn(*[_ptr(t) for t in tensors])
        sin._keep = tensors          # the struct holds raw device pointers: keep the tensors alive as long as it lives
        return sin

    def make_out(self, obs, rew=None, reset=None, rock_collision=None, extras=None, reset_ids=None, n_reset=None,
                 euler=None, heading_diff=None, ray_dist=None, wheel_dist=None, body_dist=None, stone_collision=None,
                 stone_margin=0.0, done_u8=None, ray_src=None, hit_pt=None):
        e, f, i64 = self.num_envs, torch.float32, torch.int64
        stride = 0
        if obs is not None:
            if obs.dim() != 2 or obs.shape[0] != e or obs.shape[1] != self.num_observations or obs.stride(1) != 1:
                raise RoverError(f"obs: expected [{e},{self.num_observations}] float32 rows, got {tuple(obs.shape)}")
            if obs.dtype != f or not obs.is_cuda:
                raise RoverError("obs: expected float32 on the GPU")
            stride = obs.stride(0)
        self._chk(rew, (e,), f, "rew")
        self._chk(reset, (e,), i64, "reset")
        self._chk(rock_collision, (e,), i64, "rock_collision")
        self._chk(reset_ids, (e,), i64, "reset_ids")
        self._chk(n_reset, (1,), torch.int32, "n_reset")
        self._chk(euler, (e, 3), f, "euler")
        self._chk(heading_diff, (e,), f, "heading_diff")
        self._chk(ray_dist, (e, self.P), f, "ray_dist")
        self._chk(wheel_dist, (e, 24), f, "wheel_dist")
        self._chk(body_dist, (e, 2), f, "body_dist")
        self._chk(stone_collision, (e,), i64, "stone_collision")
        self._chk(done_u8, (e,), torch.uint8, "done_u8")
        self._chk(ray_src, (e, self.P, 3), f, "ray_src")
        self._chk(hit_pt, (e, self.P, 3), f, "hit_pt")
        ex = extras or {}
        for k in EXTRAS:
            self._chk(ex.get(k), (e,), i64 if k == "collision_penalty" else f, "extras." + k)
        sout = StepOut(_ptr(obs), stride, _ptr(rew), _ptr(reset), _ptr(rock_collision),
                       *[_ptr(ex.get(k)) for k in EXTRAS], _ptr(reset_ids), _ptr(n_reset), _ptr(euler),
                       _ptr(heading_diff), _ptr(ray_dist), _ptr(wheel_dist), _ptr(body_dist), _ptr(stone_collision),
                       float(stone_margin), _ptr(done_u8), _ptr(ray_src), _ptr(hit_pt))
        sout._keep = (obs, rew, reset, rock_collision, dict(ex), reset_ids, n_reset, euler, heading_diff, ray_dist, wheel_dist,
                      body_dist, stone_collision, done_u8, ray_src, hit_pt)
        return sout

    def step(self, sin: StepIn, sout: StepOut, increment_progress=True, compact=False):
        flags = (STEP_INCREMENT_PROGRESS if increment_progress else 0) | (STEP_COMPACT if compact else 0)
        self._check(self.lib.rover_step(self._h, C.byref(sin), C.byref(sout), flags, _stream(self._dev_index)), "rover_step")

    def get_observations(self, sin, sout):
        self._check(self.lib.rover_get_observations(self._h, C.byref(sin), C.byref(sout), _stream(self._dev_index)),
                    "rover_get_observations")

    def calculate_metrics(self, sin, sout):
        self._check(self.lib.rover_calculate_metrics(self._h, C.byref(sin), C.byref(sout), _stream(self._dev_index)),
                    "rover_calculate_metrics")

    def is_done(self, sin, sout):
        self._check(self.lib.rover_is_done(self._h, C.byref(sin), C.byref(sout), _stream(self._dev_index)), "rover_is_done")

    def get_depths(self, positions, rotations):
        """Camera.get_depths (camera.py:60-145): positions [E,3], rotations [E,3] euler angles -> (distances [E,P], points [E,P,3],
        sources [E,P,3])."""
        e, f = self.num_envs, torch.float32
        self._chk(positions, (e, 3), f, "positions")
        self._chk(rotations, (e, 3), f, "rotations")
        dist = torch.empty(e, self.P, device=positions.device)
        pts = torch.empty(e, self.P, 3, device=positions.device)
        src = torch.empty(e, self.P, 3, device=positions.device)
        self._check(self.lib.rover_get_depths(self._h, _ptr(positions), _ptr(rotations), _ptr(dist), _ptr(pts), _ptr(src), _stream(self._dev_index)),
                    "rover_get_depths")
        return dist, pts, src

    def get_collisions(self, positions, rotations, joints=None):
        """Rock_Detection.get_collisions (rock_detect.py:52-149): positions [E,3], rotations [E,3] euler angles, joints [E,13] (None =
        zero) -> (wheel_dist [E,24], body_dist [E,2])."""
        e, f = self.num_envs, torch.float32
        self._chk(positions, (e, 3), f, "positions")
        self._chk(rotations, (e, 3), f, "rotations")
        if joints is not None:
            self._chk(joints, (e, 13), f, "joints")
        wheel = torch.empty(e, 24, device=positions.device)
        body = torch.empty(e, 2, device=positions.device)
        self._check(self.lib.rover_get_collisions(self._h, _ptr(positions), _ptr(rotations), _ptr(joints), _ptr(wheel), _ptr(body),
                                                  _stream(self._dev_index)), "rover_get_collisions")
        return wheel, body

    def export_rays(self):
        """The rays of the last cast in slot order (24 wheel, 2 body, P heightmap rays per env): (src [E,R,3], dir [E,R,3] — the ray
        record's direction -normalize(direction) —, cell [E,R] int32, dist [E,R]), R = 26 + P."""
        e, r = self.num_envs, 26 + self.P
        src = torch.empty(e, r, 3, device=self.device)
        dirs = torch.empty(e, r, 3, device=self.device)
        cell = torch.empty(e, r, dtype=torch.int32, device=self.device)
        dist = torch.empty(e, r, device=self.device)
        self._check(self.lib.rover_export_rays(self._h, _ptr(src), _ptr(dirs), _ptr(cell), _ptr(dist), _stream(self._dev_index)), "rover_export_rays")
        return src, dirs, cell, dist

    def cast_rays(self, src, dirs):
        """Casts caller-supplied rays (layout of export_rays: origins, record directions) through the sort + ray-cast kernels."""
        e, r, f = self.num_envs, 26 + self.P, torch.float32
        self._chk(src, (e, r, 3), f, "src")
        self._chk(dirs, (e, r, 3), f, "dir")
        dist = torch.empty(e, r, device=self.device)
        self._check(self.lib.rover_cast_rays(self._h, _ptr(src), _ptr(dirs), _ptr(dist), _stream(self._dev_index)), "rover_cast_rays")
        return dist

    def compact_resets(self, reset, reset_ids, n_reset):
        self._chk(reset, (self.num_envs,), torch.int64, "reset")
        self._chk(reset_ids, (self.num_envs,), torch.int64, "reset_ids")
        self._chk(n_reset, (1,), torch.int32, "n_reset")
        self._check(self.lib.rover_compact_resets(self._h, _ptr(reset), _ptr(reset_ids), _ptr(n_reset), _stream(self._dev_index)),
                    "rover_compact_resets")

    def quat_to_euler(self, quat, out=None):
        n = quat.shape[0]
        self._chk(quat, (n, 4), torch.float32, "quat")
        out = torch.empty(n, 3, device=self.device) if out is None else out
        self._chk(out, (n, 3), torch.float32, "euler")
        self._check(self.lib.rover_quat_to_euler(self._h, _ptr(quat), _ptr(out), n, _stream(self._dev_index)), "rover_quat_to_euler")
        return out

    def build_knn_map(self, vertices, triangles, n_x, n_y, res=0.1, k=200, ranking="exact_f32", cell_x_f16=None, cell_y_f16=None):
        """rover_utils.py:48-123 on the GPU: -> map_indices [X, Y, K] int32 (device tensor), nearest first.
        ``ranking``: "exact_f32" (exact f32 squared distance, ties by id) or "reference_fp16" (the reference's fp16 distances,
        rover_utils.py:71-102; ``cell_*_f16`` = optional fp16 coordinate tables, see rover_build_knn_map_ref)."""
        v = _host(vertices, np.float32)
        t = _host(triangles, np.int32)
        if v.ndim != 2 or v.shape[1] != 3 or t.ndim != 2 or t.shape[1] != 3:
            raise RoverError("build_knn_map: expected vertices [V,3] and triangles [T,3]")
        out = torch.empty(int(n_x), int(n_y), int(k), dtype=torch.int32, device=self.device)
        if ranking == "exact_f32":
            self._check(self.lib.rover_build_knn_map(self._h, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], int(n_x), int(n_y),
                                                     float(res), int(k), _ptr(out)), "rover_build_knn_map")
            return out
        if ranking != "reference_fp16":
            raise RoverError(f"build_knn_map: unknown ranking {ranking!r}")
        tabs = []
        for tab, n in ((cell_x_f16, n_x), (cell_y_f16, n_y)):
            if tab is None:
                tabs.append(None)
                continue
            a = np.ascontiguousarray(np.asarray(tab, dtype=np.float16)).view(np.uint16)
            if a.shape != (int(n),):
                raise RoverError(f"build_knn_map: fp16 cell table must have {n} entries")
            tabs.append(a)
        self._check(self.lib.rover_build_knn_map_ref(self._h, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], int(n_x), int(n_y),
                                                     float(res), int(k), None if tabs[0] is None else tabs[0].ctypes.data,
                                                     None if tabs[1] is None else tabs[1].ctypes.data, _ptr(out)),
                    "rover_build_knn_map_ref")
        return out

    ACTIVATIONS = {"none": 0, None: 0, "leakyrelu": 1, "tanh": 2, "relu": 3, "elu": 4}

    def linear_forward(self, x, weight, bias, activation, out):
        """out[:, :N] = act(x[:, :K] @ weight.T + bias); x / out may be column slices of wider row-major tensors."""
        m, k = x.shape
        n = weight.shape[0]
        for t, name in ((x, "x"), (out, "out")):
            if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
                raise RoverError(f"linear_forward: {name} must be a float32 GPU matrix with unit column stride")
        if k > 0:
            self._chk(weight, (n, k), torch.float32, "weight")
        self._chk(bias, (n,), torch.float32, "bias")
        if out.shape[0] != m or out.shape[1] != n:
            raise RoverError(f"linear_forward: out must be [{m},{n}]")
        self._check(self.lib.rover_linear_forward(self._h, _ptr(x), max(x.stride(0), k), m, k, _ptr(weight), _ptr(bias), n,
                                                  self.ACTIVATIONS[activation], _ptr(out), out.stride(0), _stream(self._dev_index)),
                    "rover_linear_forward")
        return out

    CHAIN_SHAPES = {2: (96, 64), 4: (256, 160, 128, 16)}        # widths the fused chain kernel is built for
    CHAIN_HIDDEN_ACTS = (None, "none", "leakyrelu", "relu")      # hidden activations the 4-layer chain is built for

    def chain_fits(self, layers):
        """True if ``layers`` (objects with .weight [n, k]) can run as one rover_mlp_chain_forward launch."""
        lim = self.CHAIN_SHAPES.get(len(layers))
        if lim is None or not all(l.weight.shape[0] <= m for l, m in zip(layers, lim)) or layers[0].weight.shape[1] <= 0:
            return False
        return len(layers) == 2 or all(l.activation in self.CHAIN_HIDDEN_ACTS for l in layers[:-1])

    def chain_forward(self, x, layers, out):
        """out = layers[-1](... layers[0](x)) in one kernel; ``layers``: objects with .weight [n, k], .bias [n], .activation."""
        m, k0 = x.shape
        for t, name in ((x, "x"), (out, "out")):
            if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
                raise RoverError(f"chain_forward: {name} must be a float32 GPU matrix with unit column stride")
        n = len(layers)
        k = k0
        for l in layers:
            self._chk(l.weight, (l.weight.shape[0], k), torch.float32, "weight")
            self._chk(l.bias, (l.weight.shape[0],), torch.float32, "bias")
            k = l.weight.shape[0]
        if out.shape[0] != m or out.shape[1] != k:
            raise RoverError(f"chain_forward: out must be [{m},{k}]")
        w = (C.c_void_p * n)(*[_ptr(l.weight) for l in layers])
        b = (C.c_void_p * n)(*[_ptr(l.bias) for l in layers])
        widths = (C.c_int32 * n)(*[l.weight.shape[0] for l in layers])
        acts = (C.c_int32 * n)(*[self.ACTIVATIONS[l.activation] for l in layers])
        self._check(self.lib.rover_mlp_chain_forward(self._h, _ptr(x), x.stride(0), m, k0, n, w, b, widths, acts, _ptr(out), out.stride(0),
                                                     _stream(self._dev_index)), "rover_mlp_chain_forward")
        return out

    def _chain_desc(self, x, layers, out, keep):
        m, k0 = x.shape
        for t, name in ((x, "x"), (out, "out")):
            if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
                raise RoverError(f"chain_pair_forward: {name} must be a float32 GPU matrix with unit column stride")
        n, k = len(layers), k0
        for l in layers:
            self._chk(l.weight, (l.weight.shape[0], k), torch.float32, "weight")
            self._chk(l.bias, (l.weight.shape[0],), torch.float32, "bias")
            k = l.weight.shape[0]
        if out.shape[0] != m or out.shape[1] != k:
            raise RoverError(f"chain_pair_forward: out must be [{m},{k}]")
        w = (C.c_void_p * n)(*[_ptr(l.weight) for l in layers])
        b = (C.c_void_p * n)(*[_ptr(l.bias) for l in layers])
        widths = (C.c_int32 * n)(*[l.weight.shape[0] for l in layers])
        acts = (C.c_int32 * n)(*[self.ACTIVATIONS[l.activation] for l in layers])
        keep.extend((w, b, widths, acts))                      # the arrays the descriptor points at live until the call returns
        return ChainDesc(x.data_ptr(), x.stride(0), k0, n, C.addressof(w), C.addressof(b), C.addressof(widths), C.addressof(acts),
                         out.data_ptr(), out.stride(0))

    def chain_pair_forward(self, xa, layers_a, out_a, xb, layers_b, out_b, copy_src=None, copy_dst=None, copy_cols=0):
        """Two 2-layer chains over the same rows (the two encoders: rover_mlp_chain_pair_forward) and, optionally,
        copy_dst[:, :copy_cols] = copy_src[:, :copy_cols] — side by side in one launch per stage at small batches."""
        if xa.shape[0] != xb.shape[0]:
            raise RoverError("chain_pair_forward: the two chains must have the same number of rows")
        keep = []
        da, db = self._chain_desc(xa, layers_a, out_a, keep), self._chain_desc(xb, layers_b, out_b, keep)
        if copy_cols:
            for t, name in ((copy_src, "copy_src"), (copy_dst, "copy_dst")):
                if (t is None or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.shape[0] != xa.shape[0]
                        or t.shape[1] < copy_cols):
                    raise RoverError(f"chain_pair_forward: {name} must be a float32 GPU matrix of the same rows with >= copy_cols columns")
        self._check(self.lib.rover_mlp_chain_pair_forward(
            self._h, xa.shape[0], C.byref(da), C.byref(db), _ptr(copy_src) if copy_cols else None, copy_src.stride(0) if copy_cols else 0,
            _ptr(copy_dst) if copy_cols else None, copy_dst.stride(0) if copy_cols else 0, int(copy_cols), _stream(self._dev_index)),
            "rover_mlp_chain_pair_forward")

    def set_option(self, name, value):
        self._check(self.lib.rover_set_option(self._h, name.encode(), int(value)), "rover_set_option")

    def set_profiling(self, enable=True, every=1):
        """Bracket the ray-cast launch of every `every`-th step with hipEvents (get_profile() sums them)."""
        self._check(self.lib.rover_set_profiling(self._h, max(1, int(every)) if enable else 0), "rover_set_profiling")

    def get_profile(self):
        p = Profile()
        self._check(self.lib.rover_get_profile(self._h, C.byref(p)), "rover_get_profile")
        return p

    def replay_raycast(self, stream=None):
        s = _stream(self._dev_index) if stream is None else C.c_void_p(stream)
        self._check(self.lib.rover_replay_raycast(self._h, s), "rover_replay_raycast")

    # ---- reset path ----------------------------------------------------------------------------
    def clearance(self, xy):
        n = xy.shape[0]
        self._chk(xy, (n, 2), torch.float32, "xy")
        out = torch.empty(n, device=self.device)
        self._check(self.lib.rover_clearance(self._h, _ptr(xy), n, _ptr(out), _stream(self._dev_index)), "rover_clearance")
        return out

    def shift_spawns(self, pos3, max_iter=100000):
        n = pos3.shape[0]
        self._chk(pos3, (n, 3), torch.float32, "pos3")
        self._check(self.lib.rover_shift_spawns(self._h, _ptr(pos3), n, int(max_iter), _stream(self._dev_index)), "rover_shift_spawns")
        return pos3

    def sample_height(self, xy):
        n = xy.shape[0]
        self._chk(xy, (n, 2), torch.float32, "xy")
        out = torch.empty(n, device=self.device)
        self._check(self.lib.rover_sample_height(self._h, _ptr(xy), n, _ptr(out), _stream(self._dev_index)), "rover_sample_height")
        return out

    def generate_goals(self, env_ids, initial_pos3, target3, radius=8.0, draws=None, max_draws=64, seed=0,
                       n_draws_used=None):
        n = env_ids.shape[0]
        self._chk(env_ids, (n,), torch.int64, "env_ids")
        self._chk(initial_pos3, (self.num_envs, 3), torch.float32, "initial_pos3")
        self._chk(target3, (self.num_envs, 3), torch.float32, "target3")
        if draws is not None:
            self._chk(draws, (draws.shape[0], n), torch.float32, "draws")
            max_draws = draws.shape[0]
        self._chk(n_draws_used, (1,), torch.int32, "n_draws_used")
        self._check(self.lib.rover_generate_goals(self._h, _ptr(env_ids), n, _ptr(initial_pos3), _ptr(target3),
                                                  float(radius), _ptr(draws), int(max_draws), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                  _ptr(n_draws_used), _stream(self._dev_index)), "rover_generate_goals")

    def _reset_io(self, reset_ids, initial_pos3, pos3, quat4, reset, progress, n_reset_dev=None, n_reset_host=0,
                  joint_pos13=None, joint_vel13=None, base_pos3=None, yaw_deg=None, target3=None, radius=8.0, draws=None,
                  max_draws=256, seed=0, n_draws_used=None, seed_dev=None):
        """Validates the arguments of rover_reset_envs and packs them (the struct keeps its tensors alive)."""
        e, f, i64 = self.num_envs, torch.float32, torch.int64
        self._chk(seed_dev, (1,), i64, "seed_dev")
        self._chk(reset_ids, (e,), i64, "reset_ids")
        for t, sh, n in ((initial_pos3, (e, 3), "initial_pos3"), (pos3, (e, 3), "pos3"), (quat4, (e, 4), "quat4"),
                         (joint_pos13, (e, 13), "joint_pos13"), (joint_vel13, (e, 13), "joint_vel13"),
                         (base_pos3, (e, 3), "base_pos3"), (target3, (e, 3), "target3")):
            self._chk(t, sh, f, n)
        self._chk(reset, (e,), i64, "reset")
        self._chk(progress, (e,), i64, "progress")
        self._chk(n_reset_dev, (1,), torch.int32, "n_reset_dev")
        self._chk(n_draws_used, (1,), torch.int32, "n_draws_used")
        if yaw_deg is not None:
            self._chk(yaw_deg, (yaw_deg.shape[0],), torch.int32, "yaw_deg")
            need = e if n_reset_dev is not None else int(n_reset_host)      # entry i belongs to reset_ids[i]; with the count on
            if yaw_deg.shape[0] < need:                                     # the device any i < num_envs may be read
                raise RoverError(f"yaw_deg: {yaw_deg.shape[0]} entries, but up to {need} may be read")
        if draws is not None:
            self._chk(draws, (draws.shape[0], int(n_reset_host)), f, "draws")
            max_draws = draws.shape[0]
        io = ResetIO(_ptr(reset_ids), _ptr(n_reset_dev), int(n_reset_host), _ptr(initial_pos3), _ptr(pos3), _ptr(quat4),
                     _ptr(joint_pos13), _ptr(joint_vel13), _ptr(base_pos3), _ptr(reset), _ptr(progress), _ptr(yaw_deg),
                     _ptr(target3), float(radius), _ptr(draws), int(max_draws), int(seed) & 0xFFFFFFFFFFFFFFFF,
                     _ptr(n_draws_used), 0 if yaw_deg is None else int(yaw_deg.shape[0]), _ptr(seed_dev))
        io._keep = (reset_ids, initial_pos3, pos3, quat4, reset, progress, n_reset_dev, joint_pos13, joint_vel13, base_pos3, yaw_deg,
                    target3, draws, n_draws_used, seed_dev)
        return io

    def reset_envs(self, *args, **kw):
        """rover_reset_envs (reset_idx + set_targets for the compacted ids, rover.py:416-453,566-584); arguments: `_reset_io`."""
        io = self._reset_io(*args, **kw)
        self._check(self.lib.rover_reset_envs(self._h, C.byref(io), _stream(self._dev_index)), "rover_reset_envs")

    def bind_reset_envs(self, *args, **kw):
        """reset_envs with everything but the seed validated and packed ONCE: returns call(seed) (for a task whose buffers persist)."""
        io = self._reset_io(*args, **kw)
        fn, check, idx, h = self.lib.rover_reset_envs, self._check, self._dev_index, self._h

        def call(seed):
            io.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            check(fn(h, C.byref(io), _stream(idx)), "rover_reset_envs")
        return call

    def bind_pre_physics(self, quat, lin_hist, ang_hist, euler_pre=None, pos_targets13=None, vel_targets13=None, actions_nn=None):
        """Validates the persistent tensors ONCE and returns call(actions), which enqueues rover_pre_physics_step on them (for a task whose
        buffers are persistent: the per-call checks are a sizeable part of a small batch's host cost).  Only the actions tensor — usually a
        fresh policy output every step — is checked per call.  The callable keeps the bound tensors alive."""
        e, f = self.num_envs, torch.float32
        for t, sh, n in ((quat, (e, 4), "quat"), (lin_hist, (e, 3), "lin_hist"),
                         (ang_hist, (e, 3), "ang_hist"), (euler_pre, (e, 3), "euler_pre"), (pos_targets13, (e, 13), "pos_targets13"),
                         (vel_targets13, (e, 13), "vel_targets13"), (actions_nn, (e, 2, 3), "actions_nn")):
            self._chk(t, sh, f, n)
        keep = (quat, lin_hist, ang_hist, euler_pre, pos_targets13, vel_targets13, actions_nn)
        rest = [_ptr(t) for t in keep]
        fn, check, idx, h, chk = self.lib.rover_pre_physics_step, self._check, self._dev_index, self._h, self._chk

        def call(actions, _keep=keep):
            chk(actions, (e, 2), f, "actions")
            check(fn(h, C.c_void_p(actions.data_ptr()), *rest, _stream(idx)), "rover_pre_physics_step")
        return call

    def pre_physics_step(self, actions, quat, lin_hist, ang_hist, euler_pre=None, pos_targets13=None, vel_targets13=None, actions_nn=None):
        e, f = self.num_envs, torch.float32
        self._chk(actions_nn, (e, 2, 3), f, "actions_nn")
        for t, sh, n in ((actions, (e, 2), "actions"), (quat, (e, 4), "quat"), (lin_hist, (e, 3), "lin_hist"),
                         (ang_hist, (e, 3), "ang_hist"), (euler_pre, (e, 3), "euler_pre"),
                         (pos_targets13, (e, 13), "pos_targets13"), (vel_targets13, (e, 13), "vel_targets13")):
            self._chk(t, sh, f, n)
        self._check(self.lib.rover_pre_physics_step(self._h, _ptr(actions), _ptr(quat), _ptr(lin_hist), _ptr(ang_hist),
                                                    _ptr(euler_pre), _ptr(pos_targets13), _ptr(vel_targets13), _ptr(actions_nn), _stream(self._dev_index)),
                    "rover_pre_physics_step")

    def ackermann(self, lin, ang):
        n = lin.shape[0]
        self._chk(lin, (n,), torch.float32, "lin")
        self._chk(ang, (n,), torch.float32, "ang")
        steer = torch.empty(n, 6, device=self.device)
        vel = torch.empty(n, 6, device=self.device)
        self._check(self.lib.rover_ackermann(self._h, _ptr(lin), _ptr(ang), n, _ptr(steer), _ptr(vel), _stream(self._dev_index)),
                    "rover_ackermann")
        return steer, vel
