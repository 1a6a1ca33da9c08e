"""TEST INFRASTRUCTURE — runs only in the build container, never on the GPU box.

Imports the reference's own hot-path modules from ``/root/reference`` (read-only)
so that golden vectors can be captured from the reference itself
(SURVEY.md §8c).  Nothing from the reference is copied: the modules are imported
where they lie, with

* stub ``sys.modules`` entries for the Isaac / USD / mesh packages the container
  lacks (``omni.*``, ``pxr``, ``open3d``, ``pymeshlab``, ``gym``, ``carb``, ``skrl``),
* ``device='cuda:0'`` coerced to CPU for tensor factories and ``Tensor.cuda()``
  made an identity (``tensor_quat_to_euler.py:12-14``, ``rover.py:651``),
* a scratch working directory holding the synthetic scene in the reference's
  on-disk layout (``camera.py:156-160``, ``rock_detect.py:153-157``,
  ``rover.py:144,210``).

``RoverTask`` methods are called *unbound* on a ``SimpleNamespace`` that carries the
fields they touch, so no simulator is needed.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"
_STUB_PREFIXES = ("omni", "pxr", "open3d", "pymeshlab", "gym", "carb", "skrl", "hydra", "omegaconf", "wandb")


class _StubBase:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _StubBase()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _StubBase()

    def initialize(self, *a, **k):      # ArticulationView.initialize: reached through super() from RoverView.initialize
        return None


class _StubModule(types.ModuleType):
    __all__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_StubBase,), {})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Idempotent: stubs, sys.path, CPU coercion."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference not present: the harness only runs in the build container")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, os.path.join(REFERENCE_ROOT, "omniisaacgymenvs"))
    sys.path.insert(0, REFERENCE_ROOT)
    # kinematics.py is @torch.jit.script: compile it before the factories are wrapped
    # (TorchScript cannot see through the *args wrappers below)
    importlib.import_module("omniisaacgymenvs.tasks.utils.kinematics")

    def _coerce(fn):
        def wrapped(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped

    for name in ("zeros", "ones", "tensor", "empty", "arange", "rand", "zeros_like", "ones_like", "full"):
        setattr(torch, name, _coerce(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    _installed = True


class _TorchFp32Proxy:
    """Rebinds ``torch.float16`` → ``torch.float32`` inside rock_detect.py (its :319 hard-codes fp16)."""

    def __getattr__(self, name):
        if name == "float16":
            return torch.float32
        return getattr(torch, name)


def write_scene(scene, root):
    """Write a synth.Scene in the reference's on-disk layout under ``root``."""
    base = os.path.join(root, "tasks", "utils", "terrain")
    for sub, m in (("knn_terrain", scene.terrain), ("knn_rocks", scene.rocks)):
        d = os.path.join(base, sub)
        os.makedirs(d, exist_ok=True)
        # rover_utils.py:108-118 saves map_indices as [K, X, Y]
        torch.save(m.map_indices.permute(2, 0, 1).contiguous(), os.path.join(d, "map_indices.pt"))
        torch.save(m.triangles, os.path.join(d, "triangles.pt"))
        torch.save(m.vertices, os.path.join(d, "vertices.pt"))
    np.save(os.path.join(base, "stone_info.npy"), scene.stone_info_raw)
    torch.save(scene.heightmap, os.path.join(base, "heightmap_tensor.pt"))


class Reference:
    """The reference's Camera / Rock_Detection / RoverTask bound to one synthetic scene."""

    def __init__(self, scene, fp32: bool, distribution=None):
        install()
        self._tmp = tempfile.TemporaryDirectory()
        write_scene(scene, self._tmp.name)
        cwd = os.getcwd()
        os.chdir(self._tmp.name)
        try:
            cam_mod = importlib.import_module("omniisaacgymenvs.tasks.utils.camera.camera")
            rd_mod = importlib.import_module("omniisaacgymenvs.tasks.utils.rock_detection.rock_detect")
            self.quat_mod = importlib.import_module("omniisaacgymenvs.tasks.utils.math.tensor_quat_to_euler")
            self.rover_mod = importlib.import_module("omniisaacgymenvs.tasks.rover")
            self.tu_mod = importlib.import_module("omniisaacgymenvs.utils.terrain_utils.terrain_utils")
            self.kin_mod = importlib.import_module("omniisaacgymenvs.tasks.utils.kinematics")
            self.shift = torch.tensor([scene.shift[0], scene.shift[1], scene.shift[2]])
            self.cam = cam_mod.Camera("cpu", self.shift)
            rd_mod.torch = _TorchFp32Proxy() if fp32 else torch
            self.rd = rd_mod.Rock_Detection("cpu", self.shift)
            self.stone_info = self.tu_mod.read_stone_info("./tasks/utils/terrain/stone_info.npy")
            self.heightmap = torch.load("tasks/utils/terrain/heightmap_tensor.pt")
        finally:
            os.chdir(cwd)
        self.fp32 = fp32
        if fp32:
            self.cam.dtype = torch.float32
            self.cam.vertices = self.cam.vertices.float()
            self.rd.dtype = torch.float32
            self.rd.vertices = self.rd.vertices.float()
        self.native_distribution = self.cam.heightmap_distribution.clone()
        self.native_sparse = self.cam.heightmap.coarse_idx.clone()
        self.native_dense = self.cam.heightmap.fine_idx.clone()
        if distribution is not None:
            self.set_distribution(*distribution)

    def set_distribution(self, pts, sparse_idx, dense_idx):
        self.cam.heightmap_distribution = torch.as_tensor(pts, dtype=torch.float64)
        self.cam.heightmap.distribution = self.cam.heightmap_distribution
        self.cam.heightmap.coarse_idx = torch.as_tensor(sparse_idx, dtype=torch.int64)
        self.cam.heightmap.fine_idx = torch.as_tensor(dense_idx, dtype=torch.int64)
        self.cam.num_exteroceptive = self.cam.heightmap_distribution.shape[0]

    # ------------------------------------------------------------------ step
    def make_task(self, st, num_envs_global=None, curriculum_level=2, rew_scales=None):
        """SimpleNamespace carrying every field rover.py:272-336,460-531,610-672 touches."""
        Rover = self.rover_mod.RoverTask
        e = st["pos"].shape[0]
        ns = int(self.cam.heightmap.coarse_idx.shape[0])
        nd = int(self.cam.heightmap.fine_idx.shape[0])
        pos, quat, joints = st["pos"].clone(), st["quat"].clone(), st["joints"].clone()
        rover_view = SimpleNamespace(
            name="rover_view",
            get_world_poses=lambda: (pos, quat),
            get_joint_positions=lambda: joints,
        )
        lin = self.rover_mod.Memory(e, 1, 3, "cpu")
        ang = self.rover_mod.Memory(e, 1, 3, "cpu")
        lin.tracker = st["lin_hist"].reshape(e, 1, 3).clone()
        ang.tracker = st["ang_hist"].reshape(e, 1, 3).clone()
        t = SimpleNamespace(
            _rover=rover_view, _device="cpu", num_envs=num_envs_global or e, _num_envs=e,
            target_positions=st["target"].clone(), Camera=self.cam, Rock_detector=self.rd,
            curriculum_level=curriculum_level, save_teacher_data=False, _num_proprioceptive=4,
            linear_velocity=lin, angular_velocity=ang,
            obs_buf=torch.zeros(e, 4 + ns + nd), rew_buf=torch.zeros(e),
            reset_buf=torch.ones(e, dtype=torch.long), progress_buf=st["progress"].clone(),
            extras={}, is_evaluation=False, max_episode_length=3000,
            rew_scales=rew_scales or dict(pos_reward=1.0, terminalReward=0, collision_reward=0.3,
                                          heading_contraint_reward=0.05, motion_contraint_reward=-0.01,
                                          goal_angle_reward=0.3, boogie_contraint_reward=0.5),
            rover_rot=st["euler_pre"].clone(), stone_info=self.stone_info,
            rock_collison=torch.zeros(e, dtype=torch.long),
        )
        t.check_collision = lambda w, b: Rover.check_collision(t, w, b)
        t._cls = Rover
        return t

    def step(self, st, **kw):
        """post_physics_step order (rl_task.py:250-257) on captured sim state ``st``."""
        t = self.make_task(st, **kw)
        Rover = t._cls
        t.progress_buf[:] += 1
        Rover.get_observations(t)
        Rover.calculate_metrics(t)
        Rover.is_done(t)
        # intermediates (recomputed through the same reference calls)
        dist, _pt, sources = self.cam.get_depths(t.rover_positions, t.rover_rotation)
        wheel, body = self.rd.get_collisions(t.rover_positions, t.rover_rotation, t._rover.get_joint_positions())
        out = dict(
            euler=t.rover_rotation, heading_diff=t.heading_diff, ray_dist=dist.float(),
            ray_sources=sources.float(), wheel_dist=wheel.float(), body_dist=body.float(),
            rock_collision=t.rock_collison, obs_buf=t.obs_buf, rew_buf=t.rew_buf,
            reset_buf=t.reset_buf, progress_buf=t.progress_buf,
        )
        for k, v in t.extras.items():
            out["extras_" + k] = v
        return out

    # ------------------------------------------------------------ reset path
    def get_pos_height(self, xy):
        Rover = self.rover_mod.RoverTask
        return Rover.get_pos_height(SimpleNamespace(), self.heightmap, xy, 0.025, 1, self.shift[0:2])

    def avoid_pos_rock_collision(self, pos):
        Rover = self.rover_mod.RoverTask
        t = SimpleNamespace(stone_info=self.stone_info)
        return Rover.avoid_pos_rock_collision(t, pos.clone())

    def clearance(self, xy):
        """nearest_rock of rover.py:536-538 / :655-658."""
        d = torch.cdist(xy, self.stone_info[:, 0:2], p=2.0)
        d[:] = d[:] - self.stone_info[:, 6]
        return torch.min(d, dim=1)[0]

    def generate_goals(self, env_ids, initial_pos, uniforms):
        """rover.py:544-564 with ``torch.rand`` fed from ``uniforms`` (list of 1-D tensors, one per draw)."""
        Rover = self.rover_mod.RoverTask
        e = initial_pos.shape[0]
        t = SimpleNamespace(stone_info=self.stone_info, _device="cpu",
                            target_positions=torch.zeros(e, 3), initial_pos=initial_pos.clone())
        t.random_goals = lambda ids, radius: Rover.random_goals(t, ids, radius)
        t.check_goal_collision = lambda ids: Rover.check_goal_collision(t, ids)
        draws = list(uniforms)
        real_rand = self.rover_mod.torch.rand
        calls = []

        def fake_rand(n, device=None):
            u = draws.pop(0)
            assert u.numel() == n
            calls.append(u.clone())
            return u.clone()

        self.rover_mod.torch.rand = fake_rand
        try:
            Rover.generate_goals(t, env_ids.clone(), radius=8)
        finally:
            self.rover_mod.torch.rand = real_rand
        return t.target_positions, len(calls)

    def ackermann(self, lin, ang):
        return self.kin_mod.Ackermann(lin, ang, "cpu")

    # ------------------------------------------------------------ pre_physics_step (f-1)
    def pre_physics_step(self, actions, quat, lin_hist, ang_hist, actions_nn, global_step=20):
        """``RoverTask.pre_physics_step`` (rover.py:338-414) called unbound with no env flagged for reset and ``global_step`` away
        from the curriculum switch (:344), on the reference's own ``RoverView`` (its ``initialize`` sets the joint index lists,
        robots/articulations/views/rover_view.py:45-46) with the PhysX setters replaced by recorders.  Returns what the method left
        in the task and what it handed to ``set_joint_position_targets`` / ``set_joint_velocity_targets``."""
        Rover = self.rover_mod.RoverTask
        rv_mod = importlib.import_module("omniisaacgymenvs.robots.articulations.views.rover_view")
        e = actions.shape[0]
        view = rv_mod.RoverView("/World/envs/.*/Rover", "rover_view")
        view.num_dof = 13
        view.initialize(None)
        pos, q = torch.zeros(e, 3), quat.clone()
        calls = {}
        view.get_world_poses = lambda: (pos, q)
        view.count = e
        view.set_joint_position_targets = lambda x, indices=None, joint_indices=None: calls.__setitem__("pos", (x.clone(), list(joint_indices)))
        view.set_joint_velocity_targets = lambda x, indices=None, joint_indices=None: calls.__setitem__("vel", (x.clone(), list(joint_indices)))
        lin = self.rover_mod.Memory(e, 1, 3, "cpu")
        ang = self.rover_mod.Memory(e, 1, 3, "cpu")
        lin.tracker = lin_hist.reshape(e, 1, 3).clone()
        ang.tracker = ang_hist.reshape(e, 1, 3).clone()
        t = SimpleNamespace(global_step=global_step, _rover=view, reset_buf=torch.zeros(e, dtype=torch.long), save_teacher_data=False,
                            _device="cpu", linear_velocity=lin, angular_velocity=ang, num_envs=e, _num_actions=2,
                            actions_nn=actions_nn.clone())
        real = self.rover_mod.Ackermann
        self.rover_mod.Ackermann = lambda l, a: real(l, a, "cpu")      # (:391 relies on the default device 'cuda:0')
        try:
            Rover.pre_physics_step(t, actions.clone())
        finally:
            self.rover_mod.Ackermann = real
        assert t.global_step == global_step + 1
        return dict(rover_rot=t.rover_rot, lin_tracker=t.linear_velocity.tracker.reshape(e, 3), ang_tracker=t.angular_velocity.tracker.reshape(e, 3),
                    actions_nn=t.actions_nn, positions=calls["pos"][0], pos_joint_indices=torch.tensor(calls["pos"][1]),
                    velocities=calls["vel"][0], vel_joint_indices=torch.tensor(calls["vel"][1]))

    # ------------------------------------------------------------ reset_idx (f-2)
    def reset_idx(self, env_ids, degrees, initial_pos, base_pos, reset_buf, progress_buf):
        """``RoverTask.reset_idx`` (rover.py:416-453) called unbound, ``random.randint`` (:429) fed from ``degrees``.
        Returns what the method handed to the RoverView setters and the buffers it changed."""
        Rover = self.rover_mod.RoverTask
        calls = {}
        view = SimpleNamespace(
            set_joint_positions=lambda x, indices=None: calls.__setitem__("joint_pos", (x.clone(), indices.clone())),
            set_joint_velocities=lambda x, indices=None: calls.__setitem__("joint_vel", (x.clone(), indices.clone())),
            set_world_poses=lambda p, q, i: calls.__setitem__("poses", (p.clone(), q.clone(), i.clone())),
        )
        t = SimpleNamespace(_rover=view, _device="cpu", save_teacher_data=False, base_pos=base_pos.clone(),
                            initial_pos=initial_pos.clone(), reset_buf=reset_buf.clone(), progress_buf=progress_buf.clone())
        feed = [int(d) for d in degrees]
        real = self.rover_mod.random.randint

        def fake_randint(a, b):
            assert (a, b) == (0, 360)
            return feed.pop(0)

        self.rover_mod.random.randint = fake_randint
        try:
            Rover.reset_idx(t, env_ids.clone())
        finally:
            self.rover_mod.random.randint = real
        assert not feed
        return dict(joint_pos=calls["joint_pos"][0], joint_pos_indices=calls["joint_pos"][1],
                    joint_vel=calls["joint_vel"][0], pose_pos=calls["poses"][0], pose_quat=calls["poses"][1],
                    pose_indices=calls["poses"][2], base_pos=t.base_pos, reset_buf=t.reset_buf, progress_buf=t.progress_buf)


def knn_triangles(vertices, triangles, file_name, res_x, res_y, res, n_triangles):
    """The reference's own map builder ``_get_knn_triangles`` (tasks/utils/rover_utils.py:52-118) on a caller-supplied mesh:
    ``o3d.io.read_triangle_mesh`` (:62, open3d is absent here) is replaced by an object carrying that mesh as float64
    vertices / int32 triangles the way open3d returns them; everything else is the reference's code, run on CPU.  Returns the
    three tensors it saved (map_indices [K, X, Y] int32, vertices fp16, triangles int32) and the fp16 cell coordinates its
    ``torch.arange(0, res_x*res, res, dtype=float16)`` produced on this host."""
    install()
    mod = importlib.import_module("omniisaacgymenvs.tasks.utils.rover_utils")
    mesh = SimpleNamespace(vertices=np.asarray(vertices, dtype=np.float64), triangles=np.asarray(triangles, dtype=np.int32))
    real = mod.o3d
    mod.o3d = SimpleNamespace(io=SimpleNamespace(read_triangle_mesh=lambda path: mesh))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        save = os.path.join("tasks", "utils", "terrain", "knn_out") + "/"
        os.makedirs(os.path.join(tmp, save))
        os.chdir(tmp)
        try:
            mod._get_knn_triangles(file_name=file_name, save_path=save, res_x=res_x, res_y=res_y, res=res, n_triangles=n_triangles)
            out = (torch.load(save + "map_indices.pt"), torch.load(save + "vertices.pt"), torch.load(save + "triangles.pt"))
        finally:
            os.chdir(cwd)
            mod.o3d = real
    xx = torch.arange(0, res_x * res, res, dtype=torch.float16)
    yy = torch.arange(0, res_y * res, res, dtype=torch.float16)
    return out + (xx, yy)


def _install_skrl_models():
    """learning/model.py subclasses skrl's Model / mixins (skrl is absent here).  Minimal stand-ins with the attributes
    model.py reads: an nn.Module base exposing num_observations / num_actions, and mixins whose __init__ takes anything."""
    import torch.nn as nn

    class Model(nn.Module):
        def __init__(self, observation_space, action_space, device=None):
            nn.Module.__init__(self)
            self.observation_space, self.action_space, self.device = observation_space, action_space, device
            self.num_observations = observation_space.shape[0]
            self.num_actions = action_space.shape[0]

    class _Mixin:
        def __init__(self, *a, **k):
            pass

    m = types.ModuleType("skrl.models.torch")
    m.Model, m.GaussianMixin, m.DeterministicMixin = Model, type("GaussianMixin", (_Mixin,), {}), type("DeterministicMixin", (_Mixin,), {})
    pkg = _StubModule("skrl"); pkg.__path__ = []
    pkg_m = _StubModule("skrl.models"); pkg_m.__path__ = []
    sys.modules.update({"skrl": pkg, "skrl.models": pkg_m, "skrl.models.torch": m})


def policy_models(num_observations, num_sparse, num_dense, seed, mlp=(256, 160, 128), encoder=(80, 60), activation="leakyrelu"):
    """The reference's StochasticActorHeightmap / DeterministicHeightmap (learning/model.py:152-241) built with the
    reference's own constructors on CPU (network sizes / activation of cfg/trainSKRL/RoverPPOSKRL.yaml:1-10)."""
    install()
    _install_skrl_models()
    mod = importlib.import_module("omniisaacgymenvs.learning.model")
    torch.manual_seed(seed)
    net = mod.NetworkInfo(list(mlp), list(encoder), list(encoder), [80, 60], activation)
    obs = mod.ObserverationInfo(num_observations - num_sparse - num_dense, num_sparse, num_dense, 0)
    osp, asp = SimpleNamespace(shape=(num_observations,)), SimpleNamespace(shape=(2,))
    actor = mod.StochasticActorHeightmap(osp, asp, net, obs, device="cpu")
    critic = mod.DeterministicHeightmap(osp, asp, net, obs, device="cpu")
    return actor, critic
