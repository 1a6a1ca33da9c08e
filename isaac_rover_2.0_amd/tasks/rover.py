"""RoverTask — the reference's rover navigation task with its per-step compute on HIP.

Drop-in for ``omniisaacgymenvs/tasks/rover.py`` (``RoverTask`` :80-672) on the env.step() path: same method
names, argument meaning, buffers and call order, so ``train.py``-style loops and a PPO learner see the same
task.  Every tensor computation of ``get_observations`` / ``calculate_metrics`` / ``is_done`` /
``pre_physics_step``'s done compaction / ``set_targets`` / ``avoid_pos_rock_collision`` /
``get_pos_height`` runs in ``librover_step.so`` (hand-written HIP, gfx950) through ``_lib.Engine``; this
class is host-side bookkeeping only and raises if the library or a GPU is missing (no CPU fallback).

What is NOT here: USD stage building (:187-270), the dead teacher/student loaders (:169-172),
teacher-data dumps (:298-317), evaluation bookkeeping (:620-641).

Deviations from the reference, all deliberate and tested:
  * ``post_physics_step`` is ONE fused ``rover_step`` launch sequence (``fused=True``, default); with
    ``fused=False`` the base-class path calls the three methods like ``rl_task.py:250-257``.  Same results.
  * the done compaction (``reset_buf.nonzero()``, :356) happens on the device inside the fused step; only
    the count is read on the host (the reference's ``len()`` is the same sync).
  * ``reset_idx`` draws its yaw on the device RNG instead of Python's ``random`` (RNG streams cannot match).
  * ``pre_physics_step`` defaults to the device-side reset orchestration (``rover_reset_envs``): same state
    transitions as ``reset_idx`` + ``set_targets``, but the reset count never travels to the host.
  * ``graph=True``: from global step 11 on (the curriculum switch of :344-353 is behind it) the launches of
    ``pre_physics_step`` and of ``post_physics_step`` are each replayed from a captured hipGraph — same kernels, same
    arguments, same results (``test_task_graph_replay_equals_eager``); the host then pays two graph launches per step
    instead of ~14 kernel launches, which is what bounds small batches (the reference's default is 512 envs).
"""
from __future__ import annotations

import math

import torch

from .. import _lib, synth
from ..config import SimConfig
from ..views import RigidPrimView, RoverView
from .base.rl_task import RLTask
from .utils.heightmap_distribution import Heightmap


_STEP_KEY = 0xD1B54A32D192ED03          # the global step's multiplier in the device RNG's key


def _to_i64(x):
    """x mod 2^64 as the int64 with the same bits (torch has no uint64 arithmetic)."""
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


class Memory:
    """3-deep action history (``rover.py:60-77``) kept in ONE persistent [E, horizon] tensor, newest first,
    so the kernels can borrow its pointer; ``input_state`` shifts in place instead of ``torch.cat``."""

    def __init__(self, num_envs, num_states, horizon, device) -> None:
        assert num_states == 1
        self.tracker = torch.zeros((num_envs, horizon), device=device)
        self.device, self.num_envs, self.num_states, self.horizon = device, num_envs, num_states, horizon

    def get_state(self, timestep):
        return self.tracker[:, timestep]

    def input_state(self, state):
        self.tracker[:, 1:] = self.tracker[:, :-1].clone()
        self.tracker[:, 0] = state.reshape(self.num_envs)


class _Camera:
    """What the task reads from ``self.Camera`` (``rover.py:92-99``): the Heightmap and the ray count."""

    def __init__(self, heightmap: Heightmap):
        self.heightmap = heightmap
        self.heightmap_distribution = heightmap.get_distribution()
        self.num_exteroceptive = self.heightmap_distribution.shape[0]
        self._engine = None                 # bound by RoverTask once the library holds the scene

    def get_num_exteroceptive(self):
        return self.num_exteroceptive

    def get_depths(self, positions, rotations):
        """camera.py:60-145: rover positions [E,3] and euler rotations [E,3] -> (distances [E,P], intersection points [E,P,3],
        ray sources [E,P,3]), through ``rover_get_depths`` (the step's own ray pipeline; the step path itself does not call this —
        it casts the heightmap rays inside ``rover_step``)."""
        if self._engine is None:
            raise _lib.RoverError("Camera.get_depths: the camera is not bound to a task yet")
        return self._engine.get_depths(positions.float().contiguous(), rotations.float().contiguous())


class _RockDetector:
    """What the task reads from ``self.Rock_detector`` (``rover.py:94,291``): ``get_collisions`` on caller-supplied poses."""

    def __init__(self):
        self._engine = None

    def get_collisions(self, positions, rotations, joint_states):
        """rock_detect.py:52-149: rover positions [E,3], euler rotations [E,3], joint positions [E,13] -> (wheel distances [E,24],
        body distances [E,2]) through ``rover_get_collisions`` (the step path casts these rays inside ``rover_step``)."""
        if self._engine is None:
            raise _lib.RoverError("Rock_detector.get_collisions: the detector is not bound to a task yet")
        return self._engine.get_collisions(positions.float().contiguous(), rotations.float().contiguous(),
                                           joint_states.float().contiguous())


class RoverTask(RLTask):
    def __init__(self, name, sim_config, env, offset=None, *, scene=None, distribution=None, fused=True,
                 device_reset=True, ray_precision="fp32", num_envs_global=None, env_offset=0, stone_mask_margin=None,
                 cell_index_mode="cuda_rcp", graph=False) -> None:
        """``scene``: a ``synth.Scene`` (or ``assets.load_reference_assets(root)``) with the terrain / rocks KNN maps,
        stone list and heightfield the reference loads from disk (:92-94,:144,:210).  ``distribution``: optional
        (points [P,3] f64, sparse_idx, dense_idx); default = the reference's native 1634-point set.
        ``stone_mask_margin``: if not None, every step also fills ``self.stone_collision`` [E] int64 with the stone_info
        occupancy mask ``nearest_rock(pos_xy) <= margin`` (the clearance of :536-539) — an ADDITIONAL output that
        the reference's step does not have and that never feeds reward or done.
        ``cell_index_mode``: how the cell lookup ``(xy - shift) / 0.1`` (camera.py:241, rock_detect.py:381, rover.py:590) is
        rounded.  Default "cuda_rcp" = the reference AS DEPLOYED (rover.py:90 pins cuda:0, where ATen multiplies by the
        reciprocal); "cpu_div" = ATen's CPU division, what the golden vectors (captured on the CPU) pin."""
        if scene is None:
            raise ValueError("RoverTask needs the terrain assets: pass scene=assets.load_reference_assets(root) "
                             "or a synth.Scene")
        if not isinstance(sim_config, SimConfig) and not hasattr(sim_config, "task_config"):
            raise TypeError("sim_config must expose .config and .task_config (rover.py:103-106)")
        self.vis_rocks = False
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self._device = self._cfg["sim_device"]
        if not str(self._device).startswith("cuda"):
            raise _lib.RoverError("RoverTask computes on the GPU only (sim_device must be cuda:N)")
        self._scene = scene
        self.shift = torch.tensor(list(scene.shift), device=self._device, dtype=torch.float32)   # :91
        hm = Heightmap(self._device) if distribution is None else Heightmap(self._device, *distribution)
        self.Camera = _Camera(hm)
        self.Rock_detector = _RockDetector()                                              # :94
        self.num_exteroceptive = self.Camera.get_num_exteroceptive()
        self.global_step = 0
        # run seed (cfg/config.yaml `seed`): mixed into every library RNG draw, so that seed sweeps re-randomise the resets
        self._seed = int(self._cfg.get("seed", 0)) if hasattr(self._cfg, "get") else 0
        self._num_proprioceptive = 4                                                      # :98
        self._num_observations = self._num_proprioceptive + hm.get_num_sparse_vector() + hm.get_num_dense_vector()
        self._num_actions = 2                                                             # :100
        self.curriculum_level = 1                                                         # :101
        self._num_envs = self._task_cfg["env"]["numEnvs"]                                # :113
        self._env_spacing = self._task_cfg["env"]["envSpacing"]
        self._rover_positions = torch.tensor([27.0, 30.0, 0.0])                           # :115
        self._reset_dist = self._task_cfg["env"]["resetDist"]
        self.max_episode_length = 3000                                                    # :119
        self.curriculum = self._task_cfg["env"]["terrain"]["curriculum"]
        self.is_evaluation = False
        self.target_positions = torch.zeros((self._num_envs, 3), device=self._device, dtype=torch.float32)   # :142
        self.stone_info = torch.from_numpy(synth.read_stone_info_array(scene.stone_info_raw)).to(self._device)  # :144
        self.save_teacher_data = self._task_cfg["collect_data"]
        self.linear_velocity = Memory(self._num_envs, 1, 3, self._device)                 # :154-155
        self.angular_velocity = Memory(self._num_envs, 1, 3, self._device)
        self.rew_scales = self._task_cfg["rewards"]                                       # :158
        self.actions_nn = torch.zeros((self._num_envs, self._num_actions, 3), device=self._device)   # :183 (the kernels keep it: :366)
        self.horizontal_scale = scene.horizontal_scale                                    # :212
        self.vertical_scale = scene.vertical_scale                                        # :213
        self.heightmap = scene.heightmap.to(self._device)                                 # :210
        self._fused = bool(fused)
        self._device_reset = bool(device_reset)
        self._use_graph = bool(graph)
        if self._use_graph and not (self._fused and self._device_reset):
            raise ValueError("graph=True replays the fused, device-reset step: it needs fused=True and device_reset=True")
        RLTask.__init__(self, name, env)                                                  # :184

        dev_index = torch.device(self._device).index or 0
        self._engine = _lib.Engine(self._num_envs, device=dev_index, num_envs_global=num_envs_global or 0,
                                   env_offset=env_offset, curriculum_level=self.curriculum_level,
                                   max_episode_length=self.max_episode_length, rewards=self.rew_scales)
        self._engine.set_scene(scene, (hm.distribution, hm.coarse_idx, hm.fine_idx))
        # "fp32" = the reference with Camera.dtype = float32 (parity target); "fp16_as_shipped" = Camera.dtype = float16
        # exactly as the reference ships (camera.py:55); "fp16_sources" = fp16 ray origins, f32 arithmetic
        self._engine.set_option("ray_precision", {"fp32": 0, "fp16_sources": 1, "fp16_as_shipped": 2}[ray_precision])
        self._engine.set_option("cell_index_mode", {"cpu_div": 0, "cuda_rcp": 1}[cell_index_mode])
        self.Camera._engine = self._engine
        self.Rock_detector._engine = self._engine
        self._env_offset = int(env_offset)

        # persistent side-state the three methods hand to each other (:274-283, :343, :667)
        e, dev = self._num_envs, self._device
        self.rover_rotation = torch.zeros(e, 3, device=dev)
        self.heading_diff = torch.zeros(e, device=dev)
        self.rover_rot = torch.zeros(e, 3, device=dev)
        self.rock_collison = torch.zeros(e, dtype=torch.long, device=dev)
        self._stone_margin = stone_mask_margin
        self.stone_collision = None if stone_mask_margin is None else torch.zeros(e, dtype=torch.long, device=dev)
        self.reset_env_ids_buf = torch.zeros(e, dtype=torch.long, device=dev)
        self._n_reset = torch.zeros(1, dtype=torch.int32, device=dev)
        self._compaction_fresh = False
        for k in _lib.EXTRAS:                                                             # :524-531
            self.extras[k] = torch.zeros(e, device=dev, dtype=torch.long if k == "collision_penalty" else torch.float32)
        self.initial_pos = torch.zeros(e, 3, device=dev)
        self.base_pos = torch.zeros(e, 3, device=dev)
        self._rover = None
        self._balls = None
        self._sin = self._sout = None
        # graph replay (graph=True): the actions land in a persistent buffer, the step counter of the device RNG lives on the device
        self._actions_buf = torch.zeros(e, self._num_actions, device=dev)
        self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._pre_graph = self._post_graph = None
        self._graph_level = self._graph_key = None
        self._bound_key = self._bound_pre = self._bound_reset = None

    # ------------------------------------------------------------------------------------------------
    # scene / views (rover.py:196-230, 455-458) — USD is out of scope; views are host pose feeders
    # ------------------------------------------------------------------------------------------------
    def set_up_scene(self, scene=None, spawn_positions=None) -> None:
        """Creates the views and the spawn poses: ``avoid_pos_rock_collision`` + heightfield z + 0.5 (:208-220).
        ``spawn_positions`` [E,3] replaces the cloner grid positions (default: a grid around :115's origin)."""
        e, dev = self._num_envs, self._device
        self._rover = RoverView(e, dev, name="rover_view")
        self._balls = RigidPrimView(e, dev, name="targets_view")
        if spawn_positions is None:
            side = int(math.ceil(math.sqrt(e)))
            ix = torch.arange(e, device=dev) // side
            iy = torch.arange(e, device=dev) % side
            spawn_positions = torch.stack((self._rover_positions[0] + (ix - side / 2) * self._env_spacing,
                                           self._rover_positions[1] + (iy - side / 2) * self._env_spacing,
                                           torch.zeros(e, device=dev)), dim=1).float()
        positions = self.avoid_pos_rock_collision(spawn_positions.to(dev).float().contiguous().clone())   # :215
        height = self.get_pos_height(self.heightmap, positions[:, 0:2], self.horizontal_scale, self.vertical_scale,
                                     self.shift[0:2])                                     # :216
        self.position_z_offset = torch.ones(height.shape, device=dev) * 0.5              # :217
        positions[:, 2] = height + self.position_z_offset                                 # :218
        self.initial_pos = positions
        self._rover.set_world_poses(self.initial_pos, self._rover.get_world_poses()[1])  # :220
        self._balls._pos = self.target_positions       # the target balls ARE the goals (:584 moves them there): one tensor, no per-step copy
        self._bind()

    def _bind(self):
        """(Re)build the C-ABI argument structs over the persistent tensors."""
        pos, quat = self._rover.get_world_poses()
        eng = self._engine
        self._sin = eng.make_in(pos, quat, self._rover.get_joint_positions(), self.target_positions,
                                self.linear_velocity.tracker, self.angular_velocity.tracker, self.rover_rot,
                                self.progress_buf)
        self._sout = eng.make_out(self.obs_buf, rew=self.rew_buf, reset=self.reset_buf, rock_collision=self.rock_collison,
                                  extras=self.extras, reset_ids=self.reset_env_ids_buf, n_reset=self._n_reset,
                                  euler=self.rover_rotation, heading_diff=self.heading_diff,
                                  stone_collision=self.stone_collision, stone_margin=self._stone_margin or 0.0)
        self.rover_positions = pos
        self._bound_key = None

    def reset(self):
        RLTask.reset(self)
        self._compaction_fresh = False

    def post_reset(self):
        self.base_pos = torch.zeros((self.num_envs, 3), dtype=torch.float, device=self._device)     # :456
        self.initial_root_pos, self.initial_root_rot = self._rover.get_world_poses(clone=True)
        self.initial_ball_pos, self.initial_ball_rot = self._balls.get_world_poses(clone=True)

    # ------------------------------------------------------------------------------------------------
    # the hot path
    # ------------------------------------------------------------------------------------------------
    def get_observations(self) -> dict:
        """rover.py:272-336 → rover_get_observations (prep + ray cast + collision mask + obs assembly)."""
        self.rover_positions = self._rover.get_world_poses()[0]
        self._engine.get_observations(self._sin, self._sout)
        return {self._rover.name: {"obs_buf": self.obs_buf}}

    def calculate_metrics(self) -> None:
        """rover.py:460-531 → rover_calculate_metrics (rew_buf + the 8 extras, in place)."""
        self._engine.calculate_metrics(self._sin, self._sout)

    def is_done(self) -> None:
        """rover.py:610-647 → rover_is_done (tilt from the PRE-physics euler ``self.rover_rot``)."""
        self._engine.is_done(self._sin, self._sout)
        self._compaction_fresh = False

    def post_physics_step(self):
        if not self._fused:
            return RLTask.post_physics_step(self)
        if self._is_playing():
            self.rover_positions = self._rover.get_world_poses()[0]
            if self._use_graph and self._post_graph is not None and self._graph_level == self.curriculum_level:
                self._post_graph.replay()
            else:
                self._engine.step(self._sin, self._sout, increment_progress=True, compact=True)
            self._compaction_fresh = True
        else:
            self.progress_buf[:] += 1
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def check_collision(self, wheel_dists, body_dists):
        """rover.py:663-668, kept for API parity on caller-supplied distances (the step path computes the mask
        inside the kernels)."""
        nearest_wheel = torch.min(wheel_dists, dim=1)[0]
        nearest_body = torch.min(body_dists, dim=1)[0]
        rc = torch.where(torch.abs(nearest_wheel) < 0.8, torch.ones_like(self.reset_buf), torch.zeros_like(self.reset_buf))
        self.rock_collison.copy_(torch.where(torch.abs(nearest_body) < 0.45, torch.ones_like(self.reset_buf), rc))

    # ------------------------------------------------------------------------------------------------
    # action side + resets (rover.py:338-453)
    # ------------------------------------------------------------------------------------------------
    def pre_physics_step(self, actions, reset_yaw_deg=None) -> None:
        """rover.py:338-414.  Default (``device_reset=True``): three launches and NO host sync —
        ``rover_pre_physics_step`` (pre-physics euler :343, history :379-380, Ackermann + joint-target scatter
        :391-414) and ``rover_reset_envs`` (reset_idx + set_targets for the compacted ids, count read on the device).
        ``device_reset=False`` keeps the reference's control flow (host ``len()`` of the id list, then the methods)."""
        self.global_step += 1
        self.rover_loc = self._rover.get_world_poses()[0]
        if self.global_step == 10:                                                               # :344-353
            self.curriculum_level = 2
            self._engine.set_curriculum_level(2)
        # done compaction (:356) — already produced by the fused step unless reset_buf was touched since
        if not self._compaction_fresh:
            self._engine.compact_resets(self.reset_buf, self.reset_env_ids_buf, self._n_reset)
        self._compaction_fresh = False
        rv = self._rover
        if self._device_reset:
            if self._use_graph and reset_yaw_deg is None and self.global_step > 10:
                self._actions_buf.copy_(actions)            # (device, dtype and layout conversions included)
                key = self._ptr_key(self._actions_nn_buf())
                if self._pre_graph is None or self._graph_level != self.curriculum_level or self._graph_key != key:
                    self._capture_graphs()
                    self._graph_key = key
                self._pre_graph.replay()
                return
            _actions = actions.to(self._device).float().contiguous()
            self._launch_pre_physics(_actions, reset_yaw_deg, seed=self._rng_seed(), seed_dev=None)
            return
        _actions = actions.to(self._device).float().contiguous()
        self.actions_nn = torch.cat((torch.reshape(_actions, (self.num_envs, self._num_actions, 1)), self.actions_nn), 2)[:, :, 0:3]   # :366
        self._engine.quat_to_euler(rv.get_world_poses()[1], out=self.rover_rot)                  # :343
        n = int(self._n_reset.item())                      # the reference's len(reset_env_ids) is the same host sync
        if n > 0:
            reset_env_ids = self.reset_env_ids_buf[:n] - self._env_offset
            self.reset_idx(reset_env_ids, yaw_deg=None if reset_yaw_deg is None else reset_yaw_deg[:n])
            self.set_targets(reset_env_ids)
        self.linear_velocity.input_state(_actions[:, 0])                                         # :379-380
        self.angular_velocity.input_state(_actions[:, 1])
        steering_angles, motor_velocities = self._engine.ackermann(_actions[:, 0].contiguous(), _actions[:, 1].contiguous())  # :391
        positions = torch.zeros((self._rover.count, 4), dtype=torch.float32, device=self._device)
        velocities = torch.zeros((self._rover.count, 6), dtype=torch.float32, device=self._device)
        positions[:, 0] = steering_angles[:, 1]            # FR  (:400-409)
        positions[:, 1] = steering_angles[:, 5]            # RR
        positions[:, 2] = steering_angles[:, 0]            # FL
        positions[:, 3] = steering_angles[:, 4]            # RL
        velocities[:, 0] = motor_velocities[:, 1]          # FR
        velocities[:, 1] = motor_velocities[:, 3]          # CR
        velocities[:, 2] = motor_velocities[:, 5]          # RR
        velocities[:, 3] = motor_velocities[:, 0]          # FL
        velocities[:, 4] = motor_velocities[:, 2]          # CL
        velocities[:, 5] = motor_velocities[:, 4]          # RL
        self._rover.set_joint_position_targets(positions, indices=None, joint_indices=self._rover.actuated_pos_indices)
        self._rover.set_joint_velocity_targets(velocities, indices=None, joint_indices=self._rover.actuated_vel_indices)

    def _launch_pre_physics(self, _actions, reset_yaw_deg, seed, seed_dev):
        """The device-reset form of :338-414 as launches on the current stream (eager, or while a graph is being captured)."""
        rv = self._rover
        ann = self._actions_nn_buf()
        # The task's buffers persist: their checks and the argument structs are made once and reused while the same tensors are in place
        # (the actions — usually a fresh policy output — are passed and checked per call).  Caller-supplied yaws (tests) and a captured graph's device seed take the plain path.
        key = self._ptr_key(ann)
        if reset_yaw_deg is None and seed_dev is None and self._bound_key == key:
            self._bound_pre(_actions)
            self._bound_reset(seed)
        else:
            # euler_pre must see the PRE-reset orientation (:343 runs before :359), so this kernel goes first
            pre_kw = dict(euler_pre=self.rover_rot, pos_targets13=rv._joint_pos_targets, vel_targets13=rv._joint_vel_targets, actions_nn=ann)
            reset_kw = dict(n_reset_dev=self._n_reset, joint_pos13=rv._joint_pos, joint_vel13=rv._joint_vel, base_pos3=self.base_pos,
                            target3=self.target_positions, radius=8.0)
            reset_args = (self.reset_env_ids_buf, self.initial_pos, rv._pos, rv._quat, self.reset_buf, self.progress_buf)
            self._engine.pre_physics_step(_actions, rv._quat, self.linear_velocity.tracker, self.angular_velocity.tracker, **pre_kw)
            self._engine.reset_envs(*reset_args, yaw_deg=reset_yaw_deg, seed=seed, seed_dev=seed_dev, **reset_kw)
            if reset_yaw_deg is None and seed_dev is None:
                self._bound_pre = self._engine.bind_pre_physics(rv._quat, self.linear_velocity.tracker,
                                                                self.angular_velocity.tracker, **pre_kw)
                self._bound_reset = self._engine.bind_reset_envs(*reset_args, **reset_kw)
                self._bound_key = key
        if self._balls._pos.data_ptr() != self.target_positions.data_ptr():                      # :584 (visual only)
            self._balls._pos.copy_(self.target_positions)

    def _ptr_key(self, ann):
        """Addresses of every tensor a cached argument struct / captured graph holds that an assignment could have replaced since (the
        actions are not among them: a bound call takes them per step, a captured graph reads the task's own actions buffer)."""
        rv = self._rover
        return (ann.data_ptr(), self.initial_pos.data_ptr(), self.base_pos.data_ptr(), self.reset_buf.data_ptr(),
                self.progress_buf.data_ptr(), self.target_positions.data_ptr(), rv._pos.data_ptr(), rv._quat.data_ptr(),
                rv._joint_pos.data_ptr())

    def _actions_nn_buf(self):
        """self.actions_nn as the contiguous float32 [E, 2, 3] device tensor the kernel shifts in place (a caller may have rebound it)."""
        a = self.actions_nn
        if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and tuple(a.shape) == (self._num_envs, self._num_actions, 3)):
            a = self.actions_nn = a.to(self._device).float().contiguous()
        return a

    def _capture_graphs(self):
        """Captures the two halves of a step as hipGraphs over the task's persistent tensors.  The device RNG's step counter is a
        device word the captured graph advances itself (`_rng_seed` is additive in the step for that), so a replay draws what the
        eager call of the same global step would."""
        torch.cuda.synchronize()
        self._seed_dev.fill_(_to_i64((self.global_step - 1) * _STEP_KEY))       # the captured add_ makes it this step's
        pre, post = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(pre):
            self._seed_dev.add_(_to_i64(_STEP_KEY))
            self._launch_pre_physics(self._actions_buf, None, seed=self._rng_seed(step=0), seed_dev=self._seed_dev)
        with torch.cuda.graph(post):
            self._engine.step(self._sin, self._sout, increment_progress=True, compact=True)
        self._pre_graph, self._post_graph, self._graph_level = pre, post, self.curriculum_level

    def reset_idx(self, env_ids, yaw_deg=None):
        """rover.py:416-453.  The reference builds scipy's (x,y,z,w) quaternion of a rotation about x and feeds
        it to Isaac as (w,x,y,z) (:429-431,:449), i.e. w = sin(d/2), z = cos(d/2): a pure yaw of (180° − d).
        Reproduced as is; ``yaw_deg`` replaces ``random.randint(0, 360)`` when given."""
        num_resets = len(env_ids)
        if yaw_deg is None:
            yaw_deg = torch.randint(0, 361, (num_resets,), device=self._device)
        half = torch.deg2rad(yaw_deg.to(self._device).float()) / 2
        reset_orientation = torch.stack((torch.sin(half), torch.zeros_like(half), torch.zeros_like(half), torch.cos(half)), dim=1)
        dof = torch.zeros((num_resets, 13), device=self._device)
        self._rover.set_joint_positions(dof, indices=env_ids)
        self._rover.set_joint_velocities(dof, indices=env_ids)
        self.base_pos[env_ids] = self.initial_pos[env_ids]
        self._rover.set_world_poses(self.base_pos[env_ids], reset_orientation.float(), env_ids)
        self.reset_buf[env_ids] = 0                                                               # :452-453
        self.progress_buf[env_ids] = 0

    def set_targets(self, env_ids, draws=None):
        """rover.py:566-584: goals at radius 8 validated against the stone list, goal z from the heightfield."""
        self.generate_goals(env_ids, radius=8, draws=draws)
        envs_long = env_ids.long()
        self._balls.set_world_poses(self.target_positions[envs_long], self.initial_ball_rot[envs_long].clone(), indices=env_ids)

    def generate_goals(self, env_ids, radius, draws=None):
        """rover.py:544-549 (+ random_goals :554-564, check_goal_collision :533-542, goal z :582-583) as ONE
        kernel, including the reference's ``env_ids = mask*env_ids`` aliasing.  ``draws`` [n_draws, n] replaces
        ``torch.rand``; otherwise the library's Philox stream keyed by the global step."""
        self._engine.generate_goals(env_ids.long().contiguous(), self.initial_pos, self.target_positions, radius=radius,
                                    draws=draws, max_draws=256, seed=self._rng_seed())

    def _rng_seed(self, step=None):
        """Philox key of this step's device-side draws: (run seed, global step); the counter is the GLOBAL env id, so every
        shard of a multi-GPU run and every run seed draws its own stream (the reference's torch.rand / random follow the
        run seed the same way).  Additive in the step, so that a captured graph can keep the step part on the device."""
        step = self.global_step if step is None else step
        return (self._seed * 0x9E3779B97F4A7C15 + step * _STEP_KEY) & 0xFFFFFFFFFFFFFFFF

    def check_goal_collision(self, env_ids):
        """rover.py:533-542."""
        c = self._engine.clearance(self.target_positions[env_ids][:, 0:2].contiguous())
        mask = (c <= 1.0).to(env_ids.dtype)
        return mask * env_ids, int(mask.sum().item())

    def get_pos_height(self, heightmap, depth_points, horizontal_scale, vertical_scale, shift):
        """rover.py:588-608 on the heightfield loaded into the engine (arguments kept for signature parity)."""
        return self._engine.sample_height(depth_points.contiguous().float())

    def avoid_pos_rock_collision(self, curr_pos):
        """rover.py:649-661: per env, x += 0.05 while the stone clearance is <= 1.4."""
        return self._engine.shift_spawns(curr_pos)

    def close(self):
        self._engine.close()
