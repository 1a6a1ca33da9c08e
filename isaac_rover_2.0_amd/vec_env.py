"""Minimal VecEnv so ``train.py``-style loops run without Isaac ("What calls the path").

Mirrors the contract of omni.isaac.gym's ``VecEnvBase`` as the reference uses it (``utils/task_util.py:45``,
``train.py:58-59``): ``set_task(task, backend, sim_params, init_sim)``, ``reset()``, ``step(actions)`` →
``(obs, rew, dones, info)`` with ``task.pre_physics_step`` before and ``task.post_physics_step`` after the
physics sub-steps, observations clipped to ``clip_obs`` and actions to ``clip_actions``.
PhysX is replaced by ``KinematicSim``: a toy unicycle integrator that moves each rover along the commanded
(linear, angular) velocity and keeps it on the heightfield — a pose FEEDER, not a simulator.
"""
from __future__ import annotations

import math

import torch


class KinematicSim:
    def __init__(self, task, dt: float = 0.05, max_lin: float = 1.0, max_ang: float = 1.0, ride_height: float = 0.3):
        self.task, self.dt, self.max_lin, self.max_ang, self.ride_height = task, dt, max_lin, max_ang, ride_height
        self._playing = True

    def is_playing(self):
        return self._playing

    def step(self, render: bool = False):
        t = self.task
        pos, quat = t._rover.get_world_poses()
        lin = t.linear_velocity.get_state(0) * self.max_lin
        ang = t.angular_velocity.get_state(0) * self.max_ang
        yaw = 2.0 * torch.atan2(quat[:, 3], quat[:, 0])
        yaw = yaw + ang * self.dt
        pos[:, 0] += lin * torch.cos(yaw) * self.dt
        pos[:, 1] += lin * torch.sin(yaw) * self.dt
        pos[:, 2] = t.get_pos_height(t.heightmap, pos[:, 0:2], t.horizontal_scale, t.vertical_scale, t.shift[0:2]) + self.ride_height
        quat[:, 0] = torch.cos(yaw / 2)
        quat[:, 1] = 0.0
        quat[:, 2] = 0.0
        quat[:, 3] = torch.sin(yaw / 2)


class VecEnv:
    def __init__(self, headless: bool = True, sim_device: int = 0):
        self._render = not headless
        self._task = None
        self._world = None

    def set_task(self, task, backend="torch", sim_params=None, init_sim=True, spawn_positions=None) -> None:
        self._task = task
        task._env = self
        self._world = KinematicSim(task, dt=(sim_params or {}).get("dt", 0.05))
        task.set_up_scene(spawn_positions=spawn_positions)
        task.post_reset()
        self.num_envs = task.num_envs
        self.observation_space = task.observation_space
        self.action_space = task.action_space

    def reset(self):
        self._task.reset()
        actions = torch.zeros((self.num_envs, self._task.num_actions), device=self._task.rl_device)
        obs, _, _, _ = self.step(actions)
        return obs

    def step(self, actions):
        task = self._task
        actions = torch.clamp(actions, -task.clip_actions, task.clip_actions).to(task.device)
        task.pre_physics_step(actions)
        for _ in range(task.control_frequency_inv):
            self._world.step(render=False)
        obs, rew, resets, extras = task.post_physics_step()
        obs = torch.clamp(obs, -task.clip_obs, task.clip_obs)
        return obs, rew, resets, extras

    def close(self):
        if self._task is not None:
            self._task.close()


def initialize_task(sim_config, env, scene, init_sim=True, **task_kw):
    """utils/task_util.py:30-47: build the task named in the config and bind it to the env."""
    from .tasks.rover import RoverTask
    task = RoverTask(name=sim_config.task_config.get("name", "Rover"), sim_config=sim_config, env=env, scene=scene, **task_kw)
    env.set_task(task=task, sim_params=sim_config.task_config.get("sim"), backend="torch", init_sim=init_sim)
    return task
