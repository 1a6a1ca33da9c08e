"""RLTask — buffers and post_physics_step orchestration of the reference's base task.

Mirrors ``omniisaacgymenvs/tasks/base/rl_task.py`` (``RLTask.__init__`` :49-96, ``cleanup`` :98-107,
properties :139-200, ``get_states`` :210, ``get_extras`` :218, ``reset`` :226-229, ``post_physics_step``
:239-259) without Isaac: the ``BaseTask`` parent, ``GridCloner`` and USD scene parts (:109-137) are out of
scope, and ``gym.spaces.Box`` (absent in this image) is replaced by a minimal ``Box``.
"""
from __future__ import annotations

import numpy as np
import torch


class Box:
    """Stand-in for ``gym.spaces.Box``: what callers of the task read (``.low/.high/.shape/.dtype``)."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class RLTask:
    """Subclasses set ``_cfg``, ``_sim_config``, ``_num_envs``, ``_num_observations``, ``_num_actions``
    before calling ``RLTask.__init__`` (like ``rover.py:99-184`` does)."""

    def __init__(self, name, env, offset=None) -> None:
        self._name = name
        self._offset = offset
        self.test = self._cfg["test"]                                                    # rl_task.py:61
        self._device = self._cfg["sim_device"]                                           # :62
        self.randomize_actions = False
        self.randomize_observations = False
        self.clip_obs = self._cfg["task"]["env"].get("clipObservations", np.inf)        # :69
        self.clip_actions = self._cfg["task"]["env"].get("clipActions", np.inf)         # :70
        self.rl_device = self._cfg.get("rl_device", "cuda:0")                            # :71
        self.control_frequency_inv = self._cfg["task"]["env"].get("controlFrequencyInv", 1)   # :73
        self._env = env
        if not hasattr(self, "_num_agents"):
            self._num_agents = 1
        if not hasattr(self, "_num_states"):
            self._num_states = 0
        if not hasattr(self, "action_space"):                                            # :85-90
            self.action_space = Box(np.ones(self.num_actions) * -1.0, np.ones(self.num_actions) * 1.0)
        if not hasattr(self, "observation_space"):
            self.observation_space = Box(np.ones(self.num_observations) * -np.inf, np.ones(self.num_observations) * np.inf)
        if not hasattr(self, "state_space"):
            self.state_space = Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.cleanup()

    def cleanup(self) -> None:
        """rl_task.py:98-107: persistent buffers; methods mutate them in place and return the same objects."""
        self.obs_buf = torch.zeros((self._num_envs, self.num_observations), device=self._device, dtype=torch.float)
        self.states_buf = torch.zeros((self._num_envs, self.num_states), device=self._device, dtype=torch.float)
        self.rew_buf = torch.zeros(self._num_envs, device=self._device, dtype=torch.float)
        self.reset_buf = torch.ones(self._num_envs, device=self._device, dtype=torch.long)
        self.progress_buf = torch.zeros(self._num_envs, device=self._device, dtype=torch.long)
        self.extras = {}

    @property
    def name(self):
        return self._name

    @property
    def device(self):
        return self._device

    @property
    def num_envs(self):
        return self._num_envs

    @property
    def num_actions(self):
        return self._num_actions

    @property
    def num_observations(self):
        return self._num_observations

    @property
    def num_states(self):
        return self._num_states

    @property
    def num_agents(self):
        return self._num_agents

    def get_states(self):
        return self.states_buf

    def get_extras(self):
        return self.extras

    def reset(self):
        """Flags all environments for reset (rl_task.py:226-229).  In place, so kernel-facing pointers stay valid."""
        self.reset_buf.fill_(1)

    def pre_physics_step(self, actions):
        pass

    def _is_playing(self) -> bool:
        world = getattr(self._env, "_world", None)
        return True if world is None else bool(world.is_playing())

    def post_physics_step(self):
        """rl_task.py:239-259."""
        self.progress_buf[:] += 1
        if self._is_playing():
            self.get_observations()
            self.get_states()
            self.calculate_metrics()
            self.is_done()
            self.get_extras()
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras
