"""Task constants of the reference's Hydra YAML (``cfg/task/Rover.yaml``, ``cfg/config.yaml``) as plain dicts,
and a ``SimConfig`` stand-in exposing the two attributes the task reads
(``sim_config.config`` / ``sim_config.task_config``, ``rover.py:103-106``, ``rl_task.py:61-73``).
Hydra / OmegaConf are not part of the hot path and are not reimplemented."""
from __future__ import annotations

import copy


def default_task_config(num_envs: int = 512) -> dict:
    """cfg/task/Rover.yaml:1-46 (only the keys the task reads)."""
    return {
        "name": "Rover",
        "collect_data": False,                                   # Rover.yaml:5
        "env": {
            "numEnvs": num_envs,                                 # :11
            "envSpacing": 1.2,                                   # :12
            "resetDist": 3.0,                                    # :13
            "clipObservations": 5.0,                             # :16
            "clipActions": 1.0,                                  # :17
            "controlFrequencyInv": 5,                            # :18
            "terrain": {"curriculum": False, "numLevels": 10, "maxInitMapLevel": 0},   # :25-29
        },
        "rewards": {                                             # :37-46
            "pos_reward": 1.0, "terminalReward": 0, "collision_reward": 0.3, "heading_contraint_reward": 0.05,
            "motion_contraint_reward": -0.01, "goal_angle_reward": 0.3, "boogie_contraint_reward": 0.5,
        },
        "sim": {"dt": 0.05},
    }


def default_config(num_envs: int = 512, device: str = "cuda:0") -> dict:
    """cfg/config.yaml keys read at rl_task.py:61-73."""
    task = default_task_config(num_envs)
    return {"test": False, "sim_device": device, "rl_device": device, "seed": 42, "task": task}


class SimConfig:
    def __init__(self, config: dict | None = None, num_envs: int = 512, device: str = "cuda:0"):
        self._config = copy.deepcopy(config) if config is not None else default_config(num_envs, device)

    @property
    def config(self):
        return self._config

    @property
    def task_config(self):
        return self._config["task"]
