"""Host-side sim-state provider: the subset of Isaac's ``ArticulationView`` / ``RigidPrimView`` the rover task
touches, backed by plain device tensors.  PhysX itself is out of scope: poses are *fed in*
(``feed``) from a recorded / synthetic trajectory or from the toy integrator in ``vec_env.py``.

Reference: ``robots/articulations/views/rover_view.py:5-49`` (index lists :45-46) and the call sites
``rover.py:208,220,274-275,291,342-343,396,412-414,439-440,449,457-458,476,584``.
"""
from __future__ import annotations

import torch


class RoverView:
    NUM_DOF = 13                                      # legend: rock_detect.py:175-187

    def __init__(self, num_envs: int, device, name: str = "rover_view"):
        self.name = name
        self.count = num_envs
        self.num_dof = self.NUM_DOF
        self._device = device
        self._pos = torch.zeros(num_envs, 3, device=device)
        self._quat = torch.zeros(num_envs, 4, device=device)
        self._quat[:, 0] = 1.0
        self._joint_pos = torch.zeros(num_envs, self.NUM_DOF, device=device)
        self._joint_vel = torch.zeros(num_envs, self.NUM_DOF, device=device)
        self._joint_pos_targets = torch.zeros(num_envs, self.NUM_DOF, device=device)
        self._joint_vel_targets = torch.zeros(num_envs, self.NUM_DOF, device=device)
        self._actuated_dof_indices = list(range(self.NUM_DOF))
        self._actuated_vel_indices = [10, 5, 12, 9, 3, 11]    # [FR, CR, RR, FL, CL, RL], rover_view.py:45
        self._actuated_pos_indices = [6, 8, 4, 7]             # [FR, RR, FL, RL],         rover_view.py:46

    actuated_dof_indices = property(lambda self: self._actuated_dof_indices)
    actuated_pos_indices = property(lambda self: self._actuated_pos_indices)
    actuated_vel_indices = property(lambda self: self._actuated_vel_indices)

    # ---- what the task reads --------------------------------------------------------------------
    def get_world_poses(self, clone: bool = False):
        return (self._pos.clone(), self._quat.clone()) if clone else (self._pos, self._quat)

    def get_joint_positions(self):
        return self._joint_pos

    def get_joint_velocities(self):
        return self._joint_vel

    # ---- what the task writes -------------------------------------------------------------------
    @staticmethod
    def _idx(indices):
        return slice(None) if indices is None else indices.long()

    def set_world_poses(self, positions=None, orientations=None, indices=None):
        i = self._idx(indices)
        if positions is not None:
            self._pos[i] = positions
        if orientations is not None:
            self._quat[i] = orientations

    def _scatter(self, dst, values, indices, joint_indices):
        cols = self._actuated_dof_indices if joint_indices is None else list(joint_indices)
        if indices is None:
            dst[:, cols] = values
        else:
            dst[indices.long()[:, None], torch.as_tensor(cols, device=self._device)] = values

    def set_joint_positions(self, positions, indices=None, joint_indices=None):
        self._scatter(self._joint_pos, positions, indices, joint_indices)

    def set_joint_velocities(self, velocities, indices=None, joint_indices=None):
        self._scatter(self._joint_vel, velocities, indices, joint_indices)

    def set_joint_position_targets(self, positions, indices=None, joint_indices=None):
        self._scatter(self._joint_pos_targets, positions, indices, joint_indices)

    def set_joint_velocity_targets(self, velocities, indices=None, joint_indices=None):
        self._scatter(self._joint_vel_targets, velocities, indices, joint_indices)

    # ---- pose feeder ----------------------------------------------------------------------------
    def feed(self, positions=None, orientations=None, joint_positions=None):
        """Overwrite the sim state in place (pointers handed to the kernels stay valid)."""
        if positions is not None:
            self._pos.copy_(positions)
        if orientations is not None:
            self._quat.copy_(orientations)
        if joint_positions is not None:
            self._joint_pos.copy_(joint_positions)


class RigidPrimView:
    """Target balls (``rover.py:221,458,584``): poses only."""

    def __init__(self, num_envs: int, device, name: str = "targets_view"):
        self.name = name
        self.count = num_envs
        self._pos = torch.zeros(num_envs, 3, device=device)
        self._quat = torch.zeros(num_envs, 4, device=device)
        self._quat[:, 0] = 1.0

    def get_world_poses(self, clone: bool = False):
        return (self._pos.clone(), self._quat.clone()) if clone else (self._pos, self._quat)

    def set_world_poses(self, positions=None, orientations=None, indices=None):
        i = slice(None) if indices is None else indices.long()
        if positions is not None:
            self._pos[i] = positions
        if orientations is not None:
            self._quat[i] = orientations
