"""Multi-GPU sharding of the rover step: one process per GPU, contiguous env blocks.

The reference is single-device (``rover.py:90`` hard-codes ``cuda:0``) and has no collective to mirror.
Envs are independent, terrain / rock / stone tables are read-only and replicated on every GPU, so the
only exchange of a step is handing (obs, reward, done) of every shard to the learner rank.  That is done
as ONE grouped point-to-point operation (``batch_isend_irecv`` = a single ncclGroup of send/recv on RCCL):
each rank sends its three buffers straight into the root's global tensors, so the 7 inbound shards of an
8-GPU node arrive over 7 distinct xGMI links and nothing is re-packed or copied afterwards.  (A ring
all-gather would be per-link bound: 7 hops instead of 1.)

The same code runs on ``gloo`` (CPU tensors) for the world_size-2 tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(num_envs_global: int, world: int, rank: int):
    """Contiguous block of envs owned by ``rank`` (rank g owns [g*E/N, (g+1)*E/N))."""
    if num_envs_global % world:
        raise ValueError(f"num_envs_global={num_envs_global} must be divisible by world size {world}")
    per = num_envs_global // world
    return rank * per, (rank + 1) * per


class StepGather:
    """Owns the (obs, rew, reset) buffers of one shard and gathers all shards on ``root``.

    On the root the local buffers are views of chunk ``rank`` of the global tensors, so the step kernels
    write the root's own shard in place; other ranks send directly into the root's chunks.
    """

    def __init__(self, num_envs_local: int, obs_dim: int, device, world: int = 1, rank: int = 0, root: int = 0,
                 group=None):
        self.E, self.W, self.world, self.rank, self.root, self.group = num_envs_local, obs_dim, world, rank, root, group
        self.is_root = rank == root
        n = num_envs_local * (world if self.is_root else 1)
        self.obs_g = torch.zeros(n, obs_dim, dtype=torch.float32, device=device)
        self.rew_g = torch.zeros(n, dtype=torch.float32, device=device)
        self.reset_g = torch.ones(n, dtype=torch.int64, device=device)     # rl_task.py:105: reset_buf starts at 1
        lo = rank * num_envs_local if self.is_root else 0
        self.obs = self.obs_g[lo:lo + num_envs_local]
        self.rew = self.rew_g[lo:lo + num_envs_local]
        self.reset = self.reset_g[lo:lo + num_envs_local]

    def local_views(self):
        return self.obs, self.rew, self.reset

    def gather(self):
        """After the step kernels of every rank were enqueued: returns the global (obs, rew, reset) on root."""
        if self.world == 1:
            return self.obs_g, self.rew_g, self.reset_g
        ops = []
        if self.is_root:
            for r in range(self.world):
                if r == self.root:
                    continue
                s = slice(r * self.E, (r + 1) * self.E)
                ops += [dist.P2POp(dist.irecv, self.obs_g[s], r, self.group),
                        dist.P2POp(dist.irecv, self.rew_g[s], r, self.group),
                        dist.P2POp(dist.irecv, self.reset_g[s], r, self.group)]
        else:
            ops = [dist.P2POp(dist.isend, self.obs, self.root, self.group),
                   dist.P2POp(dist.isend, self.rew, self.root, self.group),
                   dist.P2POp(dist.isend, self.reset, self.root, self.group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return (self.obs_g, self.rew_g, self.reset_g) if self.is_root else None
