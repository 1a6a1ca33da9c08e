"""Multi-GPU sharding of the rover step: one process per GPU, contiguous env blocks.

The reference is single-device (``rover.py:90`` hard-codes ``cuda:0``) and has no collective to mirror.
Envs are independent, terrain / rock / stone tables are read-only and replicated on every GPU, so the
only exchange of a step is handing (obs f32, reward f32, done u8) of every shard to the learner rank — done
travels as one byte per env (``rover_step_out.done_u8``, written by the is_done stage next to the int64
``reset_buf``), not as the 8-byte flag.  That is done
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
    """Owns the (obs, rew, done) buffers of one shard and gathers all shards on ``root``.

    On the root the local buffers are views of chunk ``rank`` of the global tensors, so the step kernels
    write the root's own shard in place; other ranks send directly into the root's chunks.

    ``depth`` > 1 gives that many buffer sets for an OVERLAPPED gather: ``gather(d, wait=False)`` only enqueues
    the transfer (RCCL runs it on its own stream once the step kernels that wrote set ``d`` are done) and the
    caller goes on with the next step on set ``d + 1``; ``wait(d)`` orders the current stream behind the transfer
    before set ``d`` is written again / read by the learner.  With 7 x 11 MB per step converging on one GPU the
    transfer is a sizeable fraction of the 1.3 ms step; hidden under the next step it costs nothing.
    """

    def __init__(self, num_envs_local: int, obs_dim: int, device, world: int = 1, rank: int = 0, root: int = 0,
                 group=None, depth: int = 1):
        self.E, self.W, self.world, self.rank, self.root, self.group = num_envs_local, obs_dim, world, rank, root, group
        self.is_root = rank == root
        self.depth = depth
        n = num_envs_local * (world if self.is_root else 1)
        lo = rank * num_envs_local if self.is_root else 0
        self._sets = []
        for _ in range(depth):
            obs_g = torch.zeros(n, obs_dim, dtype=torch.float32, device=device)
            rew_g = torch.zeros(n, dtype=torch.float32, device=device)
            reset_g = torch.ones(n, dtype=torch.uint8, device=device)     # done flags, 1 B per env (reset_buf starts at 1, rl_task.py:105)
            self._sets.append((obs_g, rew_g, reset_g))
        self._pending = [None] * depth
        self._lo = lo
        # depth-1 interface (kept): the global tensors and the local views of set 0
        self.obs_g, self.rew_g, self.reset_g = self._sets[0]
        self.obs, self.rew, self.reset = self.local_views(0)

    def local_views(self, d: int = 0):
        o, r, z = self._sets[d]
        s = slice(self._lo, self._lo + self.E)
        return o[s], r[s], z[s]

    def global_views(self, d: int = 0):
        """The gathered (obs f32 [N*E, W], rew f32 [N*E], done u8 [N*E]) of set ``d`` on the root (None elsewhere); valid after
        ``wait(d)``."""
        return self._sets[d] if self.is_root else None

    def wait(self, d: int = 0):
        """Order the current stream behind the outstanding transfer of set ``d`` (no-op if there is none)."""
        works = self._pending[d]
        if works:
            for w in works:
                w.wait()
        self._pending[d] = None

    def gather(self, d: int = 0, wait: bool = True):
        """After the step kernels of every rank were enqueued: returns the global (obs, rew, reset) on root."""
        if self.world == 1:
            return self._sets[d]
        self.wait(d)
        obs_g, rew_g, reset_g = self._sets[d]
        ops = []
        if self.is_root:
            for r in range(self.world):
                if r == self.root:
                    continue
                s = slice(r * self.E, (r + 1) * self.E)
                ops += [dist.P2POp(dist.irecv, obs_g[s], r, self.group),
                        dist.P2POp(dist.irecv, rew_g[s], r, self.group),
                        dist.P2POp(dist.irecv, reset_g[s], r, self.group)]
        else:
            obs, rew, reset = self.local_views(d)
            ops = [dist.P2POp(dist.isend, obs, self.root, self.group),
                   dist.P2POp(dist.isend, rew, self.root, self.group),
                   dist.P2POp(dist.isend, reset, self.root, self.group)]
        self._pending[d] = dist.batch_isend_irecv(ops)
        if wait:
            self.wait(d)
        return self._sets[d] if self.is_root else None
