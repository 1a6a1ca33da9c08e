"""Multi-GPU sharding of the rover step: one process per GPU, contiguous env blocks.

The reference is single-device (``rover.py:90`` hard-codes ``cuda:0``) and has no collective to mirror.
Envs are independent, terrain / rock / stone tables are read-only and replicated on every GPU, so the
only exchange of a step is handing (obs f32, reward f32, done u8) of every shard to the learner rank — done
travels as one byte per env (``rover_step_out.done_u8``, written by the is_done stage next to the int64
``reset_buf``), not as the 8-byte flag.

A rank's three outputs live in ONE contiguous byte buffer, laid out by the step kernels themselves (they take plain
pointers: ``obs`` with its row stride, ``rew`` and ``done_u8`` behind it), so a step costs one message per peer: every
non-root rank posts ONE send, the root ONE receive per peer, straight into that rank's slice of the root's buffer —
7 receives on an 8-GPU node, over 7 distinct xGMI links, nothing re-packed or copied afterwards.  (A ring all-gather
would be per-link bound: 7 hops instead of 1; three messages per peer — obs, rew, done apart — were 21 receives.)
The sends and receives of a step go out as one ``batch_isend_irecv`` (a single ncclGroup on RCCL).

The same code runs on ``gloo`` (CPU tensors) for the world_size-2 tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bytes(num_envs_local: int, obs_dim: int) -> int:
    """Bytes of one rank's packed (obs f32 [E, W] | rew f32 [E] | done u8 [E]) shard, padded to 256 B so that every shard (and every
    view inside it) starts aligned."""
    return (num_envs_local * (4 * obs_dim + 4 + 1) + 255) // 256 * 256


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
        self.sb = shard_bytes(num_envs_local, obs_dim)
        n_shards = world if self.is_root else 1
        self._mine = rank if self.is_root else 0             # index of this rank's own shard in its buffer
        self._bufs = []
        for _ in range(depth):
            buf = torch.zeros(n_shards * self.sb, dtype=torch.uint8, device=device)
            self._bufs.append(buf)
            for k in range(n_shards):
                self._views(buf, k)[2].fill_(1)              # done flags start at 1 (reset_buf, rl_task.py:105)
        self._pending = [None] * depth
        self._flat = [None] * depth                          # flat_views(): allocated on first use, root only
        # the local views of set 0 (what a single-set caller hands to the step)
        self.obs, self.rew, self.reset = self.local_views(0)

    def _views(self, buf, k):
        """(obs f32 [E, W], rew f32 [E], done u8 [E]) of shard ``k`` of a packed buffer: views, no copies."""
        e, w, base = self.E, self.W, k * self.sb
        o = buf[base: base + 4 * e * w].view(torch.float32).view(e, w)
        r = buf[base + 4 * e * w: base + 4 * e * w + 4 * e].view(torch.float32)
        z = buf[base + 4 * e * w + 4 * e: base + 4 * e * w + 5 * e]
        return o, r, z

    def local_views(self, d: int = 0):
        """What the step kernels of this rank write for buffer set ``d``: obs [E, W] (contiguous rows), rew [E], done u8 [E]."""
        return self._views(self._bufs[d], self._mine)

    def global_views(self, d: int = 0):
        """The gathered shards of set ``d`` on the root (None elsewhere) as strided views of the packed buffer — obs f32 [N, E, W],
        rew f32 [N, E], done u8 [N, E], shard r = rank r's envs —; valid after ``wait(d)``.  (``.reshape(N * E, ...)`` copies: a
        learner that wants one flat batch pays that copy once, or indexes by [rank, env].)"""
        if not self.is_root:
            return None
        buf, e, w, n, sb = self._bufs[d], self.E, self.W, self.world, self.sb
        f = buf.view(torch.float32)                          # (sb is a multiple of 256: every shard starts on a float)
        obs = torch.as_strided(f, (n, e, w), (sb // 4, w, 1))
        rew = torch.as_strided(f, (n, e), (sb // 4, 1), storage_offset=e * w)
        done = torch.as_strided(buf, (n, e), (sb, 1), storage_offset=4 * e * w + 4 * e)
        return obs, rew, done

    def flat_views(self, d: int = 0):
        """The learner's batch of set ``d`` on the root (None elsewhere): obs f32 [N * E, W], rew f32 [N * E], done u8 [N * E] in global env
        order (rank r's envs are rows [r * E, (r + 1) * E)); valid after ``wait(d)``.  With one rank these ARE the local views (no copy).  With
        N > 1 a rank's three outputs travel as one packed message, so the shards' obs blocks are not adjacent in the receive buffer: this
        accessor copies them into persistent flat tensors (three strided device copies: at 262 144 envs x 41 floats 43 MB, tens of
        microseconds) — a learner that can index [rank, env] takes ``global_views`` and pays nothing."""
        if not self.is_root:
            return None
        if self.world == 1:
            return self.local_views(d)
        og, rg, dg = self.global_views(d)
        if self._flat[d] is None:
            n = self.world * self.E
            self._flat[d] = (torch.empty(n, self.W, dtype=torch.float32, device=og.device), torch.empty(n, dtype=torch.float32, device=og.device),
                             torch.empty(n, dtype=torch.uint8, device=og.device))
        fo, fr, fd = self._flat[d]
        fo.view(self.world, self.E, self.W).copy_(og)
        fr.view(self.world, self.E).copy_(rg)
        fd.view(self.world, self.E).copy_(dg)
        return fo, fr, fd

    def wait(self, d: int = 0):
        """Order the current stream behind the outstanding transfer of set ``d`` (no-op if there is none)."""
        works = self._pending[d]
        if works:
            for w in works:
                w.wait()
        self._pending[d] = None

    def gather(self, d: int = 0, wait: bool = True):
        """After the step kernels of every rank were enqueued: ONE message per peer — the rank's packed shard — to the root."""
        if self.world == 1:
            return self.global_views(d)
        self.wait(d)
        buf, sb = self._bufs[d], self.sb
        if self.is_root:
            ops = [dist.P2POp(dist.irecv, buf[r * sb:(r + 1) * sb], r, self.group) for r in range(self.world) if r != self.root]
        else:
            ops = [dist.P2POp(dist.isend, buf, self.root, self.group)]
        self._pending[d] = dist.batch_isend_irecv(ops)
        if wait:
            self.wait(d)
        return self.global_views(d)
