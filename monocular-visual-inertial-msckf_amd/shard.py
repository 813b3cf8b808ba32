"""Feature-sharded measurement update: one process (rank) per GPU.

The feature batch of one `MSCKF.update` call is split into contiguous shards
(balanced by stacked rows).  Every rank holds the full filter state and runs
K1-K5 on its shard; the only exchange is ONE gather of the compressed blocks
[R | Q^T r] (6N x (6N+1) doubles per rank, 260 KB at N = 30) to rank 0, which
QR-merges them and runs the serial K6-K7; dx and P+ are then broadcast so every
rank holds the state for the next update.

Two drivers:
 * `RcclShardedUpdate` -- the product path: the gather / broadcast run on librccl BEHIND the C-ABI
   (`msckf_comm_*`, enqueued on the engine's stream between its kernels; no PyTorch, no host round trip
   inside a step); the 128-byte RCCL id travels from rank 0 to the others through a file.
 * `ShardedUpdate` -- the same update over a `torch.distributed` process group ("gloo" in the CPU tests and
   in the two-process GPU test, where both ranks share the one GPU of the box and RCCL refuses duplicate
   devices); torch is plumbing only, the compute is behind the `backend` object.
"""
from __future__ import annotations

import os
import time
from typing import List, Tuple

import numpy as np

from .synth import UpdateProblem


def partition_features(view_ptr: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous shards [lo, hi) with near-equal stacked-row bounds sum(2 M_j)."""
    F = int(len(view_ptr) - 1)
    rows = 2 * np.diff(np.asarray(view_ptr, dtype=np.int64))
    csum = np.concatenate([[0], np.cumsum(rows)])
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.searchsorted(csum, target, side="left")))
    cuts.append(F)
    cuts = [min(max(c, 0), F) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


class HipShardBackend:
    """Compute side of a shard on one MI355X (the product path).

    What a rank sends depends on the WHOLE batch (every rank sees it, so all agree): when it qualifies for
    the band pipeline (`UpdateEngine.band_ok`) the ranks stop in front of their root sweep and send the
    triangles of their first-slot groups (one record); rank 0 folds them group by group and runs ONE root
    sweep.  Otherwise they send their root blocks [R | Q^T r] and rank 0 merges those with the fold tree."""

    def __init__(self, engine, device_buffers: bool = False):
        self.engine = engine
        self.device_buffers = device_buffers
        self.groups = False

    def prepare(self, prob: UpdateProblem):
        self.groups = bool(self.engine.band_ok(prob))
        self.engine.set_group_exchange(self.groups)

    def compress(self, local: UpdateProblem):
        self.engine.load(local)
        self.engine.run_compress()
        blk, n_acc = self.engine.export_groups() if self.groups else self.engine.export_block()
        acc = self.engine.result().accepted
        return blk, n_acc, acc

    def merge_gain(self, state: UpdateProblem, blocks: np.ndarray, total_accepted: int):
        if self.groups:
            self.engine.merge_groups(blocks, -1)                      # the counts ride in the records
        else:
            self.engine.merge_gain(blocks, total_accepted)
        res = self.engine.result()
        return res.status, res.dx, res.P_new


class ShardedUpdate:
    """Drives one sharded update over a torch.distributed process group."""

    def __init__(self, backend, rank: int, world: int, dist=None, device=None):
        self.backend, self.rank, self.world, self.dist, self.device = backend, rank, world, dist, device

    def update(self, prob: UpdateProblem):
        """Every rank passes the same full problem (state + all features) and gets
        back (status, dx, P_new, accepted[F])."""
        import torch
        if hasattr(self.backend, "prepare"):
            self.backend.prepare(prob)                                # exchange format, from the whole batch
        shards = partition_features(prob.view_ptr, self.world)
        lo, hi = shards[self.rank]
        local = prob.subset(lo, hi)
        blk, n_acc, acc = self.backend.compress(local)
        dc, d, F = 6 * prob.N, prob.d, prob.F
        dev = self.device if self.device is not None else "cpu"
        if self.world == 1:
            blocks = np.asarray(blk)[None]
            total = n_acc
            acc_all = acc
        else:
            dist = self.dist
            t_blk = torch.from_numpy(np.ascontiguousarray(blk)).to(dev)
            t_meta = torch.tensor([n_acc], dtype=torch.int64, device=dev)
            t_acc = torch.zeros(F, dtype=torch.uint8, device=dev)
            t_acc[lo:hi] = torch.from_numpy(acc.astype(np.uint8)).to(dev)
            gathered = [torch.empty_like(t_blk) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(t_blk, gather_list=gathered, dst=0)          # the one data-path collective
            dist.all_reduce(t_meta)                                   # total accepted (8 bytes)
            dist.all_reduce(t_acc)                                    # accepted mask (disjoint slices)
            total = int(t_meta.item())
            acc_all = t_acc.cpu().numpy()
            blocks = torch.stack(gathered).cpu().numpy() if self.rank == 0 else None
        out = torch.zeros(1 + d + d * d, dtype=torch.float64, device=dev)
        if self.rank == 0:
            status, dx, P_new = self.backend.merge_gain(prob, blocks, total)
            out[0] = float(status)
            out[1:1 + d] = torch.from_numpy(dx).to(dev)
            out[1 + d:] = torch.from_numpy(P_new.reshape(-1)).to(dev)
        if self.world > 1:
            self.dist.broadcast(out, src=0)                           # dx and P+ for the next update
        o = out.cpu().numpy()
        return int(o[0]), o[1:1 + d].copy(), o[1 + d:].reshape(d, d).copy(), acc_all


def shard_group_flags(prob: UpdateProblem, shards: List[Tuple[int, int]]) -> np.ndarray:
    """(world, N) uint8: shard r has tracks whose first clone slot is s.  Every rank sees the whole batch, so the
    merging rank needs no read-back of the records' heads (`msckf_run_merge_groups_flags`)."""
    vp = np.asarray(prob.view_ptr)
    slots = np.asarray(prob.obs_slot).reshape(-1)
    first = np.minimum.reduceat(slots, vp[:-1]) if prob.F else np.zeros(0, dtype=np.int64)
    flags = np.zeros((len(shards), prob.N), dtype=np.uint8)
    for r, (lo, hi) in enumerate(shards):
        flags[r, np.unique(first[lo:hi])] = 1
    return flags


def exchange_unique_id(engine, rank: int, world: int, path: str, timeout_s: float = 120.0) -> bytes:
    """Bootstrap of the RCCL communicator without any other communication layer: rank 0 writes the id to
    `path` (atomically), the others poll for it."""
    if rank == 0:
        uid = engine.comm_unique_id()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == 128:
                return uid
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise TimeoutError("no RCCL id at " + path)


class RcclShardedUpdate:
    """One sharded update per `step()`: K1-K5 on the local shard, ONE RCCL gather of the group records (or root
    blocks) to rank 0, merge + K6-K7 there, ONE RCCL broadcast of dx | P+.  Everything is enqueued on the
    engine's stream; a step returns without waiting for the device."""

    def __init__(self, engine, rank: int, world: int, uid: bytes):
        self.e, self.rank, self.world = engine, rank, world
        engine.comm_init(rank, world, uid)
        self.groups = False
        self.count = 0
        self.flags = None
        self.recv = 0

    def load(self, prob: UpdateProblem):
        """Every rank passes the same full problem; it keeps its shard resident."""
        e = self.e
        shards = partition_features(prob.view_ptr, self.world)
        self.groups = bool(e.band_ok(prob))
        e.set_group_exchange(self.groups)
        lo, hi = shards[self.rank]
        e.load(prob.subset(lo, hi))
        self.d = prob.d
        self.shard = (lo, hi)
        if self.groups:
            self.count = e.group_record_doubles()
            self.flags = shard_group_flags(prob, shards)
            self.recv = e.comm_buffer(self.count * self.world) if self.rank == 0 else 0
        else:
            # fallback exchange (tracks wider than the sweep tiles, merge tree forced): root blocks [R | Q^T r] plus the
            # shard's accepted count, staged through host memory on both sides
            self.count = e.block_doubles() + 1
            buf = e.comm_buffer(self.count * (self.world + 1))
            self.recv, self.send = buf, buf + 8 * self.count * self.world

    def step(self):
        e = self.e
        e.run_compress()
        if self.groups:
            e.comm_gather(e.device_pointer(3), self.recv, self.count, 0)        # the record lies at the head of the workspace
            if self.rank == 0:
                e.merge_groups_flags(self.recv, self.world, self.flags)          # no read-back: flags from the partition
        else:
            blk, n = e.export_block()
            e.comm_put(self.send, np.concatenate([blk.reshape(-1), [float(n)]]))
            e.comm_gather(self.send, self.recv, self.count, 0)
            if self.rank == 0:
                g = e.comm_get(self.recv, self.count * self.world).reshape(self.world, self.count)
                dc = 6 * e.n_clones
                e.merge_gain(g[:, :-1].reshape(self.world, dc, dc + 1), int(round(g[:, -1].sum())))
        e.comm_broadcast(e.device_pointer(0), self.d + self.d * self.d, 0)       # dx | P+ into every rank's result range

    def result(self):
        """(status, dx, P_new) on every rank (the broadcast filled the result range); syncs."""
        e = self.e
        e.sync()
        if self.rank == 0:
            r = e.result()
            return r.status, r.dx, r.P_new
        dx, P = e.result_host()
        return 0, dx, P

    def close(self):
        self.e.comm_destroy()
