"""Feature-sharded measurement update: one process (rank) per GPU.

The feature batch of one `MSCKF.update` call is split into contiguous shards
(balanced by stacked rows).  Every rank holds the full filter state and runs
K1-K5 on its shard; the only exchange is ONE gather of the compressed blocks
[R | Q^T r] (6N x (6N+1) doubles per rank, 260 KB at N = 30) to rank 0, which
QR-merges them and runs the serial K6-K7; dx and P+ are then broadcast so every
rank holds the state for the next update.

The exchange goes through `torch.distributed` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests); torch is plumbing only -- the compute is
behind the `backend` object (the HIP engine in production).
"""
from __future__ import annotations

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
