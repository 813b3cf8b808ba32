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
        self.engine.set_exchange_span(self.engine.max_span(prob) if self.groups else 0)   # every rank: the mode of the whole batch

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


def _run_tag() -> bytes:
    """16 bytes that every rank of ONE launch shares and an earlier launch almost surely does not: the launcher's
    rendezvous (run id, master address / port) and its process id (the ranks' common parent under
    `torch.distributed.run` / `mpirun`).  Callers with their own side channel pass a tag explicitly."""
    import hashlib
    e = os.environ
    ident = [e.get("TORCHELASTIC_RUN_ID", ""), e.get("MASTER_ADDR", ""), e.get("MASTER_PORT", ""), e.get("MSCKF_RUN_TAG", "")]
    # the parent's pid only when nothing else names the launch: ranks started through per-rank wrappers (`mpirun` / `srun`
    # with a script or `bash -c` per rank) or on several nodes have DIFFERENT parents and must still agree on the tag
    if not any(ident):
        ident.append(str(os.getppid()))
    return hashlib.sha256("|".join(ident).encode()).digest()[:16]


def exchange_unique_id(engine, rank: int, world: int, path: str, timeout_s: float = 120.0, tag: bytes = None) -> bytes:
    """Bootstrap of the RCCL communicator without any other communication layer: rank 0 writes `tag | id` to `path`
    (atomically), the others poll for a file that carries THEIR launch's tag -- a file left over from an earlier or
    crashed run has another tag and is ignored (joining with a stale id would hang in ncclCommInitRank).  Rank 0
    removes the file in `RcclShardedUpdate.__init__` once the communicator is up (every rank has read it by then)."""
    tag = _run_tag() if tag is None else bytes(tag)[:16].ljust(16, b"\0")
    if rank == 0:
        uid = engine.comm_unique_id()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(tag + uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == 16 + 128 and blob[:16] == tag:
                return blob[16:]
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise TimeoutError("no RCCL id of this launch at " + path + ": the ranks derive the launch's tag from TORCHELASTIC_RUN_ID / "
                       "MASTER_ADDR + MASTER_PORT / MSCKF_RUN_TAG (else the parent's pid); set MSCKF_RUN_TAG to the same value on "
                       "every rank, or pass tag=, if the launcher provides none of them")


class RcclShardedUpdate:
    """One sharded update per `step()`: K1-K5 on the local shard, ONE RCCL gather of the group records (or root
    blocks) to rank 0, merge + K6-K7 there, ONE RCCL broadcast of the result range
    `status | dx | P+ | gate bytes of the whole batch` (`msckf_set_exchange_mask`: the shards' gate results ride in
    their records).  Everything is enqueued on the engine's stream; a step returns without waiting for the device.
    `result()` is the same call on every rank and yields the same status everywhere (0 updated / 1 no-op; a
    non-SPD innovation covariance raises on every rank, none keeps the garbage)."""

    def __init__(self, engine, rank: int, world: int, uid: bytes, id_path: str = None):
        self.e, self.rank, self.world = engine, rank, world
        engine.comm_init(rank, world, uid)
        if rank == 0 and id_path:
            try:
                os.unlink(id_path)                # every rank has joined: the id file has served
            except OSError:
                pass
        self.groups = False
        self.count = 0
        self.flags = None
        self.recv = 0

    def load(self, prob: UpdateProblem):
        """Every rank passes the same full problem; it keeps its shard resident."""
        e = self.e
        shards = partition_features(prob.view_ptr, self.world)
        self.bounds = np.array([s[0] for s in shards] + [shards[-1][1]], dtype=np.int32)
        self.groups = bool(e.band_ok(prob))
        e.set_group_exchange(self.groups)
        e.set_exchange_span(e.max_span(prob) if self.groups else 0)    # every rank: the sweep mode of the whole batch
        e.set_exchange_mask(self.bounds)
        lo, hi = shards[self.rank]
        e.load(prob.subset(lo, hi))
        self.d = prob.d
        self.shard = (lo, hi)
        self.F_total = int(prob.F)
        self.mask_doubles = (int(np.diff(self.bounds).max()) + 7) // 8
        if self.groups:
            self.count = e.group_record_doubles()
            self.flags = shard_group_flags(prob, shards)
        else:
            # fallback exchange (tracks wider than the sweep tiles, merge tree forced): root blocks [R | Q^T r], the
            # shard's accepted count and its gate bytes, staged through host memory on both sides
            self.count = e.block_doubles() + 1 + self.mask_doubles
        # ONE allocation (a second, larger request would move the buffer under the pointers handed out here):
        # receive side of the gather | staging record of the fallback exchange | 8 doubles of scratch for the caller
        buf = e.comm_buffer(self.count * (self.world + 1) + 8)
        self.recv, self.send = buf, buf + 8 * self.count * self.world
        self.scratch = buf + 8 * self.count * (self.world + 1)

    def step(self):
        e = self.e
        e.run_compress()
        if self.groups:
            e.comm_gather(e.device_pointer(3), self.recv, self.count, 0)        # the record lies at the head of the workspace
            if self.rank == 0:
                e.merge_groups_flags(self.recv, self.world, self.flags)          # no read-back: flags from the partition
        else:
            blk, n = e.export_block()
            gate = np.zeros(8 * self.mask_doubles, dtype=np.uint8)
            lo, hi = self.shard
            if hi > lo:
                gate[:hi - lo] = e.gate_bytes()
            e.comm_put(self.send, np.concatenate([blk.reshape(-1), [float(n)], gate.view(np.float64)]))
            e.comm_gather(self.send, self.recv, self.count, 0)
            if self.rank == 0:
                g = e.comm_get(self.recv, self.count * self.world).reshape(self.world, self.count)
                dc = 6 * e.n_clones
                nb = dc * (dc + 1)
                e.merge_gain(g[:, :nb].reshape(self.world, dc, dc + 1), int(round(g[:, nb].sum())))
                allg = np.zeros((self.F_total + 7) // 8 * 8, dtype=np.uint8)
                for r in range(self.world):
                    a, b = int(self.bounds[r]), int(self.bounds[r + 1])
                    allg[a:b] = np.ascontiguousarray(g[r, nb + 1:]).view(np.uint8)[:b - a]
                e.comm_put(e.device_pointer(6), allg.view(np.float64))
        # status | dx | P+ | gate bytes into every rank's result range
        e.comm_broadcast(e.device_pointer(5), e.result_range_doubles(), 0)

    def result(self):
        """(status, dx, P_new, accepted[F], n_rejected) -- identical on every rank; syncs."""
        r = self.e.shared_result()
        return r.status, r.dx, r.P_new, r.accepted, int(r.stats["n_rejected"])

    def close(self):
        self.e.comm_destroy()
