"""The N>1 path on CPU: world_size-2 (and 3) `gloo` process groups run the
product's shard driver (partition, gather of the compressed blocks, merge on
rank 0, broadcast).  The compute backend here is the oracle (tests may use it as
the checker); on the GPU box the same driver runs with the HIP backend."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err


class OracleShardBackend:
    """compress = QR of the shard's accepted stack (oracle), merge = QR of the
    stacked triangles + the reference's gain/Joseph formulas."""

    def compress(self, local):
        from oracle import msckf_oracle as oracle
        out = oracle.update(local, dense_noise=False)
        dc = 6 * local.N
        blk = np.zeros((dc, dc + 1))
        if out["status"] == 0:
            A = np.hstack([out["H_X"][:, 15:], out["r_o"][:, None]])
            R = np.linalg.qr(A, mode="r")
            n = min(R.shape[0], dc)
            blk[:n, :] = R[:n, :]
        return blk, int(out["accepted"].sum()), out["accepted"]

    def merge_gain(self, state, blocks, total_accepted):
        d, dc = state.d, 6 * state.N
        if total_accepted == 0:
            return 1, np.zeros(d), state.P.copy()
        R = np.linalg.qr(blocks.reshape(-1, dc + 1), mode="r")[:dc]
        T = np.zeros((dc, d))
        T[:R.shape[0], 15:] = R[:, :dc]
        rn = np.zeros(dc)
        rn[:R.shape[0]] = R[:, dc]
        P, s2 = state.P, state.sigma ** 2
        S = T @ P @ T.T + s2 * np.eye(dc)
        K = P @ T.T @ np.linalg.inv(S)
        dx = K @ rn
        A = np.eye(d) - K @ T
        Pn = A @ P @ A.T + s2 * K @ K.T
        return 0, dx, (Pn + Pn.T) / 2


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import msckf_amd  # noqa: F401
    from msckf_amd.shard import ShardedUpdate
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prob, ref = load_golden(case)
        drv = ShardedUpdate(OracleShardBackend(), rank, world, dist=dist)
        status, dx, P_new, acc = drv.update(prob)
        q.put((rank, status, rel_err(dx, ref["dx"]), rel_err(P_new, ref["P_new"]),
               bool(np.array_equal(acc, ref["accepted"]))))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,case", [(2, "cfg1_A"), (2, "edge_some_rejected"), (3, "cfg1_B"), (2, "edge_all_rejected")])
def test_sharded_update_gloo(world, case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, ref = load_golden(case)
    for rank, status, edx, eP, acc_ok in results:
        assert status == int(ref["status"]), (rank, status)
        assert acc_ok
        assert edx < 1e-9 and eP < 1e-11, (rank, edx, eP)      # every rank holds the broadcast result
