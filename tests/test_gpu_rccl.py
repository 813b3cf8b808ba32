"""The sharded update's exchange on the HIP path.
 * RCCL behind the C-ABI (msckf_comm_*): a one-rank communicator on the box's single GPU exercises the loader,
   the bootstrap and every collective; the multi-GPU bench uses the same driver (RcclShardedUpdate).
 * ShardedUpdate(HipShardBackend) end to end at world 2: two processes share the GPU and exchange over gloo
   (RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-8


_RCCL_CODE = r"""
import os, sys, tempfile
sys.path.insert(0, %(root)r)
import numpy as np
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from msckf_amd.shard import RcclShardedUpdate, exchange_unique_id
from oracle import msckf_oracle as oracle
assert "torch" not in sys.modules                 # the product exchange runs without PyTorch

# every collective of the C-ABI on a one-rank communicator
with UpdateEngine(max_clones=4, max_features=8, max_track=4) as e:
    uid = e.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    e.comm_init(0, 1, uid)
    buf = e.comm_buffer(3 * 1000)
    x = np.arange(1000, dtype=np.float64) * 0.5 - 7.0
    e.comm_put(buf, x)
    e.comm_gather(buf, buf + 8 * 1000, 1000, 0)              # world 1: the root receives its own block
    e.sync()
    assert np.array_equal(e.comm_get(buf + 8 * 1000, 1000), x)
    e.comm_allreduce(buf, 1000, "sum")
    e.comm_allreduce(buf, 1000, "max")
    e.comm_broadcast(buf, 1000, 0)
    e.sync()
    assert np.array_equal(e.comm_get(buf, 1000), x)
    e.comm_destroy()
    e.comm_destroy()                                          # idempotent
print("RCCL_ABI_OK", flush=True)

# RcclShardedUpdate (the driver bench.py --gpus N runs): compress -> RCCL gather -> merge + gain -> RCCL broadcast, all on
# the engine's stream; group records where the batch runs the 60-column band pipeline, root blocks otherwise
for (N, F, M, groups) in [(30, 2000, 10, True), (16, 120, 14, True), (20, 120, 18, False)]:
    prob = synth.make_problem(N, F, M, seed=81, outlier_fraction=0.05, outlier_px=400.0)
    ref = oracle.update(prob, dense_noise=False)
    with tempfile.TemporaryDirectory() as td, UpdateEngine(max_clones=N, max_features=F, max_track=M) as e:
        idp = os.path.join(td, "id")
        with open(idp, "wb") as fh:                               # a file left over by an earlier launch (another tag): ignored
            fh.write(b"\x01" * 16 + b"\x02" * 128)
        try:
            exchange_unique_id(e, 1, 1, idp, timeout_s=0.2)
            raise AssertionError("a stale id file was accepted")
        except TimeoutError:
            pass
        uid = exchange_unique_id(e, 0, 1, idp)
        assert exchange_unique_id(e, 1, 1, idp) == uid            # what another rank of THIS launch would read
        drv = RcclShardedUpdate(e, 0, 1, uid, id_path=idp)
        assert not os.path.exists(idp)                            # rank 0 removed it once the communicator was up
        drv.load(prob)
        assert drv.groups == groups
        for _ in range(3):                                        # steps chain on the stream, no sync in between
            drv.step()
        status, dx, P, acc, n_rej = drv.result()
        e_dx = np.linalg.norm(dx - ref["dx"]) / np.linalg.norm(ref["dx"])
        e_P = np.linalg.norm(P - ref["P_new"]) / np.linalg.norm(ref["P_new"])
        assert status == 0 and e_dx < 1e-8 and e_P < 1e-8, (status, e_dx, e_P)
        assert np.array_equal(acc, ref["accepted"]) and n_rej == prob.F - int(ref["accepted"].sum())
        dx2, P2 = e.result_host()                                 # dx and P_out read from their own addresses
        assert np.array_equal(dx2, dx) and np.array_equal(P2, P)
        drv.close()
    print("RCCL_STEP_OK", N, F, M, flush=True)
"""


def test_rccl_exchange_through_the_abi():
    """In a fresh interpreter (this pytest process may have imported PyTorch, whose bundled HIP / RCCL runtimes do
    not mix with the system ones inside one process; the product path never imports it)."""
    r = subprocess.run([sys.executable, "-c", _RCCL_CODE % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "RCCL_ABI_OK" in out and out.count("RCCL_STEP_OK") == 3, out[-3000:]


_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch                                   # before the HIP library: both ship a libamdhip64
import torch.distributed as dist
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", init_method="file://" + sys.argv[3], rank=rank, world_size=world)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from msckf_amd.shard import ShardedUpdate, HipShardBackend
from oracle import msckf_oracle as oracle
ok = True
for (N, F, M, kw) in [(30, 600, 10, {}), (12, 80, 12, {"variable_tracks": True}), (20, 100, 18, {})]:
    prob = synth.make_problem(N, F, M, seed=91, **kw)
    with UpdateEngine(max_clones=N, max_features=F, max_track=M, device=0) as e:
        drv = ShardedUpdate(HipShardBackend(e), rank, world, dist)
        status, dx, P, acc = drv.update(prob)
    ref = oracle.update(prob, dense_noise=False)
    e_dx = np.linalg.norm(dx - ref["dx"]) / np.linalg.norm(ref["dx"])
    e_P = np.linalg.norm(P - ref["P_new"]) / np.linalg.norm(ref["P_new"])
    ok = ok and status == 0 and e_dx < 1e-8 and e_P < 1e-8 and np.array_equal(acc, ref["accepted"])
    print("rank", rank, N, F, M, "dx", e_dx, "P", e_P, flush=True)
dist.barrier()
dist.destroy_process_group()
print("SHARD_WORLD2_OK" if ok else "SHARD_WORLD2_FAIL", flush=True)
"""


def test_sharded_driver_world2_on_the_hip_path():
    """ShardedUpdate(HipShardBackend) with a real process group of two ranks: both compress their shard on the
    GPU, rank 0 merges (group records for the first two batches -- 60- and 90-column slots --, root blocks for the
    tracks of 18 slots), everyone gets dx / P+ / the mask."""
    with tempfile.TemporaryDirectory() as td:
        code = _WORKER % {"root": ROOT}
        store = os.path.join(td, "store")
        procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", store], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=400)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
            outs.append(o)
        for r, o in enumerate(outs):
            assert "SHARD_WORLD2_OK" in o, "rank %d:\n%s" % (r, o[-3000:])
