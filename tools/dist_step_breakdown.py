#!/usr/bin/env python3
"""Host-clock breakdown of one sharded step at world size 1 (nccl process group of one rank): where the
time between the kernels' own time and the step time goes."""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
N, F, M = 30, 2000, 10
prob = synth.make_problem(N, F, M, seed=0)
eng = UpdateEngine(max_clones=N, max_features=F, max_track=M)
eng.set_group_exchange(True)
eng.load(prob)
rec = eng.group_record_doubles(); d = prob.d
mine = torch.zeros(rec, dtype=torch.float64, device="cuda")
gathered = torch.zeros(rec, dtype=torch.float64, device="cuda")
glist = [gathered]
out = torch.zeros(d + d * d, dtype=torch.float64, device="cuda")
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); eng.sync()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
def step(timed):
    t = time.perf_counter()
    eng.run_compress();                                  t = tick("run_compress (K1-K4, leaves, merges)", t) if timed else t
    eng.export_groups(dst_ptr=mine.data_ptr(), count=False);          t = tick("export_groups", t) if timed else t
    dist.gather(mine, gather_list=glist, dst=0);         t = tick("gather", t) if timed else t
    torch.cuda.current_stream().synchronize()
    eng.merge_groups(int(gathered.data_ptr()), -1, n_records=1); t = tick("merge_groups (root sweep + K6-K7)", t) if timed else t
    eng.sync()
    eng.export_result(out.data_ptr(), out.data_ptr() + d * 8); t = tick("export_result", t) if timed else t
    dist.broadcast(out, src=0);                          t = tick("broadcast", t) if timed else t
for _ in range(10): step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): step(False)
torch.cuda.synchronize(); eng.sync()
print(f"step (untimed pieces): {(time.perf_counter() - t0) / 100 * 1e6:.0f} us")
for _ in range(100): step(True)
for k, v in acc.items(): print(f"  {k:42s} {v / 100 * 1e6:7.0f} us")
dist.destroy_process_group()
