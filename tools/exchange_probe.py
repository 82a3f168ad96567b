#!/usr/bin/env python3
"""development aid (needs a GPU; run under torch.distributed.run --nproc-per-node 1): what one small RCCL all-gather through
torch.distributed costs from Python, piece by piece -- bench.py's _exchange"""
import os, time
import numpy as np
import torch
import torch.distributed as dist
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
world = dist.get_world_size()
n = 1304
src = torch.empty(n, dtype=torch.uint8).pin_memory(); dst = torch.empty(world * n, dtype=torch.uint8).pin_memory()
d_src = torch.empty(n, dtype=torch.uint8, device=dev); d_dst = torch.empty(world * n, dtype=torch.uint8, device=dev)
xs = torch.cuda.Stream(device=dev, priority=-1)
blob = bytes(n)
acc = {}
def T(k, t0): acc[k] = acc.get(k, 0.0) + time.perf_counter() - t0
for it in range(220):
    if it == 20: acc.clear()
    t = time.perf_counter(); src.numpy()[:] = np.frombuffer(blob, dtype=np.uint8); T("fill pinned", t)
    t0 = time.perf_counter()
    with torch.cuda.stream(xs):
        t = time.perf_counter(); d_src.copy_(src, non_blocking=True); T("h2d launch", t)
        t = time.perf_counter(); dist.all_gather_into_tensor(d_dst, d_src); T("all_gather call", t)
        t = time.perf_counter(); dst.copy_(d_dst, non_blocking=True); T("d2h launch", t)
    t = time.perf_counter(); xs.synchronize(); T("synchronize", t)
    t = time.perf_counter(); raw = dst.numpy().tobytes(); T("tobytes", t)
    T("total", t0)
if dist.get_rank() == 0:
    for k, v in acc.items(): print("%-18s %7.1f us" % (k, v / 200 * 1e6))
dist.barrier(); dist.destroy_process_group()
