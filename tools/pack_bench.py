#!/usr/bin/env python3
"""Times the `.clusters` layout kernel alone (development aid): scan the bench workload once, then e2s_pack_clusters (synchronous) N times.
    tools/pack_bench.py [workload] [copies]      (E2S_LIB_PATH: another build of the library)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ebwt2snp_amd import capi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
R, L = bench.WORKLOADS[wl][:2]
copies = int(sys.argv[2]) if len(sys.argv) > 2 else (bench.WORKLOADS[wl][2] if len(bench.WORKLOADS[wl]) > 2 else 1)
dev = torch.device("cuda", 0)
reads, n1, blk0 = bench.build_block(R, L, dev)
blk, n1 = bench.tile_block(blk0, n1, copies)
torch.cuda.synchronize()
n = blk["lcp"].numel()
h = capi.Handle(0)
h.attach_soa(capi.Handle.desc(n), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
ts = []
for i in range(20):
    h.cluster_scan(16, 2, 0)                                   # (an image is built once per scan: scan again, time the layout only)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.pack_clusters(async_=False)
    ts.append((time.perf_counter() - t0) * 1e3)
b = h.boundary()
h.stitch([b], 0)
_, nrec = h.clusters_image()
nbytes = b.n_ends * 6 + nrec * 10
med = float(np.median(ts[3:]))
import hashlib
img = h.fetch_clusters()
print("pack: median %.4f ms  min %.4f (wall, one synchronous call) -> %.0f GB/s (6 B read per raw record + 10 B written per kept one; %d raw, %d kept)  image sha1 %s"
      % (med, min(ts[3:]), nbytes / med / 1e6, b.n_ends, nrec, hashlib.sha1(bytes(img)).hexdigest()[:16]), flush=True)
