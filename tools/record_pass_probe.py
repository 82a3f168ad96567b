#!/usr/bin/env python3
"""development aid: the per-cluster pass over records loaded from a `.clusters` image (what the stand-alone clust2snp does),
steady state (buffers allocated): E2S_RECORD_PREFILTER=0/1 tools/record_pass_probe.py [workload]"""
import sys, time, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ebwt2snp_amd import capi
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
R, L = bench.WORKLOADS[wl][:2]
reads, n1, blk = bench.build_block(R, L, dev)
m = blk["lcp"].numel()
torch.cuda.synchronize()   # the arrays are built on torch's stream; the library runs on its own
h = capi.Handle(0)
d = capi.Handle.desc(m)
h.attach_soa(d, blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
h.cluster_scan(16, 2, 0); h.stitch([h.boundary()], 0)
recs = h.fetch_clusters().copy()
mcl, _, _ = capi.auto_max_clust_length([h.hist()], 5, 0.99)
prm = capi.C2SParams(n1, 31, 30, 5, 20, mcl, 2)
want = h.find_candidates(prm, 0)
g = capi.Handle(0)
g.attach_soa(d, blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
g.load_clusters(recs)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nc = g.find_candidates(prm, 0)
    t1 = time.perf_counter()
print("records %d, candidates %d (streaming pass: %d), per-cluster pass over loaded records: %.3f ms" % (recs.size // 10, nc, want, (t1 - t0) * 1e3))
