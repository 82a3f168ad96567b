#!/usr/bin/env python3
"""development aid: the pipeline on the same EGSA with 1-byte and 4-byte LCP columns (the latter narrowed on attach)"""
import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from ebwt2snp_amd import capi
dev = torch.device("cuda", 0)
R, L = bench.WORKLOADS["c2"][:2]
reads, n1, blk = bench.build_block(R, L, dev)
m = blk["lcp"].numel()
torch.cuda.synchronize()   # the arrays are built on torch's stream; the library runs on its own
for x in (1, 4):
    lcp = blk["lcp"] if x == 1 else blk["lcp"].to(torch.int32).contiguous()
    h = capi.Handle(0)
    h.set_timing(2)
    d = capi.Handle.desc(m, x=x, y=4, z=1)
    h.attach_soa(d, blk["bwt"].data_ptr(), lcp.data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr(), keep=(lcp,))
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.cluster_scan(16, 2, 0)
        t1 = time.perf_counter()
        h.stitch([h.boundary()], 0)
        mcl, _, _ = capi.auto_max_clust_length([h.hist()], 5, 0.99)
        t2 = time.perf_counter()
        prm = capi.C2SParams(n1, 31, 30, 5, 20, mcl, 2)
        nc = h.find_candidates(prm, 0)
        t3 = time.perf_counter()
    print("x=%d: scan %.3f ms, stitch+hist %.3f ms, candidates %.3f ms (%d)" % (x, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, nc))
    h.close()
