#!/usr/bin/env python3
"""development aid: is K2b bound by the bytes it writes to the host?  the per-cluster pass with list caps 20 / 10 / 4 / 1"""
import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from ebwt2snp_amd import capi
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
R, L = bench.WORKLOADS[wl][:2]
copies = bench.WORKLOADS[wl][2] if len(bench.WORKLOADS[wl]) > 2 else 1
reads, n1, blk = bench.build_block(R, L, dev)
blk, n1 = bench.tile_block(blk, n1, copies)
m = blk["lcp"].numel()
torch.cuda.synchronize()   # the arrays are built on torch's stream; the library runs on its own
h = capi.Handle(0)
h.set_timing(2)
h.attach_soa(capi.Handle.desc(m), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
for c in (20, 10, 4, 1):
    ts = []
    for it in range(6):
        h.cluster_scan(16, 2, 0); h.prefilter_begin(5, 2); h.stitch([h.boundary()], 0)
        mcl, _, _ = capi.auto_max_clust_length([h.hist()], 5, 0.99)
        nc = h.find_candidates(capi.C2SParams(n1, 31, 30, 5, c, mcl, 2), 0)
        tm = h.timing(); ts.append((tm.cand_ms, tm.cand_main_ms))
    st, off8, n, pw = h.candidates_packed()
    print("c=%d: candidates %d, stream %.2f MB, pre-filter %.1f us, pre-filter + K2b %.1f us" % (c, nc, int(off8[n]) * 8 / 1e6, np.median([t[1] for t in ts]) * 1e3, np.median([t[0] for t in ts]) * 1e3))
