#!/usr/bin/env python3
"""development aid: what the Python binding itself costs per call in bench.py's step (host-only calls, GPU idle)"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ebwt2snp_amd import capi
dev = torch.device("cuda", 0)
reads, n1, blk = bench.build_block(200000, 100, dev)
m = blk["lcp"].numel()
torch.cuda.synchronize()
h = capi.Handle(0)
h.attach_soa(capi.Handle.desc(m), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
h.prefilter_arm(5); h.cluster_scan(16, 2, 0); b = h.boundary(); h.stitch([b], 0)
mcl, _, _ = capi.auto_max_clust_length([h.hist()], 5, 0.99)
prm = capi.C2SParams(n1, 31, 30, 5, 20, mcl, 2)
h.find_candidates(prm, 0)
def t(name, f, n=2000):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    print("%-28s %.2f us" % (name, (time.perf_counter() - t0) / n * 1e6))
t("prefilter_arm", lambda: h.prefilter_arm(5))
t("boundary", lambda: h.boundary())
t("stitch", lambda: h.stitch([b], 0))
t("hist", lambda: h.hist())
hs = h.hist()
t("auto_max_clust_length", lambda: capi.auto_max_clust_length([hs], 5, 0.99))
t("C2SParams", lambda: capi.C2SParams(n1, 31, 30, 5, 20, mcl, 2))
t("candidates_packed", lambda: h.candidates_packed())
t("timing", lambda: h.timing())
