#!/usr/bin/env python3
"""Kernel micro-bench (development aid): times e2s_cluster_scan alone on the bench workload, so that
diagnostic (ablation) builds whose results are wrong can still be timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ebwt2snp_amd import capi
import bench

R, L = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"][:2]
dev = torch.device("cuda", 0)
reads, n1, blk = bench.build_block(R, L, dev)
torch.cuda.synchronize()
n = blk["lcp"].numel()
h = capi.Handle(0)
torch.cuda.synchronize()
h.attach_soa(capi.Handle.desc(n), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
flush = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)       # > Infinity Cache: evicts the LCP between scans, as a full step does
for mode in ("warm (LCP may sit in the 256 MiB Infinity Cache)", "cold (1 GiB written in between)"):
    ts = []
    for i in range(25):
        if mode.startswith("cold"):
            flush.add_(1)
            torch.cuda.synchronize()
        h.cluster_scan(16, 2, 0)
        ts.append(h.timing().scan_main_ms)
    print("scan_main_ms median %.4f min %.4f  (%.0f GB/s LCP)  %s" % (np.median(ts[3:]), min(ts[3:]), n / np.median(ts[3:]) / 1e6, mode))
