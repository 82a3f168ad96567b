#!/usr/bin/env python3
"""Kernel micro-bench (development aid): times the scan kernel alone (K1) and with the pre-filter inside (fused) on the bench
workload, so that variant / ablation builds of the library (E2S_LIB_PATH) can be compared -- ablation builds' results are wrong.
    tools/kbench.py [workload] [copies]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ebwt2snp_amd import capi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
R, L = bench.WORKLOADS[wl][:2]
copies = int(sys.argv[2]) if len(sys.argv) > 2 else (bench.WORKLOADS[wl][2] if len(bench.WORKLOADS[wl]) > 2 else 1)
dev = torch.device("cuda", 0)
reads, n1, blk0 = bench.build_block(R, L, dev)
blk, n1 = bench.tile_block(blk0, n1, copies)
torch.cuda.synchronize()
n = blk["lcp"].numel()
h = capi.Handle(0)
h.set_timing(2)
h.attach_soa(capi.Handle.desc(n), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
flush = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)       # > Infinity Cache: evicts the columns between scans, as a full step does
for fused in (False, True):
    for mode in ("warm", "cold (1 GiB written in between)"):
        ts = []
        for i in range(15):
            if mode.startswith("cold"):
                flush.add_(1)
                torch.cuda.synchronize()
            if fused:
                h.prefilter_arm(5)
            h.cluster_scan(16, 2, 0)
            ts.append(h.timing().scan_main_ms)
        b = h.boundary()
        nbytes = n * (2 if fused else 1) + b.n_ends * 6
        med = float(np.median(ts[3:]))
        print("%-5s %-32s median %.4f ms  min %.4f  -> %.0f GB/s algorithmic = %.3f of 8 TB/s  (%d entries, %d records)"
              % ("fused" if fused else "K1", mode, med, min(ts[3:]), nbytes / med / 1e6, nbytes / med / 1e6 / 8000.0, n, b.n_ends), flush=True)
