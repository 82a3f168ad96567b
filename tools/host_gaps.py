#!/usr/bin/env python3
"""development aid (needs a GPU): wall time of every C-ABI call of the resident step beside the GPU time its events report -- where the
host side of a step goes.  usage: tools/host_gaps.py [c2|c3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from ebwt2snp_amd import capi

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda", 0)
R, L = bench.WORKLOADS[name][:2]
copies = bench.WORKLOADS[name][2] if len(bench.WORKLOADS[name]) > 2 else 1
reads, n1b, blk0 = bench.build_block(R, L, dev)
blk, n1 = bench.tile_block(blk0, n1b, copies)
m = blk["lcp"].numel()
rd = bench.tile_reads(reads.to(dev), n1b, copies)
h = capi.Handle(0)
h.set_timing(2)
h.attach_reads(rd.data_ptr(), None, None, rd.shape[0], L, keep=(rd,))
h.attach_soa(capi.Handle.desc(m), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr(), keep=tuple(blk.values()))
acc = {}
def T(label, fn):
    t0 = time.perf_counter(); r = fn(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; return r
N = 30
for it in range(N + 3):
    if it == 3:
        acc.clear(); gpu = dict(scan=0.0, cand=0.0, tail=0.0); torch.cuda.synchronize(); t_all = time.perf_counter()
    T("prefilter_arm", lambda: h.prefilter_arm(5))
    T("cluster_scan", lambda: h.cluster_scan(16, 2, 0))
    b = T("boundary", lambda: h.boundary())
    T("stitch", lambda: h.stitch([b], 0))
    mcl, _, _ = T("hist + auto length", lambda: capi.auto_max_clust_length([h.hist()], 5, 0.99))
    T("pack_clusters(async)", lambda: h.pack_clusters(async_=True))
    prm = capi.C2SParams(n1, 31, 30, 5, 20, mcl, 2)
    nc = T("find_candidates", lambda: h.find_candidates(prm, 0))
    T("call_variants", lambda: h.call_variants(31, 30, 10, 2, first_id=1))
    T("clusters_image", lambda: h.clusters_image())
    if it >= 3:
        tm = h.timing(); gpu["scan"] += tm.scan_ms; gpu["cand"] += tm.cand_ms; gpu["tail"] += tm.tail_ms
tot = time.perf_counter() - t_all
print("%s: step %.1f us" % (name, tot / N * 1e6))
for k, v in acc.items():
    print("  %-22s %8.1f us" % (k, v / N * 1e6))
print("  GPU events: scan phase %.1f, candidates %.1f, tail %.1f us" % tuple(gpu[k] / N * 1e3 for k in ("scan", "cand", "tail")))
