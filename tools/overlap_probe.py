#!/usr/bin/env python3
"""Development aid (round 4): what does the chip give when the bandwidth-bound scan and the latency-bound per-cluster pass + tail
run SIDE BY SIDE?  Two handles on one GPU attach the same resident EGSA (zero copy): handle A repeats the scan with the pre-filter inside
(k_scan_seg<1,true>), handle B repeats the per-cluster pass + GPU tail (K2b -> ordering -> consensus -> text) of an already scanned shard.
Timed: A alone, B alone, A and B from two host threads at once.  If the pair takes about max(A, B), overlapping a piece's per-cluster pass
with the next piece's scan would pay; if it takes about A + B, the two kinds of work compete for the same wave slots / registers.
    tools/overlap_probe.py [workload=c3] [iters=20]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ebwt2snp_amd import capi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
R, L = bench.WORKLOADS[wl][:2]
copies = bench.WORKLOADS[wl][2] if len(bench.WORKLOADS[wl]) > 2 else 1
dev = torch.device("cuda", 0)
reads, n1b, blk0 = bench.build_block(R, L, dev)
blk, n1 = bench.tile_block(blk0, n1b, copies)
rd = bench.tile_reads(reads.to(dev), n1b, copies)
torch.cuda.synchronize()
n = blk["lcp"].numel()


def handle():
    h = capi.Handle(0)
    h.attach_reads(rd.data_ptr(), None, None, rd.shape[0], L, keep=(rd,))
    h.attach_soa(capi.Handle.desc(n), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
    return h


A, B = handle(), handle()
MC = 5


def scan_once(h):
    h.prefilter_arm(MC)
    h.cluster_scan(16, 2, 0)


def prepare_b(h):
    scan_once(h)
    b = h.boundary()
    h.stitch([b], 0)
    mcl, _, _ = capi.auto_max_clust_length([h.hist()], MC, 0.99)
    return mcl


def percluster_once(h, mcl):
    nc = h.find_candidates(capi.C2SParams(n1, 31, 30, MC, 20, mcl, 2), 0)
    h.call_variants(31, 30, 10, 2, first_id=1)
    return nc


mcl = prepare_b(B)
for _ in range(3):
    scan_once(A); percluster_once(B, mcl)
torch.cuda.synchronize()


def loop(fn, k, out, key):
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    out[key] = (time.perf_counter() - t0) / k * 1e3


res = {}
loop(lambda: scan_once(A), iters, res, "A alone (scan + pre-filter, one wait)")
loop(lambda: percluster_once(B, mcl), iters, res, "B alone (K2b + order + consensus + text, two waits)")
ta = threading.Thread(target=loop, args=(lambda: scan_once(A), iters, res, "A beside B"))
tb = threading.Thread(target=loop, args=(lambda: percluster_once(B, mcl), iters, res, "B beside A"))
t0 = time.perf_counter()
ta.start(); tb.start(); ta.join(); tb.join()
res["pair, wall per iteration of both"] = (time.perf_counter() - t0) / iters * 1e3
for k, v in res.items():
    print("%-60s %.4f ms" % (k, v))
a, b, p = res["A alone (scan + pre-filter, one wait)"], res["B alone (K2b + order + consensus + text, two waits)"], res["pair, wall per iteration of both"]
print("A + B = %.4f ms, max = %.4f ms, pair = %.4f ms -> overlap efficiency %.2f (1 = perfect overlap, 0 = serial)" % (a + b, max(a, b), p, (a + b - p) / min(a, b)))
