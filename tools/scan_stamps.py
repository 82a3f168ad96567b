#!/usr/bin/env python3
"""Development aid (round 5): when does every scan worker start and end?  Needs the diagnostic build of the library
(make EXTRA=-DE2S_DIAG_STAMPS LIB=build/libe2s_stamps.so lib; E2S_LIB_PATH=build/libe2s_stamps.so), which stamps s_memrealtime (100 MHz)
into a buffer of its own when a worker starts and when it has written its descriptor, with the XCC / CU / SIMD it ran on.
    tools/scan_stamps.py [workload=c3] [fused=1] [copies]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ebwt2snp_amd import capi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
R, L = bench.WORKLOADS[wl][:2]
copies = int(sys.argv[3]) if len(sys.argv) > 3 else (bench.WORKLOADS[wl][2] if len(bench.WORKLOADS[wl]) > 2 else 1)
dev = torch.device("cuda", 0)
reads, n1, blk0 = bench.build_block(R, L, dev)
blk, n1 = bench.tile_block(blk0, n1, copies)
torch.cuda.synchronize()
n = blk["lcp"].numel()
h = capi.Handle(0)
h.set_timing(2)
h.attach_soa(capi.Handle.desc(n), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
lib = C.CDLL(capi.LIB_PATH)
NW = 32768
buf = np.zeros(NW * 8, dtype=np.uint64)
SLOTS = 4096 if fused else 5120          # waves resident at once (4 / 5 per SIMD x 1024 SIMDs)


def one():
    if fused:
        h.prefilter_arm(5)
    h.cluster_scan(16, 2, 0)
    ms = h.timing().scan_main_ms
    rc = lib.e2s_diag_scan_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
    assert rc == 0, rc
    return ms, buf.reshape(NW, 8).copy()


for _ in range(5):
    one()
for rep in range(3):
    ms, st = one()
    st = st[st[:, 1] != 0]
    w = len(st)
    t0 = st[:, 0].astype(np.int64); t1 = st[:, 1].astype(np.int64)
    base = t0.min()
    s_us = (t0 - base) / 100.0; e_us = (t1 - base) / 100.0
    klen = e_us.max()
    xcc = (st[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
    tiles = (st[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64); recs = (st[:, 3] >> np.uint64(32)).astype(np.int64)
    life = e_us - s_us
    print("rep %d: %s %s  kernel (events) %.4f ms; %d ranges (one wave each), %d..%d tiles each; first start -> last end %.1f us" %
          (rep, wl, "fused" if fused else "K1", ms, w, tiles.min(), tiles.max(), klen))
    print("  wave slots busy: %.3f of %d x the kernel's length   (sum of the ranges' lives %.0f us)" % (life.sum() / (SLOTS * klen), SLOTS, life.sum()))
    print("  a range's life: min %.1f, 10%% %.1f, median %.1f, 90%% %.1f, max %.1f us" %
          (life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max()))
    print("  last range started at %.1f us: the tail (ranges only ending) is %.1f %% of the kernel" % (s_us.max(), 100.0 * (klen - s_us.max()) / klen))
    print("  ends: 50%% %.1f, 90%% %.1f, 99%% %.1f, max %.1f   corr(life, records) = %.3f" %
          (np.median(e_us), np.percentile(e_us, 90), np.percentile(e_us, 99), e_us.max(), np.corrcoef(life, recs)[0, 1]))
    # resident ranges over time (20 bins)
    edges = np.linspace(0, klen, 21)
    occ = [(np.minimum(e_us, edges[i + 1]) - np.maximum(s_us, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(20)]
    print("  ranges resident, by twentieth of the kernel: " + " ".join("%.0f" % x for x in occ))
    print("  per XCC: " + "  ".join("x%d n=%d last end %.0f" % (x, (xcc == x).sum(), e_us[xcc == x].max()) for x in sorted(set(xcc))))
    ph = [(st[:, c].astype(np.int64) - base) / 100.0 for c in (4, 5, 6, 7)]       # after the range's carries | tiles done | queue flushed | histogram flushed
    ok = st[:, 4] != 0
    if ok.any():
        seg = [ph[0] - s_us, ph[1] - ph[0], ph[2] - ph[1], e_us - ph[2], ph[3] - e_us]
        names = ["start -> carries read", "tiles", "flagged list flushed (returning atomic)", "descriptor written", "histogram flushed"]
        print("  phases of a range, median / mean us: " + "; ".join("%s %.1f / %.1f" % (nm, np.median(x[ok]), x[ok].mean()) for nm, x in zip(names, seg)))
        print("  per tile: median %.2f us" % np.median((seg[1] / np.maximum(tiles, 1))[ok]))
    order = np.argsort(s_us)
    print("  median life by tenth of the start order: " + " ".join("%.0f" % np.median(life[i]) for i in np.array_split(order, 10)))
