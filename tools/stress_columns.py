#!/usr/bin/env python3
"""development aid (needs a GPU): random EGSA columns, the whole pipeline against the oracle -- unsharded, and cut into 2-6
shards at random positions with the three ways of starting the pre-filter.  usage: tools/stress_columns.py FIRST LAST
(seeds).  Round 1: 400 sharded + 300 unsharded seeds, no mismatch."""
import sys, os, pathlib, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import test_gpu_parity as T
import oracle_lib as O
from ebwt2snp_amd import capi, egsa_io as IO

def run(seed):
    en, reads, mcov, Rc = T._fuzz_columns(seed)
    n = en["lcp"].size
    d = pathlib.Path(tempfile.mkdtemp())
    try:
        oc, omcl = T._oracle_candidates(d, reads, en, 1000, mcov=mcov, R=Rc, tag="s")
        ocl = open(str(d / "s.fasta.clusters"), "rb").read()
        rnd = np.random.RandomState(seed + 7)
        nsh = int(rnd.randint(2, 7))
        cuts = sorted(set(rnd.randint(300, n - 300, nsh - 1).tolist()))
        bounds = [0] + cuts + [n]
        psa = np.zeros(n, dtype=np.dtype([("s", "u1"), ("t", "<u4")])); psa["s"] = en["suff"]; psa["t"] = en["text"]
        psa = psa.view(np.uint8)
        hs = []
        for i in range(len(bounds) - 1):
            g0, g1 = bounds[i], bounds[i + 1]
            hl, hr = min(g0, 152), min(n - g1, 1)
            h = capi.Handle(0)
            lo, hi = g0 - hl, g1 + hr
            h.upload_bcr(capi.Handle.desc(n, offset=g0, n_entries=g1 - g0, halo_left=hl, halo_right=hr),
                         en["bwt"][lo:hi].copy(), en["lcp"][lo:hi].copy(), psa[lo * 5:hi * 5].copy())
            if seed % 3 == 1:
                h.prefilter_arm(mcov)
            h.cluster_scan(16, 2, 0)
            if seed % 3 == 2:
                h.prefilter_begin(mcov, 2)
            hs.append(h)
        allb = [h.boundary() for h in hs]
        alls = [h.scan_stats() for h in hs]
        recs = []
        for r, h in enumerate(hs):
            h.stitch(allb, r)
            recs.append(h.fetch_clusters())
        assert np.concatenate(recs).tobytes() == ocl, "clusters differ"
        mcl, _, _ = capi.auto_max_clust_length_ranks(allb, alls, n, 16, 1, mcov, 0.99)
        assert mcl == omcl, "mcl"
        prm = capi.C2SParams(1000, 31, Rc, mcov, 20, mcl, 2)
        cands = []
        for r, h in enumerate(hs):
            h.find_candidates(prm, r << 40)
            buf, nc, cap, stride = h.fetch_candidates(prm)
            cands += capi.parse_candidates(buf, nc, cap, stride)
        assert len(cands) == len(oc), "count %d vs %d (shards %s)" % (len(cands), len(oc), bounds)
        for a, b in zip(cands, oc):
            for key in ("right_idx", "right_pos", "idx0", "pos0", "idx1", "pos1"):
                assert a[key] == b[key], key
        for h in hs:
            h.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)

if __name__ == "__main__":
    bad = 0
    H = capi.Handle(0)
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        try:
            run(seed)
            d = pathlib.Path(tempfile.mkdtemp())
            try:
                T.test_fused_streaming_fuzz_columns(H, d, seed)
            finally:
                shutil.rmtree(d, ignore_errors=True)
        except AssertionError as ex:
            bad += 1
            print("SEED", seed, "FAILED", str(ex)[:300])
    print("stress done, failures:", bad)
    sys.exit(1 if bad else 0)
