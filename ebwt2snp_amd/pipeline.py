"""The resident job -- both tools back to back on an EGSA shard that already lives in HBM -- as one function over the C ABI.
What bench.py times and what the full-size parity tests run; the executables do the same through files.

  arm -> scan (+ pre-filter in the same kernel) -> `.clusters` image laid out on the second stream || [descriptor exchange] -> stitch
  (finishes the image in place) -> statistics -> per-cluster pass -> tail on the GPU (consensus, distance, KisSNP2 text) -> both images in HBM
"""
import ctypes

import numpy as np

from . import capi


def exchange_none(blob):
    return [blob]


def resident_job(h, n1, n_total, rank=0, world=1, exchange=exchange_none, k=16, min_len=2, mcov=5, pval=0.99, k_left=31, k_right=30,
                 max_gap=10, max_err=2, cons_reads=20, fused=True, tail=True, pack=True, ids_exchange=None, tick=None, prefilter=None,
                 allgather=None, native=None):
    """One pass of the hot path.  Two forms:
      * e2s_resident_job (native: the whole step enqueued at once, plans and statistics on the device, ONE wait) -- taken when the
        pre-filter goes into the scan kernel, both outputs are wanted and, for world > 1, `allgather(d_send, d_recv, nbytes, stream)` is
        given (an all-gather ENQUEUED on the handle's stream); where it does not apply (ERR_STATE) or a buffer was too small
        (ERR_CAPACITY, on every rank alike) the step runs through
      * the stepwise calls: exchange(blob) -> list of every rank's blob (ONE all-gather of descriptor + statistics);
        ids_exchange(n_events) -> events on the ranks before this one (a second, 16-byte exchange: `.snp` ids are sequential).
    prefilter: "arm" (inside the scan kernel; = fused), "begin" (a kernel of its own right after the scan; = not fused) or
    "inside" (run by e2s_find_candidates).  Returns a dict of counts; the images stay in HBM (h.clusters_image(), h.fetch_snp())."""
    t = tick or (lambda name: None)
    prefilter = prefilter or ("arm" if fused else "begin")
    if native is None:
        native = prefilter == "arm" and tail and pack and (world == 1 or allgather is not None)
    if native:
        try:
            rep = capi.resident_job(h, n1, rank, world, allgather, k, min_len, mcov, pval, k_left, k_right, max_gap, max_err, cons_reads, want_image=pack)
            t("native job")
            out = dict(n_ends=rep.n_ends, n_cand=rep.n_candidates, mcl=rep.max_clust_length, n_clust=rep.n_clust, n_variants=rep.n_variants,
                       n_events=rep.n_events, snp_bytes=rep.snp_bytes, native=True, n_flagged=rep.n_flagged)
            if pack:
                out["n_records"] = rep.n_records                   # (known without waiting; h.clusters_image() waits for the layout on the second stream)
            return out
        except capi.E2SError as ex:
            if ex.code not in (3, 4):                              # ERR_STATE: does not apply here; ERR_CAPACITY: every rank reruns stepwise
                raise
            t("native job (redo stepwise)")
    if prefilter == "arm":
        h.prefilter_arm(mcov)                                      # the pre-filter runs inside the scan kernel
    h.cluster_scan(k, min_len, 0)                                  # scan + statistics of the shard's own records
    if pack:
        h.pack_clusters(async_=True)                               # the `.clusters` image laid out on the second stream beside everything below
    if prefilter == "begin":
        h.prefilter_begin(mcov, min_len)                           # ... or beside the exchange and the statistics below
    t("scan")
    b = h.boundary()
    if world > 1 or exchange is not exchange_none:
        st = h.scan_stats()
        sb, ss = ctypes.sizeof(capi.Boundary), ctypes.sizeof(capi.ScanStats)
        parts = exchange(bytes(b) + bytes(st))
        blob = np.frombuffer(b"".join(parts), dtype=np.uint8).reshape(world, sb + ss)
        allb = (capi.Boundary * world).from_buffer_copy(np.ascontiguousarray(blob[:, :sb]).tobytes())
        alls = (capi.ScanStats * world).from_buffer_copy(np.ascontiguousarray(blob[:, sb:]).tobytes())
        h.stitch_array(allb, world, rank)
        t("stitch")
        # every rank derives the same statistics from the gathered blobs (pure host function, one call)
        mcl, n_clust, _ = capi.auto_max_clust_length_ranks(allb, alls, n_total, k, 1, mcov, pval)
    else:
        h.stitch([b], 0)
        t("stitch")
        mcl, n_clust, _ = capi.auto_max_clust_length([h.hist()], mcov, pval)
    t("hist")
    prm = capi.C2SParams(n1, k_left, k_right, mcov, cons_reads, mcl, min_len)
    ncand = h.find_candidates(prm, rank << 40)
    t("candidates")
    out = dict(n_ends=b.n_ends, n_cand=ncand, mcl=mcl, n_clust=n_clust)
    if tail:
        # (a shard without candidates has no events; only a JOB without candidates is where the reference crashes)
        if ids_exchange is None:
            nv, ne, nb = h.call_variants(k_left, k_right, max_gap, max_err, first_id=1) if ncand else (0, 0, 0)
        else:
            # several ranks: the ids follow from the event counts of the ranks before -- the text is written once, after that exchange
            nv, ne, nb = h.call_variants(k_left, k_right, max_gap, max_err, first_id=0) if ncand else (0, 0, 0)
            before = ids_exchange(ne)
            if ncand:
                nb = h.renumber_snp(1 + before)
        out.update(n_variants=nv, n_events=ne, snp_bytes=nb)
        t("tail")
    if pack:
        _, nrec = h.clusters_image()                               # waits for the pack on the second stream
        out["n_records"] = nrec
        t("image")
    return out
