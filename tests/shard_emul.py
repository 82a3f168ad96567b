"""Test infrastructure: numpy emulation of what one shard's scan kernel reports (cluster records that
END in the shard + its boundary descriptor), straight from the per-entry definition of
cluster_lm's tests (ebwt2clust.cpp:83-84,98-102,116).  Lets the host-side stitch logic be checked on
CPU (single process and gloo world_size 2) without a GPU."""
import numpy as np

from ebwt2snp_amd import capi

NONE = capi.E2S_NONE


def flags(L, k):
    L = np.asarray(L, dtype=np.int64)
    n = L.size
    ge = L >= k
    opn = ge.copy()
    if n > 1:
        opn[1] = ge[0] | ge[1]
    gt = np.zeros(n, dtype=bool)
    gt[1:] = L[:-1] > L[1:]
    end = np.zeros(n, dtype=bool)
    if n > 2:
        j = np.arange(1, n - 1)
        end[j] = opn[j] & ((gt[j] & ~gt[j + 1]) | ~ge[j + 1])
    start = opn.copy()
    if n > 1:
        start[1:] = opn[1:] & (~opn[:-1] | end[:-1])
    return opn, end, start


def shard_scan(L, bwt, g0, g1, k):
    """(starts, lens16, Boundary) of the shard owning [g0, g1): records for the ends it owns; record 0
    is provisional (start unknown) when the shard begins inside a cluster."""
    L = np.asarray(L, dtype=np.int64)
    n = L.size
    opn, end, start = flags(L, k)
    E = np.nonzero(end[g0:g1])[0] + g0
    S = np.nonzero(start[g0:g1])[0] + g0
    begins_inside = bool(opn[g0] and not start[g0])
    last = g1 - 1
    open_after = bool(opn[last] and not end[last])
    b = capi.Boundary()
    b.n_ends = int(E.size)
    b.begins_inside = int(begins_inside)
    b.open_after = int(open_after)
    b.pass_through = int(E.size == 0 and begins_inside and open_after)
    own_open = open_after and not (E.size == 0 and begins_inside)
    b.open_start = int(S[-1]) if own_open else NONE
    b.last_end_start = NONE
    starts = np.zeros(E.size, dtype=np.uint64)
    lens = np.zeros(E.size, dtype=np.uint16)
    off = 1 if begins_inside else 0
    for r, e in enumerate(E):
        if r - off >= 0:
            s = int(S[r - off])
            starts[r] = s
            lens[r] = (int(e) - s + 1) & 0xFFFF
        else:
            lens[r] = (int(e) + 1) & 0xFFFF          # provisional
    if E.size:
        b.first_end_low16 = (int(E[0]) + 1) & 0xFFFF
        if not (E.size == 1 and begins_inside):
            b.last_end_start = int(starts[-1])
    b.end_before_last = int(g1 == n and n >= 2 and bool(end[n - 2]))
    b.tail_lcp = int(L[last])
    b.tail_prev_lcp = int(L[last - 1]) if last >= 1 else 0
    b.tail_bwt = int(bwt[last])
    return starts, lens, b


def apply_plan(starts, lens, plan):
    starts = starts.copy()
    lens = lens.copy()
    if plan.patch_first:
        starts[0] = plan.first_start
        lens[0] = plan.first_len
    ts = [plan.tail_start[i] for i in range(plan.n_tail)]
    tl = [plan.tail_len[i] for i in range(plan.n_tail)]
    return np.concatenate([starts, np.asarray(ts, dtype=np.uint64)]), np.concatenate([lens, np.asarray(tl, dtype=np.uint16)])


def hist_of(lens, m):
    """e2s_hist of a record list (lengths as stored), min_len filter m"""
    ln = np.asarray(lens).astype(np.int64)
    keep = ln[ln >= m]
    hs = capi.Hist()
    for v in keep[keep <= capi.MAX_CLUST_LEN]:
        hs.freq[int(v)] += 1
    hs.n_records = int(keep.size)
    hs.n_bases = int(keep.sum())
    hs.last_len = int(keep[-1]) if keep.size else 0
    return hs


def raw_stats(lens, m):
    """what the scan phase leaves (e2s_scan_stats): statistics of the shard's own records as stored before the
    stitch, record 0 possibly provisional"""
    ln = np.asarray(lens).astype(np.int64)
    st = capi.ScanStats()
    st.hist = hist_of(ln, m)
    q = np.nonzero(ln >= m)[0]
    st.last_none = int(q.size == 0)
    st.last_is_first = int(q.size > 0 and q[-1] == 0)
    st.min_len = m
    return st
