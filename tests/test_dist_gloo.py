"""N > 1 path on CPU: two gloo ranks, each holding one SA range.  Every rank produces its shard's records and
boundary descriptor (numpy emulation of the scan kernel, tests/shard_emul.py), the descriptors travel with
ONE all-gather, each rank runs the product's e2s_stitch_plan, and the concatenated result must equal the
sequential oracle on the whole array.  The statistics travel with the descriptors (one all-gather) and are
checked against the oracle's auto-detected max cluster length."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import shard_emul as SE
from ebwt2snp_amd import capi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, L, bwt, cuts, k, m, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = len(L)
        bounds = [0] + list(cuts) + [n]
        st, ln, b = SE.shard_scan(L, bwt, bounds[rank], bounds[rank + 1], k)
        # ONE all-gather: boundary descriptor + the statistics the scan phase left (record 0 still provisional)
        raw = SE.raw_stats(ln, m)
        sz, zz = ctypes.sizeof(capi.Boundary), ctypes.sizeof(capi.ScanStats)
        mine = torch.frombuffer(bytearray(bytes(b) + bytes(raw)), dtype=torch.uint8)
        allt = torch.empty(world * (sz + zz), dtype=torch.uint8)
        dist.all_gather_into_tensor(allt, mine)
        blob = allt.numpy().tobytes()
        allb = [capi.Boundary.from_buffer_copy(blob[i * (sz + zz):i * (sz + zz) + sz]) for i in range(world)]
        alls = [capi.ScanStats.from_buffer_copy(blob[i * (sz + zz) + sz:(i + 1) * (sz + zz)]) for i in range(world)]
        plan = capi.stitch_plan(allb, rank, n, k, 1)
        st, ln = SE.apply_plan(st, ln, plan)
        closed = st.size
        # every rank evaluates every rank's plan and stitched statistics (pure host functions): no second exchange
        hists = [capi.hist_after_stitch(alls[r], allb[r], capi.stitch_plan(allb, r, n, k, 1)) for r in range(world)]
        mine_direct = SE.hist_of(ln, m)                               # ... equal to the statistics of the final records
        assert list(hists[rank].freq) == list(mine_direct.freq)
        assert (hists[rank].n_records, hists[rank].n_bases, hists[rank].last_len) == \
               (mine_direct.n_records, mine_direct.n_bases, mine_direct.last_len)
        keep = ln.astype(np.int64) >= m
        st, ln = st[keep], ln[keep]
        mcl, n_clust, _ = capi.auto_max_clust_length(hists, 5, 0.99)
        assert capi.auto_max_clust_length_ranks(allb, alls, n, k, 1, 5, 0.99)[:2] == (mcl, n_clust)    # the same in one call
        q.put((rank, st.tolist(), ln.tolist(), closed, mcl, n_clust))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_two_rank_stitch_and_statistics(seed, tmp_path):
    rnd = np.random.RandomState(seed)
    n = 20000
    L = np.where(rnd.rand(n) < 0.15, rnd.randint(0, 16, n), rnd.randint(16, 60, n)).astype(np.uint32)
    if seed == 2:                      # one cluster covering the whole second shard (pass-through) and more
        L[6000:19000] = 30
    if seed == 3:
        L[-1] = 0
    L[0] = 0
    bwt = rnd.choice([36, 65, 67, 71, 84], n).astype(np.uint8)
    ph = O.phantom_lcp_for_scan(L, bwt, 16, 1)
    es, el, ec = O.cluster_lm_mem(L, ph, 16, 2)
    cut = 9000 if seed != 2 else 8000
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, L, bwt, [cut], 16, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    S = sum((r[1] for r in res), [])
    Ln = sum((r[2] for r in res), [])
    assert sum(r[3] for r in res) == ec
    assert S == es.tolist() and Ln == el.tolist()
    # statistics: every rank derives the same value, equal to the sequential rule on the whole record list
    hs = capi.Hist()
    for v in el:
        if v <= 150:
            hs.freq[int(v)] += 1
    hs.n_records = len(el)
    hs.n_bases = int(el.astype(np.int64).sum())
    hs.last_len = int(el[-1])
    want = capi.auto_max_clust_length([hs], 5, 0.99)
    assert all(r[4] == want[0] and r[5] == want[1] for r in res)
