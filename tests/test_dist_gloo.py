"""N > 1 path on CPU: two gloo ranks, each holding one SA range.  Every rank produces its shard's records and
boundary descriptor (numpy emulation of the scan kernel, tests/shard_emul.py), the descriptors travel with
ONE all-gather, each rank runs the product's e2s_stitch_plan, and the concatenated result must equal the
sequential oracle on the whole array.  The statistics step (one more all-gather of histogram partials) is
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
        sz = ctypes.sizeof(capi.Boundary)
        mine = torch.frombuffer(bytearray(bytes(b)), dtype=torch.uint8)
        allt = torch.empty(world * sz, dtype=torch.uint8)
        dist.all_gather_into_tensor(allt, mine)                       # the boundary stitch: one all-gather
        raw = allt.numpy().tobytes()
        allb = [capi.Boundary.from_buffer_copy(raw[i * sz:(i + 1) * sz]) for i in range(world)]
        plan = capi.stitch_plan(allb, rank, n, k, 1)
        st, ln = SE.apply_plan(st, ln, plan)
        closed = st.size
        keep = ln.astype(np.int64) >= m
        st, ln = st[keep], ln[keep]
        hs = capi.Hist()
        for v in ln:
            if v <= 150:
                hs.freq[int(v)] += 1
        hs.n_records = len(ln)
        hs.n_bases = int(ln.astype(np.int64).sum())
        hs.last_len = int(ln[-1]) if len(ln) else 0
        hz = ctypes.sizeof(capi.Hist)
        allh = torch.empty(world * hz, dtype=torch.uint8)
        dist.all_gather_into_tensor(allh, torch.frombuffer(bytearray(bytes(hs)), dtype=torch.uint8))
        rawh = allh.numpy().tobytes()
        hists = [capi.Hist.from_buffer_copy(rawh[i * hz:(i + 1) * hz]) for i in range(world)]
        mcl, n_clust, _ = capi.auto_max_clust_length(hists, 5, 0.99)
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
