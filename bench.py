#!/usr/bin/env python3
"""bench.py -- EGSA entries/sec through ebwt2clust+clust2snp on MI355X (BASELINE.json metric).

A step = one pass of the hot path over a synthetic EGSA that is already resident in HBM:
  cluster-boundary scan (ebwt2clust) -> boundary stitch -> cluster-length statistics ->
  per-cluster allele-count/candidate pass (clust2snp) -> candidates copied to the host.
The host tail (consensus / distance / KisSNP2 text for the ~1-per-12k-entries candidates) is outside the
step; DESIGN.md quotes it separately.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

N > 1: weak scaling.  The global EGSA is N copies of the block; rank r owns a contiguous SA range whose
cuts fall INSIDE clusters, so the stitch is real: ONE all-gather of the 88-byte boundary descriptors
(plus one of the 1.2 KB histogram partials for the statistics step), RCCL over xGMI.
"""
import argparse
import ctypes
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from ebwt2snp_amd import capi, synth
from ebwt2snp_amd import egsa_io as IO

WORKLOADS = {
    # name: (reads per individual, read length)  -- BASELINE.json configs
    "c1": (1000, 100),            # 2x1k x100bp, 2.02e5 entries (the reference's own CPU-runnable case)
    "c2": (1000000, 100),         # 2x1M x100bp, 2.02e8 entries  <- the metric's 1-GPU configuration
    "c2s": (250000, 100),         # quarter-size c2 for quick checks
    # 3.02e9 entries: ten copies of an exactly sorted 2x1M x150bp block laid end to end, read ids remapped per copy
    # ("C2-style block tiled with read-id remap") -- BASELINE.json configs[2], 21 GB of EGSA in HBM
    "c3": (1000000, 150, 10),
    "c3x": (1000000, 150, 15),    # 4.53e9 entries: SA positions beyond 2^32 on one GPU (stress check, 32 GB of EGSA)
}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s


def build_block(R, L, device):
    reads, n1, e = synth.synth_case(R, L, seed=1, device=device, tail_lcp0=True)
    bwt = e["bwt"].contiguous()
    lcp = e["lcp"].to(torch.uint8).contiguous()
    pos = e["suff"].to(torch.uint8).contiguous()
    da = e["text"].to(torch.int32).contiguous()
    return reads, n1, dict(bwt=bwt, lcp=lcp, pos=pos, da=da)


def tile_block(blk, n1, copies):
    """`copies` blocks end to end; copy b holds reads b*n1 .. (b+1)*n1-1 of individual 1 and the same range of
    individual 2 (ids >= copies*n1), so that every read id is distinct and `text < nr_reads1` still tells the
    individuals apart.  The block ends with LCP 0, so no cluster crosses a seam."""
    if copies == 1:
        return blk, n1
    out = {k: blk[k].repeat(copies) for k in ("bwt", "lcp", "pos")}
    da = blk["da"]
    m = da.numel()
    big = torch.empty(m * copies, dtype=torch.int32, device=da.device)
    for b in range(copies):
        big[b * m:(b + 1) * m] = torch.where(da < n1, da + b * n1, da - n1 + copies * n1 + b * n1)
    out["da"] = big
    return out, n1 * copies


def pick_cut(lcp, approx):
    """an SA position near `approx` that lies strictly inside a cluster (both neighbours share >= 16)"""
    w = lcp[approx:approx + 100000].to(torch.int32)
    # c-1, c, c+1 all >= 16 and c-1 not a local minimum: the cluster runs through the cut
    inside = (w[1:-2] >= 16) & (w[2:-1] >= 16) & (w[3:] >= 16) & (w[1:-2] >= w[:-3])
    idx = torch.nonzero(inside)[0].item()
    return approx + idx + 2


def shard_arrays(blk, m, rank, world):
    """rank's slice of the N-block global EGSA: [rank*m + d_r, (rank+1)*m + d_{r+1}) plus halos"""
    cuts = [0] * (world + 1)
    for r in range(1, world):
        cuts[r] = pick_cut(blk["lcp"], m // 3 + 1000 * r)
    lo_in, hi_in = cuts[rank], cuts[rank + 1]
    hl = min(152, lo_in) if rank > 0 else 0
    hr = 1 if rank < world - 1 else 0
    out = {}
    for k, v in blk.items():
        parts = [v[lo_in - hl:]]
        if hi_in + hr > 0:
            parts.append(v[:hi_in + hr])
        out[k] = torch.cat(parts).contiguous()
    g0 = rank * m + lo_in
    n_own = (m - lo_in) + hi_in
    return out, g0, n_own, hl, hr


PHASES = {}


def _tick(name, t0):
    t1 = time.perf_counter()
    PHASES[name] = PHASES.get(name, 0.0) + (t1 - t0)
    return t1


_XCHG = {}
# E2S_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- exercises the N-rank logic (shards, single
# exchange, stitch plans, merged statistics) on a one-GPU box; not a measurement
REHEARSAL = os.environ.get("E2S_BENCH_REHEARSAL", "0") == "1"
# E2S_BENCH_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): take the N-rank code path -- RCCL process group,
# exchange, merged statistics -- with a single rank, to exercise it on a one-GPU box
FORCE_DIST = os.environ.get("E2S_BENCH_FORCE_DIST", "0") == "1"
# E2S_BENCH_OVERLAP=1: the timed steps arm the pre-filter before the scan (e2s_prefilter_arm): it then runs INSIDE the scan kernel
# (k_scan_seg<1, true>; with E2S_FUSED=0 beside it on a second stream) -- one pass over the LCP, 13-18 % more throughput on one GPU.
# E2S_BENCH_OVERLAP=0 keeps the two kernels apart: with several ranks the pre-filter then runs while the descriptors are exchanged.
# Default "auto": inside the scan kernel unless there is an exchange AND the shard is small (see main).  Either way the other mode
# is measured after the timed run and reported as "other_mode", so that the separate kernels are also seen against their rooflines.
OVERLAP = os.environ.get("E2S_BENCH_OVERLAP", "auto")             # "1" / "0" / "auto" (resolved in main once the shard size is known)
FUSED = os.environ.get("E2S_FUSED", "1") != "0"
SECOND_LEG = os.environ.get("E2S_BENCH_SECOND_LEG", "1") == "1"      # also measure the other mode (reported as other_mode)


def _exchange(blob, world, dist, dev):
    """ONE all-gather (RCCL) of a small fixed-size blob per rank; persistent pinned / device buffers."""
    n = len(blob)
    x = _XCHG.get(n)
    if x is None:
        x = dict(src=torch.empty(n, dtype=torch.uint8).pin_memory(), dst=torch.empty(world * n, dtype=torch.uint8).pin_memory(),
                 d_src=torch.empty(n, dtype=torch.uint8, device=dev), d_dst=torch.empty(world * n, dtype=torch.uint8, device=dev))
        _XCHG[n] = x
    x["src"].numpy()[:] = np.frombuffer(blob, dtype=np.uint8)
    if REHEARSAL:                                                  # gloo: host tensors
        dist.all_gather_into_tensor(x["dst"], x["src"])
        raw = x["dst"].numpy().tobytes()
        return [raw[i * n:(i + 1) * n] for i in range(world)]
    xs = x.get("stream")
    if xs is None:
        xs = x["stream"] = torch.cuda.Stream(device=dev, priority=-1)      # beside the pre-filter kernel, not behind it
    with torch.cuda.stream(xs):
        x["d_src"].copy_(x["src"], non_blocking=True)
        dist.all_gather_into_tensor(x["d_dst"], x["d_src"])
        x["dst"].copy_(x["d_dst"], non_blocking=True)
    xs.synchronize()
    raw = x["dst"].numpy().tobytes()
    return [raw[i * n:(i + 1) * n] for i in range(world)]


def run_step(h, prm_base, world, rank, dist, dev, k=16, min_len=2, n_total=None, overlap=None):
    t = time.perf_counter()
    if OVERLAP if overlap is None else overlap:
        h.prefilter_arm(prm_base["m"])                             # the scan starts the pre-filter beside itself (second stream)
    h.cluster_scan(k, min_len, 0)                                  # scan + statistics of the shard's own records
    h.prefilter_begin(prm_base["m"], min_len)                      # (no-op if armed) runs beside the exchange and the statistics below
    t = _tick("scan", t)
    b = h.boundary()
    if dist is not None:
        # the boundary stitch: ONE all-gather carries descriptor + statistics of every rank
        st = h.scan_stats()
        sb, ss = ctypes.sizeof(capi.Boundary), ctypes.sizeof(capi.ScanStats)
        parts = _exchange(bytes(b) + bytes(st), world, dist, dev)
        blob = np.frombuffer(b"".join(parts), dtype=np.uint8).reshape(world, sb + ss)
        allb = (capi.Boundary * world).from_buffer_copy(np.ascontiguousarray(blob[:, :sb]).tobytes())
        alls = (capi.ScanStats * world).from_buffer_copy(np.ascontiguousarray(blob[:, sb:]).tobytes())
        h.stitch_array(allb, world, rank)
    else:
        h.stitch([b], rank)
    t = _tick("stitch", t)
    if dist is not None:
        # every rank derives the same statistics from the gathered blobs (pure host function, one call)
        mcl, n_clust, _ = capi.auto_max_clust_length_ranks(allb, alls, n_total, k, 1, prm_base["m"], 0.99)
    else:
        mcl, n_clust, _ = capi.auto_max_clust_length([h.hist()], prm_base["m"], 0.99)
    t = _tick("hist", t)
    prm = capi.C2SParams(prm_base["n1"], 31, 30, prm_base["m"], 20, mcl, min_len)
    ncand = h.find_candidates(prm, rank << 40)
    t = _tick("candidates", t)
    stream, off8, ncand, _pw = h.candidates_packed()                  # pinned host memory, packed, SA order (what the host tail consumes)
    t = _tick("view", t)
    return dict(n_ends=b.n_ends, n_cand=ncand, mcl=mcl, n_clust=n_clust)


def cpu_baseline(blk, reads, n1, sample_entries):
    """the reference's own executables (oracle/_ref) -- or, without them, the C restatement -- timed on
    one host core over the first `sample_entries` entries of the same EGSA, through files like the
    reference works (BCR triple + FASTA)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n = int(min(sample_entries, blk["lcp"].numel()))
    d = tempfile.mkdtemp(prefix="e2s_cpu_")
    try:
        p = os.path.join(d, "reads.fasta")
        IO.write_fasta_fast(p, reads.cpu().numpy())
        lcp = blk["lcp"][:n].cpu().numpy().copy()
        lcp[-1] = 0
        IO.write_bcr(p, blk["da"][:n].cpu().numpy(), blk["pos"][:n].cpu().numpy(), lcp, blk["bwt"][:n].cpu().numpy())
        if O.have_ref():
            kind = "reference"
            t0 = time.time()
            rc, out = O.ref_ebwt2clust(p)
            t1 = time.time()
            assert rc == 0, out[-300:]
            rc, out = O.ref_clust2snp(p, n1)
            t2 = time.time()
            assert rc == 0, out[-300:]
        else:
            kind = "port"
            O.build()
            t0 = time.time()
            rc, _ = O.ebwt2clust(p)
            t1 = time.time()
            assert rc == 0
            rc, _ = O.clust2snp(p, O.c2s_opts(n1))
            t2 = time.time()
            assert rc == 0
        res = dict(value=n / (t2 - t0), unit="EGSA entries/s", cores=1, kind=kind,
                   sample="first %d entries of the same EGSA (BCR triple + FASTA on local disk): ebwt2clust %.2f s + clust2snp %.2f s"
                          % (n, t1 - t0, t2 - t1))
        # the drop-in executables of this repo on the very same files, end to end (process start, file reads, upload,
        # kernels, host tail, file writes), and whether their outputs are byte-identical to the baseline's
        try:
            e2c, c2s = os.path.join(ROOT, "bin", "ebwt2clust"), os.path.join(ROOT, "bin", "clust2snp")
            if os.path.exists(e2c) and os.path.exists(c2s):
                snp = IO.snp_out_path(p)
                want_cl, want_snp = open(p + ".clusters", "rb").read(), open(snp, "rb").read()
                os.remove(p + ".clusters"); os.remove(snp)
                u0 = time.time()
                r1 = subprocess.run([e2c, "-i", p], capture_output=True)
                u1 = time.time()
                r2 = subprocess.run([c2s, "-i", p, "-n", str(n1)], capture_output=True)
                u2 = time.time()
                same = (r1.returncode == 0 and r2.returncode == 0 and open(p + ".clusters", "rb").read() == want_cl
                        and open(snp, "rb").read() == want_snp)
                res["drop_in"] = {"ebwt2clust_s": round(u1 - u0, 3), "clust2snp_s": round(u2 - u1, 3), "value": n / (u2 - u0),
                                  "unit": "EGSA entries/s end to end incl. file I/O", "outputs_identical": bool(same),
                                  "clusters_bytes": len(want_cl), "snp_bytes": len(want_snp)}
        except Exception as ex:
            res["drop_in"] = {"error": repr(ex)}
        # ... and the library pipeline exactly as the timed steps run it (pre-filter inside the scan kernel, packed candidates in
        # pinned memory) on the same sample, finished by the host tail: `.clusters` and `.snp` bytes against the baseline's
        try:
            want_cl, want_snp = open(p + ".clusters", "rb").read(), open(IO.snp_out_path(p), "rb").read()
            sub = {k2: blk[k2][:n].clone() for k2 in ("bwt", "lcp", "pos", "da")}
            sub["lcp"][-1] = 0
            torch.cuda.synchronize()                                # (the handle's stream is not ordered after torch's)
            h = capi.Handle(blk["lcp"].device.index or 0)
            h.attach_soa(capi.Handle.desc(n), sub["bwt"].data_ptr(), sub["lcp"].data_ptr(), sub["pos"].data_ptr(), sub["da"].data_ptr(),
                         keep=tuple(sub.values()))
            same, secs, nc = True, {}, 0
            for mode in ("arm", "begin", "inside"):                 # pre-filter inside the scan kernel / after the scan / inside find_candidates
                u0 = time.time()
                if mode == "arm":
                    h.prefilter_arm(5)
                h.cluster_scan(16, 2, 0)
                if mode == "begin":
                    h.prefilter_begin(5, 2)
                h.stitch([h.boundary()], 0)
                got_cl = h.fetch_clusters().tobytes()
                mcl, _, _ = capi.auto_max_clust_length([h.hist()], 5, 0.99)
                h.find_candidates(capi.C2SParams(n1, 31, 30, 5, 20, mcl, 2), 0)
                stream, off8, nc, pw = h.candidates_packed()
                out_snp = os.path.join(d, "library_%s.snp" % mode)
                capi.snp_from_packed(stream, off8, nc, p, out_snp, pos_bytes=pw)
                secs[mode] = round(time.time() - u0, 3)
                same = same and got_cl == want_cl and open(out_snp, "rb").read() == want_snp
            res["library_pipeline"] = {"what": "attach -> [arm] -> scan -> [begin] -> stitch -> statistics -> candidates -> host tail on the same sample "
                                               "(resident EGSA), the three ways of running the pre-filter",
                                       "seconds": secs, "candidates": int(nc), "outputs_identical": bool(same)}
            h.close()
        except Exception as ex:
            res["library_pipeline"] = {"error": repr(ex)}
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


PMC_FILES = {"c2": "profiles/round1_fused_pmc_traffic.json", "c3": "profiles/round1_fused_c3_pmc_traffic.json"}
PMC_FILE = PMC_FILES["c2"]


def pmc_traffic(workload, kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled for gfx950 + WRITE_SIZE; collected by
    tools/profile.sh in separate --pmc runs of this same command, summarised by tools/profile_summary.py).  None when the
    committed profile is of another workload -- counters cannot be read from inside the timed run."""
    try:
        j = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_FILES.get(workload, PMC_FILE))))
        if j.get("workload") != workload:
            return None
        names = (kernel,) if isinstance(kernel, str) else kernel
        for k2, v in j["kernels"].items():
            if k2.replace(" ", "") in [x.replace(" ", "") for x in names]:
                return float(v["hbm_bytes_per_launch"])
        return None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=100000000, help="entries of the CPU baseline sample (0 = skip)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and a.gpus > 1:
        # not launched by torch.distributed.run: start it as a child (one rank per GPU) before touching the GPU
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    dist = None
    if world > 1 or (FORCE_DIST and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")    # the small all-gather runs beside the pre-filter kernel
        if REHEARSAL:
            local = 0
            import faulthandler
            faulthandler.dump_traceback_later(600, exit=True)     # a rehearsal that stalls says where
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # the box's hostname may not resolve
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    R, L = WORKLOADS[a.workload][:2]
    copies = WORKLOADS[a.workload][2] if len(WORKLOADS[a.workload]) > 2 else 1
    t0 = time.time()
    reads, n1_block, blk0 = build_block(R, L, dev)
    blk, n1 = tile_block(blk0, n1_block, copies)
    m = blk["lcp"].numel()
    arrs, g0, n_own, hl, hr = shard_arrays(blk, m, rank, world)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    n_total = m * world

    h = capi.Handle(local)
    d = capi.Handle.desc(n_total, offset=g0, n_entries=n_own, halo_left=hl, halo_right=hr)
    h.attach_soa(d, arrs["bwt"].data_ptr(), arrs["lcp"].data_ptr(), arrs["pos"].data_ptr(), arrs["da"].data_ptr(),
                 keep=tuple(arrs.values()))
    prm_base = dict(n1=n1, m=5)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(overlap, warmup):
        """W untimed + K timed steps of the whole job, bracketed by barrier + synchronize; max over ranks"""
        info = None
        for _ in range(warmup):
            info = run_step(h, prm_base, world, rank, dist, dev, n_total=n_total, overlap=overlap)
        r = dict(scan_ms=[], cand_ms=[], scan_all_ms=[], hist_ms=[])
        barrier()
        PHASES.clear()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            info = run_step(h, prm_base, world, rank, dist, dev, n_total=n_total, overlap=overlap)
            tm = h.timing()
            r["scan_ms"].append(tm.scan_main_ms); r["cand_ms"].append(tm.cand_main_ms); r["scan_all_ms"].append(tm.scan_ms); r["hist_ms"].append(tm.hist_ms)
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            rdev = torch.device("cpu") if REHEARSAL else dev
            t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        r.update(elapsed=elapsed, info=info, phases=dict(PHASES))
        return r

    # the timed run `value` comes from, then -- reported beside it, never as `value` -- the same job in the other mode
    # (pre-filter inside the scan kernel / the two kernels apart, each against its own roofline)
    # auto: the pre-filter goes inside the scan kernel unless there is an exchange to hide behind it AND the shard is small
    # enough for that to matter (c2-size shards, N > 1: 0.37 against 0.38 ms per job; c3-size: the fusion's 0.4 ms win)
    global OVERLAP
    OVERLAP = (dist is None or n_own >= 10 ** 9) if OVERLAP == "auto" else (OVERLAP == "1")
    prim = measure(OVERLAP, a.warmup)
    other = measure(not OVERLAP, min(a.warmup, 2) + 1) if (FUSED and SECOND_LEG) else None
    info = prim["info"]
    if other is not None:
        oi = other["info"]
        assert (oi["n_ends"], oi["n_cand"], oi["mcl"]) == (info["n_ends"], info["n_cand"], info["mcl"]), "the two modes differ"
    if dist is not None:
        rdev = torch.device("cpu") if REHEARSAL else dev
        cnt = torch.tensor([info["n_ends"], info["n_cand"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(cnt)
        tot_ends, tot_cand = int(cnt[0].item()), int(cnt[1].item())
    else:
        tot_ends, tot_cand = info["n_ends"], info["n_cand"]

    if rank == 0:
        ref_pattern_bytes = n_own * 7 + info["n_ends"] * 10

        def rooflines(r, overlap):
            """roofline objects of one mode.  Algorithmic bytes per launch:
            pre-filter fused into the scan (k_scan_seg<1,true>): LCP 1 + BWT 1 per provided entry read once, 10 B per record written;
            apart: the scan (k_scan_seg<1,false>) LCP 1 B/entry + 10 B/record, the pre-filter (k_cluster_filter2) LCP 1 + BWT 1 B/entry.
            SURVEY 8d's figure for the reference's access pattern (7 B/entry + 10 B/record) is reported as reference_pattern_bytes."""
            scan_avg = float(np.mean(r["scan_ms"]))
            if overlap and FUSED:
                fb = n_own * 2 + info["n_ends"] * 10
                fg = fb / (scan_avg * 1e-3) / 1e9
                tr = pmc_traffic(a.workload, "k_scan_seg<1,true>")
                return {"roofline": {"bound": "hbm", "kernel": "k_scan_seg<1,true> (cluster scan + per-cluster pre-filter in one pass)", "achieved": fg,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fg / HBM_PEAK_GBS, "traffic": tr, "launch_ms": scan_avg,
                                     "algorithmic_bytes": fb, "reference_pattern_bytes": ref_pattern_bytes + n_own,
                                     "traffic_source": PMC_FILES.get(a.workload) if tr else None, "hist_ms": float(np.mean(r["hist_ms"]))}}
            cand_avg = float(np.mean(r["cand_ms"]))
            cand_bytes = n_own * 2
            cand_gbs = cand_bytes / (cand_avg * 1e-3) / 1e9
            scan_bytes = n_own * 1 + info["n_ends"] * 10
            scan_gbs = scan_bytes / (scan_avg * 1e-3) / 1e9
            k2_traffic, k1_traffic = pmc_traffic(a.workload, "k_cluster_filter2"), pmc_traffic(a.workload, ("k_scan_seg<1>", "k_scan_seg<1,false>"))
            return {"roofline": {"bound": "hbm", "kernel": "k_cluster_filter2", "achieved": cand_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": cand_gbs / HBM_PEAK_GBS, "traffic": k2_traffic, "launch_ms": cand_avg, "algorithmic_bytes": cand_bytes,
                                 "reference_pattern_bytes": ref_pattern_bytes, "traffic_source": PMC_FILES.get(a.workload) if k2_traffic else None},
                    "roofline_scan": {"bound": "hbm", "kernel": "k_scan_seg<1,false>", "achieved": scan_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": scan_gbs / HBM_PEAK_GBS, "traffic": k1_traffic, "launch_ms": scan_avg,
                                      "algorithmic_bytes": scan_bytes, "scan_all_kernels_ms": float(np.mean(r["scan_all_ms"])),
                                      "hist_ms": float(np.mean(r["hist_ms"]))}}

        def mode_name(overlap):
            if not overlap:
                return "separate kernels (E2S_BENCH_OVERLAP=0): scan, then pre-filter"
            return ("pre-filter inside the scan kernel (e2s_prefilter_arm)" if FUSED
                    else "pre-filter beside the scan on a second stream (E2S_FUSED=0; per-kernel times include the sharing)")

        elapsed = prim["elapsed"]
        out = {
            "metric": "EGSA entries/sec through ebwt2clust+clust2snp", "value": n_total * a.steps / elapsed, "unit": "EGSA entries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %s2x%d reads x %dbp synthetic EGSA per GPU, %d entries/GPU, BCR widths 1/4/1"
                                   % (a.workload, ("%d copies of " % copies) if copies > 1 else "", R, L, m), "entries_total": n_total, "k": 16, "min_len": 2,
                       "parallelism": "sa-range x%d" % world, "cluster_records": tot_ends, "candidates": tot_cand,
                       "max_clust_length": info["mcl"], "input_build_s": round(t_build, 2), "mode": mode_name(OVERLAP),
                       "host_phase_ms": {k2: round(v / a.steps * 1e3, 4) for k2, v in prim["phases"].items()}},
        }
        out.update(rooflines(prim, OVERLAP))
        if other is not None:
            om = {"mode": mode_name(not OVERLAP), "value": n_total * a.steps / other["elapsed"], "unit": "EGSA entries/s",
                  "ms_per_step": other["elapsed"] / a.steps * 1e3}
            om.update(rooflines(other, not OVERLAP))
            out["other_mode"] = om
        if world == 1 and a.cpu_sample > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(blk0, reads, n1_block, a.cpu_sample)
            except Exception as ex:           # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "EGSA entries/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out))
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
