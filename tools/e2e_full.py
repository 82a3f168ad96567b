#!/usr/bin/env python3
"""Round 4: ONE full-size end-to-end number with its breakdown -- the executables on the WHOLE c3 files (what pipeline.sh:115,122
runs: 3.02x10^9 entries = 21 GB of BCR files + 3 GB of FASTA), not on one block:
    bin/ebwt2clust + bin/clust2snp (two processes, the drop-in pair), bin/ebwt2snp_mgpu -G 1 in its two forms (one process)
with E2S_CLI_TIMING=1 / E2S_CLI_JSON, every pair of outputs compared (SHA-256) with the resident job's images of the same data.
    tools/e2e_full.py [copies=10] [reads=1000000] [len=150] [dir=/tmp]     (needs ~25 GB of local disk / page cache)"""
import hashlib, json, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ebwt2snp_amd import capi, pipeline, egsa_io as IO

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
base = sys.argv[4] if len(sys.argv) > 4 else "/tmp"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda", 0)
t0 = time.time()
reads, n1b, blk0 = bench.build_block(R, L, dev)
m = blk0["lcp"].numel()
n, n1 = m * copies, n1b * copies
d = tempfile.mkdtemp(prefix="e2s_e2e_", dir=base)
p = os.path.join(d, "reads.fasta")
print("block built in %.1f s: %d entries x %d copies = %d entries" % (time.time() - t0, m, copies, n), flush=True)


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 26)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


try:
    # ---- the files: the block `copies` times end to end, read ids remapped per copy as bench.tile_block does
    t0 = time.time()
    bw, lc, ps = (blk0[k].cpu().numpy() for k in ("bwt", "lcp", "pos"))
    da0 = blk0["da"].cpu().numpy().astype(np.int64)
    rec = np.dtype([("s", "<u1"), ("t", "<u4")])
    with open(p + ".out", "wb") as fb, open(p + ".out.lcp", "wb") as fl, open(p + ".out.pairSA", "wb") as fp:
        for b in range(copies):
            bw.tofile(fb); lc.tofile(fl)
            a = np.empty(m, dtype=rec)
            a["s"] = ps
            a["t"] = np.where(da0 < n1b, da0 + b * n1b, da0 - n1b + copies * n1b + b * n1b).astype(np.uint32)
            a.tofile(fp)
            del a
    rd = reads.cpu().numpy()
    with open(p, "wb") as ff:                                     # individual 1 of every copy, then individual 2 of every copy (tile_reads)
        hdr_len = 12
        for part in ([rd[:n1b]] * copies) + ([rd[n1b:]] * copies):
            k = part.shape[0]
            recf = np.empty((k, hdr_len + L + 1), dtype=np.uint8)
            recf[:, :hdr_len] = np.frombuffer(b">r000000000\n", dtype=np.uint8)
            recf[:, hdr_len:hdr_len + L] = part
            recf[:, hdr_len + L] = 10
            recf.tofile(ff)
    files_gb = sum(os.path.getsize(p + s) for s in ("", ".out", ".out.lcp", ".out.pairSA")) / 1e9
    print("files written in %.1f s: %.2f GB" % (time.time() - t0, files_gb), flush=True)
    # ---- the resident job's images of the same data: the bytes every form below must reproduce
    blk, _ = bench.tile_block(blk0, n1b, copies)
    rdt = bench.tile_reads(reads.to(dev), n1b, copies)
    torch.cuda.synchronize()
    h = capi.Handle(0)
    h.attach_reads(rdt.data_ptr(), None, None, rdt.shape[0], L, keep=(rdt,))
    h.attach_soa(capi.Handle.desc(n), blk["bwt"].data_ptr(), blk["lcp"].data_ptr(), blk["pos"].data_ptr(), blk["da"].data_ptr())
    r = pipeline.resident_job(h, n1, n)
    want = (hashlib.sha256(h.fetch_clusters().tobytes()).hexdigest(), hashlib.sha256(h.fetch_snp().tobytes()).hexdigest())
    print("resident job: %d records in the file, %d events, %d bytes of .snp" % (r["n_records"], r["n_events"], r["snp_bytes"]), flush=True)
    h.close()
    del blk, rdt, blk0, reads
    torch.cuda.empty_cache()
    snp = IO.snp_out_path(p)
    rows = []

    def run(tag, cmds, extra_env=None):
        for f in (p + ".clusters", snp):
            if os.path.exists(f):
                os.remove(f)
        total = 0.0
        detail = []
        for i, cmd in enumerate(cmds):
            jp = os.path.join(d, "perf_%s_%d.json" % (tag, i))
            env = dict(os.environ, E2S_CLI_TIMING="1", E2S_UPLOAD_PROFILE="1", E2S_CLI_JSON=jp)
            env.update(extra_env or {})
            t1 = time.time()
            rr = subprocess.run(cmd, capture_output=True, text=True, env=env)
            t2 = time.time()
            dt = t2 - t1
            total += dt
            assert rr.returncode == 0, (cmd, rr.stdout[-600:], rr.stderr[-1200:])
            jr = json.load(open(jp)) if os.path.exists(jp) else {}
            tl = [l for l in rr.stderr.splitlines() if l.startswith("[timing]") or l.startswith("[upload]")]
            if "epoch_main" in jr:                               # where the time outside the job's own clock goes
                tl.append("[outside] process start -> main() %.1f ms; main() -> report %.1f ms; report -> process reaped (exit: mappings, GPU memory) %.1f ms"
                          % ((jr["epoch_main"] - t1) * 1e3, (jr["epoch_report"] - jr["epoch_main"]) * 1e3, (t2 - jr["epoch_report"]) * 1e3))
            detail.append((os.path.basename(cmd[0]), dt, tl, jr))
        ok = (sha(p + ".clusters"), sha(snp)) == want
        rows.append((tag, total, ok))
        print("== %s: %.3f s wall in all = %.3g entries/s; outputs %s the resident job's images (%.2f GB .clusters, %.1f MB .snp)"
              % (tag, total, n / total, "byte-identical to" if ok else "DIFFER from", os.path.getsize(p + ".clusters") / 1e9, os.path.getsize(snp) / 1e6), flush=True)
        for name, dt, tl, jr in detail:
            print("   %s: %.3f s wall" % (name, dt))
            for l in tl:
                print("      " + l)
            if jr:
                print("      json: " + json.dumps({k: v for k, v in jr.items() if k not in ("input", "library")}))
        assert ok, tag

    e2c, c2s, mg = (os.path.join(root, "bin", x) for x in ("ebwt2clust", "clust2snp", "ebwt2snp_mgpu"))
    for it in range(0 if os.environ.get("E2E_ONLY") == "mgpu" else 2):     # (the second pass finds the files in the page cache)
        run("two tools, run %d" % it, [[e2c, "-i", p], [c2s, "-i", p, "-n", str(n1)]])
    for it in range(2):                                            # (the first run of the process that links RCCL pages the library in)
        run("ebwt2snp_mgpu -G 1 (all four columns resident, e2s_resident_job), run %d" % it, [[mg, "-i", p, "-n", str(n1), "-G", "1"]], {"E2S_MGPU_FULL_COLUMNS": "1"})
        run("ebwt2snp_mgpu -G 1 (LCP + BWT uploaded, flagged clusters gathered), run %d" % it, [[mg, "-i", p, "-n", str(n1), "-G", "1"]])
    # E2E_VARIANTS="tag:K=V,K=V;tag2:K=V": the one-process driver again under other settings (development)
    for var in [v for v in os.environ.get("E2E_VARIANTS", "").split(";") if v]:
        vt, _, kv = var.partition(":")
        envv = dict(x.split("=", 1) for x in kv.split(",") if x)
        for it in range(2):
            run("ebwt2snp_mgpu -G 1 [%s], run %d" % (vt, it), [[mg, "-i", p, "-n", str(n1), "-G", "1"]], envv)
    print("summary: %d entries, files %.2f GB" % (n, files_gb))
    for tag, total, ok in rows:
        print("   %-78s %.3f s  %.3g entries/s" % (tag, total, n / total))
finally:
    shutil.rmtree(d, ignore_errors=True)
