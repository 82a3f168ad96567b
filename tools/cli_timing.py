#!/usr/bin/env python3
"""development aid: phase times of bin/clust2snp on a 10^8-entry sample (the files bench.py's cpu_baseline uses)"""
import os, subprocess, sys, tempfile, shutil, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ebwt2snp_amd import egsa_io as IO
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
dev = torch.device("cuda", 0)
reads, n1, blk = bench.build_block(1000000, 100, dev)
d = tempfile.mkdtemp(prefix="e2s_cli_")
try:
    p = os.path.join(d, "reads.fasta")
    IO.write_fasta_fast(p, reads.cpu().numpy())
    lcp = blk["lcp"][:n].cpu().numpy().copy(); lcp[-1] = 0
    IO.write_bcr(p, blk["da"][:n].cpu().numpy(), blk["pos"][:n].cpu().numpy(), lcp, blk["bwt"][:n].cpu().numpy())
    del blk; torch.cuda.empty_cache()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, E2S_CLI_TIMING="1")
    for it in range(2):
        t0 = time.time(); r1 = subprocess.run([os.path.join(root, "bin", "ebwt2clust"), "-i", p], capture_output=True, text=True, env=env); t1 = time.time()
        r2 = subprocess.run([os.path.join(root, "bin", "clust2snp"), "-i", p, "-n", str(n1)], capture_output=True, text=True, env=env); t2 = time.time()
        print("run", it, "ebwt2clust %.3f s, clust2snp %.3f s" % (t1 - t0, t2 - t1))
        print(r1.stderr[-600:]); print(r2.stderr[-900:])
finally:
    shutil.rmtree(d, ignore_errors=True)
