#!/usr/bin/env python3
"""development aid: phase times of bin/ebwt2clust and bin/clust2snp (E2S_CLI_TIMING=1) on files written here:
   tools/cli_timing.py [entries=10^8] [reads=10^6] [len=100] [copies=1]   (copies > 1: the block laid end to end, bench.py's c3)"""
import os, subprocess, sys, tempfile, shutil, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ebwt2snp_amd import egsa_io as IO
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
dev = torch.device("cuda", 0)
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
copies = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reads, n1, blk = bench.build_block(R, L, dev)
if copies > 1:
    reads = torch.cat([reads[:n1].repeat(copies, 1), reads[n1:].repeat(copies, 1)])      # read ids as tile_block numbers them
    blk, n1 = bench.tile_block(blk, n1, copies)
n = min(n, blk["lcp"].numel())
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
