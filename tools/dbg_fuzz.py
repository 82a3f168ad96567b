#!/usr/bin/env python3
"""development aid: one case of tests/test_gpu_parity.py::test_gpu_tail_fuzz_columns outside pytest, the differing lines printed
   tools/dbg_fuzz.py <seed> <L> <letters> <rate>"""
import os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ.setdefault("E2S_DEV", "1")
import numpy as np
import test_gpu_parity as T
from ebwt2snp_amd import egsa_io as IO, tools, capi
seed, L, letters, rate = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], float(sys.argv[4])
en, _, mcov, Rc = T._fuzz_columns(seed)
rnd = np.random.RandomState(seed)
n = en["lcp"].size
reads = np.full((2000, 100), 65, dtype=np.uint8)
mut = rnd.rand(2000, 100) < rate
main, _, rare = letters.partition("|")
reads[mut] = rnd.choice(np.frombuffer(main.encode(), dtype=np.uint8), int(mut.sum()))
if rare:
    for r in range(7, 2000, 23):
        reads[r, rnd.randint(0, 100, 6)] = rnd.choice(np.frombuffer(rare.encode(), dtype=np.uint8), 6)
suff = en["suff"].copy()
far = rnd.rand(n) < 0.2
suff[far] = rnd.randint(96, 101, int(far.sum()))
text = en["text"].copy()
text[rnd.rand(n) < 0.1] = 1999
en = dict(en, suff=suff.astype(np.uint8), text=text)
out = {}
H = capi.Handle(0)
for tag in ("hip", "orc"):
    d = tempfile.mkdtemp()
    p = os.path.join(d, "reads.fasta")
    IO.write_fasta(p, reads)
    IO.write_bcr(p, en["text"], en["suff"], en["lcp"], en["bwt"])
    if tag == "hip":
        tools.ebwt2clust(p, handle=H)
        tools.clust2snp(p, 1000, handle=H, L=L, R=Rc, m=mcov)
    else:
        assert T.O.ebwt2clust(p)[0] == 0
        assert T.O.clust2snp(p, T.O.c2s_opts(1000, L=L, R=Rc, m=mcov))[0] == 0
    out[tag] = open(IO.snp_out_path(p), "rb").read().split(b"\n")
print("lines", len(out["hip"]), len(out["orc"]))
k = 0
for i, (a, b) in enumerate(zip(out["hip"], out["orc"])):
    if a != b:
        print(i, "hip:", a[:200]); print(i, "orc:", b[:200]); k += 1
        if k > 6: break
print("differing lines shown:", k)
