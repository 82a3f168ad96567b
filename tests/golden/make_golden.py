#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Needs oracle/_ref/{ebwt2clust,clust2snp} (built by `make -C oracle ref` from the sources
under /root/reference; only possible in the build container).  For every case the script
 1. writes the inputs, runs the reference executables, records their outputs;
 2. runs the CPU restatement (oracle/liboracle.so) on the same inputs and asserts that
    it is byte-identical -- this is what pins the oracle.
The fixtures are data only: inputs (or the seed that regenerates them plus a checksum)
and the outputs the reference produced.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import random
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O                      # noqa: E402
from ebwt2snp_amd import egsa_io as IO      # noqa: E402
from ebwt2snp_amd import synth              # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def last_count(stdout):
    return int(stdout.strip().split("\n")[-1].split()[1])


# --------------------------------------------------------------------------------------
# A. ebwt2clust known-answer table + fuzz, every field width, both formats
# --------------------------------------------------------------------------------------
def e2c_cases():
    A = 65
    T = 84
    tab = [
        dict(lcp=[0, 20, 20, 20, 0]),
        dict(lcp=[0, 20, 25, 20, 25, 20, 3, 20, 20]),
        dict(lcp=[0, 20, 20, 19, 19, 21, 0]),
        dict(lcp=[0, 20, 21, 22, 0]),
        dict(lcp=[20, 3, 20, 20, 0]),
        dict(lcp=[0, 20, 20, 20]),
        dict(lcp=[0, 20, 20, 19]),
        dict(lcp=[0, 20, 20, 19], m=1),
        dict(lcp=[0, 20, 20, 90, 80]),
        dict(lcp=[0, 20, 20, 90, 80], bwt=[T] * 5),
        dict(lcp=[0] + [20] * 70000 + [0, 0]),
        dict(lcp=[0] + [20] * 65535 + [0, 0]),
        dict(lcp=[0] + [20] * 65536 + [0, 0]),
        dict(lcp=[0] + [20] * 65537 + [0, 0], m=1),
        dict(lcp=[20]), dict(lcp=[3], bwt=[T], m=1), dict(lcp=[20], m=1),
        dict(lcp=[0, 12, 12, 12, 0], k=1, m=1), dict(lcp=[0, 12, 12, 12, 0], k=2, m=1),
        dict(lcp=[0, 0, 0, 0, 12, 0], k=4), dict(lcp=[0, 0, 0, 0, 12, 0], k=5),
        dict(lcp=[0, 20, 20, 20, 17], x=2, bwt=[T] * 5, m=1),
        dict(lcp=[0, 20, 20, 20, 15], x=4, bwt=[T] * 5, m=1),
        dict(lcp=[0, 300, 300, 700, 300, 20, 0], x=2),
        dict(lcp=[0, 70000, 70000, 5, 70000, 70001, 70000, 0], x=4, k=1000),
    ]
    rnd = random.Random(20181)
    for _ in range(400):
        n = rnd.choice([1, 3, 4, 5, 6, 8, 12, 30, 100, 300])     # n = 2: reference is non-deterministic
        x = rnd.choice([1, 1, 1, 2, 4, 8])
        y = rnd.choice([4, 4, 1, 2, 8])
        z = rnd.choice([1, 1, 2, 4, 8])
        k = rnd.choice([16, 16, 1, 2, 5, 200])
        m = rnd.choice([2, 2, 1, 3, 8])
        mode = rnd.choice(["lo", "hi", "mix"])

        def val():
            if mode == "lo":
                return rnd.choice([0, 1, max(k - 1, 0), k, k + 1, k + 2])
            if mode == "hi":
                return rnd.randrange(0, 256)
            return rnd.choice([0, k, k + 1, k + 3, rnd.randrange(0, 256)])
        lcp = [min(val(), 255 if x == 1 else 60000) for _ in range(n)]
        tab.append(dict(lcp=lcp, bwt=[rnd.choice([36, A, 67, 71, T, 1, 200]) for _ in range(n)],
                        x=x, y=y, z=z, k=k, m=m, gesa=rnd.random() < 0.5))
    return tab


def run_e2c_case(c, wd):
    n = len(c["lcp"])
    x, y, z = c.get("x", 1), c.get("y", 4), c.get("z", 1)
    bwt = c.get("bwt", [65] * n)
    text = [(7 * i + 3) % 251 for i in range(n)]
    suff = [(5 * i + 1) % 101 for i in range(n)]
    outs = {}
    for tag in ("ref", "orc"):
        d = os.path.join(wd, tag)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        p = os.path.join(d, "t.fa")
        (IO.write_gesa if c.get("gesa") else IO.write_bcr)(p, text, suff, c["lcp"], bwt, x, y, z)
        if tag == "ref":
            rc, out = O.ref_ebwt2clust(p, ["-k", c.get("k", 0), "-m", c.get("m", 0), "-x", x, "-y", y, "-z", z])
            assert rc == 0, (c, rc)
            cnt = last_count(out)
        else:
            rc, cnt = O.ebwt2clust(p, c.get("k", 0), c.get("m", 0), x, y, z)
            assert rc == 0
        outs[tag] = (open(p + ".clusters", "rb").read(), cnt)
    assert outs["ref"] == outs["orc"], ("oracle != reference", {k: v for k, v in c.items() if k != "lcp"}, len(c["lcp"]))
    s, l = IO.read_clusters(os.path.join(wd, "ref", "t.fa.clusters"))
    return dict(n_printed=outs["ref"][1], starts=s.tolist(), lens=l.tolist())


def compress_lcp(lcp):
    """run-length encode long constant runs so the 70000-entry cases stay small in JSON"""
    out = []
    i = 0
    while i < len(lcp):
        j = i
        while j < len(lcp) and lcp[j] == lcp[i]:
            j += 1
        out.append([lcp[i], j - i])
        i = j
    return out


# --------------------------------------------------------------------------------------
# B. end-to-end synthetic cases (ebwt2clust -> clust2snp) through the reference
# --------------------------------------------------------------------------------------
E2E = [
    # name, R, L, seed, gesa, (x,y,z), e2c args, c2s args, tail_lcp0, coverage, store_inputs
    ("tiny_bcr", 300, 100, 11, False, (1, 4, 1), {}, {}, True, 20, True),
    ("c1_bcr", 1000, 100, 1, False, (1, 4, 1), {}, {}, True, 20, False),
    ("c1_gesa", 1000, 100, 1, True, (1, 4, 1), {}, {}, True, 20, False),
    ("m8_bcr", 600, 100, 2, False, (1, 4, 1), {"m": 8}, {}, True, 20, False),
    ("w444_gesa", 600, 100, 3, True, (4, 4, 4), {}, {}, True, 20, False),
    ("w282_bcr", 600, 100, 4, False, (2, 8, 2), {"k": 12}, {"L": 25, "R": 20}, True, 20, False),
    ("opts_bcr", 600, 100, 5, False, (1, 4, 1), {}, {"m": 3, "c": 5, "e": 1, "g": 3, "p": 0.9}, True, 20, False),
    ("cov30_bcr", 600, 100, 6, False, (1, 4, 1), {}, {"m": 4, "L": 40, "R": 35}, True, 30, False),
    ("len150_bcr", 2000, 150, 7, False, (1, 4, 1), {}, {}, True, 20, False),
    ("tail_bcr", 1000, 100, 8, False, (1, 4, 1), {}, {}, False, 20, False),
    ("tail_gesa", 1000, 100, 9, True, (1, 4, 1), {}, {}, False, 20, False),
    ("indel_bcr", 3000, 100, 12, False, (1, 4, 1), {}, {}, True, 20, False),
    # reads of DIFFERENT lengths (trimmed reads: 40-100 % of L, one in sixteen shorter than -L, the last read the shortest; clust2snp.cpp:147-212
    # reads whatever the FASTA holds): a 13th field = the FASTA line width (0: one line per read)
    ("rag_bcr", 1500, 100, 21, False, (1, 4, 1), {}, {}, True, 30, False, 0),
    ("rag150_gesa", 1500, 150, 22, True, (1, 4, 1), {}, {}, False, 30, False, 70),
    ("rag_opts_bcr", 1200, 120, 23, False, (1, 4, 1), {}, {"L": 20, "R": 25, "m": 3}, True, 30, False, 0),
]


def e2e_inputs(R, L, seed, cov, tail, xyz, ragged=False):
    if ragged:
        reads, n1, e, _ = synth.synth_case_ragged(R, L, seed=seed, coverage=cov, tail_lcp0=tail)
        return reads, n1, synth.egsa_to_numpy(e, *xyz)
    reads, n1, e = synth.synth_case(R, L, seed=seed, coverage=cov, tail_lcp0=tail)
    en = synth.egsa_to_numpy(e, *xyz)
    return reads.numpy(), n1, en


def input_digest(reads, en):
    h = hashlib.sha256()
    h.update(b"\n".join(reads) if isinstance(reads, list) else reads.tobytes())
    for k in ("text", "suff", "lcp", "bwt"):
        h.update(en[k].astype(np.uint64).tobytes())
    return h.hexdigest()


def run_e2e(name, R, L, seed, gesa, xyz, ea, ca, tail, cov, store, *rest):
    wd = rest[-1]
    ragged = len(rest) > 1
    width = rest[0] if ragged else 0
    reads, n1, en = e2e_inputs(R, L, seed, cov, tail, xyz, ragged)
    x, y, z = xyz
    res = {}
    for tag in ("ref", "orc"):
        d = os.path.join(wd, tag)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        p = os.path.join(d, "reads.fasta")
        IO.write_fasta(p, reads, width=width or None)
        (IO.write_gesa if gesa else IO.write_bcr)(p, en["text"], en["suff"], en["lcp"], en["bwt"], x, y, z)
        if tag == "ref":
            a = []
            for k, v in ea.items():
                a += ["-" + k, v]
            rc, out = O.ref_ebwt2clust(p, a + ["-x", x, "-y", y, "-z", z])
            assert rc == 0
            nprinted = last_count(out)
            a = []
            for k, v in ca.items():
                a += ["-" + k, v]
            rc, out = O.ref_clust2snp(p, n1, a + ["-x", x, "-y", y, "-z", z])
            assert rc == 0, (name, rc, out[-400:])
            mcl = int(re.search(r"Cluster sizes allowed: \[(\d+),(\d+)\]", out).group(2))
            ncand = int(re.search(r"Done. (\d+) potential", out).group(1))
        else:
            rc, nprinted = O.ebwt2clust(p, ea.get("k", 0), ea.get("m", 0), x, y, z)
            assert rc == 0
            opts = O.c2s_opts(n1, L=ca.get("L", 0), R=ca.get("R", 0), g=ca.get("g", 0), v=ca.get("v", 0),
                              c=ca.get("c", 0), e=ca.get("e", 0), m=ca.get("m", 0), p=ca.get("p", 0.0), x=x, y=y, z=z)
            rc, st = O.clust2snp(p, opts)
            assert rc == 0, (name, rc)
            mcl, ncand = st.max_clust_length, st.n_candidates
        res[tag] = dict(clusters=open(p + ".clusters", "rb").read(), snp=open(os.path.join(d, "reads.snp"), "rb").read(),
                        n_printed=nprinted, mcl=mcl, ncand=ncand)
    assert res["ref"] == res["orc"], ("oracle != reference", name)
    r = res["ref"]
    meta = dict(name=name, R=R, L=L, seed=seed, gesa=gesa, xyz=list(xyz), e2c=ea, c2s=ca, tail_lcp0=tail, coverage=cov,
                nr_reads1=n1, n_entries=int(en["lcp"].size), input_sha256=input_digest(reads, en),
                clusters_sha256=sha(r["clusters"]), n_records=len(r["clusters"]) // 10, n_printed=r["n_printed"],
                max_clust_length=r["mcl"], n_candidates=r["ncand"], snp_file=name + ".snp",
                n_events=r["snp"].count(b">") // 2, n_indel_events=r["snp"].count(b">INDEL_higher"))
    if ragged:
        meta.update(ragged=True, fasta_width=width, min_read_len=min(len(q) for q in reads), max_read_len=max(len(q) for q in reads))
    with open(os.path.join(HERE, name + ".snp"), "wb") as f:
        f.write(r["snp"])
    if store:
        np.savez_compressed(os.path.join(HERE, name + ".inputs.npz"), reads=reads, **en)
        with open(os.path.join(HERE, name + ".clusters"), "wb") as f:
            f.write(r["clusters"])
        meta["inputs_file"] = name + ".inputs.npz"
        meta["clusters_file"] = name + ".clusters"
    return meta


# --------------------------------------------------------------------------------------
# C. crafted clust2snp cases: hand-made EGSA entries + .clusters (tail phantom, filters, H20)
# --------------------------------------------------------------------------------------
def mkreads(nreads=400, L=100, holes=((45, 50), (66, 71)), seed=5):
    rnd = random.Random(seed)
    reads = []
    for _ in range(nreads):
        s = [rnd.choice("ACGT") for _ in range(L)]
        for a, b in holes:
            for p in range(a, b):
                s[p] = "A"
        reads.append("".join(s))
    return reads


def crafted_cases():
    pad = (9, 0, 0, "A")
    A4 = [(0, 50, 9, "C"), (1, 50, 9, "C"), (4, 50, 9, "G"), (5, 50, 9, "G")]
    small = dict(L=5, R=5, m=2, g=2)
    cases = []

    def tailcase(name, nreal, last=(5, 50, 9, "G"), gesa=False, xyz=(1, 4, 1)):
        Z = [(0, 50, 9, "C"), (1, 50, 9, "C")] + [(4, 50, 9, "G")] * (nreal - 3)
        entries = [pad] + A4 + [pad, pad, pad] + Z + [last]
        n = len(entries)
        cases.append(dict(name=name, n1=4, entries=entries, clusters=[(1, 4), (8, n - 8 + 1)], args=small, gesa=gesa, xyz=xyz))
    # EOF phantom inside the last cluster (slot model); nreal = 2^m+1 would abort the reference
    tailcase("tail_phantom_4", 4)
    tailcase("tail_phantom_4_gesa", 4, gesa=True)
    tailcase("tail_phantom_7", 7)
    tailcase("tail_phantom_12", 12)
    tailcase("tail_phantom_24", 24)
    tailcase("tail_phantom_text300", 4, last=(300, 50, 9, "G"))
    tailcase("tail_phantom_T", 4, last=(5, 50, 9, "T"))
    tailcase("tail_phantom_lcp80", 4, last=(5, 50, 80, "G"))
    tailcase("tail_phantom_w444", 4, xyz=(4, 4, 4))
    tailcase("tail_phantom_w282", 6, xyz=(2, 8, 2))
    tailcase("tail_phantom_w218_gesa", 6, xyz=(2, 1, 8), gesa=True)
    # SURVEY §4 crafted-cluster known answer: '$' counts as A -> 3 frequent chars in sample 0
    ent = [pad]
    ent += [(0, 50, 40, "$")] * 5 + [(1, 50, 40, "C")] * 5 + [(2, 50, 40, "T")] * 5 + [(150, 50, 40, "G")] * 5
    ent += [pad, pad]
    ent += [(3, 50, 40, "C")] * 5 + [(151, 50, 40, "G")] * 5
    ent += [pad]
    cases.append(dict(name="survey_crafted", n1=100, entries=ent, clusters=[(1, 20), (23, 10)], args={}, gesa=False, xyz=(1, 4, 1)))
    # two alleles per individual -> up to 4 (c0,c1) pairs, c0-major ascending (H18); lowercase bwt
    ent = [pad]
    ent += [(0, 60, 45, "A")] * 6 + [(1, 61, 44, "c")] * 6 + [(200, 62, 43, "C")] * 6 + [(201, 63, 50, "G")] * 6
    ent += [pad]
    cases.append(dict(name="two_by_two", n1=100, entries=ent, clusters=[(1, 24)], args={}, gesa=False, xyz=(1, 4, 1)))
    # consensus_reads cap (H13), suff < k_left excluded, lcp < k_right excluded
    ent = [pad]
    ent += [(i, 40 + (i % 3) * 20, 25 + (i % 2) * 10, "C") for i in range(30)] + [(250 + i, 55, 31, "T") for i in range(25)]
    ent += [pad]
    cases.append(dict(name="caps", n1=100, entries=ent, clusters=[(1, 55)], args=dict(c=7), gesa=False, xyz=(1, 4, 1)))
    # cluster below 2*mcov / above max length are skipped; min-coverage filter
    ent = [pad] + [(0, 50, 40, "C")] * 4 + [(150, 50, 40, "G")] * 5 + [pad] + [(1, 50, 40, "C")] * 5 + [(151, 50, 40, "G")] * 5 + [pad]
    cases.append(dict(name="mincov", n1=100, entries=ent, clusters=[(1, 9), (11, 10)], args={}, gesa=False, xyz=(1, 4, 1)))
    return cases


def run_crafted(c, reads, wd):
    x, y, z = c["xyz"]
    ent = c["entries"]
    text = [e[0] for e in ent]
    suff = [e[1] for e in ent]
    lcp = [e[2] for e in ent]
    bwt = [ord(e[3]) for e in ent]
    res = {}
    for tag in ("ref", "orc"):
        d = os.path.join(wd, tag)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        p = os.path.join(d, "v.fasta")
        IO.write_fasta(p, reads)
        (IO.write_gesa if c["gesa"] else IO.write_bcr)(p, text, suff, lcp, bwt, x, y, z)
        IO.write_clusters(p + ".clusters", [s for s, _ in c["clusters"]], [l for _, l in c["clusters"]])
        a = c["args"]
        if tag == "ref":
            al = []
            for k, v in a.items():
                al += ["-" + k, v]
            rc, out = O.ref_clust2snp(p, c["n1"], al + ["-x", x, "-y", y, "-z", z])
            assert rc == 0, (c["name"], rc, out[-300:])
            mcl = int(re.search(r"Cluster sizes allowed: \[(\d+),(\d+)\]", out).group(2))
            ncand = int(re.search(r"Done. (\d+) potential", out).group(1))
        else:
            opts = O.c2s_opts(c["n1"], L=a.get("L", 0), R=a.get("R", 0), g=a.get("g", 0), c=a.get("c", 0),
                              e=a.get("e", 0), m=a.get("m", 0), p=a.get("p", 0.0), x=x, y=y, z=z)
            rc, st = O.clust2snp(p, opts)
            assert rc == 0, (c["name"], O.RC_NAMES.get(rc))
            mcl, ncand = st.max_clust_length, st.n_candidates
        res[tag] = dict(snp=open(os.path.join(d, "v.snp"), "rb").read().decode(), mcl=mcl, ncand=ncand)
    assert res["ref"] == res["orc"], ("oracle != reference", c["name"], res)
    out = dict(c)
    out["entries"] = [[e[0], e[1], e[2], ord(e[3])] for e in ent]
    out.update(expected_snp=res["ref"]["snp"], max_clust_length=res["ref"]["mcl"], n_candidates=res["ref"]["ncand"])
    return out


def main():
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle ref` in the build container"
    O.build()
    wd = tempfile.mkdtemp(prefix="e2s_golden_")
    try:
        tab = []
        for c in e2c_cases():
            r = run_e2c_case(c, wd)
            cc = dict(c)
            cc["lcp_rle"] = compress_lcp(cc.pop("lcp"))
            cc["expect"] = r
            tab.append(cc)
        with open(os.path.join(HERE, "e2c_table.json"), "w") as f:
            json.dump(tab, f, separators=(",", ":"))
        print("e2c table:", len(tab), "cases, oracle == reference")

        metas = [run_e2e(*row, wd) for row in E2E]
        with open(os.path.join(HERE, "e2e_cases.json"), "w") as f:
            json.dump(metas, f, indent=1)
        for m in metas:
            print("e2e", m["name"], "records", m["n_records"], "events", m["n_events"], "indels", m["n_indel_events"], "mcl", m["max_clust_length"])

        reads = mkreads()
        cr = [run_crafted(c, reads, wd) for c in crafted_cases()]
        with open(os.path.join(HERE, "crafted_c2s.json"), "w") as f:
            json.dump(dict(reads=reads, cases=cr), f, separators=(",", ":"))
        print("crafted clust2snp:", len(cr), "cases, oracle == reference")

        # the two worked examples of clust2snp.cpp:250-251
        assert O.distance("ACCTACTG", "TTACTTAC", 10) == (1, 2)
        assert O.distance("TTACTTAC", "ACCTACTG", 10) == (1, -2)
    finally:
        shutil.rmtree(wd, ignore_errors=True)


if __name__ == "__main__":
    main()
