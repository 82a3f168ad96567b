"""The CPU restatement (oracle/) against the outputs the reference itself produced (tests/golden/).
CPU only.  This is what pins the oracle on the GPU box, where the reference cannot travel."""
import hashlib
import os

import numpy as np
import pytest

import goldens
import oracle_lib as O
from ebwt2snp_amd import egsa_io as IO


def test_e2c_table(tmp_path):
    tab = goldens.e2c_table()
    assert len(tab) >= 400
    for c in tab:
        n = len(c["lcp"])
        x, y, z = c.get("x", 1), c.get("y", 4), c.get("z", 1)
        p = str(tmp_path / "t.fa")
        for ext in (".gesa", ".out", ".out.lcp", ".out.pairSA"):
            if os.path.exists(p + ext):
                os.remove(p + ext)
        text = [(7 * i + 3) % 251 for i in range(n)]
        suff = [(5 * i + 1) % 101 for i in range(n)]
        (IO.write_gesa if c.get("gesa") else IO.write_bcr)(p, text, suff, c["lcp"], c.get("bwt", [65] * n), x, y, z)
        rc, cnt = O.ebwt2clust(p, c.get("k", 0), c.get("m", 0), x, y, z)
        assert rc == 0
        s, l = IO.read_clusters(p + ".clusters")
        assert cnt == c["expect"]["n_printed"]
        assert s.tolist() == c["expect"]["starts"] and l.tolist() == c["expect"]["lens"]


def test_cluster_lm_mem_matches_file_scan():
    """the in-memory scan + phantom model used by the GPU parity tests agrees with the golden table"""
    for c in goldens.e2c_table():
        x = c.get("x", 1)
        k = c.get("k", 0) or 16
        m = c.get("m", 0) or 2
        lcp = np.asarray(c["lcp"], dtype=np.uint32)
        bwt = np.asarray(c.get("bwt", [65] * len(lcp)), dtype=np.uint8)
        ph = O.phantom_lcp_for_scan(lcp, bwt, k, x)
        s, l, closed = O.cluster_lm_mem(lcp, ph, k, m)
        assert closed == c["expect"]["n_printed"]
        assert s.tolist() == c["expect"]["starts"] and l.tolist() == c["expect"]["lens"]


@pytest.mark.parametrize("meta", goldens.e2e_cases(), ids=lambda m: m["name"])
def test_e2e(meta, tmp_path):
    reads, n1, en = goldens.e2e_inputs(meta)
    p = goldens.write_case_files(meta, reads, en, str(tmp_path))
    x, y, z = meta["xyz"]
    rc, cnt = O.ebwt2clust(p, meta["e2c"].get("k", 0), meta["e2c"].get("m", 0), x, y, z)
    assert rc == 0 and cnt == meta["n_printed"]
    cl = open(p + ".clusters", "rb").read()
    assert hashlib.sha256(cl).hexdigest() == meta["clusters_sha256"]
    if "clusters_file" in meta:
        assert cl == open(os.path.join(goldens.GOLD, meta["clusters_file"]), "rb").read()
    a = meta["c2s"]
    opts = O.c2s_opts(n1, L=a.get("L", 0), R=a.get("R", 0), g=a.get("g", 0), c=a.get("c", 0), e=a.get("e", 0),
                      m=a.get("m", 0), p=a.get("p", 0.0), x=x, y=y, z=z)
    rc, st = O.clust2snp(p, opts)
    assert rc == 0
    assert st.max_clust_length == meta["max_clust_length"] and st.n_candidates == meta["n_candidates"]
    assert st.n_clust == meta["n_records"] + 1          # H4: last record counted twice
    assert open(IO.snp_out_path(p), "rb").read() == goldens.golden_snp(meta)


def test_crafted_c2s(tmp_path):
    g = goldens.crafted()
    for c in g["cases"]:
        d = tmp_path / c["name"]
        d.mkdir()
        p = str(d / "v.fasta")
        IO.write_fasta(p, g["reads"])
        x, y, z = c["xyz"]
        ent = np.asarray(c["entries"])
        (IO.write_gesa if c["gesa"] else IO.write_bcr)(p, ent[:, 0], ent[:, 1], ent[:, 2], ent[:, 3], x, y, z)
        IO.write_clusters(p + ".clusters", [s for s, _ in c["clusters"]], [l for _, l in c["clusters"]])
        a = c["args"]
        opts = O.c2s_opts(c["n1"], L=a.get("L", 0), R=a.get("R", 0), g=a.get("g", 0), c=a.get("c", 0),
                          e=a.get("e", 0), m=a.get("m", 0), p=a.get("p", 0.0), x=x, y=y, z=z)
        rc, st = O.clust2snp(p, opts)
        assert rc == 0, c["name"]
        assert open(str(d / "v.snp")).read() == c["expected_snp"], c["name"]
        assert st.max_clust_length == c["max_clust_length"] and st.n_candidates == c["n_candidates"]


def test_distance_worked_examples():
    # the only known answers in the reference tree: clust2snp.cpp:250-251
    assert O.distance("ACCTACTG", "TTACTTAC", 10) == (1, 2)
    assert O.distance("TTACTTAC", "ACCTACTG", 10) == (1, -2)
    assert O.distance("ACGTACGT", "ACGTACGT", 10) == (0, 0)
    # H9: a tie between no-indel and indel goes to the indel
    d, l = O.distance("AAAAAAAC", "AAAAAAAA", 10)
    assert (d, l) != (1, 0)


def test_cons_tie_break():
    # include.hpp:349-358: C[i] only changes on a strictly greater count (H8)
    assert O.cons(["C", "G"], 1) == "C"
    assert O.cons(["G", "C"], 1) == "G"
    assert O.cons(["G", "C", "C"], 1) == "C"
    assert O.cons(["$"], 1) == "A"          # non-ACGT counts as A and never displaces the initial 'A'
    assert O.cons(["c", "c"], 1) == "c"     # the raw character is kept


def test_failure_modes(tmp_path):
    p = str(tmp_path / "x.fasta")
    assert O.ebwt2clust(p)[0] == 2                       # missing index files
    IO.write_bcr(p, [0] * 4, [0] * 4, [0, 20, 20, 0], [65] * 4)
    assert O.ebwt2clust(p)[0] == 0
    os.remove(p + ".clusters")
    assert O.clust2snp(p, O.c2s_opts(2))[0] == 3         # missing .clusters
    open(p + ".clusters", "wb").close()
    assert O.clust2snp(p, O.c2s_opts(2))[0] == 4         # empty .clusters: reference SIGFPE (H6)
    IO.write_clusters(p + ".clusters", [1], [2])
    IO.write_fasta(p, ["ACGT" * 10] * 4)
    assert O.clust2snp(p, O.c2s_opts(2))[0] == 5         # zero candidates: reference SIGSEGV (H6)
