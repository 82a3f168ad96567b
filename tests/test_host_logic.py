"""CPU-only tests of the product's host-side pieces (everything in libe2s_hip.so that needs no GPU)
and of the kernel's SWAR stencil compiled for the host."""
import ctypes as C
import os
import random
import re
import subprocess

import numpy as np
import pytest

import goldens
import oracle_lib as O
import shard_emul as SE
from ebwt2snp_amd import capi
from ebwt2snp_amd import egsa_io as IO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = capi.lib()
    hdr = open(os.path.join(ROOT, "include", "e2s_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(e2s_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(capi.SYMBOLS) == declared
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.e2s_version()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.E2SError) as ei:
        capi.Handle(0)
    assert "no CPU path" in str(ei.value)


def test_swar_stencil_host_build(tmp_path):
    exe = str(tmp_path / "swar_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "swar_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_fasta_chunked_index_equals_whole_file_index(tmp_path):
    """the index built chunk by chunk while the FASTA streams to the GPU against the one over the whole file (sanitizers on)"""
    exe = str(tmp_path / "fasta_chunks_check")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "fasta_chunks_check.cpp"),
                    os.path.join(ROOT, "ebwt2snp_amd", "host", "fasta_index.cpp"), "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_fasta_index_open_one_pass_equals_two_pass(tmp_path):
    """e2s_fasta_index_open (host only: no GPU): a file of the plain shape is indexed in ONE pass over its mapping, chunk by chunk; every other shape --
    and every file under E2S_FASTA_TWO_PASS=1 -- by the two-pass index.  Same reads either way; a multi-line file joins its lines; an irregular one is
    E2S_ERR_INPUT"""
    import ctypes as C
    L = C.CDLL(capi.LIB_PATH)
    L.e2s_fasta_index_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.e2s_fasta_index_close.argtypes = [C.c_void_p]

    def index(path):
        ix, text, st, ln = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        tb, nr, ml = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        rc = L.e2s_fasta_index_open(path.encode(), C.byref(ix), C.byref(text), C.byref(tb), C.byref(st), C.byref(ln), C.byref(nr), C.byref(ml))
        if rc:
            return rc, None
        n = nr.value
        starts = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint64)), shape=(n,)).copy()
        lens = np.ctypeslib.as_array(C.cast(ln, C.POINTER(C.c_uint32)), shape=(n,)).copy()
        buf = np.ctypeslib.as_array(C.cast(text, C.POINTER(C.c_uint8)), shape=(tb.value,))
        reads = [buf[int(a):int(a) + int(b)].tobytes() for a, b in zip(starts, lens)]
        L.e2s_fasta_index_close(ix)
        return 0, (reads, ml.value)

    rnd = np.random.RandomState(5)
    saved = os.environ.get("E2S_FASTA_TWO_PASS")
    try:
        for case in range(12):
            nrec = int(rnd.randint(1, 4000))
            same = case % 2 == 0
            seqs = ["".join("ACGTN"[i] for i in rnd.randint(0, 5, 50 if same else rnd.randint(0, 120))) for _ in range(nrec)]
            multi = case % 4 == 3
            p = str(tmp_path / ("f%d.fasta" % case))
            with open(p, "w") as f:
                for i, q in enumerate(seqs):
                    f.write(">read_%d some text\n" % i)
                    if multi and len(q) > 3 and i % 7 == 0:
                        f.write(q[:len(q) // 2] + "\n" + q[len(q) // 2:] + "\n")
                    else:
                        f.write(q + ("\n" if (i + 1 < nrec or case % 3) else ""))
            got = {}
            for mode in ("one", "two"):
                if mode == "two":
                    os.environ["E2S_FASTA_TWO_PASS"] = "1"
                else:
                    os.environ.pop("E2S_FASTA_TWO_PASS", None)
                rc, res = index(p)
                assert rc == 0, (case, mode, rc)
                got[mode] = res
            want = [q.encode() for q in seqs]
            assert got["one"][0] == want and got["two"][0] == want and got["one"][1] == got["two"][1] == max(len(q) for q in seqs), case
        p = str(tmp_path / "bad.fasta")
        open(p, "w").write(">a\n>b\nACGT\n")
        os.environ.pop("E2S_FASTA_TWO_PASS", None)
        assert index(p)[0] == 6                                      # E2S_ERR_INPUT
    finally:
        if saved is None:
            os.environ.pop("E2S_FASTA_TWO_PASS", None)
        else:
            os.environ["E2S_FASTA_TWO_PASS"] = saved


def _stitch_all(L, bwt, cuts, k, m, x):
    """records of the whole EGSA assembled from emulated shards + the product's e2s_stitch_plan"""
    n = len(L)
    bounds = [0] + list(cuts) + [n]
    shards = [SE.shard_scan(L, bwt, bounds[i], bounds[i + 1], k) for i in range(len(bounds) - 1)]
    allb = [s[2] for s in shards]
    S, Ln, closed = [], [], 0
    for r, (st, ln, _) in enumerate(shards):
        plan = capi.stitch_plan(allb, r, n, k, x)
        st2, ln2 = SE.apply_plan(st, ln, plan)
        # the statistics the scan phase leaves + the plan (e2s_hist_after_stitch, a pure host function) must equal the
        # statistics of the records the shard holds after the stitch -- for every min_len that matters
        for mm in (m, 1, 3, 40):
            got, want = capi.hist_after_stitch(SE.raw_stats(ln, mm), allb[r], plan), SE.hist_of(ln2, mm)
            assert list(got.freq) == list(want.freq)
            assert (got.n_records, got.n_bases, got.last_len) == (want.n_records, want.n_bases, want.last_len)
        S.append(st2)
        Ln.append(ln2)
        closed += st2.size
    S = np.concatenate(S)
    Ln = np.concatenate(Ln)
    keep = Ln.astype(np.int64) >= m
    return S[keep], Ln[keep], closed


def test_stencil_and_stitch_plan_against_golden_table():
    rnd = random.Random(5)
    for c in goldens.e2c_table():
        L = c["lcp"]
        n = len(L)
        if n < 1:
            continue
        x = c.get("x", 1)
        k = c.get("k", 0) or 16
        m = c.get("m", 0) or 2
        bwt = c.get("bwt", [65] * n)
        for nsh in (1, 2, 3):
            if nsh > 1 and n < 2 * nsh + 2:
                continue
            cuts = sorted(rnd.sample(range(1, n - 1), nsh - 1)) if nsh > 1 else []
            if nsh > 1 and n - cuts[-1] < 2:
                continue
            S, Ln, closed = _stitch_all(L, bwt, cuts, k, m, x)
            assert closed == c["expect"]["n_printed"], (c.get("k"), nsh, cuts)
            assert S.tolist() == c["expect"]["starts"] and Ln.tolist() == c["expect"]["lens"], (nsh, cuts)


def test_stitch_plan_random_long_clusters():
    """clusters spanning whole shards (pass-through) and u16 wrap across a cut"""
    rnd = np.random.RandomState(3)
    for it in range(60):
        n = int(rnd.randint(50, 4000))
        L = np.where(rnd.rand(n) < 0.02, rnd.randint(0, 16, n), rnd.randint(16, 40, n)).astype(np.uint32)
        if it % 3 == 0:
            L[:] = 20
            L[0] = 0
        bwt = rnd.choice([36, 65, 67, 71, 84], n).astype(np.uint8)
        ph = O.phantom_lcp_for_scan(L, bwt, 16, 1)
        es, el, ec = O.cluster_lm_mem(L, ph, 16, 2)
        nsh = int(rnd.randint(1, 6))
        cuts = sorted(set(rnd.randint(1, n - 2, nsh - 1).tolist())) if nsh > 1 else []
        S, Ln, closed = _stitch_all(L, bwt, cuts, 16, 2, 1)
        assert closed == ec
        assert S.tolist() == es.tolist() and Ln.tolist() == el.tolist()


def test_auto_max_clust_length_matches_reference_statistics():
    for meta in goldens.e2e_cases():
        if "clusters_file" not in meta:
            continue
        s, l = IO.read_clusters(os.path.join(goldens.GOLD, meta["clusters_file"]))
        hs = capi.Hist()
        for v in l:
            if v <= 150:
                hs.freq[int(v)] += 1
        hs.n_records = len(l)
        hs.n_bases = int(l.astype(np.int64).sum())
        hs.last_len = int(l[-1])
        mcl, nc, nb = capi.auto_max_clust_length([hs], meta["c2s"].get("m", 0) or 5, meta["c2s"].get("p", 0) or 0.99)
        assert mcl == meta["max_clust_length"] and nc == meta["n_records"] + 1
        # split into two partial histograms (what two shards would report)
        h1, h2 = capi.Hist(), capi.Hist()
        half = len(l) // 2
        for hh, part in ((h1, l[:half]), (h2, l[half:])):
            for v in part:
                if v <= 150:
                    hh.freq[int(v)] += 1
            hh.n_records = len(part)
            hh.n_bases = int(part.astype(np.int64).sum())
            hh.last_len = int(part[-1])
        assert capi.auto_max_clust_length([h1, h2], 5, 0.99)[0] == mcl
    # empty .clusters: the reference dies with SIGFPE (H6)
    with pytest.raises(capi.E2SError):
        capi.auto_max_clust_length([capi.Hist()], 5, 0.99)


def _pack_candidates(cands, cap):
    hs = C.sizeof(capi.CandHeader)
    stride = hs + 16 * cap
    buf = np.zeros(len(cands) * stride, dtype=np.uint8)
    for i, c in enumerate(cands):
        h = capi.CandHeader(i, 0, len(c["idx0"]), len(c["idx1"]), c["right_idx"], c["right_pos"], 0)
        buf[i * stride:i * stride + hs] = np.frombuffer(bytes(h), dtype=np.uint8)
        l = buf[i * stride + hs:(i + 1) * stride].view(np.uint32)
        l[0:len(c["idx0"])] = c["idx0"]
        l[cap:cap + len(c["pos0"])] = c["pos0"]
        l[2 * cap:2 * cap + len(c["idx1"])] = c["idx1"]
        l[3 * cap:3 * cap + len(c["pos1"])] = c["pos1"]
    return buf


def _pack_stream(cands, pw):
    """the packed layout of e2s_candidates_packed: u32 n0, n1, right_idx, right_pos | idx0 | idx1 | pos0 | pos1 (pw bytes each),
    padded to 8 bytes; offsets in 8-byte units"""
    pdt = {1: np.uint8, 2: "<u2", 4: "<u4"}[pw]
    parts, off8, at = [], [], 0
    for c in cands:
        hdr = np.asarray([len(c["idx0"]), len(c["idx1"]), c["right_idx"], c["right_pos"]], dtype="<u4").view(np.uint8)
        idx = np.asarray(list(c["idx0"]) + list(c["idx1"]), dtype="<u4").view(np.uint8)
        pos = np.asarray(list(c["pos0"]) + list(c["pos1"]), dtype=np.uint64).astype(pdt).view(np.uint8)
        rec = np.concatenate([hdr, idx, pos])
        rec = np.concatenate([rec, np.zeros((-rec.size) % 8, dtype=np.uint8)])
        off8.append(at // 8)
        parts.append(rec)
        at += rec.size
    off8.append(at // 8)
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8), np.asarray(off8, dtype=np.uint32)


@pytest.mark.parametrize("meta", [m for m in goldens.e2e_cases() if m["name"] in ("tiny_bcr", "c1_bcr", "opts_bcr", "w282_bcr", "indel_bcr")],
                         ids=lambda m: m["name"])
def test_host_tail_on_oracle_candidates(meta, tmp_path):
    """product host tail (extract_variants + distance + to_file) fed with the oracle's candidates
    must reproduce the reference's .snp byte for byte"""
    reads, n1, en = goldens.e2e_inputs(meta)
    p = goldens.write_case_files(meta, reads, en, str(tmp_path))
    x, y, z = meta["xyz"]
    assert O.ebwt2clust(p, meta["e2c"].get("k", 0), meta["e2c"].get("m", 0), x, y, z)[0] == 0
    a = meta["c2s"]
    opts = O.c2s_opts(n1, L=a.get("L", 0), R=a.get("R", 0), g=a.get("g", 0), c=a.get("c", 0), e=a.get("e", 0),
                      m=a.get("m", 0), p=a.get("p", 0.0), x=x, y=y, z=z)
    rc, cands, mcl = O.candidates(p, opts)
    assert rc == 0 and len(cands) == meta["n_candidates"]
    cap = a.get("c", 0) or 20
    out = str(tmp_path / "product.snp")
    nv, ne = capi.snp_from_candidates(_pack_candidates(cands, cap), len(cands), cap, p, out,
                                      a.get("L", 0) or 31, a.get("R", 0) or 30, a.get("g", 0) or 10, a.get("e", 0) or 2)
    assert open(out, "rb").read() == goldens.golden_snp(meta)
    # the same tail over the packed stream (the layout the GPU writes into the pinned buffer)
    maxpos = max([0] + [v for c in cands for v in list(c["pos0"]) + list(c["pos1"])])
    for pw in (1, 2, 4):
        if maxpos >= 1 << (8 * pw):
            continue
        stream, off8 = _pack_stream(cands, pw)
        out2 = str(tmp_path / ("product_packed%d.snp" % pw))
        nv2, ne2 = capi.snp_from_packed(stream, off8, len(cands), p, out2, a.get("L", 0) or 31, a.get("R", 0) or 30, a.get("g", 0) or 10,
                                        a.get("e", 0) or 2, pos_bytes=pw)
        assert (nv2, ne2) == (nv, ne) and open(out2, "rb").read() == goldens.golden_snp(meta)
    assert ne == meta["n_events"]


def test_host_tail_crafted(tmp_path):
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
        rc, cands, _ = O.candidates(p, opts)
        assert rc == 0
        cap = a.get("c", 0) or 20
        out = str(d / "product.snp")
        capi.snp_from_candidates(_pack_candidates(cands, cap), len(cands), cap, p, out,
                                 a.get("L", 0) or 31, a.get("R", 0) or 30, a.get("g", 0) or 10, a.get("e", 0) or 2)
        assert open(out).read() == c["expected_snp"], c["name"]


def test_host_tail_fasta_layouts(tmp_path):
    """multi-line FASTA records and a missing final newline parse like the reference's get_reads"""
    meta = [m for m in goldens.e2e_cases() if m["name"] == "tiny_bcr"][0]
    reads, n1, en = goldens.e2e_inputs(meta)
    p = goldens.write_case_files(meta, reads, en, str(tmp_path))
    assert O.ebwt2clust(p)[0] == 0
    rc, cands, _ = O.candidates(p, O.c2s_opts(n1))
    for variant in ("wrapped", "nonl"):
        q = str(tmp_path / (variant + ".fasta"))
        rl = [bytes(r).decode() for r in reads]
        if variant == "wrapped":
            IO.write_fasta(q, rl, width=37)
        else:
            IO.write_fasta(q, rl)
            data = open(q, "rb").read()
            open(q, "wb").write(data[:-1])
        # the oracle on the same layout
        for ext in (".out", ".out.lcp", ".out.pairSA", ".clusters"):
            os.link(p + ext, q + ext)
        assert O.clust2snp(q, O.c2s_opts(n1))[0] == 0
        out = str(tmp_path / (variant + ".product.snp"))
        capi.snp_from_candidates(_pack_candidates(cands, 20), len(cands), 20, q, out)
        assert open(out, "rb").read() == open(IO.snp_out_path(q), "rb").read() == goldens.golden_snp(meta)


@pytest.mark.parametrize("case", ["tiny_bcr", "c1_bcr"])
def test_host_tail_threads_and_parallel_fasta(case, tmp_path, monkeypatch):
    """the tail with worker threads (consensus per candidate; the FASTA cut into pieces at line starts) gives the bytes of the
    single-threaded reference whatever the thread count and the FASTA layout: one line per read, wrapped at 37 and at 1
    base per line, no final newline, CRLF line ends -- and layouts the piece-wise parser must hand back to the sequential
    restatement of get_reads: a header directly after a header, a header as the last line"""
    meta = [m for m in goldens.e2e_cases() if m["name"] == case][0]
    reads, n1, en = goldens.e2e_inputs(meta)
    p = goldens.write_case_files(meta, reads, en, str(tmp_path))
    assert O.ebwt2clust(p)[0] == 0
    rc, cands, _ = O.candidates(p, O.c2s_opts(n1))
    assert rc == 0 and cands
    rl = [bytes(r).decode() for r in reads]
    layouts = {}
    for name, width in (("plain", None), ("w37", 37), ("w1", 1)):
        q = str(tmp_path / (name + ".fasta"))
        IO.write_fasta(q, rl) if width is None else IO.write_fasta(q, rl, width=width)
        layouts[name] = open(q, "rb").read()
    layouts["nonl"] = layouts["w37"][:-1]
    layouts["crlf"] = layouts["plain"].replace(b"\n", b"\r\n")
    layouts["tail_header"] = layouts["plain"] + b">one more header\n"
    layouts["double_header"] = layouts["w37"] + b">empty\n>last\nACGT\n"
    for name, data in layouts.items():
        q = str(tmp_path / (name + ".fasta"))
        open(q, "wb").write(data)
        for ext in (".out", ".out.lcp", ".out.pairSA", ".clusters"):
            if not os.path.exists(q + ext):
                os.link(p + ext, q + ext)
        assert O.clust2snp(q, O.c2s_opts(n1))[0] == 0, name
        want = open(IO.snp_out_path(q), "rb").read()
        assert len(want) > 0
        for threads in ("1", "3", "8"):
            monkeypatch.setenv("E2S_TAIL_THREADS", threads)
            monkeypatch.setenv("E2S_TAIL_PAR_MIN_BYTES", "1")
            out = str(tmp_path / ("%s.t%s.snp" % (name, threads)))
            capi.snp_from_candidates(_pack_candidates(cands, 20), len(cands), 20, q, out)
            assert open(out, "rb").read() == want, (name, threads)


def test_integration_snippets_compile():
    """INTEGRATION.md's C++ snippets are what a maintainer of the reference would paste: every ```cpp block of the document must
    appear verbatim (its #include line apart) in tests/integration_snippet.cpp, and that file must compile against
    include/e2s_hip.h -- so the document cannot drift from the header"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src_path = os.path.join(ROOT, "tests", "integration_snippet.cpp")
    src = open(src_path).read()
    blocks = re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)
    assert len(blocks) >= 2
    for b in blocks:
        body = "\n".join(l for l in b.split("\n") if not l.startswith("#include"))
        assert body.strip() and body in src, "INTEGRATION.md block not found verbatim in tests/integration_snippet.cpp:\n" + body[:300]
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src_path],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_sanitizer_builds_over_goldens(tmp_path):
    """`make sanitize`: the host tail (worker threads, mmap, pointer walks), the FASTA index and the oracle's C restatement built with
    -fsanitize=address,undefined and with -fsanitize=thread, every end-to-end golden through both with 8 tail threads and the parallel
    FASTA cut forced on (E2S_TAIL_PAR_MIN_BYTES=1): no report, and the host tail's bytes equal the oracle's, which equal the golden's"""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", ROOT, "-s", "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for meta in goldens.e2e_cases():
        reads, n1, en = goldens.e2e_inputs(meta)
        d = tmp_path / meta["name"]
        d.mkdir()
        p = goldens.write_case_files(meta, reads, en, str(d))
        x, y, z = meta["xyz"]
        e, a = meta["e2c"], meta["c2s"]
        args = [p, str(n1), str(e.get("k", 0)), str(e.get("m", 0)), str(x), str(y), str(z), str(a.get("L", 0)), str(a.get("R", 0)), str(a.get("g", 0)),
                str(a.get("c", 0)), str(a.get("e", 0)), str(a.get("m", 0)), str(a.get("p", 0.0))]
        for exe, env in (("san_asan", {"ASAN_OPTIONS": "halt_on_error=1:detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}),
                         ("san_tsan", {"TSAN_OPTIONS": "halt_on_error=1:exitcode=66"})):
            env = dict(os.environ, E2S_TAIL_THREADS="8", E2S_TAIL_PAR_MIN_BYTES="1", **env)
            r = subprocess.run([os.path.join(ROOT, "build", exe)] + args, capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode == 0 and r.stdout.startswith("ok:"), (meta["name"], exe, r.returncode, r.stdout[-500:], r.stderr[-3000:])
            assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, (meta["name"], exe, r.stderr[-3000:])
            assert open(p + ".oracle.snp", "rb").read() == goldens.golden_snp(meta), meta["name"]


def test_executables_without_a_gpu_say_why_and_leave_no_partial_output(tmp_path):
    """The jobs own their handles: their failure text is the thread's handle-less message, and a failed ebwt2clust removes the
    `.clusters` it had opened (a partial file used to stay behind, and only a number was printed)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    meta = [m for m in goldens.e2e_cases() if m["name"] == "tiny_bcr"][0]
    reads, n1, en = goldens.e2e_inputs(meta)
    p = goldens.write_case_files(meta, reads, en, str(tmp_path))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "bin", "ebwt2clust"), "-i", p], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device available" in r.stderr and "no CPU path" in r.stderr, (r.returncode, r.stderr)
    assert not os.path.exists(p + ".clusters")
    with open(p + ".clusters", "wb") as f:
        f.write(open(os.path.join(goldens.GOLD, "tiny_bcr.clusters"), "rb").read())
    r = subprocess.run([os.path.join(root, "bin", "clust2snp"), "-i", p, "-n", str(n1)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device available" in r.stderr, (r.returncode, r.stderr)
    assert not os.path.exists(p[:p.rfind(".fast")] + ".snp")
