#!/usr/bin/env python3
"""development aid (needs a GPU and oracle/_ref): random synthetic cases and random options through bin/ebwt2clust + bin/clust2snp
and through the reference's own executables; `.clusters`, `.snp`, exit codes and the lines both print must agree.
usage: tools/stress_cli.py FIRST LAST (seeds)"""
import os, random, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as O
from ebwt2snp_amd import synth, egsa_io as IO

E2C, C2S = os.path.join(ROOT, "bin", "ebwt2clust"), os.path.join(ROOT, "bin", "clust2snp")
RAGGED = os.environ.get("E2S_STRESS_RAGGED", "0") == "1"       # (off: the seeds of rounds 1-2 produce the cases they produced then)
ERRORS = os.environ.get("E2S_STRESS_ERRORS", "0") == "1"       # sequencing errors in half of the cases (0.5 - 3 %): ties and near-ties in the consensus


def case(seed, d):
    rnd = random.Random(seed)
    R, L = rnd.choice([300, 800, 1500, 3000]), rnd.choice([60, 100, 150])
    xyz = rnd.choice([(1, 4, 1), (1, 4, 1), (2, 4, 1), (4, 4, 4), (1, 8, 2), (2, 8, 2), (8, 8, 8), (1, 4, 4)])
    cov, t0 = rnd.choice([10, 20, 35]), rnd.random() < 0.7
    ragged = RAGGED and rnd.random() < 0.5                      # reads of different lengths: the (start, length) index path of the GPU tail
    if ragged:
        reads, n1, e, _ = synth.synth_case_ragged(R, L, seed=seed, coverage=cov, tail_lcp0=t0)
    else:
        err = rnd.choice([0.005, 0.01, 0.03]) if ERRORS and rnd.random() < 0.5 else 0.0
        reads, n1, e = synth.synth_case(R, L, seed=seed, coverage=cov, tail_lcp0=t0, error_rate=err)
        reads = reads.numpy()
    en = synth.egsa_to_numpy(e, *xyz)
    ext = rnd.choice(["fasta", "fa", "fastq.txt"])              # H14: output name derived from the last ".fast"
    p = os.path.join(d, "reads." + ext)
    IO.write_fasta(p, reads, width=rnd.choice([None, None, 60]))
    (IO.write_gesa if rnd.random() < 0.4 else IO.write_bcr)(p, en["text"], en["suff"], en["lcp"], en["bwt"], *xyz)
    wopt = ["-x", xyz[0], "-y", xyz[1], "-z", xyz[2]] if xyz != (1, 4, 1) or rnd.random() < 0.3 else []
    a1 = list(wopt)
    if rnd.random() < 0.5: a1 += ["-k", rnd.choice([8, 12, 20, 31])]
    if rnd.random() < 0.4: a1 += ["-m", rnd.choice([1, 3, 8])]
    a2 = list(wopt)
    for opt, vals, pr in (("-L", [20, 25, 40, 1, 64, 65, 80], .35), ("-R", [10, 20, 45, 200], .3), ("-m", [2, 3, 7], .4), ("-c", [4, 6, 30, 150], .3),
                          ("-p", [0.9, 0.5, 0.999], .3), ("-g", [0, 3, 12, 70], .3), ("-e", [0, 1, 4], .3), ("-v", [1, 5], .2)):
        if rnd.random() < pr: a2 += [opt, rnd.choice(vals)]
    nn = rnd.choice([n1, n1, n1 // 2, n1 + 10])
    # the executables stream the EGSA in chunks: sometimes small ones (cuts anywhere, a last chunk of a few entries)
    chunk = rnd.choice([None, None, 5000, 20011, 70000])
    return p, a1, a2, nn, chunk


def outputs(p):
    cl = p + ".clusters"
    snp = IO.snp_out_path(p)
    return (open(cl, "rb").read() if os.path.exists(cl) else None, open(snp, "rb").read() if os.path.exists(snp) else None)


def clean(p):
    for q in (p + ".clusters", IO.snp_out_path(p)):
        if os.path.exists(q): os.remove(q)


def norm_rc(rc):        # the reference dies by signal (SIGFPE / SIGSEGV / abort); the drop-in exits with 128 + signal
    return rc if rc >= 0 else 128 - rc


def run(seed):
    d = tempfile.mkdtemp(prefix="e2s_cli_")
    try:
        p, a1, a2, nn, chunk = case(seed, d)
        env = dict(os.environ, E2S_CHUNK_ENTRIES=str(chunk)) if chunk else dict(os.environ)
        r1, o1 = O.ref_ebwt2clust(p, a1)
        want_cl = outputs(p)[0]
        r2, o2 = O.ref_clust2snp(p, nn, a2) if r1 == 0 else (None, "")
        want = outputs(p)
        clean(p)
        g1 = subprocess.run([E2C, "-i", p, *map(str, a1)], capture_output=True, text=True, env=env)
        assert norm_rc(g1.returncode) == norm_rc(r1), ("ebwt2clust exit", g1.returncode, r1, a1)
        assert outputs(p)[0] == want_cl, ("clusters differ", a1)
        assert [l for l in g1.stdout.splitlines() if l.startswith("Done.")] == [l for l in o1.splitlines() if l.startswith("Done.")], "ebwt2clust stdout"
        if r1 == 0:
            g2 = subprocess.run([C2S, "-i", p, "-n", str(nn), *map(str, a2)], capture_output=True, text=True, env=env)
            assert norm_rc(g2.returncode) == norm_rc(r2), ("clust2snp exit", g2.returncode, r2, a2, g2.stderr[-300:])
            got = outputs(p)
            assert got[1] == want[1], ("snp differs", a1, a2, nn, chunk, None if got[1] is None else len(got[1]), None if want[1] is None else len(want[1]))
        return True
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    assert O.have_ref(), "oracle/_ref missing (make -C oracle ref)"
    bad = 0
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        try:
            run(seed)
        except AssertionError as ex:
            bad += 1
            print("SEED", seed, "FAILED", str(ex)[:400], flush=True)
        if seed % 25 == 0:
            print("... seed", seed, flush=True)
    print("cli stress done, failures:", bad)
    sys.exit(1 if bad else 0)
