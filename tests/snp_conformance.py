"""Format conformance of `.snp` files against the reference's own consumers (test infrastructure):
filter_snp (filter_snp.cpp:53-95) and snp2fastq (snp2fastq.cpp:70-122) parse the KisSNP2 headers our writer emits,
snp_vs_vcf (snp_vs_vcf.cpp:108-573) scores the calls against a planted truth.  The three executables are the reference's,
compiled from the sources where they lie into oracle/_ref (oracle/Makefile)."""
import os
import re
import subprocess

import oracle_lib as O

HDR = re.compile(rb"^>(SNP|INDEL)_(higher|lower)_path_(\d+)\|P_1:(\d+)_([ACGTNacgtn]*)/([ACGTNacgtn]*)\|(\d+)\|nb_pol_1$")


def have_tools():
    return all(os.path.exists(os.path.join(O.REF_DIR, t)) for t in ("filter_snp", "snp2fastq", "snp_vs_vcf"))


def parse_snp(data):
    """independent parse of the writer's grammar (clust2snp.cpp:654-762): list of events
    (kind, id, rlen, a, b, cov0, dna0, cov1, dna1)"""
    lines = data.split(b"\n")
    assert lines[-1] == b"" and (len(lines) - 1) % 4 == 0, "4 lines per event, newline-terminated"
    ev = []
    for i in range(0, len(lines) - 1, 4):
        m0, m1 = HDR.match(lines[i]), HDR.match(lines[i + 2])
        assert m0 and m1, (lines[i], lines[i + 2])
        assert m0.group(2) == b"higher" and m1.group(2) == b"lower"
        assert m0.group(1) == m1.group(1) and m0.group(3) == m1.group(3) and m0.group(4) == m1.group(4)
        assert int(m0.group(3)) == len(ev) + 1, "ids are sequential from 1 (clust2snp.cpp:637,764)"
        ev.append((m0.group(1), int(m0.group(3)), int(m0.group(4)), m0.group(5), m0.group(6), int(m0.group(7)), lines[i + 1],
                   int(m1.group(7)), lines[i + 3]))
    return ev


def check_consumers(snp_path, tmp_dir):
    """run the reference's filter_snp and snp2fastq over the file and compare what they understood with parse_snp"""
    data = open(snp_path, "rb").read()
    ev = parse_snp(data)
    # filter_snp file M: keeps the 4-line records whose two coverages are >= M, byte for byte
    for M in (1, 3, 5, 8, 1000):
        r = subprocess.run([os.path.join(O.REF_DIR, "filter_snp"), snp_path, str(M)], capture_output=True)
        assert r.returncode == 0
        lines = data.split(b"\n")
        want = b"".join(b"\n".join(lines[4 * i:4 * i + 4]) + b"\n" for i, e in enumerate(ev) if e[5] >= M and e[7] >= M)
        assert r.stdout == want, "filter_snp M=%d" % M
    # snp2fastq file: one read per event, name = type_id_pos_event_cov0_cov1_DNA2, read = DNA1
    local = os.path.join(tmp_dir, "conf.snp")
    with open(local, "wb") as f:
        f.write(data)
    r = subprocess.run([os.path.join(O.REF_DIR, "snp2fastq"), local], capture_output=True)
    assert r.returncode == 0
    fq = open(local + ".fastq", "rb").read().split(b"\n")
    assert len(fq) - 1 == 4 * len(ev)
    for i, e in enumerate(ev):
        kind, eid, rlen, a, b, c0, d0, c1, d1 = e
        name = b"@%s_%d_%d_%s/%s_%d_%d_%s" % (kind, eid, rlen, a, b, c0, c1, d1)
        assert fq[4 * i] == name and fq[4 * i + 1] == d0 and fq[4 * i + 2] == b"+" and fq[4 * i + 3] == b"I" * len(d0), (i, fq[4 * i], name)
    # (with -i the reference's own line dispatch is off by one event -- snp2fastq.cpp:70,108: the second header is appended before
    #  the first one is parsed -- so only "runs and emits one read per event" is checked)
    r = subprocess.run([os.path.join(O.REF_DIR, "snp2fastq"), local, "-i"], capture_output=True)
    assert r.returncode == 0 and len(open(local + ".fastq", "rb").read().split(b"\n")) - 1 == 4 * len(ev)
    return len(ev)


def score_against_truth(snp_path, truth, read_len, tmp_dir):
    """snp_vs_vcf -v truth.vcf -c calls.snp -f genome.fasta -l read_len: (TP, FP, FN, sensitivity %, precision %)"""
    fa, vcf = os.path.join(tmp_dir, "genome.fasta"), os.path.join(tmp_dir, "truth.vcf")
    with open(fa, "wb") as f:
        f.write(b">chr1\n" + truth["genome"].tobytes() + b"\n")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\n")
        for p, r, a in zip(truth["snp_pos"], truth["ref"], truth["alt"]):
            f.write("chr1\t%d\t.\t%s\t%s\n" % (int(p) + 1, chr(r), chr(a)))
    r = subprocess.run([os.path.join(O.REF_DIR, "snp_vs_vcf"), "-v", vcf, "-c", snp_path, "-f", fa, "-l", str(read_len)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    g = lambda k: float(re.search(r"%s\s*=\s*(?:[A-Z/()+]+\s*=\s*)?([0-9.eE+-]+|nan|-nan)" % re.escape(k), r.stdout).group(1))
    return dict(TP=int(g("TP")), FP=int(g("FP")), FN=int(g("FN")), sensitivity=g("sensitivity"), precision=g("precision"))
