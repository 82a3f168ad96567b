"""the reference's own `.snp` consumers -- filter_snp, snp2fastq (format conformance of the KisSNP2 writer)
and snp_vs_vcf (sensitivity / precision against the planted truth) -- run over the golden `.snp` files here (no GPU) and over a
freshly produced HIP `.snp` on the GPU box.  The consumers are the reference's executables in oracle/_ref."""
import os

import pytest

import goldens
import snp_conformance as SC
from ebwt2snp_amd import synth

needs_tools = pytest.mark.skipif(not SC.have_tools(), reason="oracle/_ref consumers not built (make -C oracle ref)")


@needs_tools
@pytest.mark.parametrize("meta", goldens.e2e_cases(), ids=lambda m: m["name"])
def test_reference_consumers_parse_golden_snp(meta, tmp_path):
    n = SC.check_consumers(os.path.join(goldens.GOLD, meta["snp_file"]), str(tmp_path))
    assert n == meta["n_events"]


@needs_tools
def test_planted_snps_are_found_in_goldens(tmp_path):
    """snp_vs_vcf on every default-option golden: the planted SNPs come back (sensitivity / precision as the reference's own
    validation tool computes them); the figures are properties of the algorithm at 2 x 20x coverage, not of this implementation"""
    tp = fp = fn = 0
    for meta in goldens.e2e_cases():
        if meta["c2s"] or meta["e2c"]:
            continue
        tr = {}
        synth.make_reads(meta["R"], meta["L"], coverage=meta["coverage"], seed=meta["seed"], truth=tr)
        s = SC.score_against_truth(os.path.join(goldens.GOLD, meta["snp_file"]), tr, meta["L"], str(tmp_path))
        assert s["TP"] + s["FN"] == len(tr["snp_pos"])
        tp += s["TP"]; fp += s["FP"]; fn += s["FN"]
    assert tp > 100 and tp / (tp + fn) > 0.9 and tp / (tp + fp) > 0.9


@needs_tools
@pytest.mark.gpu
def test_reference_consumers_parse_fresh_hip_snp(tmp_path):
    """a `.snp` written by the HIP path for an input no golden covers (SNPs and 3-bp deletions planted): parsed by filter_snp /
    snp2fastq, scored by snp_vs_vcf, and byte-identical to the oracle's"""
    import oracle_lib as O
    from ebwt2snp_amd import egsa_io as IO, tools
    tr = {}
    reads, n1 = synth.make_reads(6000, 100, seed=77, truth=tr)
    e = synth.build_egsa(reads)
    e["lcp"][-1] = 0
    en = synth.egsa_to_numpy(e)
    out = {}
    for tag in ("hip", "orc"):
        os.makedirs(str(tmp_path / tag))
        p = str(tmp_path / tag / "reads.fasta")
        IO.write_fasta(p, reads.numpy())
        IO.write_bcr(p, en["text"], en["suff"], en["lcp"], en["bwt"])
        if tag == "hip":
            tools.ebwt2clust(p)
            tools.clust2snp(p, n1)
        else:
            assert O.ebwt2clust(p)[0] == 0 and O.clust2snp(p, O.c2s_opts(n1))[0] == 0
        out[tag] = IO.snp_out_path(p)
    assert open(out["hip"], "rb").read() == open(out["orc"], "rb").read()
    n = SC.check_consumers(out["hip"], str(tmp_path))
    s = SC.score_against_truth(out["hip"], tr, 100, str(tmp_path))
    assert n > 80 and s["TP"] + s["FN"] == len(tr["snp_pos"]) and s["sensitivity"] > 90 and s["precision"] > 90
