"""Loaders for tests/golden/ (data produced by tests/golden/make_golden.py from the reference)."""
import hashlib
import json
import os

import numpy as np

from ebwt2snp_amd import egsa_io as IO
from ebwt2snp_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def e2c_table():
    with open(os.path.join(GOLD, "e2c_table.json")) as f:
        tab = json.load(f)
    for c in tab:
        lcp = []
        for v, r in c["lcp_rle"]:
            lcp.extend([v] * r)
        c["lcp"] = lcp
    return tab


def e2e_cases():
    with open(os.path.join(GOLD, "e2e_cases.json")) as f:
        return json.load(f)


def crafted():
    with open(os.path.join(GOLD, "crafted_c2s.json")) as f:
        return json.load(f)


def e2e_inputs(meta):
    """(reads uint8 [2R,L], nr_reads1, dict of numpy EGSA arrays).  Regenerated from the seed and
    checked against the digest recorded when the reference ran on them."""
    if "inputs_file" in meta:
        z = np.load(os.path.join(GOLD, meta["inputs_file"]))
        reads = z["reads"]
        en = {k: z[k] for k in ("text", "suff", "lcp", "bwt")}
    elif meta.get("ragged"):
        reads, n1, e, _ = synth.synth_case_ragged(meta["R"], meta["L"], seed=meta["seed"], coverage=meta["coverage"], tail_lcp0=meta["tail_lcp0"])
        en = synth.egsa_to_numpy(e, *meta["xyz"])
    else:
        r, n1, e = synth.synth_case(meta["R"], meta["L"], seed=meta["seed"], coverage=meta["coverage"],
                                    tail_lcp0=meta["tail_lcp0"])
        reads = r.numpy()
        en = synth.egsa_to_numpy(e, *meta["xyz"])
    h = hashlib.sha256()
    h.update(b"\n".join(reads) if isinstance(reads, list) else reads.tobytes())
    for k in ("text", "suff", "lcp", "bwt"):
        h.update(en[k].astype(np.uint64).tobytes())
    assert h.hexdigest() == meta["input_sha256"], "synthetic generator drifted from the golden inputs"
    return reads, meta["nr_reads1"], en


def write_case_files(meta, reads, en, d, name="reads.fasta"):
    p = os.path.join(d, name)
    IO.write_fasta(p, reads, width=meta.get("fasta_width") or None)
    x, y, z = meta["xyz"]
    (IO.write_gesa if meta["gesa"] else IO.write_bcr)(p, en["text"], en["suff"], en["lcp"], en["bwt"], x, y, z)
    return p


def golden_snp(meta):
    with open(os.path.join(GOLD, meta["snp_file"]), "rb") as f:
        return f.read()
