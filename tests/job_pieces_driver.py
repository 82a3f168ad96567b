"""Child process of test_gpu_parity.py::test_resident_job_in_pieces: the library reads E2S_JOB_PIECES once, when it is loaded, so the
native job with its per-cluster pass cut into pieces needs a process of its own.  usage: job_pieces_driver.py <out prefix>
Writes <prefix>.clusters / <prefix>.snp (the images of the native job)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ebwt2snp_amd import capi, pipeline, synth  # noqa: E402
from test_gpu_parity import _attach_job_inputs  # noqa: E402


def main(prefix):
    dev = torch.device("cuda", 0)
    reads, n1, e = synth.synth_case(5000, 100, seed=71)
    en = synth.egsa_to_numpy(e)
    n = en["lcp"].size
    h = capi.Handle(0)
    _attach_job_inputs(h, reads.numpy(), en, n, 0, n, 0, 0, dev)
    expect_fallback = int(os.environ.get("E2S_TEST_EXPECT_FALLBACK", "0"))        # the first N steps are redone stepwise
    for it in range(2):                                          # (twice: the second step finds the buffers and counters of the first)
        r = pipeline.resident_job(h, n1, n, native=True)
        # (the testing aids make the first step overflow a buffer: it reports it and is redone stepwise; the second fits -- the segments'
        #  estimate apart, which is made anew for every scan)
        assert bool(r.get("native")) == (it >= expect_fallback), (it, r)
    cl, snp = h.fetch_clusters().tobytes(), h.fetch_snp().tobytes()
    h.close()
    open(prefix + ".clusters", "wb").write(cl)
    open(prefix + ".snp", "wb").write(snp)


if __name__ == "__main__":
    main(sys.argv[1])
