"""Python mirror of the two drop-in executables (bin/ebwt2clust, bin/clust2snp): same options,
same defaults (0 = "use default", ebwt2clust.cpp:175-180, clust2snp.cpp:1032-1043), same files
in and out.  All work is done by libe2s_hip.so through the C ABI (ebwt2snp_amd.capi); this module
only reads/writes files and applies the option defaults.  Used by the GPU parity tests."""
import os

import numpy as np

from . import capi
from .egsa_io import snp_out_path


class MissingIndex(RuntimeError):
    """include.hpp:74-75: 'Error: missing index files.' exit(1)"""


def _widths(x, y, z):
    return (x or 1, y or 4, z or 1)          # lcp_def, da_def, pos_def


def _open_shard(h, input_path, x, y, z, lcp_only=False):
    """egsa_stream's constructor (include.hpp:42-81): .gesa first, then the BCR triple."""
    if os.path.exists(input_path + ".gesa"):
        rec = y + z + x + 1
        aos = np.fromfile(input_path + ".gesa", dtype=np.uint8)
        n = aos.size // rec
        d = capi.Handle.desc(n, x=x, y=y, z=z)
        h.upload_gesa(d, aos[:n * rec])
        return n
    paths = [input_path + e for e in (".out.lcp", ".out", ".out.pairSA")]
    if not all(os.path.exists(p) for p in paths):
        raise MissingIndex("Error: missing index files.")
    lcp = np.fromfile(paths[0], dtype=np.uint8)
    n = lcp.size // x
    d = capi.Handle.desc(n, x=x, y=y, z=z)
    if lcp_only:
        with open(paths[1], "rb") as f:
            f.seek(max(n - 1, 0))
            tb = f.read(1)
        h.upload_lcp(d, lcp[:n * x], tb[0] if tb else 0)
    else:
        bwt = np.fromfile(paths[1], dtype=np.uint8)
        psa = np.fromfile(paths[2], dtype=np.uint8)
        h.upload_bcr(d, bwt[:n], lcp[:n * x], psa[:n * (y + z)])
    return n


def ebwt2clust(input_path, k=0, m=0, x=0, y=0, z=0, device=0, scan_mode=0, handle=None):
    """ebwt2clust -i input [-k -m -x -y -z].  Writes input.clusters; returns the count the reference prints."""
    x, y, z = _widths(x, y, z)
    k = k or 16
    m = m or 2
    h = handle or capi.Handle(device)
    try:
        _open_shard(h, input_path, x, y, z, lcp_only=True)
        h.scan_single(k, m, scan_mode)
        n_closed, _ = h.counts()
        h.fetch_clusters().tofile(input_path + ".clusters")
        return n_closed
    finally:
        if handle is None:
            h.close()


def clust2snp(input_path, n, L=0, R=0, g=0, v=0, c=0, e=0, m=0, p=0.0, x=0, y=0, z=0, device=0, handle=None,
              return_candidates=False, tail="auto"):
    """clust2snp -i input -n N [...].  Reads input.clusters + the EGSA + the FASTA; writes the .snp file.
    tail: "auto" = what the executable does (reads made resident, consensus / distance / text on the GPU, the host tail where
    only it reproduces the reference), "host" = the host tail (snp_tail.cpp) over the packed candidates."""
    x, y, z = _widths(x, y, z)
    L = L or 31
    R = R or 30
    g = g or 10
    c = c or 20
    e = e or 2
    m = m or 5
    p = p or 0.99
    nr_reads1 = int(np.int32(n)) & ((1 << 64) - 1)          # atoi -> uint64_t
    cl_path = input_path + ".clusters"
    if not os.path.exists(cl_path):
        raise FileNotFoundError('ERROR: Could not find BWT clusters file "%s"' % cl_path)
    h = handle or capi.Handle(device)
    try:
        _open_shard(h, input_path, x, y, z)
        h.attach_reads(None, None, None, 0, 0)                      # (a handle reused across jobs: forget the previous job's reads)
        if tail != "host":
            try:
                h.upload_reads_fasta(input_path)
            except capi.E2SError as ex:
                if ex.code not in (6, 8):                           # ERR_INPUT: irregular FASTA, ERR_IO: unreadable -> the host tail decides
                    raise
        recs = np.fromfile(cl_path, dtype=np.uint8)
        h.load_clusters(recs[:(recs.size // 10) * 10])
        mcl, n_clust, n_bases = capi.auto_max_clust_length([h.hist()], m, p)
        prm = capi.C2SParams(nr_reads1, L, R, m, c, mcl, 0)
        ncand = h.find_candidates(prm)
        out = snp_out_path(input_path)
        if tail == "host":
            stream, off8, ncand, pw = h.candidates_packed()
            nv, ne = capi.snp_from_packed(stream, off8, ncand, input_path, out, L, R, g, e, pos_bytes=pw)
        else:
            nv, ne = h.write_snp(input_path, out, L, R, g, e)
        res = dict(max_clust_length=mcl, n_clust=n_clust, n_bases=n_bases, n_candidates=ncand, n_variants=nv, n_events=ne,
                   out_path=out)
        if return_candidates:
            buf, ncand, cap, stride = h.fetch_candidates(prm)
            res["candidates"] = capi.parse_candidates(buf, ncand, cap, stride)
        return res
    finally:
        if handle is None:
            h.close()
