"""On-disk formats of the hot path (numpy plumbing for tests and bench).

Formats follow the reference's reader, include.hpp:120-199:
  * ``<fasta>.gesa``      array-of-structs  [text:y][suff:z][lcp:x][bwt:1]  (include.hpp:126-155)
  * BCR triple            ``<fasta>.out`` (bwt, 1 B), ``<fasta>.out.lcp`` (x B),
                          ``<fasta>.out.pairSA`` [suff:z][text:y]            (include.hpp:159-188)
  * ``<fasta>.clusters``  packed {u64 start; u16 length}, 10 B, no header     (ebwt2clust.cpp:58-59)
All integers little-endian.
"""
import os
import numpy as np

_DT = {1: "<u1", 2: "<u2", 4: "<u4", 8: "<u8"}

CLUSTER_DTYPE = np.dtype([("start", "<u8"), ("length", "<u2")])  # itemsize 10, packed


def write_bcr(prefix, text, suff, lcp, bwt, x=1, y=4, z=1):
    n = len(lcp)
    np.asarray(bwt, dtype=np.uint8).tofile(prefix + ".out")
    np.asarray(lcp).astype(_DT[x]).tofile(prefix + ".out.lcp")
    rec = np.dtype([("s", _DT[z]), ("t", _DT[y])])
    a = np.zeros(n, dtype=rec)
    a["s"] = suff
    a["t"] = text
    a.tofile(prefix + ".out.pairSA")


def write_gesa(prefix, text, suff, lcp, bwt, x=1, y=4, z=1):
    n = len(lcp)
    rec = np.dtype([("t", _DT[y]), ("s", _DT[z]), ("l", _DT[x]), ("b", "u1")])
    a = np.zeros(n, dtype=rec)
    a["t"] = text
    a["s"] = suff
    a["l"] = lcp
    a["b"] = bwt
    a.tofile(prefix + ".gesa")


def read_egsa(prefix, x=1, y=4, z=1):
    """Returns dict(text, suff, lcp, bwt) of numpy arrays (auto-detect like include.hpp:42-81)."""
    if os.path.exists(prefix + ".gesa"):
        rec = np.dtype([("t", _DT[y]), ("s", _DT[z]), ("l", _DT[x]), ("b", "u1")])
        a = np.fromfile(prefix + ".gesa", dtype=rec)
        return dict(text=a["t"].copy(), suff=a["s"].copy(), lcp=a["l"].copy(), bwt=a["b"].copy())
    rec = np.dtype([("s", _DT[z]), ("t", _DT[y])])
    g = np.fromfile(prefix + ".out.pairSA", dtype=rec)
    return dict(text=g["t"].copy(), suff=g["s"].copy(),
                lcp=np.fromfile(prefix + ".out.lcp", dtype=_DT[x]),
                bwt=np.fromfile(prefix + ".out", dtype=np.uint8))


def write_clusters(path, starts, lens):
    a = np.zeros(len(starts), dtype=CLUSTER_DTYPE)
    a["start"] = starts
    a["length"] = lens
    a.tofile(path)


def read_clusters(path):
    a = np.fromfile(path, dtype=CLUSTER_DTYPE)
    return a["start"].copy(), a["length"].copy()


def write_fasta(path, reads, width=None):
    """reads: list of str/bytes, or a 2-D uint8 array of ASCII codes (one read per row)."""
    with open(path, "wb") as f:
        if isinstance(reads, np.ndarray):
            R, L = reads.shape
            hdr = [b">r%d\n" % i for i in range(R)]
            for i in range(R):
                f.write(hdr[i])
                f.write(reads[i].tobytes())
                f.write(b"\n")
        else:
            for i, r in enumerate(reads):
                if isinstance(r, str):
                    r = r.encode()
                f.write(b">r%d\n" % i)
                if width:
                    for p in range(0, len(r), width):
                        f.write(r[p:p + width] + b"\n")
                else:
                    f.write(r + b"\n")


def write_fasta_fast(path, reads):
    """reads: uint8 [R, L] ASCII.  Fixed-width headers so the whole file is one array write."""
    R, L = reads.shape
    hdr = np.char.zfill(np.arange(R).astype(str), 9)
    rec = np.empty((R, 2 + 9 + 1 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord(">")
    rec[:, 1] = ord("r")
    rec[:, 2:11] = np.frombuffer("".join(hdr.tolist()).encode(), dtype=np.uint8).reshape(R, 9)
    rec[:, 11] = 10
    rec[:, 12:12 + L] = reads
    rec[:, 12 + L] = 10
    rec.tofile(path)


def snp_out_path(input_path):
    """clust2snp.cpp:1074-1076: input.substr(0, rfind(".fast")) + ".snp"."""
    p = input_path.rfind(".fast")
    return (input_path[:p] if p >= 0 else input_path) + ".snp"
