"""ctypes binding of oracle/liboracle.so (the CPU restatement).  Checker only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

RC_NAMES = {0: "OK", 1: "HELP", 2: "MISSING_INDEX", 3: "MISSING_CLUSTERS", 4: "REF_SIGFPE",
            5: "REF_SIGSEGV", 6: "REF_UB_ABORT", 7: "BAD_WIDTH", 8: "IO"}


class E2COpts(C.Structure):
    _fields_ = [("k", C.c_int), ("min_len", C.c_int), ("lcp", C.c_int), ("da", C.c_int), ("pos", C.c_int)]


class C2SOpts(C.Structure):
    _fields_ = [("nr_reads1", C.c_int64), ("k_left", C.c_int), ("k_right", C.c_int), ("max_gap", C.c_int),
                ("max_snvs", C.c_int), ("consensus_reads", C.c_int), ("max_err", C.c_int),
                ("mcov_out", C.c_int), ("pval", C.c_double), ("lcp", C.c_int), ("da", C.c_int), ("pos", C.c_int)]


class C2SStats(C.Structure):
    _fields_ = [("n_clust", C.c_uint64), ("n_bases", C.c_uint64), ("max_clust_length", C.c_int),
                ("n_candidates", C.c_uint64), ("n_variants", C.c_uint64), ("n_events", C.c_uint64)]


_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_ebwt2clust.argtypes = [C.c_char_p, C.POINTER(E2COpts), C.POINTER(C.c_uint64)]
        L.orc_clust2snp.argtypes = [C.c_char_p, C.POINTER(C2SOpts), C.c_char_p, C.POINTER(C2SStats), C.c_int]
        L.orc_candidates.argtypes = [C.c_char_p, C.POINTER(C2SOpts), C.POINTER(C.c_uint64), C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.orc_distance.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_distance.restype = None
        L.orc_cons.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
        L.orc_cons.restype = None
        L.orc_cluster_lm_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.orc_cluster_lm_mem.restype = C.c_size_t
        L.orc_phantom_lcp_for_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.orc_phantom_lcp_for_scan.restype = C.c_uint32
        _lib = L
    return _lib


def ebwt2clust(input_path, k=0, m=0, x=0, y=0, z=0):
    o = E2COpts(k, m, x, y, z)
    n = C.c_uint64(0)
    rc = lib().orc_ebwt2clust(input_path.encode(), C.byref(o), C.byref(n))
    return rc, n.value


def c2s_opts(n, L=0, R=0, g=0, v=0, c=0, e=0, m=0, p=0.0, x=0, y=0, z=0):
    return C2SOpts(n, L, R, g, v, c, e, m, p, x, y, z)


def clust2snp(input_path, opts, out_path=None, verbose=False):
    st = C2SStats()
    rc = lib().orc_clust2snp(input_path.encode(), C.byref(opts), out_path.encode() if out_path else None,
                             C.byref(st), 1 if verbose else 0)
    return rc, st


def candidates(input_path, opts):
    """Returns (rc, list of candidate dicts, max_clust_length)."""
    import numpy as np
    nw = C.c_size_t(0)
    nc = C.c_uint64(0)
    mcl = C.c_int(0)
    rc = lib().orc_candidates(input_path.encode(), C.byref(opts), None, 0, C.byref(nw), C.byref(nc), C.byref(mcl))
    buf = np.zeros(max(nw.value, 1), dtype=np.uint64)
    rc = lib().orc_candidates(input_path.encode(), C.byref(opts), buf.ctypes.data_as(C.POINTER(C.c_uint64)),
                              buf.size, C.byref(nw), C.byref(nc), C.byref(mcl))
    out = []
    w = 0
    for _ in range(nc.value):
        n0, n1, ri, rp = (int(v) for v in buf[w:w + 4])
        w += 4
        d = dict(right_idx=ri, right_pos=rp)
        d["idx0"] = buf[w:w + n0].tolist(); w += n0
        d["pos0"] = buf[w:w + n0].tolist(); w += n0
        d["idx1"] = buf[w:w + n1].tolist(); w += n1
        d["pos1"] = buf[w:w + n1].tolist(); w += n1
        out.append(d)
    return rc, out, mcl.value


def distance(a, b, max_gap=10):
    d, l = C.c_int(0), C.c_int(0)
    lib().orc_distance(a.encode(), b.encode(), max_gap, C.byref(d), C.byref(l))
    return d.value, l.value


def cons(rows, k):
    buf = C.create_string_buffer(k + 1)
    lib().orc_cons("".join(rows).encode(), len(rows), k, buf)
    return buf.value.decode()


def cluster_lm_mem(lcp_u32, phantom, k=16, min_len=2):
    import numpy as np
    lcp_u32 = np.ascontiguousarray(lcp_u32, dtype=np.uint32)
    n = lcp_u32.size
    cap = n + 2
    starts = np.zeros(cap, dtype=np.uint64)
    lens = np.zeros(cap, dtype=np.uint16)
    closed = C.c_uint64(0)
    nrec = lib().orc_cluster_lm_mem(lcp_u32.ctypes.data, n, phantom, k, min_len,
                                    starts.ctypes.data, lens.ctypes.data, cap, C.byref(closed))
    return starts[:nrec].copy(), lens[:nrec].copy(), closed.value


def phantom_lcp_for_scan(lcp_u32, bwt_u8, k=16, x=1):
    import numpy as np
    lcp_u32 = np.ascontiguousarray(lcp_u32, dtype=np.uint32)
    bwt_u8 = np.ascontiguousarray(bwt_u8, dtype=np.uint8)
    return lib().orc_phantom_lcp_for_scan(lcp_u32.ctypes.data, bwt_u8.ctypes.data, lcp_u32.size, k, x)


# ---- the reference itself (oracle/_ref), when built ----
def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "ebwt2clust")) and os.path.exists(os.path.join(REF_DIR, "clust2snp"))


def ref_ebwt2clust(input_path, args=()):
    r = subprocess.run([os.path.join(REF_DIR, "ebwt2clust"), "-i", input_path, *map(str, args)],
                       capture_output=True, text=True)
    return r.returncode, r.stdout


def ref_clust2snp(input_path, n, args=()):
    r = subprocess.run([os.path.join(REF_DIR, "clust2snp"), "-i", input_path, "-n", str(n), *map(str, args)],
                       capture_output=True, text=True)
    return r.returncode, r.stdout
