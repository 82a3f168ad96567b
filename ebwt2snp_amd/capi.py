"""ctypes binding of libe2s_hip.so (include/e2s_hip.h).  Plumbing only: every compute call goes to
the HIP library; there is no Python or CPU fallback -- without the library or without a GPU the
calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libe2s_hip.so")
if os.environ.get("E2S_LIB_PATH"):            # development aid: an ablation build of the library (tools/kbench.py)
    LIB_PATH = os.environ["E2S_LIB_PATH"]

E2S_NONE = (1 << 64) - 1
MAX_CLUST_LEN = 150

ERR = {0: "OK", 1: "ERR_ARG", 2: "ERR_HIP", 3: "ERR_STATE", 4: "ERR_CAPACITY", 5: "ERR_5_UNUSED",
       6: "ERR_INPUT", 7: "ERR_REF_CRASH", 8: "ERR_IO"}


class E2SError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR.get(code, code), msg))
        self.code = code


class ShardDesc(C.Structure):
    _fields_ = [("global_offset", C.c_uint64), ("n_entries", C.c_uint64), ("n_total", C.c_uint64),
                ("halo_left", C.c_uint32), ("halo_right", C.c_uint32),
                ("lcp_bytes", C.c_int32), ("da_bytes", C.c_int32), ("pos_bytes", C.c_int32), ("reserved", C.c_int32)]


class Boundary(C.Structure):
    _fields_ = [("n_ends", C.c_uint64), ("open_start", C.c_uint64), ("last_end_start", C.c_uint64),
                ("begins_inside", C.c_uint32), ("open_after", C.c_uint32), ("pass_through", C.c_uint32),
                ("end_before_last", C.c_uint32), ("first_end_low16", C.c_uint32), ("tail_lcp", C.c_uint32),
                ("tail_prev_lcp", C.c_uint32), ("tail_bwt", C.c_uint32), ("reserved", C.c_uint64 * 2)]


class StitchPlan(C.Structure):
    _fields_ = [("patch_first", C.c_uint32), ("n_tail", C.c_uint32), ("first_start", C.c_uint64),
                ("tail_start", C.c_uint64 * 2), ("first_len", C.c_uint16), ("tail_len", C.c_uint16 * 2),
                ("reserved16", C.c_uint16), ("phantom_lcp", C.c_uint32), ("reserved32", C.c_uint32)]


class Hist(C.Structure):
    _fields_ = [("freq", C.c_uint64 * (MAX_CLUST_LEN + 1)), ("n_records", C.c_uint64), ("n_bases", C.c_uint64),
                ("last_len", C.c_uint64)]


class ScanStats(C.Structure):
    _fields_ = [("hist", Hist), ("last_is_first", C.c_uint32), ("last_none", C.c_uint32), ("min_len", C.c_int32),
                ("reserved", C.c_uint32)]


class C2SParams(C.Structure):
    _fields_ = [("nr_reads1", C.c_uint64), ("k_left", C.c_int32), ("k_right", C.c_int32), ("mcov_out", C.c_int32),
                ("consensus_reads", C.c_int32), ("max_clust_length", C.c_int32), ("min_len", C.c_int32)]


class CandHeader(C.Structure):
    _fields_ = [("record_index", C.c_uint64), ("pair_index", C.c_uint32), ("n0", C.c_uint32), ("n1", C.c_uint32),
                ("right_idx", C.c_uint32), ("right_pos", C.c_uint32), ("reserved", C.c_uint32)]


class TailParams(C.Structure):
    _fields_ = [("k_left", C.c_int32), ("k_right", C.c_int32), ("max_gap", C.c_int32), ("max_err", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("scan_ms", C.c_float), ("scan_main_ms", C.c_float), ("hist_ms", C.c_float), ("cand_ms", C.c_float),
                ("cand_main_ms", C.c_float), ("upload_ms", C.c_float), ("tail_ms", C.c_float),
                ("scan_bytes_read", C.c_uint64), ("scan_bytes_written", C.c_uint64),
                ("cand_bytes_read", C.c_uint64), ("cand_bytes_written", C.c_uint64), ("enqueue_ms", C.c_float), ("reserved", C.c_float)]


# every symbol include/e2s_hip.h declares (checked by tests/test_capi_symbols.py)
BYTE_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)     # e2s_byte_sink
SYMBOLS = ["e2s_create", "e2s_destroy", "e2s_last_error", "e2s_version", "e2s_upload_bcr", "e2s_upload_gesa",
           "e2s_attach_soa", "e2s_upload_lcp", "e2s_upload_lcp_bwt", "e2s_upload_bcr_fd", "e2s_upload_gesa_fd", "e2s_upload_lcp_bwt_fd", "e2s_set_upload_staging", "e2s_cluster_scan", "e2s_boundary_desc", "e2s_stitch_plan", "e2s_stitch",
           "e2s_cluster_counts", "e2s_fetch_clusters", "e2s_stream_clusters", "e2s_load_clusters", "e2s_cluster_hist", "e2s_scan_stats_desc",
           "e2s_hist_after_stitch", "e2s_auto_max_clust_length_ranks",
           "e2s_auto_max_clust_length", "e2s_prefilter_begin", "e2s_prefilter_arm", "e2s_find_candidates", "e2s_candidate_stride", "e2s_fetch_candidates", "e2s_candidates_view",
           "e2s_candidates_packed", "e2s_snp_from_candidates", "e2s_snp_from_packed", "e2s_last_timing", "e2s_set_timing", "e2s_stream", "e2s_synchronize",
           "e2s_pack_clusters", "e2s_clusters_image", "e2s_upload_reads", "e2s_upload_reads_of_candidates", "e2s_share_reads", "e2s_attach_reads", "e2s_upload_reads_fasta", "e2s_call_variants", "e2s_renumber_snp", "e2s_fetch_snp", "e2s_stream_snp", "e2s_write_snp",
           "e2s_ebwt2clust_chunked", "e2s_clust2snp_chunked", "e2s_ebwt2snp_chunked", "e2s_find_candidates_gathered", "e2s_hist_of_records", "e2s_resident_job", "e2s_fasta_index_open", "e2s_fasta_index_close"]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP extension not built: %s missing (run `make lib` or __graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.e2s_create.argtypes = [C.c_int, C.POINTER(H)]
    L.e2s_destroy.argtypes = [H]
    L.e2s_last_error.argtypes = [H]
    L.e2s_last_error.restype = C.c_char_p
    L.e2s_version.restype = C.c_char_p
    L.e2s_upload_bcr.argtypes = [H, C.POINTER(ShardDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.e2s_upload_gesa.argtypes = [H, C.POINTER(ShardDesc), C.c_void_p]
    if hasattr(L, "e2s_upload_bcr_fd"):                          # (absent from older builds: tools/ab_libs.sh loads those too)
        L.e2s_upload_bcr_fd.argtypes = [H, C.POINTER(ShardDesc), C.c_int, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_uint64]
        L.e2s_upload_gesa_fd.argtypes = [H, C.POINTER(ShardDesc), C.c_int, C.c_uint64]
        L.e2s_upload_lcp_bwt_fd.argtypes = [H, C.POINTER(ShardDesc), C.c_int, C.c_uint64, C.c_int, C.c_uint64]
    if hasattr(L, "e2s_set_upload_staging"):
        L.e2s_set_upload_staging.argtypes = [H, C.c_int]
    L.e2s_upload_lcp_bwt.argtypes = [H, C.POINTER(ShardDesc), C.c_void_p, C.c_void_p]
    L.e2s_attach_soa.argtypes = [H, C.POINTER(ShardDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.e2s_upload_lcp.argtypes = [H, C.POINTER(ShardDesc), C.c_void_p, C.c_uint8]
    L.e2s_cluster_scan.argtypes = [H, C.c_int, C.c_int, C.c_int]
    L.e2s_boundary_desc.argtypes = [H, C.POINTER(Boundary)]
    L.e2s_stitch.argtypes = [H, C.POINTER(Boundary), C.c_int, C.c_int]
    L.e2s_stitch_plan.argtypes = [C.POINTER(Boundary), C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(StitchPlan)]
    L.e2s_cluster_counts.argtypes = [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_fetch_clusters.argtypes = [H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.e2s_stream_clusters.argtypes = [H, BYTE_SINK, C.c_void_p, C.POINTER(C.c_uint64)]
    L.e2s_load_clusters.argtypes = [H, C.c_void_p, C.c_uint64]
    L.e2s_cluster_hist.argtypes = [H, C.POINTER(Hist)]
    L.e2s_scan_stats_desc.argtypes = [H, C.POINTER(ScanStats)]
    L.e2s_hist_after_stitch.argtypes = [C.POINTER(ScanStats), C.POINTER(Boundary), C.POINTER(StitchPlan), C.POINTER(Hist)]
    L.e2s_auto_max_clust_length_ranks.argtypes = [C.POINTER(Boundary), C.POINTER(ScanStats), C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                                  C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_auto_max_clust_length.argtypes = [C.POINTER(Hist), C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_prefilter_begin.argtypes = [H, C.c_int, C.c_int]
    L.e2s_prefilter_arm.argtypes = [H, C.c_int]
    L.e2s_find_candidates.argtypes = [H, C.POINTER(C2SParams), C.c_uint64, C.POINTER(C.c_uint64)]
    L.e2s_candidate_stride.argtypes = [H, C.POINTER(C2SParams), C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]
    L.e2s_fetch_candidates.argtypes = [H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.e2s_candidates_view.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]
    L.e2s_snp_from_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(TailParams), C.c_char_p,
                                          C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_candidates_packed.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32)]
    L.e2s_snp_from_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(TailParams), C.c_char_p, C.c_char_p,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_pack_clusters.argtypes = [H, C.c_int]
    L.e2s_clusters_image.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.e2s_upload_reads.argtypes = [H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    if hasattr(L, "e2s_upload_reads_of_candidates"):             # (E2S_LIB_PATH may name a build from before round 5: tools/ab_libs.sh)
        L.e2s_upload_reads_of_candidates.argtypes = [H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_attach_reads.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    L.e2s_upload_reads_fasta.argtypes = [H, C.c_char_p, C.POINTER(C.c_uint64)]
    L.e2s_call_variants.argtypes = [H, C.POINTER(TailParams), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_renumber_snp.argtypes = [H, C.c_uint64, C.POINTER(C.c_uint64)]
    L.e2s_fetch_snp.argtypes = [H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.e2s_stream_snp.argtypes = [H, BYTE_SINK, C.c_void_p]
    L.e2s_write_snp.argtypes = [H, C.POINTER(TailParams), C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.e2s_last_timing.argtypes = [H, C.POINTER(Timing)]
    L.e2s_set_timing.argtypes = [H, C.c_int]
    L.e2s_stream.argtypes = [H, C.POINTER(C.c_void_p)]
    L.e2s_synchronize.argtypes = [H]
    _lib = L
    return L


def _ptr(a):
    """host numpy array or integer device pointer -> void*"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


class Handle:
    """One GPU, one EGSA shard.  Thin object wrapper over the C ABI."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.e2s_create(device, C.byref(self.h))
        if rc:
            raise E2SError(rc, self.L.e2s_last_error(None).decode())
        self._keep = []
        self._cbuf = None

    def close(self):
        self._packed_views = None
        if self.h:
            self.L.e2s_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise E2SError(rc, self.L.e2s_last_error(self.h).decode())

    @staticmethod
    def desc(n_total, offset=0, n_entries=None, halo_left=0, halo_right=0, x=1, y=4, z=1):
        return ShardDesc(offset, n_total if n_entries is None else n_entries, n_total, halo_left, halo_right, x, y, z, 0)

    def upload_bcr(self, d, bwt, lcp, pairsa):
        self._ck(self.L.e2s_upload_bcr(self.h, C.byref(d), _ptr(bwt), _ptr(lcp), _ptr(pairsa)))

    def upload_gesa(self, d, aos):
        self._ck(self.L.e2s_upload_gesa(self.h, C.byref(d), _ptr(aos)))

    def upload_bcr_fd(self, d, bwt_fd, bwt_off, lcp_fd, lcp_off, pairsa_fd, pairsa_off):
        """the columns from open files (descriptor, byte offset of the first provided entry): pread into pinned buffers"""
        self._ck(self.L.e2s_upload_bcr_fd(self.h, C.byref(d), bwt_fd, bwt_off, lcp_fd, lcp_off, pairsa_fd, pairsa_off))

    def set_upload_staging(self, through_pinned_buffers):
        self._ck(self.L.e2s_set_upload_staging(self.h, 1 if through_pinned_buffers else 0))

    def upload_gesa_fd(self, d, fd, off):
        self._ck(self.L.e2s_upload_gesa_fd(self.h, C.byref(d), fd, off))

    def upload_lcp_bwt(self, d, lcp, bwt):
        self._ck(self.L.e2s_upload_lcp_bwt(self.h, C.byref(d), _ptr(lcp), _ptr(bwt)))

    def upload_lcp_bwt_fd(self, d, lcp_fd, lcp_off, bwt_fd, bwt_off):
        self._ck(self.L.e2s_upload_lcp_bwt_fd(self.h, C.byref(d), lcp_fd, lcp_off, bwt_fd, bwt_off))

    def upload_lcp(self, d, lcp, tail_bwt):
        self._ck(self.L.e2s_upload_lcp(self.h, C.byref(d), _ptr(lcp), int(tail_bwt)))

    def attach_soa(self, d, d_bwt, d_lcp, d_pos, d_da, keep=()):
        self._keep = list(keep)
        self._ck(self.L.e2s_attach_soa(self.h, C.byref(d), _ptr(d_bwt), _ptr(d_lcp), _ptr(d_pos), _ptr(d_da)))

    def cluster_scan(self, k=16, min_len=2, scan_mode=0):
        self._ck(self.L.e2s_cluster_scan(self.h, k, min_len, scan_mode))

    def boundary(self):
        b = Boundary()
        self._ck(self.L.e2s_boundary_desc(self.h, C.byref(b)))
        return b

    def stitch(self, all_b, rank):
        arr = (Boundary * len(all_b))(*all_b)
        self._ck(self.L.e2s_stitch(self.h, arr, len(all_b), rank))

    def stitch_array(self, arr, n, rank):
        """stitch with the gathered descriptors already in a ctypes array"""
        self._ck(self.L.e2s_stitch(self.h, arr, n, rank))

    def scan_single(self, k=16, min_len=2, scan_mode=0):
        """scan + the (trivial) one-rank stitch"""
        self.cluster_scan(k, min_len, scan_mode)
        self.stitch([self.boundary()], 0)

    def counts(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.e2s_cluster_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fetch_clusters(self):
        n = C.c_uint64(0)
        self._ck(self.L.e2s_fetch_clusters(self.h, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1) * 10, dtype=np.uint8)
        self._ck(self.L.e2s_fetch_clusters(self.h, _ptr(buf), n.value, C.byref(n)))
        return buf[:n.value * 10]

    def fetch_clusters_into(self, out_ptr, cap_records):
        """the records into the caller's host buffer (an address; pinned memory is written by one DMA); returns the number of records"""
        n = C.c_uint64(0)
        self._ck(self.L.e2s_fetch_clusters(self.h, C.c_void_p(out_ptr), cap_records, C.byref(n)))
        return n.value

    def pack_clusters(self, async_=False):
        """build the `.clusters` image in HBM (async_: on the second stream, beside the per-cluster pass)"""
        self._ck(self.L.e2s_pack_clusters(self.h, 1 if async_ else 0))

    def clusters_image(self):
        """(device pointer of the image, records): waits for a pack in flight"""
        p, n = C.c_void_p(), C.c_uint64(0)
        self._ck(self.L.e2s_clusters_image(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def stream_clusters(self, sink):
        """the `.clusters` bytes handed to sink(bytes) chunk by chunk (whole records, file order); sink returns None / 0 to go on,
        anything else stops the stream (E2S_ERR_IO).  Returns the number of records."""
        def cb(_user, data, nbytes):
            r = sink(C.string_at(data, nbytes))
            return int(r) if r else 0
        n = C.c_uint64(0)
        self._ck(self.L.e2s_stream_clusters(self.h, BYTE_SINK(cb), None, C.byref(n)))
        return n.value

    def load_clusters(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        self._ck(self.L.e2s_load_clusters(self.h, _ptr(packed), packed.size // 10))

    def hist(self):
        hs = Hist()
        self._ck(self.L.e2s_cluster_hist(self.h, C.byref(hs)))
        return hs

    def scan_stats(self):
        """statistics of the scan's own records, available right after cluster_scan (no GPU work)"""
        st = ScanStats()
        self._ck(self.L.e2s_scan_stats_desc(self.h, C.byref(st)))
        return st

    def prefilter_begin(self, mcov_out=5, min_len=2):
        """start the pre-filter of the per-cluster pass ahead of find_candidates (runs beside the exchange / statistics)"""
        self._ck(self.L.e2s_prefilter_begin(self.h, mcov_out, min_len))

    def prefilter_arm(self, mcov_out=5):
        """the next cluster_scan starts the pre-filter on the second stream beside its own kernels"""
        self._ck(self.L.e2s_prefilter_arm(self.h, mcov_out))

    def find_candidates(self, params, record_index_base=0):
        n = C.c_uint64(0)
        self._ck(self.L.e2s_find_candidates(self.h, C.byref(params), record_index_base, C.byref(n)))
        return n.value

    def fetch_candidates(self, params):
        cap, stride = C.c_uint32(0), C.c_size_t(0)
        self._ck(self.L.e2s_candidate_stride(self.h, C.byref(params), C.byref(cap), C.byref(stride)))
        n = C.c_uint64(0)
        self._ck(self.L.e2s_fetch_candidates(self.h, None, 0, C.byref(n)))
        need = max(n.value, 1) * stride.value
        if self._cbuf is None or self._cbuf.size < need:
            self._cbuf = np.empty(need + need // 4, dtype=np.uint8)
        self._ck(self.L.e2s_fetch_candidates(self.h, _ptr(self._cbuf), n.value, C.byref(n)))
        return self._cbuf[:n.value * stride.value], n.value, cap.value, stride.value

    def candidates_view(self):
        """zero-copy: (records uint8 view into the library's pinned buffer, order uint32 view, n, list_cap, stride)"""
        recs, order = C.c_void_p(), C.c_void_p()
        n, cap, stride = C.c_uint64(0), C.c_uint32(0), C.c_size_t(0)
        self._ck(self.L.e2s_candidates_view(self.h, C.byref(recs), C.byref(order), C.byref(n), C.byref(cap), C.byref(stride)))
        if not n.value:
            return np.zeros(0, np.uint8), np.zeros(0, np.uint32), 0, cap.value, stride.value
        r = np.ctypeslib.as_array(C.cast(recs, C.POINTER(C.c_uint8)), shape=(n.value * stride.value,))
        o = np.ctypeslib.as_array(C.cast(order, C.POINTER(C.c_uint32)), shape=(n.value,))
        return r, o, n.value, cap.value, stride.value

    def candidates_packed(self):
        """zero-copy: (packed stream uint8 view into the handle's pinned buffer, offsets8 uint32 view with n + 1 entries, n,
        bytes per position in the stream)"""
        st, off = C.c_void_p(), C.c_void_p()
        n, nbytes, pw = C.c_uint64(0), C.c_uint64(0), C.c_uint32(4)
        self._ck(self.L.e2s_candidates_packed(self.h, C.byref(st), C.byref(off), C.byref(n), C.byref(nbytes), C.byref(pw)))
        if not n.value:
            return np.zeros(0, np.uint8), np.zeros(1, np.uint32), 0, pw.value
        # views of the handle's pinned buffer; building them costs ~25 us (np.ctypeslib.as_array creates a ctypes type per shape),
        # and a job repeated on the same handle gets the same addresses and sizes back: keep the last pair
        key = (st.value, nbytes.value, off.value, n.value)
        cached = getattr(self, "_packed_views", None)
        if cached is None or cached[0] != key:
            r = np.frombuffer((C.c_uint8 * nbytes.value).from_address(st.value), dtype=np.uint8)
            o = np.frombuffer((C.c_uint32 * (n.value + 1)).from_address(off.value), dtype=np.uint32)
            cached = self._packed_views = (key, r, o)
        return cached[1], cached[2], n.value, pw.value

    def upload_reads_fasta(self, path):
        n = C.c_uint64(0)
        self._ck(self.L.e2s_upload_reads_fasta(self.h, path.encode(), C.byref(n)))
        return n.value

    def upload_reads(self, text, start, length):
        """reads as host arrays: text uint8, start uint64 [n], length uint32 [n]"""
        text = np.ascontiguousarray(text, dtype=np.uint8); start = np.ascontiguousarray(start, dtype=np.uint64)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        self._ck(self.L.e2s_upload_reads(self.h, _ptr(text), text.size, _ptr(start), _ptr(length), start.size, int(length.max()) if length.size else 0))

    def upload_reads_of_candidates(self, text, start, length):
        """only the reads the handle's candidates name (candidates delivered to the host): (reads gathered, bytes uploaded)"""
        text = np.ascontiguousarray(text, dtype=np.uint8); start = np.ascontiguousarray(start, dtype=np.uint64)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        ng, nb = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.e2s_upload_reads_of_candidates(self.h, _ptr(text), text.size, _ptr(start), _ptr(length), start.size,
                                                       int(length.max()) if length.size else 0, C.byref(ng), C.byref(nb)))
        return ng.value, nb.value

    def attach_reads(self, d_text, d_start, d_len, n_reads, max_len, keep=()):
        self._keep_reads = list(keep)
        self._ck(self.L.e2s_attach_reads(self.h, _ptr(d_text), _ptr(d_start), _ptr(d_len), n_reads, max_len))

    def call_variants(self, k_left=31, k_right=30, max_gap=10, max_err=2, first_id=1):
        """the tail on the GPU: (n_variants, n_events, snp_bytes); raises E2SError (ERR_STATE) where only the host tail applies"""
        p = TailParams(k_left, k_right, max_gap, max_err)
        nv, ne, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.e2s_call_variants(self.h, C.byref(p), first_id, C.byref(nv), C.byref(ne), C.byref(nb)))
        return nv.value, ne.value, nb.value

    def renumber_snp(self, first_id):
        nb = C.c_uint64(0)
        self._ck(self.L.e2s_renumber_snp(self.h, first_id, C.byref(nb)))
        return nb.value

    def fetch_snp(self):
        n = C.c_uint64(0)
        self._ck(self.L.e2s_fetch_snp(self.h, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        self._ck(self.L.e2s_fetch_snp(self.h, _ptr(buf), n.value, C.byref(n)))
        return buf[:n.value]

    def snp_size(self):
        n = C.c_uint64(0)
        self._ck(self.L.e2s_fetch_snp(self.h, None, 0, C.byref(n)))
        return n.value

    def fetch_snp_into(self, out_ptr, cap_bytes):
        n = C.c_uint64(0)
        self._ck(self.L.e2s_fetch_snp(self.h, C.c_void_p(out_ptr), cap_bytes, C.byref(n)))
        return n.value

    def write_snp(self, fasta_path, out_path, k_left=31, k_right=30, max_gap=10, max_err=2):
        p = TailParams(k_left, k_right, max_gap, max_err)
        nv, ne = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.e2s_write_snp(self.h, C.byref(p), fasta_path.encode(), out_path.encode(), C.byref(nv), C.byref(ne)))
        return nv.value, ne.value

    def set_timing(self, level):
        """0: no timing events (default), 1: around the scan kernel only, 2: every phase (each event costs stream time)"""
        self._ck(self.L.e2s_set_timing(self.h, int(level)))

    def timing(self):
        t = Timing()
        self._ck(self.L.e2s_last_timing(self.h, C.byref(t)))
        return t

    def stream(self):
        s = C.c_void_p()
        self._ck(self.L.e2s_stream(self.h, C.byref(s)))
        return s.value

    def synchronize(self):
        self._ck(self.L.e2s_synchronize(self.h))


ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)     # e2s_allgather_fn
JOB_BLOB_BYTES = 1344


class EgsaSource(C.Structure):
    pass


EGSA_GET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
EGSA_RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
EgsaSource._fields_ = [("get", EGSA_GET), ("release", EGSA_RELEASE), ("user", C.c_void_p), ("gesa", C.c_int32), ("flags", C.c_int32)]


class ChunkParams(C.Structure):
    _fields_ = [("device", C.c_int32), ("k", C.c_int32), ("min_len", C.c_int32), ("lcp_bytes", C.c_int32), ("da_bytes", C.c_int32),
                ("pos_bytes", C.c_int32), ("n_total", C.c_uint64), ("chunk_entries", C.c_uint64)]


class C2SOptions(C.Structure):
    _fields_ = [("nr_reads1", C.c_uint64), ("k_left", C.c_int32), ("k_right", C.c_int32), ("mcov_out", C.c_int32),
                ("consensus_reads", C.c_int32), ("max_gap", C.c_int32), ("max_err", C.c_int32), ("pval", C.c_double)]


class JobReport(C.Structure):
    _fields_ = [("n_closed", C.c_uint64), ("n_records", C.c_uint64), ("max_clust_length", C.c_int32), ("n_chunks", C.c_int32),
                ("n_clust", C.c_uint64), ("n_bases", C.c_uint64), ("n_candidates", C.c_uint64), ("n_variants", C.c_uint64),
                ("n_events", C.c_uint64), ("hist", Hist), ("seconds_upload", C.c_double), ("seconds_total", C.c_double), ("snp_bytes", C.c_uint64),
                ("n_ends", C.c_uint64), ("n_flagged", C.c_uint64)]


class ResidentParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("min_len", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("o", C2SOptions), ("want_image", C.c_int32),
                ("probe_only", C.c_int32)]


def resident_job(h, n1, rank=0, world=1, allgather=None, k=16, min_len=2, mcov=5, pval=0.99, k_left=31, k_right=30, max_gap=10, max_err=2,
                 cons_reads=20, want_image=True):
    """e2s_resident_job on Handle h: the whole step enqueued at once, one wait.  allgather(d_send, d_recv, nbytes_per_rank, stream) -> None
    enqueues an all-gather on the given HIP stream (world > 1).  Returns JobReport; raises E2SError (ERR_STATE: not applicable, nothing ran;
    ERR_CAPACITY: rerun through the stepwise calls)."""
    key = (n1, rank, world, allgather, k, min_len, mcov, pval, k_left, k_right, max_gap, max_err, cons_reads, want_image)
    c = getattr(h, "_job_cache", None)
    if c is None or c[0] != key:                                   # (a job repeated on a handle: the ctypes objects are built once)
        L = lib()
        L.e2s_resident_job.argtypes = [C.c_void_p, C.POINTER(ResidentParams), ALLGATHER, C.c_void_p, C.POINTER(JobReport)]
        p = ResidentParams(k, min_len, rank, world, C2SOptions(n1, k_left, k_right, mcov, cons_reads, max_gap, max_err, pval), 1 if want_image else 0, 0)

        def _ag(_user, d_send, d_recv, nbytes, stream):
            try:
                allgather(d_send, d_recv, nbytes, stream)
                return 0
            except Exception:                                      # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLGATHER(_ag) if allgather is not None else ALLGATHER(0)
        c = h._job_cache = (key, L.e2s_resident_job, C.byref(p), cb, JobReport(), p)
    rep = c[4]
    h._ck(c[1](h.h, c[2], c[3], None, C.byref(rep)))
    return rep


def ebwt2snp_chunked(get, n_total, chunk_entries, n1, d_text=None, n_reads=0, read_len=0, clusters_sink=None, snp_sink=None, device=0, k=16, min_len=2,
                     mcov=5, k_left=31, k_right=30, max_gap=10, max_err=2, cons_reads=20, pval=0.99, widths=(1, 4, 1), fasta_path=None, gesa=False):
    """both tools over an EGSA handed out chunk by chunk by get(slot, lo, n, want_all) -> (bwt_ptr, lcp_ptr, pairsa_ptr) (host addresses,
    BCR layout) or -> aos_ptr (gesa=True); the reads as a dense matrix in HBM (d_text) or a FASTA file (fasta_path).
    Sinks: callables (address, nbytes) -> None.  Returns JobReport."""
    L = lib()
    L.e2s_ebwt2snp_chunked.argtypes = [C.POINTER(ChunkParams), C.POINTER(C2SOptions), C.POINTER(EgsaSource), C.c_char_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint64, C.c_uint32, BYTE_SINK, C.c_void_p, BYTE_SINK, C.c_void_p, C.POINTER(JobReport)]

    def _get(_user, slot, lo, n, want_all, pb, pl, pp, pa):
        try:
            if gesa:
                pa[0] = get(slot, lo, n, want_all)
            else:
                b, l, p = get(slot, lo, n, want_all)
                pb[0], pl[0], pp[0] = b, l, p
            return 0
        except Exception:                                          # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    def _sink(fn):
        def cb(_user, data, nbytes):
            if fn is not None:
                fn(data, nbytes)
            return 0
        return BYTE_SINK(cb)

    src = EgsaSource(EGSA_GET(_get), EGSA_RELEASE(0), None, 1 if gesa else 0, 0)
    p = ChunkParams(device, k, min_len, widths[0], widths[1], widths[2], n_total, chunk_entries)
    o = C2SOptions(n1, k_left, k_right, mcov, cons_reads, max_gap, max_err, pval)
    rep = JobReport()
    cs, ss = _sink(clusters_sink), _sink(snp_sink)
    rc = L.e2s_ebwt2snp_chunked(C.byref(p), C.byref(o), C.byref(src), fasta_path.encode() if fasta_path else None,
                                C.c_void_p(int(d_text)) if d_text else None, None, None, n_reads, read_len, cs, None, ss, None, C.byref(rep))
    if rc:
        raise E2SError(rc, "e2s_ebwt2snp_chunked")
    return rep


def stitch_plan(all_b, rank, n_total, k=16, lcp_bytes=1):
    """Host-only: what `rank` must patch/append once it holds every shard's boundary descriptor."""
    arr = (Boundary * len(all_b))(*all_b)
    pl = StitchPlan()
    rc = lib().e2s_stitch_plan(arr, len(all_b), rank, n_total, k, lcp_bytes, C.byref(pl))
    if rc:
        raise E2SError(rc, "e2s_stitch_plan: inconsistent boundary descriptors")
    return pl


def hist_after_stitch(raw, boundary, plan):
    """Host-only: a rank's statistics once its stitch plan is applied (what Handle.hist() returns after stitch)."""
    hs = Hist()
    rc = lib().e2s_hist_after_stitch(C.byref(raw), C.byref(boundary), C.byref(plan), C.byref(hs))
    if rc:
        raise E2SError(rc, "e2s_hist_after_stitch")
    return hs


def auto_max_clust_length_ranks(all_b, all_stats, n_total, k=16, lcp_bytes=1, mcov_out=5, pval=0.99):
    """Host-only: plans + stitched statistics of every rank + the auto-detection, in one call.  all_b / all_stats may be
    ctypes arrays (as gathered) or lists."""
    nr = len(all_b)
    ab = all_b if isinstance(all_b, C.Array) else (Boundary * nr)(*all_b)
    st = all_stats if isinstance(all_stats, C.Array) else (ScanStats * nr)(*all_stats)
    mcl, nc, nb = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
    rc = lib().e2s_auto_max_clust_length_ranks(ab, st, nr, n_total, k, lcp_bytes, mcov_out, pval, C.byref(mcl), C.byref(nc), C.byref(nb))
    if rc:
        raise E2SError(rc, "e2s_auto_max_clust_length_ranks")
    return mcl.value, nc.value, nb.value


def auto_max_clust_length(hists, mcov_out=5, pval=0.99):
    """Host-only: statistics() of clust2snp.cpp:877-966 over per-shard partial histograms."""
    arr = (Hist * len(hists))(*hists)
    mcl, nc, nb = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
    rc = lib().e2s_auto_max_clust_length(arr, len(hists), mcov_out, pval, C.byref(mcl), C.byref(nc), C.byref(nb))
    if rc:
        raise E2SError(rc, "statistics(): the reference crashes on this input")
    return mcl.value, nc.value, nb.value


def snp_from_candidates(cands, n_cand, list_cap, fasta_path, out_path, k_left=31, k_right=30, max_gap=10, max_err=2,
                        order=None):
    p = TailParams(k_left, k_right, max_gap, max_err)
    nv, ne = C.c_uint64(0), C.c_uint64(0)
    cands = np.ascontiguousarray(cands, dtype=np.uint8)
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
    rc = lib().e2s_snp_from_candidates(_ptr(cands), _ptr(order), n_cand, list_cap, C.byref(p), fasta_path.encode(),
                                       out_path.encode(), C.byref(nv), C.byref(ne))
    if rc:
        raise E2SError(rc, "e2s_snp_from_candidates")
    return nv.value, ne.value


def snp_from_packed(stream, offsets8, n_cand, fasta_path, out_path, k_left=31, k_right=30, max_gap=10, max_err=2, pos_bytes=4):
    p = TailParams(k_left, k_right, max_gap, max_err)
    nv, ne = C.c_uint64(0), C.c_uint64(0)
    rc = lib().e2s_snp_from_packed(_ptr(stream), _ptr(offsets8), n_cand, pos_bytes, C.byref(p), fasta_path.encode(), out_path.encode(),
                                   C.byref(nv), C.byref(ne))
    if rc:
        raise E2SError(rc, "e2s_snp_from_packed")
    return nv.value, ne.value


def parse_candidates(buf, n_cand, list_cap, stride):
    """fetch_candidates bytes -> list of dicts (same shape as tests/oracle_lib.candidates)."""
    out = []
    hs = C.sizeof(CandHeader)
    for c in range(n_cand):
        rec = buf[c * stride:(c + 1) * stride]
        h = CandHeader.from_buffer_copy(rec[:hs].tobytes())
        l = rec[hs:].view(np.uint32)
        out.append(dict(record_index=h.record_index, pair_index=h.pair_index, right_idx=h.right_idx, right_pos=h.right_pos,
                        idx0=l[0:h.n0].tolist(), pos0=l[list_cap:list_cap + h.n0].tolist(),
                        idx1=l[2 * list_cap:2 * list_cap + h.n1].tolist(), pos1=l[3 * list_cap:3 * list_cap + h.n1].tolist()))
    return out
