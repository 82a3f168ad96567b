// integration_snippet.cpp -- the C++ snippets of INTEGRATION.md, compiled (never run) against include/e2s_hip.h by
// tests/test_host_logic.py::test_integration_snippets_compile, which also checks that every ```cpp block of the document
// appears here verbatim: the document cannot drift from the header again.
// The declarations in front of each block stand for what the reference's main() has in scope at that seam.
#include <cstdint>
#include <fstream>
#include <iostream>
#include <fcntl.h>
#include <unistd.h>
#include <string>
#include <vector>
#include "e2s_hip.h"      // (the snippets' own #include line, at file scope where the reference would have it)
using namespace std;

int snippet_ebwt2clust(int lcp, int da, int pos, int k, int min_len, uint64_t n, const void* lcp_bytes, uint8_t bwt_last_byte, ofstream& out) {
// [snippet 1: ebwt2clust.cpp]
// after parsing options; lcp/da/pos/k/min_len already defaulted (:175-180)
e2s_handle h; if (e2s_create(0, &h)) { cerr << e2s_last_error(NULL) << endl; return 2; }
e2s_shard_desc d = {0, n, n, 0, 0, lcp, da, pos, 0};           // whole EGSA on one GPU
// BCR triple mapped/read by the caller (X.out.lcp, last byte of X.out):
e2s_upload_lcp(h, &d, lcp_bytes, bwt_last_byte);               // cluster_lm only ever looks at the LCP
e2s_cluster_scan(h, k, min_len, 0);
e2s_boundary b; e2s_boundary_desc(h, &b); e2s_stitch(h, &b, 1, 0);   // one rank: trivial stitch, resolves the EOF rule
uint64_t n_closed, n_rec; e2s_cluster_counts(h, &n_closed, &n_rec);
std::vector<char> recs(n_rec * 10); e2s_fetch_clusters(h, recs.data(), n_rec, &n_rec);
out.write(recs.data(), n_rec * 10);                             // same bytes append_entry (:54-63) would have written
// (large outputs: e2s_stream_clusters(h, sink, user, &n_rec) hands the same bytes to a sink chunk by chunk out of pinned
//  staging buffers -- what this repo's ebwt2clust uses, with write(2) as the sink)
cout << "Done. " << (unsigned)n_closed << " clusters saved to output file." << endl;   // :137
// [end]
    return 0;
}

int snippet_clust2snp(e2s_handle h, e2s_shard_desc d, const void* bwt, const void* lcp, const void* pairsa, const void* clusters_bytes, uint64_t n_records,
                      int mcov_out, double pval, uint64_t nr_reads1, int k_left, int k_right, int consensus_reads, int max_gap, int max_err,
                      string input, string filename_out) {
    int max_clust_length; uint64_t n_clust, n_bases;
// [snippet 2: clust2snp.cpp]
e2s_upload_bcr(h, &d, bwt, lcp, pairsa);        // or e2s_upload_gesa(h, &d, gesa)
e2s_upload_reads_fasta(h, input.c_str(), NULL);  // get_reads once, for all reads: they stay in HBM (E2S_ERR_INPUT: irregular FASTA, the host tail decides)
e2s_load_clusters(h, clusters_bytes, n_records); // the .clusters file as is
e2s_hist hs; e2s_cluster_hist(h, &hs);
e2s_auto_max_clust_length(&hs, 1, mcov_out, pval, &max_clust_length, &n_clust, &n_bases);   // :936-946 incl. the H4 double count
e2s_c2s_params p = {nr_reads1, k_left, k_right, mcov_out, consensus_reads, max_clust_length, 0};
uint64_t n_cand; e2s_find_candidates(h, &p, 0, &n_cand);        // find_events + find_variants (:788-857, :367-500)
e2s_tail_params t = {k_left, k_right, max_gap, max_err};
uint64_t nv, ne;
e2s_write_snp(h, &t, input.c_str(), filename_out.c_str(), &nv, &ne);   // extract_variants + to_file (:505-780): on the GPU, or the host tail where only it applies
// [end]
    {
// [snippet 3: the host tail]
const void* stream; const uint32_t* off8; uint64_t bytes; uint32_t pos_bytes;
e2s_candidates_packed(h, &stream, &off8, &n_cand, &bytes, &pos_bytes);   // host memory: packed, SA order
e2s_snp_from_packed(stream, off8, n_cand, pos_bytes, &t, input.c_str(), filename_out.c_str(), &nv, &ne);   // :505-780 on the host
// [end]
    }
    return (int)(nv + ne);
}

int snippet_chunked(const uint8_t* bwt_bytes, const uint8_t* lcp_bytes, const uint8_t* pairsa_bytes, int lcp, int da, int pos, int k, int min_len, uint64_t n,
                    e2s_byte_sink sink, void* sink_user) {
// [snippet 4: inputs of any size]
struct Mapped { const uint8_t *bwt, *lcp, *pairsa; int x, y, z; };          // the three BCR files, memory-mapped
auto get = [](void* user, int, uint64_t lo, uint64_t, int, const void** bwt, const void** lcp, const void** pairsa, const void**) -> int {
    const Mapped* m = (const Mapped*)user;
    *bwt = m->bwt + lo; *lcp = m->lcp + lo * m->x; *pairsa = m->pairsa + lo * (m->y + m->z);
    return 0;
};
Mapped files = {bwt_bytes, lcp_bytes, pairsa_bytes, lcp, da, pos};
e2s_egsa_source src = {get, NULL, &files, 0, 0};
e2s_chunk_params cp = {0, k, min_len, lcp, da, pos, n, 0};      // device 0, chunk size: the library's default
e2s_job_report rep;
e2s_ebwt2clust_chunked(&cp, &src, sink, sink_user, &rep);       // sink: int (*)(void*, const void*, size_t) -- write(2) in bin/ebwt2clust
cout << "Done. " << (unsigned)rep.n_closed << " clusters saved to output file." << endl;
// [end]
    return 0;
}

int snippet_resident_job(e2s_handle h, int k, int min_len, int rank, int world, uint64_t nr_reads1, int k_left, int k_right, int mcov_out, int consensus_reads,
                         int max_gap, int max_err, double pval) {
// [snippet 5: the whole step as one call per rank]
struct Comm { /* ncclComm_t comm; */ };
auto allgather = [](void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream) -> int {
    (void)user; (void)d_send; (void)d_recv; (void)bytes_per_rank; (void)hip_stream;
    // return ncclAllGather(d_send, d_recv, bytes_per_rank, ncclChar, ((Comm*)user)->comm, (hipStream_t)hip_stream) != ncclSuccess;
    return 0;                                                   // ENQUEUED on the handle's stream: the library never waits for it on the host
};
Comm comm;
e2s_resident_params rp = {k, min_len, rank, world, {nr_reads1, k_left, k_right, mcov_out, consensus_reads, max_gap, max_err, pval}, 1, 0};
e2s_job_report jr;
rp.probe_only = 1;                                                 // PREPARE: does the form apply? + every buffer, stream and event of the step put in place
int can = e2s_resident_job(h, &rp, allgather, &comm, NULL);        // ... the ranks agree on `can` (and on their set-up) BEFORE any of them enqueues:
rp.probe_only = 0;                                                 // from here to the second exchange a rank allocates nothing and cannot drop out for want of memory
int rc = e2s_resident_job(h, &rp, allgather, &comm, &jr);        // shard + reads resident (e2s_upload_bcr, e2s_upload_reads_fasta): ONE enqueue, ONE wait
if (rc == E2S_ERR_STATE || rc == E2S_ERR_CAPACITY) {
    // this form does not apply to the input / a buffer was too small -- on every rank alike (the verdict travels in the second exchange
    // and is evaluated before the reference's crash cases): the stepwise calls of section 3 run the step
}
// (a rank whose enqueue fails on a HIP error after its scan is on the stream still enters both exchanges, with a zeroed share: its peers
//  come back with E2S_ERR_CAPACITY instead of waiting for it; a failing exchange itself -- E2S_ERR_IO -- is for the driver: ncclCommAbort)
// jr.n_ends, jr.n_records, jr.max_clust_length, jr.n_candidates, jr.n_events, jr.snp_bytes; both images in HBM:
const void* d_clusters; uint64_t n_rec; e2s_clusters_image(h, &d_clusters, &n_rec);      // or e2s_stream_clusters / e2s_stream_snp in rank order
// [end]
    return rc + can + (int)n_rec;
}

int snippet_reads_of_candidates(e2s_handle h, const std::string& input, int k_left, int k_right, int max_gap, int max_err, uint64_t first_id) {
    e2s_fasta_index* fix; const void* text; uint64_t text_bytes, n_reads; const uint64_t* start; const uint32_t* len; uint32_t max_len;
    e2s_tail_params t = {k_left, k_right, max_gap, max_err};
    uint64_t nv, ne, snp_bytes;
// [snippet 6: a rank that does not hold every read]
e2s_fasta_index_open(input.c_str(), &fix, &text, &text_bytes, &start, &len, &n_reads, &max_len);   // once per process, shared by the ranks
uint64_t n_named, up_bytes;
e2s_upload_reads_of_candidates(h, text, text_bytes, start, len, n_reads, max_len, &n_named, &up_bytes);   // the named reads as a compact table + the candidates, ranks rewritten
e2s_call_variants(h, &t, first_id, &nv, &ne, &snp_bytes);        // as with all reads resident; e2s_candidates_packed still hands out the original ranks
// [end]
    e2s_fasta_index_close(fix);
    return (int)(n_named + up_bytes + nv + ne + snp_bytes);
}

int snippet_upload_from_files(e2s_handle h, const std::string& input, uint64_t n, uint64_t g0, uint64_t g1, int k, int min_len) {
// [snippet 7: a shard's columns straight from the index files]
int fd_lcp = open((input + ".out.lcp").c_str(), O_RDONLY), fd_bwt = open((input + ".out").c_str(), O_RDONLY);
e2s_shard_desc d = {g0, g1 - g0, n, (uint32_t)(g0 < 152 ? g0 : 152), (uint32_t)(g1 < n ? 1 : 0), 1, 4, 1, 0};   // SA range [g0, g1) of n entries, halos 152 / 1
const uint64_t first = g0 - d.halo_left;                          // the first PROVIDED entry: the byte offset in a 1-byte-per-entry file
e2s_upload_lcp_bwt_fd(h, &d, fd_lcp, first, fd_bwt, first);       // pread into pinned buffers, several threads; no mapping
e2s_prefilter_arm(h, 5);
e2s_cluster_scan(h, k, min_len, 0);
close(fd_lcp); close(fd_bwt);
// [end]
    return 0;
}

