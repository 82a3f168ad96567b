/*
 * e2s_hip.h -- C ABI of the MI355X (gfx950) implementation of ebwt2snp's eBWT
 * positional-clustering hot path.
 *
 * The reference (nicolaprezza/ebwt2snp) has no FFI or plugin interface: the hot
 * path is reachable only through the two executables ebwt2clust / clust2snp and
 * their on-disk formats.  The entry points below are therefore cut at the
 * reference's internal seams, each one replacing the function(s) cited:
 *
 *   e2s_upload_* / e2s_attach_soa   egsa_stream::read_el           include.hpp:120-199
 *   e2s_cluster_scan                cluster_lm                     ebwt2clust.cpp:68-139
 *   e2s_stitch / e2s_fetch_clusters append_entry + tail flush      ebwt2clust.cpp:54-63,127-135
 *   e2s_cluster_hist                statistics (histogram pass)    clust2snp.cpp:877-909
 *   e2s_auto_max_clust_length       statistics (auto max length)   clust2snp.cpp:936-946
 *   e2s_find_candidates             find_events + find_variants    clust2snp.cpp:788-857,367-500
 *   e2s_upload_reads[_fasta]        get_reads                      clust2snp.cpp:147-212
 *   e2s_call_variants               extract_variants + distance + to_file (on the GPU)   clust2snp.cpp:505-780, 254-302
 *   e2s_snp_from_candidates         the same on the host           clust2snp.cpp:505-780
 *
 * Conventions: one handle per GPU; the caller is single-threaded per handle; no
 * exceptions cross the boundary; every call returns E2S_OK (0) or an error code
 * (e2s_last_error gives text); the caller owns every host buffer it passes, the
 * library owns device buffers unless attached.  Plain pointers and sizes only.
 * All file-format integers are little-endian.
 */
#ifndef E2S_HIP_H
#define E2S_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2S_OK               0
#define E2S_ERR_ARG          1   /* bad argument / unsupported byte width */
#define E2S_ERR_HIP          2   /* a HIP runtime call failed (no GPU, OOM, ...) */
#define E2S_ERR_STATE        3   /* call out of order (e.g. scan before upload) */
#define E2S_ERR_CAPACITY     4   /* output buffer too small; required count returned */
#define E2S_ERR_INPUT        6   /* malformed input (.clusters not ascending / past EOF) */
#define E2S_ERR_REF_CRASH    7   /* the reference crashes on this input (H6); nothing written */
#define E2S_ERR_IO           8

#define E2S_NONE (~(uint64_t)0)
#define E2S_MAX_CLUST_LEN 150    /* max_clust_length_def, clust2snp.cpp:32; -M is unparseable (H5) */

typedef struct e2s_ctx* e2s_handle;

/* ---- lifetime ---------------------------------------------------------------------------- */
int  e2s_create(int device_id, e2s_handle* out);
int  e2s_destroy(e2s_handle h);
const char* e2s_last_error(e2s_handle h);   /* h may be NULL: this thread's last error of a call without a handle */
const char* e2s_version(void);

/* ---- EGSA shard ---------------------------------------------------------------------------
 * A handle holds one contiguous suffix-array range [global_offset, global_offset+n_entries)
 * of an EGSA with n_total entries, plus halos: halo_left entries before (>= E2S_MAX_CLUST_LEN+2,
 * or up to the array start: a cluster is handled by the shard that holds its END) and halo_right
 * entries after (>= 1, or none at the array end).
 * Source pointers address the FIRST PROVIDED entry, i.e. global index global_offset-halo_left.
 * Byte widths follow -x/-y/-z of the reference (1,2,4,8); 8-byte fields are truncated to
 * 32 bits exactly as t_GSA does (include.hpp:131,140,149).                                     */
typedef struct {
    uint64_t global_offset;
    uint64_t n_entries;
    uint64_t n_total;
    uint32_t halo_left;
    uint32_t halo_right;
    int32_t  lcp_bytes;   /* -x */
    int32_t  da_bytes;    /* -y */
    int32_t  pos_bytes;   /* -z */
    int32_t  reserved;
} e2s_shard_desc;

/* BCR triple (include.hpp:159-188): bwt = <f>.out (1 B/entry), lcp = <f>.out.lcp (x B),
 * pairsa = <f>.out.pairSA ([pos:z][da:y] per entry).  Host pointers; copied to HBM and
 * de-interleaved there. */
int e2s_upload_bcr(e2s_handle h, const e2s_shard_desc* d, const void* bwt, const void* lcp, const void* pairsa);
/* .gesa array-of-structs (include.hpp:126-155): [da:y][pos:z][lcp:x][bwt:1] per entry. */
int e2s_upload_gesa(e2s_handle h, const e2s_shard_desc* d, const void* aos);
/* Struct-of-arrays already in HBM (device pointers, widths as in d, 8-byte widths not allowed);
 * zero-copy: the caller keeps ownership and must keep them alive until detach/destroy.  The arrays must be COMPLETE when
 * this is called: the handle works on a stream of its own (e2s_stream) that is not ordered after the caller's streams --
 * synchronise whatever produced them first (torch.cuda.synchronize(), hipStreamSynchronize, or an event the caller makes
 * e2s_stream wait for). */
int e2s_attach_soa(e2s_handle h, const e2s_shard_desc* d, const void* d_bwt, const void* d_lcp,
                   const void* d_pos, const void* d_da);
/* LCP column only (host pointer, x bytes per entry) -- all ebwt2clust needs: cluster_lm reads and
 * discards the other fields.  tail_bwt = BWT byte of the shard's last owned entry (it feeds the
 * reference's EOF phantom on the last rank; ignored elsewhere). */
int e2s_upload_lcp(e2s_handle h, const e2s_shard_desc* d, const void* lcp, uint8_t tail_bwt);
/* The LCP and the BWT columns (host pointers, x bytes and 1 byte per entry): what the scan WITH clust2snp's pre-filter inside reads
 * (e2s_prefilter_arm + e2s_cluster_scan) -- 2 of the 7 bytes per entry.  The document array and the positions are needed only for the
 * clusters the pre-filter lets through (~0.5 % of the entries): the chunked jobs gather those from the host afterwards. */
int e2s_upload_lcp_bwt(e2s_handle h, const e2s_shard_desc* d, const void* lcp, const void* bwt);
/* The same three uploads with the columns in open FILES -- what egsa_stream reads with ifstream (include.hpp:42-81, :126-188): a file
 * descriptor and the byte offset of the shard's first PROVIDED entry (global_offset - halo_left, times the bytes per entry of that file)
 * for every pointer of the memory form.  The columns are read with pread into pinned buffers by a few threads (E2S_UPLOAD_THREADS) and
 * sent from there: a caller that would map the file and pass the pointer pays a page fault per 4 KB of it inside the copy (6 GB of
 * LCP + BWT out of the page cache: 0.28-0.30 s through a mapping) and again for tearing the mapping down.  E2S_ERR_IO: a file is
 * shorter than the shard needs.  The descriptors are used with pread only (no seek, not closed). */
/* How the uploads out of HOST MEMORY (the pointer forms above, e2s_upload_reads) stage large copies: 0 (default) the runtime's own staging of a
 * copy out of pageable memory -- best for ordinary memory (48 GB/s from eight threads) --, 1: through the handle's pinned buffers, filled by
 * those threads -- for pointers into file MAPPINGS, whose pages fault inside the copy (6 GB of LCP + BWT: 0.25 -> 0.16-0.17 s). */
int e2s_set_upload_staging(e2s_handle h, int through_pinned_buffers);
int e2s_upload_bcr_fd(e2s_handle h, const e2s_shard_desc* d, int bwt_fd, uint64_t bwt_off, int lcp_fd, uint64_t lcp_off, int pairsa_fd, uint64_t pairsa_off);
int e2s_upload_gesa_fd(e2s_handle h, const e2s_shard_desc* d, int fd, uint64_t off);
int e2s_upload_lcp_bwt_fd(e2s_handle h, const e2s_shard_desc* d, int lcp_fd, uint64_t lcp_off, int bwt_fd, uint64_t bwt_off);

/* ---- ebwt2clust: cluster boundaries ------------------------------------------------------- */
/* scan_mode: must be 0 (the segmented single-pass scan: every wave a worker over a contiguous range of 4-KiB tiles; the
 * argument stays in the signature for ABI stability -- the count/scan/emit implementation it used to select is gone). */
int e2s_cluster_scan(e2s_handle h, int k, int min_len, int scan_mode);

/* What one shard tells the others (fixed size; exchanged with ONE all-gather). */
typedef struct {
    uint64_t n_ends;           /* clusters that END in this shard (before the EOF tail is resolved)        */
    uint64_t open_start;       /* start of the cluster still open at the shard's end, if it began in this
                                  shard; E2S_NONE otherwise                                                 */
    uint64_t last_end_start;   /* start of the cluster closed by the shard's last end (E2S_NONE: none, or
                                  it began in an earlier shard)                                             */
    uint32_t begins_inside;    /* the first owned entry continues a cluster opened in an earlier shard       */
    uint32_t open_after;       /* a cluster is open after the last owned entry                               */
    uint32_t pass_through;     /* no start and no end at all, and inside a cluster throughout                */
    uint32_t end_before_last;  /* last rank only: a cluster ends at SA position n_total-2 (EOF phantom rule) */
    uint32_t first_end_low16;  /* ((position of the first end)+1) & 0xFFFF: lets the owner of the end finish
                                  the u16 length once the start is known                                     */
    uint32_t tail_lcp;         /* LCP of the last owned entry                                                */
    uint32_t tail_prev_lcp;    /* LCP of the entry before it                                                 */
    uint32_t tail_bwt;         /* BWT byte of the last owned entry                                           */
    uint64_t reserved[2];
} e2s_boundary;

int e2s_boundary_desc(e2s_handle h, e2s_boundary* out);

/* What rank `rank` must do to its own record list once it has everybody's descriptor. */
typedef struct {
    uint32_t patch_first;      /* 1: this shard's record 0 becomes (first_start, first_len)          */
    uint32_t n_tail;           /* last rank: 0..2 records appended after its n_ends records           */
    uint64_t first_start;
    uint64_t tail_start[2];
    uint16_t first_len;
    uint16_t tail_len[2];
    uint16_t reserved16;
    uint32_t phantom_lcp;      /* last rank: the EOF phantom LCP of the reference (hazard H1)         */
    uint32_t reserved32;
} e2s_stitch_plan_t;
/* Host-only, pure function of the gathered descriptors (all[nranks], rank order = SA order). */
int e2s_stitch_plan(const e2s_boundary* all, int nranks, int rank, uint64_t n_total, int k, int lcp_bytes,
                    e2s_stitch_plan_t* out);
/* e2s_stitch_plan + apply it to the device-resident records of this shard. */
int e2s_stitch(e2s_handle h, const e2s_boundary* all, int nranks, int rank);

/* After e2s_stitch: n_closed = this shard's share of the count ebwt2clust prints
 * (ebwt2clust.cpp:137, before the min_len filter); n_records = records it writes. */
int e2s_cluster_counts(e2s_handle h, uint64_t* n_closed, uint64_t* n_records);
/* The `.clusters` image of this shard -- packed 10-byte {u64 start; u16 length} records (ebwt2clust.cpp:58-59), ascending,
 * append_entry's length filter applied -- built in HBM.  async != 0: on the handle's second stream, returning at once, so that it runs
 * beside the per-cluster pass (the pack is bandwidth bound, that pass latency bound); e2s_clusters_image, e2s_fetch_clusters and
 * e2s_stream_clusters wait for it (and build the image themselves if nobody has). */
int e2s_pack_clusters(e2s_handle h, int async);
int e2s_clusters_image(e2s_handle h, const void** d_image, uint64_t* n_records);
/* The same bytes copied to the host: one DMA where `out` is pinned host memory (hipHostMalloc / hipHostRegister), through a pinned
 * double buffer and a memcpy otherwise. */
int e2s_fetch_clusters(e2s_handle h, void* out, uint64_t cap_records, uint64_t* n_records);
/* The same bytes handed to a sink chunk by chunk (whole records per call, in file order) straight out of pinned staging
 * buffers, the next chunk's copy running meanwhile: what the ebwt2clust executable writes its output file with.  The sink
 * returns 0 to continue; anything else stops the stream (E2S_ERR_IO). */
typedef int (*e2s_byte_sink)(void* user, const void* data, size_t bytes);
int e2s_stream_clusters(e2s_handle h, e2s_byte_sink sink, void* user, uint64_t* n_records);

/* ---- clust2snp: per-cluster pass ---------------------------------------------------------- */
/* Use a .clusters file instead of the device-resident scan result (stand-alone clust2snp).
 * recs: packed 10-byte records whose start lies in this shard, ascending, non-overlapping. */
int e2s_load_clusters(e2s_handle h, const void* recs, uint64_t n_records);

/* Partial statistics of this shard's records (clust2snp.cpp:889-909 without the H4 double count). */
typedef struct {
    uint64_t freq[E2S_MAX_CLUST_LEN + 1];
    uint64_t n_records;
    uint64_t n_bases;
    uint64_t last_len;        /* length of this shard's last record (valid if n_records > 0) */
} e2s_hist;
int e2s_cluster_hist(e2s_handle h, e2s_hist* out);

/* The same statistics as the scan left them, available right after e2s_cluster_scan (the scan computes them in
 * the same phase, no further GPU work): together with e2s_boundary this is everything a rank has to publish, so
 * descriptor and statistics travel in ONE all-gather.  `hist` counts the shard's own n_ends records with the
 * lengths as stored before e2s_stitch (record 0 provisional if the shard begins inside a cluster). */
typedef struct {
    e2s_hist hist;
    uint32_t last_is_first;   /* hist.last_len was taken from record 0                                 */
    uint32_t last_none;       /* no record reaches min_len                                              */
    int32_t  min_len;         /* the -m the scan ran with                                               */
    uint32_t reserved;
} e2s_scan_stats;
int e2s_scan_stats_desc(e2s_handle h, e2s_scan_stats* out);
/* Host-only, pure: rank's statistics once its stitch plan (e2s_stitch_plan) is applied -- what e2s_cluster_hist
 * returns after e2s_stitch.  b = the rank's own descriptor. */
int e2s_hist_after_stitch(const e2s_scan_stats* raw, const e2s_boundary* b, const e2s_stitch_plan_t* plan, e2s_hist* out);
/* Host-only, pure: the whole statistics step of N ranks in one call -- every rank's stitch plan, its statistics after
 * the stitch, and e2s_auto_max_clust_length over them (all[r] / stats[r] as gathered, rank order = SA order). */
int e2s_auto_max_clust_length_ranks(const e2s_boundary* all, const e2s_scan_stats* stats, int nranks, uint64_t n_total, int k,
                                    int lcp_bytes, int mcov_out, double pval, int* max_clust_length, uint64_t* n_clust,
                                    uint64_t* n_bases);

/* Host-only: merge per-rank partials in rank order, apply the reference's double count of the
 * last record (H4) and its auto-detection loop incl. the -Ofast arithmetic (H19).
 * Returns E2S_ERR_REF_CRASH where the reference divides by zero (no record / no length<=150). */
int e2s_auto_max_clust_length(const e2s_hist* parts, int nparts, int mcov_out, double pval,
                              int* max_clust_length, uint64_t* n_clust, uint64_t* n_bases);

typedef struct {
    uint64_t nr_reads1;        /* -n as the reference stores it: (uint64_t)atoi(arg)            */
    int32_t  k_left;           /* -L (31) */
    int32_t  k_right;          /* -R (30) */
    int32_t  mcov_out;         /* -m (5)  */
    int32_t  consensus_reads;  /* -c (20) */
    int32_t  max_clust_length; /* from e2s_auto_max_clust_length */
    int32_t  min_len;          /* ebwt2clust -m when records come from e2s_cluster_scan; 0 for a .clusters file */
} e2s_c2s_params;

/* Fixed-size candidate (clust2snp.cpp:101-117).  Lists keep SA order, at most `cap` entries. */
typedef struct {
    uint64_t record_index;     /* global index of the cluster record it came from */
    uint32_t pair_index;       /* position in the (c0,c1) loop order of clust2snp.cpp:431-435 */
    uint32_t n0, n1;
    uint32_t right_idx, right_pos;
    uint32_t reserved;
} e2s_cand_header;
/* Optional: start the pre-filter of the per-cluster pass right after e2s_cluster_scan.  It needs only -m (mcov_out) and
 * ebwt2clust's -m (min_len), so it runs while the ranks exchange descriptors and the host derives max_clust_length;
 * e2s_find_candidates with the same values picks its result up (and applies the exact length bound), otherwise it
 * runs the pre-filter itself.  No-op (E2S_OK) where the streaming pass does not apply. */
int e2s_prefilter_begin(e2s_handle h, int mcov_out, int min_len);
/* Earlier still: armed before e2s_cluster_scan, the pre-filter runs INSIDE the scan kernel (it needs the same start / end
 * planes plus the BWT: one pass over the LCP instead of two; 1-byte LCPs, 16-byte aligned arrays -- otherwise, or with
 * E2S_FUSED=0 in the environment, the scan starts it on the handle's second stream beside its own kernels).  What a job
 * that runs ebwt2clust and clust2snp back to back should call; e2s_find_candidates picks the result up as above. */
int e2s_prefilter_arm(e2s_handle h, int mcov_out);
/* out layout: n_cand blocks of [e2s_cand_header][u32 idx0[cap]][u32 pos0[cap]][u32 idx1[cap]][u32 pos1[cap]] */
int e2s_find_candidates(e2s_handle h, const e2s_c2s_params* p, uint64_t record_index_base, uint64_t* n_cand);
int e2s_candidate_stride(e2s_handle h, const e2s_c2s_params* p, uint32_t* list_cap, size_t* stride_bytes);
int e2s_fetch_candidates(e2s_handle h, void* out, uint64_t cap_cand, uint64_t* n_cand);  /* copied out in SA order */
/* The same fixed-stride records without a copy of the caller's (host memory owned by the handle) and a
 * permutation `order` (the i-th candidate is recs[order[i]]; the identity since the packed transfer below;
 * record_index / pair_index of the headers are device-side ordering keys and come back as the SA rank / 0).
 * Valid until the next e2s_find_candidates / e2s_destroy on this handle. */
int e2s_candidates_view(e2s_handle h, const void** recs, const uint32_t** order, uint64_t* n_cand,
                        uint32_t* list_cap, size_t* stride_bytes);
/* Zero-copy, what e2s_find_candidates itself delivers: the candidates in SA order as a packed stream in the
 * handle's pinned host buffer, written there by the GPU.  Candidate i starts at stream + 8 * offsets8[i]
 * (offsets8 has n_cand + 1 entries, the last one = end of the stream):
 *   u32 n0, n1, right_idx, right_pos | u32 idx0[n0] | u32 idx1[n1] | pos0[n0] | pos1[n1] | padding to 8 bytes
 * with positions of *pos_bytes bytes each (1, 2 or 4: the width of the suffix field, little endian). */
int e2s_candidates_packed(e2s_handle h, const void** stream, const uint32_t** offsets8, uint64_t* n_cand, uint64_t* stream_bytes,
                          uint32_t* pos_bytes);

/* ---- host tail: candidates -> KisSNP2 text (extract_variants + distance + to_file) -------- */
typedef struct {
    int32_t k_left, k_right, max_gap, max_err;   /* -L -R -g -e */
} e2s_tail_params;
/* cands: e2s_fetch_candidates layout (all ranks concatenated in rank order); order: NULL, or the
 * permutation of e2s_candidates_view.  fasta_path is the reads file; out_path the .snp to write.
 * Host only (no GPU work). */
int e2s_snp_from_candidates(const void* cands, const uint32_t* order, uint64_t n_cand, uint32_t list_cap,
                            const e2s_tail_params* p, const char* fasta_path, const char* out_path,
                            uint64_t* n_variants, uint64_t* n_events);
/* The same over the packed stream of e2s_candidates_packed. */
int e2s_snp_from_packed(const void* stream, const uint32_t* offsets8, uint64_t n_cand, uint32_t pos_bytes, const e2s_tail_params* p,
                        const char* fasta_path, const char* out_path, uint64_t* n_variants, uint64_t* n_events);

/* ---- the tail on the GPU: reads resident in HBM -------------------------------- */
/* get_reads (clust2snp.cpp:147-212) done once for ALL reads: read r = text[start[r] .. start[r] + len[r]), max_len = the longest.
 * e2s_upload_reads copies host arrays to HBM (reads of one length are gathered into a dense matrix there); e2s_attach_reads takes
 * device pointers (the caller keeps ownership; d_start = d_len = NULL: a dense matrix, read r = d_text[r * max_len .. +max_len));
 * e2s_upload_reads_fasta indexes a FASTA file on the host (worker threads) and uploads it -- E2S_ERR_INPUT for a file whose
 * line structure is irregular (a header after a header, no header first, ...), where the reference's line juggling yields
 * text this index cannot express: keep to e2s_snp_from_packed then.  A file of the plain shape (header, ONE line of DNA, header, ...)
 * goes up in a single pass: chunks read with pread into pinned buffers, indexed there while their copy is under way; multi-line records
 * are mapped, joined and indexed whole (E2S_READS_MAPPED=1: always).
 * Make the reads resident BEFORE e2s_find_candidates: the candidates then stay in HBM instead of crossing PCIe.
 * Both calls also keep the text a second time as 2-bit planes (a quarter of its bytes; one pass over it, ~1.7 ms for 3 GB) from which the
 * consensus kernel gathers the left contexts of reads made of "ACGT" alone: attached reads must not change while they are attached. */
/* dst uses the reads resident on src (same device) without a copy; src keeps ownership and must outlive dst's use of them */
int e2s_share_reads(e2s_handle dst, e2s_handle src);
int e2s_upload_reads(e2s_handle h, const void* text, uint64_t text_bytes, const uint64_t* start, const uint32_t* len, uint64_t n_reads,
                     uint32_t max_len);
int e2s_attach_reads(e2s_handle h, const void* d_text, const uint64_t* d_start, const uint32_t* d_len, uint64_t n_reads, uint32_t max_len);
/* get_reads touches the reads the candidates name and no others (clust2snp.cpp:519-528 -> :147-212).  For a handle whose candidates were
 * delivered to the HOST (no reads resident during e2s_find_candidates): exactly those reads, from the host-side index (arrays as for
 * e2s_upload_reads), become resident as a compact table, the candidates follow them into HBM with their read ranks rewritten, and
 * e2s_call_variants then runs as with all reads resident; e2s_candidates_packed keeps handing out the ORIGINAL ranks (the host tail's).
 * What a rank of a multi-GPU job uploads instead of every read.  *n_gathered: reads made resident; *bytes_uploaded: text + index + candidates. */
int e2s_upload_reads_of_candidates(e2s_handle h, const void* text, uint64_t text_bytes, const uint64_t* start, const uint32_t* len, uint64_t n_reads,
                                   uint32_t max_len, uint64_t* n_gathered, uint64_t* bytes_uploaded);
int e2s_upload_reads_fasta(e2s_handle h, const char* fasta_path, uint64_t* n_reads);
/* Host only: the same index built ONCE for several handles to upload from (e2s_upload_reads with the arrays handed out here) -- what a
 * driver of several GPUs does instead of parsing the file once per GPU.  E2S_ERR_INPUT for an irregular file (see above), E2S_ERR_IO. */
typedef struct e2s_fasta_index e2s_fasta_index;
int e2s_fasta_index_open(const char* fasta_path, e2s_fasta_index** out, const void** text, uint64_t* text_bytes, const uint64_t** start,
                         const uint32_t** len, uint64_t* n_reads, uint32_t* max_len);
void e2s_fasta_index_close(e2s_fasta_index* ix);
/* extract_variants + distance + to_file (clust2snp.cpp:505-780) over the candidates of the last e2s_find_candidates, on the
 * GPU: one wave per candidate gathers the left contexts, builds the two consensus strings (cons' tie-break, include.hpp:334-371),
 * counts the supports, runs distance() and the KisSNP2 text of every event is written into an image in HBM, ids counting
 * from first_id (1 for a whole job; a later rank / chunk continues where the previous one stopped).  n_variants = candidates
 * with both supports > 0, n_events = records of the `.snp`, snp_bytes = size of the image.
 * E2S_ERR_REF_CRASH where the reference crashes (no candidate; a right context beyond its read); E2S_ERR_STATE where only the
 * host tail reproduces the reference (-L outside 1..64, a read rank past the last read): use e2s_snp_from_packed then. */
int e2s_call_variants(e2s_handle h, const e2s_tail_params* p, uint64_t first_id, uint64_t* n_variants, uint64_t* n_events, uint64_t* snp_bytes);
/* The same events with ids counting from another first_id (several ranks: every rank learns how many events the ranks before it
 * hold only after its own e2s_call_variants); consensus and distance results are kept, only the text is written again.
 * e2s_call_variants with first_id 0 defers the text altogether (snp_bytes comes back 0; fetch / stream refuse): with several ranks the
 * text is then written ONCE, by e2s_renumber_snp after the exchange of the event counts. */
int e2s_renumber_snp(e2s_handle h, uint64_t first_id, uint64_t* snp_bytes);
/* the image, copied out whole or handed to a sink chunk by chunk (as e2s_stream_clusters) */
int e2s_fetch_snp(e2s_handle h, void* out, uint64_t cap_bytes, uint64_t* bytes);
int e2s_stream_snp(e2s_handle h, e2s_byte_sink sink, void* user);
/* What the clust2snp executable calls after e2s_find_candidates: the GPU tail when the reads are resident and it applies,
 * otherwise the host tail over the packed candidates; writes out_path either way (nothing where the reference crashes). */
int e2s_write_snp(e2s_handle h, const e2s_tail_params* p, const char* fasta_path, const char* out_path, uint64_t* n_variants, uint64_t* n_events);

/* ---- whole jobs over an EGSA that comes from the host in chunks (EGSAs larger than HBM) --------------------------
 * The array goes through ONE GPU in chunks of chunk_entries SA positions (left halo 152, right halo 1 -- or max cluster length + 1
 * where records come from a file), chunk i + 1 uploading on a second handle while chunk i is processed.  BCR files of the default
 * widths stream only LCP + BWT (2 of the 7 bytes per entry: what the scan with clust2snp's pre-filter inside reads); the document
 * array and the positions are read for the ~0.5 % of the entries that lie in clusters the pre-filter lets through.  Every chunk is a rank of
 * the boundary stitch (e2s_stitch_plan / e2s_hist_after_stitch), so the outputs are byte for byte those of one pass over the whole
 * array: no new semantics.  The reference streams the same files with O(cluster) memory (include.hpp:120-199, ebwt2clust.cpp:90-122). */
typedef struct {
    /* Hands out host memory holding the entries [lo, lo + n) of the EGSA (global SA positions).  BCR layout (gesa == 0): *bwt (1 byte per
     * entry), *lcp (x bytes), *pairsa (z + y bytes: [pos][da]); .gesa layout: *aos (y + z + x + 1 bytes per entry: [da][pos][lcp][bwt]).
     * want_all == 0: only the LCP column and *bwt are needed (ebwt2clust reads and discards the rest).  The pointers stay valid until
     * release(user, slot); slots 0 and 1 are in use alternately by the streaming uploads, never the same one twice at a time; slot 2
     * (want_all = 1) is random access to the few entries the per-cluster pass needs from all columns -- the clusters the pre-filter
     * let through, gathered on the host -- possibly while a streaming slot is in use.  Returns 0, or non-zero to abort. */
    int (*get)(void* user, int slot, uint64_t lo, uint64_t n, int want_all, const void** bwt, const void** lcp, const void** pairsa, const void** aos);
    void (*release)(void* user, int slot);                     /* may be NULL */
    void* user;
    int32_t gesa;
    int32_t flags;                                             /* bit 0: the pointers are into file MAPPINGS (their copies then go through the handles' pinned
                                                                  buffers, e2s_set_upload_staging); other bits: 0 */
} e2s_egsa_source;
typedef struct {
    int32_t device;
    int32_t k, min_len;                  /* ebwt2clust -k -m */
    int32_t lcp_bytes, da_bytes, pos_bytes;
    uint64_t n_total;
    uint64_t chunk_entries;              /* 0: a default that fits one MI355X comfortably */
} e2s_chunk_params;
typedef struct {
    uint64_t nr_reads1;                  /* -n */
    int32_t k_left, k_right, mcov_out, consensus_reads, max_gap, max_err;   /* -L -R -m -c -g -e */
    double pval;                         /* -p */
} e2s_c2s_options;
typedef struct {
    uint64_t n_closed, n_records;        /* ebwt2clust's printed count / records of the `.clusters` file */
    int32_t max_clust_length; int32_t n_chunks;
    uint64_t n_clust, n_bases;           /* statistics() incl. the H4 double count */
    uint64_t n_candidates, n_variants, n_events;
    e2s_hist hist;                       /* the merged histogram (before the H4 double count) */
    double seconds_upload, seconds_total;
    uint64_t snp_bytes, n_ends;          /* e2s_resident_job: bytes of this rank's `.snp` text; clusters that END in this shard */
    uint64_t n_flagged;                  /* e2s_resident_job: clusters the pre-filter let through to the exact per-cluster test */
} e2s_job_report;
/* ebwt2clust, chunked: the `.clusters` bytes go to clusters_sink chunk by chunk (file order). */
int e2s_ebwt2clust_chunked(const e2s_chunk_params* p, const e2s_egsa_source* src, e2s_byte_sink clusters_sink, void* sink_user, e2s_job_report* rep);
/* clust2snp, chunked, for records taken from a `.clusters` file (recs: all of them, packed 10-byte records): the EGSA passes through
 * HBM once; the reads (fasta_path) are resident throughout; max_clust_length from e2s_auto_max_clust_length over e2s_hist_of_records.
 * Writes out_path (nothing where the reference crashes: E2S_ERR_REF_CRASH). */
int e2s_clust2snp_chunked(const e2s_chunk_params* p, const e2s_c2s_options* o, int max_clust_length, const e2s_egsa_source* src, const void* recs,
                          uint64_t n_records, const char* fasta_path, const char* out_path, e2s_job_report* rep);
/* both tools in one process: pass A streams the LCP column (scan, `.clusters` bytes to clusters_sink -- may be NULL --, statistics);
 * pass B streams all four columns (scan + pre-filter in one kernel, stitch with the descriptors of pass A, per-cluster pass, tail).
 * The reads: fasta_path, or (d_text, d_start, d_len, n_reads, max_len) already in HBM when fasta_path is NULL (e2s_attach_reads).
 * E2S_ERR_STATE after part of the `.snp` text has reached snp_sink: a later chunk needs the host tail (a read rank past the last read),
 * which works on all candidates at once -- e2s_clust2snp_chunked, which owns its output file, starts over by itself in that case. */
int e2s_ebwt2snp_chunked(const e2s_chunk_params* p, const e2s_c2s_options* o, const e2s_egsa_source* src, const char* fasta_path,
                         const void* d_text, const uint64_t* d_start, const uint32_t* d_len, uint64_t n_reads, uint32_t max_len,
                         e2s_byte_sink clusters_sink, void* clusters_user, e2s_byte_sink snp_sink, void* snp_user, e2s_job_report* rep);
/* The per-cluster pass of ONE shard that holds only LCP + BWT (e2s_upload_lcp_bwt -> e2s_prefilter_arm -> e2s_cluster_scan -> e2s_stitch on
 * h_scan): the clusters its pre-filter let through -- and the records the EOF rule adds (plan = this rank's e2s_stitch_plan) -- are gathered
 * from all four columns of `src` (slot 2: random access) on host threads and passed through find_variants (clust2snp.cpp:367-500) on
 * h_pass, where the candidates then live (e2s_call_variants / e2s_candidates_packed on h_pass; make the reads resident there first).
 * What the chunked jobs do per chunk, for drivers that hold their shards themselves (bin/ebwt2snp_mgpu: 2 of the 7 bytes per entry
 * uploaded per rank).  p: n_total, widths, min_len.  *n_gathered (may be NULL): entries gathered.  E2S_ERR_STATE: not applicable to this
 * shard (.gesa source, other widths, a window too dense for the pre-filter, a cluster of 65536+ entries): upload all four columns and
 * use e2s_find_candidates. */
int e2s_find_candidates_gathered(e2s_handle h_scan, e2s_handle h_pass, const e2s_chunk_params* p, const e2s_c2s_options* o, const e2s_egsa_source* src,
                                 const e2s_stitch_plan_t* plan, int max_clust_length, uint64_t* n_cand, uint64_t* n_gathered);
/* host-only: the partial statistics of packed 10-byte records (clust2snp.cpp:889-909 without the H4 double count), worker threads */
int e2s_hist_of_records(const void* recs, uint64_t n_records, e2s_hist* out);

/* ---- both tools back to back on a RESIDENT shard, without a host round trip between the phases ------------------------------
 * cluster_lm (ebwt2clust.cpp:68-139) with clust2snp's pre-filter in the same kernel -> [exchange of the ranks' descriptors + statistics]
 * -> the boundary stitch and statistics() (clust2snp.cpp:877-966) -> find_events / find_variants (clust2snp.cpp:788-857, 367-500) ->
 * extract_variants + distance + to_file on the GPU (clust2snp.cpp:505-780) with ids sequential over the ranks -- ALL ENQUEUED AT ONCE:
 * the stitch plans, the merged statistics and max_clust_length are worked out on the device by the very functions behind
 * e2s_stitch_plan / e2s_hist_after_stitch / e2s_auto_max_clust_length (csrc/e2s_plan.h: one source, host and device), every kernel
 * takes its counts from device memory, and the host waits once, at the end.  What e2s_cluster_scan + e2s_stitch +
 * e2s_find_candidates + e2s_call_variants (+ e2s_pack_clusters) leave behind is left behind here too: images, counts, timings.
 *
 * The two exchanges are the caller's: an all-gather of bytes_per_rank bytes from d_send into d_recv (rank order), ENQUEUED on the
 * given HIP stream (hipStream_t as void*) -- ncclAllGather, or anything that orders itself after the stream's work.  world == 1:
 * never called.  The first carries E2S_JOB_BLOB_BYTES per rank (e2s_boundary + e2s_scan_stats), the second 16 (events, ok).
 * Every rank of a job must call e2s_resident_job with the same options.
 * Returns E2S_ERR_STATE, BEFORE anything is enqueued, where this form does not apply (the reads not resident, -L outside 1..64, wide or
 * unaligned columns: conditions the ranks of a job share): use the stepwise calls.  E2S_ERR_CAPACITY after the whole step has run --
 * on EVERY rank of the job, they tell each other in the second exchange -- where a buffer of some rank was too small or its input
 * needs a path only the stepwise calls have: run the step again through them (they size the buffers; later jobs fit).             */
#define E2S_JOB_BLOB_BYTES 1344
typedef int (*e2s_allgather_fn)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);
typedef struct {
    int32_t k, min_len;                  /* ebwt2clust -k -m */
    int32_t rank, world;                 /* this shard's place in SA order; world <= 64 */
    e2s_c2s_options o;                   /* clust2snp's options */
    int32_t want_image;                  /* != 0: the `.clusters` image is laid out on the second stream beside the rest (e2s_clusters_image waits for it) */
    int32_t probe_only;                  /* != 0: only answer whether this form applies (E2S_OK / E2S_ERR_STATE), enqueue nothing -- for drivers whose
                                            ranks agree on the form before any of them enters an exchange.  The probe puts the step's buffers in place:
                                            results of an earlier scan / pass on the handle are dropped (as by a new scan) */
} e2s_resident_params;
int e2s_resident_job(e2s_handle h, const e2s_resident_params* p, e2s_allgather_fn allgather, void* user, e2s_job_report* rep);

/* ---- measurement --------------------------------------------------------------------------- */
typedef struct {
    float scan_ms;        /* e2s_cluster_scan kernels incl. the statistics (HIP events on the handle's stream) */
    float scan_main_ms;   /* the dominant scan kernel alone */
    float hist_ms;        /* the scan's follow-up kernels (segment fix-up + statistics); e2s_cluster_hist: its own kernel */
    float cand_ms;        /* e2s_find_candidates kernels */
    float cand_main_ms;   /* the dominant per-cluster kernel alone */
    float upload_ms;
    float tail_ms;        /* e2s_call_variants kernels */
    uint64_t scan_bytes_read, scan_bytes_written;   /* algorithmic bytes of the last call */
    uint64_t cand_bytes_read, cand_bytes_written;
    float enqueue_ms;     /* e2s_resident_job: host time of the step's whole enqueue (every launch of both tools, the exchanges' callbacks), wall clock --
                             it runs ahead of the GPU: the step's length is the kernels', not the launches' (always measured, no event involved) */
    float reserved;
} e2s_timing;
int e2s_last_timing(e2s_handle h, e2s_timing* out);
/* Kernel timing is OPT-IN: an event recorded between two dependent kernels costs stream time (5-10 us per boundary on gfx950
 * against ~2 us for a bare boundary), so by default the library records none.  level 0: none (e2s_last_timing reports zeros
 * for the kernel times); 1: two events around the dominant scan kernel only (scan_main_ms); 2: every phase (all fields).
 * A handle starts at the level of the environment variable E2S_TIMING (default 0). */
int e2s_set_timing(e2s_handle h, int level);
/* The HIP stream the handle launches on (hipStream_t as void*), for callers that bracket calls
 * with their own events. */
int e2s_stream(e2s_handle h, void** stream);
int e2s_synchronize(e2s_handle h);

#ifdef __cplusplus
}
#endif
#endif
