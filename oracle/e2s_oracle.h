/*
 * oracle/e2s_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single-threaded) of the ebwt2snp hot path
 * (ebwt2clust + clust2snp of nicolaprezza/ebwt2snp).  It exists to CHECK the
 * HIP path; it is never linked, imported or executed by the product
 * (ebwt2snp_amd/, bin/ebwt2clust, bin/clust2snp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED.  The restatement is checked byte-for-byte against
 * the reference itself (oracle/_ref/, built by oracle/Makefile from the
 * sources under /root/reference) in tests/golden/make_golden.py, and against
 * the committed outputs of that run in tests/test_oracle_golden.py.
 */
#ifndef E2S_ORACLE_H
#define E2S_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes */
#define ORC_OK               0
#define ORC_HELP             1   /* reference would print help and exit(0) */
#define ORC_MISSING_INDEX    2   /* "Error: missing index files." exit(1)  (include.hpp:74-75) */
#define ORC_MISSING_CLUSTERS 3   /* clust2snp.cpp:1063-1068 */
#define ORC_REF_SIGFPE       4   /* reference divides by zero here (clust2snp.cpp:849,921,200) */
#define ORC_REF_SIGSEGV      5   /* reference indexes an empty vector (clust2snp.cpp:531) */
#define ORC_REF_UB_ABORT     6   /* EOF phantom read behind a reallocating push_back: reference aborts */
#define ORC_BAD_WIDTH        7   /* byte size not in {1,2,4,8}: reference leaves the field uninitialised */
#define ORC_IO               8

/* ebwt2clust.cpp:141-198 (main) + :68-139 (cluster_lm).  0 for an option = "use default". */
typedef struct {
    int k;        /* -k, default 16 */
    int min_len;  /* -m, default 2  */
    int lcp, da, pos; /* -x -y -z, defaults 1 4 1 */
} orc_e2c_opts;

/* Writes <input>.clusters.  *n_clust_out = the count the reference prints
 * ("Done. N clusters saved", ebwt2clust.cpp:137; counted BEFORE the min_len filter). */
int orc_ebwt2clust(const char* input, const orc_e2c_opts* o, uint64_t* n_clust_out);

/* clust2snp.cpp:969-1085 (main).  0 = "use default" for every field, as in the reference. */
typedef struct {
    int64_t nr_reads1;      /* -n (atoi -> uint64_t) */
    int k_left;             /* -L 31 */
    int k_right;            /* -R 30 */
    int max_gap;            /* -g 10 */
    int max_snvs;           /* -v 3 (parsed, never used: to_file tests max_snvs_def, clust2snp.cpp:648) */
    int consensus_reads;    /* -c 20 */
    int max_err;            /* -e 2 */
    int mcov_out;           /* -m 5 */
    double pval;            /* -p 0.99 */
    int lcp, da, pos;       /* -x -y -z */
} orc_c2s_opts;

typedef struct {
    uint64_t n_clust;           /* statistics(): records + 1 (last record counted twice) */
    uint64_t n_bases;
    int      max_clust_length;  /* auto-detected, clust2snp.cpp:938-946 */
    uint64_t n_candidates;      /* candidate_variants.size() */
    uint64_t n_variants;        /* output_variants.size() (supp0>0 && supp1>0) */
    uint64_t n_events;          /* records written to .snp (distance <= 3) */
} orc_c2s_stats;

/* Reads <input>.clusters, the EGSA of <input>, the FASTA <input>; writes the .snp
 * file at the reference's path rule (clust2snp.cpp:1074-1076) unless out_path != NULL. */
int orc_clust2snp(const char* input, const orc_c2s_opts* o, const char* out_path,
                  orc_c2s_stats* stats, int verbose);

/* Candidate dump for parity checks of the device stage (find_events + find_variants,
 * clust2snp.cpp:788-857, 367-500).  Flat u64 stream per candidate:
 *   [n0, n1, right_idx, right_pos, idx0[n0].., pos0[n0].., idx1[n1].., pos1[n1]..]
 * Returns number of u64 written (or needed, if cap too small) in *n_words. */
int orc_candidates(const char* input, const orc_c2s_opts* o, uint64_t* out, size_t cap,
                   size_t* n_words, uint64_t* n_cand, int* max_clust_length);

/* small pure functions for unit tests */
/* clust2snp.cpp:254-302; returns D in *d, L in *l */
void orc_distance(const char* a, const char* b, int max_gap, int* d, int* l);
/* clust2snp.cpp:218-232 */
int  orc_dH(const char* a, size_t la, const char* b, size_t lb);
/* include.hpp:334-371: consensus of n strings of length k (row-major), order-dependent tie-break */
void orc_cons(const char* rows, int n, int k, char* out);

/* ebwt2clust.cpp:68-139 on an in-memory LCP array with an explicit EOF phantom
 * (used to cross-check the stencil/sharded formulations).  recs: (start,len) pairs
 * AFTER the min_len filter; returns number of records, *n_closed = printed count. */
size_t orc_cluster_lm_mem(const uint32_t* lcp, uint64_t n, uint32_t phantom, int k, int min_len,
                          uint64_t* starts, uint16_t* lens, size_t cap, uint64_t* n_closed);

/* The EOF phantom LCP the g++ 11.4 -Ofast reference build produces (see e2s_oracle.c, "slot model"). */
uint32_t orc_phantom_lcp_for_scan(const uint32_t* lcp, const uint8_t* bwt, uint64_t n, int k, int x);

#ifdef __cplusplus
}
#endif
#endif
