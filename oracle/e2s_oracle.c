/*
 * oracle/e2s_oracle.c -- TEST INFRASTRUCTURE ONLY (see e2s_oracle.h).
 *
 * Plain-C restatement of the reference's sequential algorithm for the hot path.
 * Every function cites the reference file:line it follows (paths relative to
 * the reference repository root).  Nothing here is used by the product.
 *
 * ---- "slot model" of the reference's EOF behaviour (hazard H1) ----
 * egsa_stream::read_el (include.hpp:120-199) reads each field into a block-scope
 * temporary (x8/x16/x32/x64) and never checks the stream state.  In the
 * reference build used as ground truth here (g++ 11.4, CMake Release flags
 * -Ofast, see oracle/Makefile) read_el is an out-of-line function whose
 * temporaries ALL live in one 8-byte stack slot at (%rsp) (objdump of both
 * executables).  A read that fails at EOF leaves the slot as it was, so the
 * "phantom" entry returned past the end is whatever the slot last held:
 *   - normally the bytes written by the last successful read_el (low byte =
 *     last BWT byte, bytes 1.. from the previous wider reads);
 *   - in ebwt2clust, if append_entry (ebwt2clust.cpp:54-63) ran between the
 *     last successful read and the phantom read, the slot holds `start` (its
 *     by-reference spill lands on the same stack address) -- measured;
 *   - in clust2snp, if the vector push_back between the two reads reallocated
 *     (clust2snp.cpp:829), the slot is clobbered by the allocator and the
 *     reference aborts or emits garbage -- reported as ORC_REF_UB_ABORT.
 * All of this is undefined behaviour of the reference; the model reproduces
 * the one build that serves as oracle and is verified against it in
 * tests/golden/make_golden.py (tail fixtures).
 */
#define _GNU_SOURCE
#include "e2s_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* EGSA stream: include.hpp:20-27 (t_GSA), :32-219 (egsa_stream)              */
/* ------------------------------------------------------------------------- */

typedef struct { uint32_t text, suff, lcp; uint8_t bwt; } o_gsa;   /* include.hpp:20-27 */

typedef struct {
    FILE* f;
    int eof, fail;            /* ifstream eofbit / failbit */
} o_ifs;

typedef struct {
    int egsa, bcr;                       /* include.hpp:203-204 */
    int lcp_size, da_size, suff_size;    /* include.hpp:207-209 */
    o_ifs EGSA, LCP, BWT, GSA;           /* include.hpp:212-217 */
    uint8_t slot[8];                     /* the shared stack slot of read_el's temporaries */
} o_stream;

static void ifs_open(o_ifs* s, const char* path) { s->f = fopen(path, "rb"); s->eof = s->fail = 0; }
static void ifs_close(o_ifs* s) { if (s->f) fclose(s->f); s->f = NULL; }

/* istream::read(buf, w): on a failed stream does nothing; a short read sets eof+fail,
 * the bytes that did arrive are stored. */
static void ifs_read(o_ifs* s, uint8_t* buf, size_t w) {
    if (!s->f || s->fail) return;
    size_t got = fread_unlocked(buf, 1, w, s->f);
    if (got < w) { s->eof = 1; s->fail = 1; }
}

static char* path_cat(const char* a, const char* ext) {
    size_t la = strlen(a), le = strlen(ext);
    char* p = (char*)malloc(la + le + 1);
    memcpy(p, a, la); memcpy(p + la, ext, le + 1);
    return p;
}

/* include.hpp:42-81 */
static int stream_open(o_stream* S, const char* input) {
    memset(S, 0, sizeof(*S));
    S->lcp_size = 1; S->da_size = 4; S->suff_size = 1;
    char* p = path_cat(input, ".gesa");
    ifs_open(&S->EGSA, p); free(p);
    if (S->EGSA.f) { S->egsa = 1; return ORC_OK; }
    char* pl = path_cat(input, ".out.lcp");
    char* pb = path_cat(input, ".out");
    char* pg = path_cat(input, ".out.pairSA");
    ifs_open(&S->LCP, pl); ifs_open(&S->BWT, pb); ifs_open(&S->GSA, pg);
    free(pl); free(pb); free(pg);
    if (S->LCP.f && S->BWT.f && S->GSA.f) { S->bcr = 1; return ORC_OK; }
    ifs_close(&S->LCP); ifs_close(&S->BWT); ifs_close(&S->GSA);
    return ORC_MISSING_INDEX;
}
static void stream_close(o_stream* S) {
    ifs_close(&S->EGSA); ifs_close(&S->LCP); ifs_close(&S->BWT); ifs_close(&S->GSA);
}
/* include.hpp:92-107 */
static int stream_eof(const o_stream* S) { return S->egsa ? S->EGSA.eof : S->LCP.eof; }

static int width_ok(int w) { return w == 1 || w == 2 || w == 4 || w == 8; }

/* value of the temporary after the read attempt, truncated to the uint32_t field (include.hpp:128-149) */
static uint32_t slot_field(const o_stream* S, int w) {
    uint64_t v = 0;
    memcpy(&v, S->slot, (size_t)w);     /* little-endian host, as the reference assumes */
    return (uint32_t)v;
}

/* include.hpp:120-199 */
static o_gsa read_el(o_stream* S) {
    o_gsa e;
    if (S->egsa) {                                   /* :124-155  [text:y][suff:z][lcp:x][bwt:1] */
        ifs_read(&S->EGSA, S->slot, (size_t)S->da_size);   e.text = slot_field(S, S->da_size);
        ifs_read(&S->EGSA, S->slot, (size_t)S->suff_size); e.suff = slot_field(S, S->suff_size);
        ifs_read(&S->EGSA, S->slot, (size_t)S->lcp_size);  e.lcp  = slot_field(S, S->lcp_size);
        ifs_read(&S->EGSA, S->slot, 1);                    e.bwt  = S->slot[0];
    } else {                                         /* :157-188  pairSA=[suff:z][text:y], lcp, bwt */
        ifs_read(&S->GSA, S->slot, (size_t)S->suff_size);  e.suff = slot_field(S, S->suff_size);
        ifs_read(&S->GSA, S->slot, (size_t)S->da_size);    e.text = slot_field(S, S->da_size);
        ifs_read(&S->LCP, S->slot, (size_t)S->lcp_size);   e.lcp  = slot_field(S, S->lcp_size);
        ifs_read(&S->BWT, S->slot, 1);                     e.bwt  = S->slot[0];
    }
    return e;
}

/* include.hpp:250-263 */
static unsigned char int_to_base(int i) {
    switch (i) { case 0: return 'A'; case 1: return 'C'; case 2: return 'G'; case 3: return 'T'; }
    return 'A';
}
/* include.hpp:265-279 */
static int base_to_int(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        case 'N': case 'n': return rand() % 4;   /* non-deterministic in the reference (srand(time)) */
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* ebwt2clust                                                                 */
/* ------------------------------------------------------------------------- */

typedef struct { FILE* out; int min_len; } o_sink;

/* ebwt2clust.cpp:54-63.  `length >= min_len` compares uint16_t with int. */
static void append_entry(o_sink* k, uint64_t start, uint16_t length, o_stream* S) {
    if ((int)length >= k->min_len) {
        fwrite(&start, sizeof(uint64_t), 1, k->out);
        fwrite(&length, sizeof(uint16_t), 1, k->out);
    }
    /* slot model: `start` is spilled onto read_el's temporary slot (measured, with or
     * without the write taking place) */
    if (S) memcpy(S->slot, &start, 8);
}

/* ebwt2clust.cpp:68-139 */
static uint64_t cluster_lm(o_stream* EGSA, o_sink* out, int k) {
    const uint64_t null = ~(uint64_t)0;
    uint64_t start = null, i = 0;
    o_gsa e1 = read_el(EGSA);                                   /* :79 */
    o_gsa e2 = read_el(EGSA);                                   /* :80 */
    o_gsa e3;
    /* `e1.lcp >= k`: uint32_t vs int -> k converted to unsigned */
    start = e1.lcp >= (uint32_t)k ? 0 : e2.lcp >= (uint32_t)k ? 1 : null;   /* :83-84 */
    i = 1;
    unsigned int n_clust_out = 0;
    while (!stream_eof(EGSA)) {                                 /* :90 */
        e3 = read_el(EGSA);                                     /* :93 */
        if (start != null &&
            ((e1.lcp > e2.lcp && e2.lcp <= e3.lcp) || e3.lcp < (uint32_t)k)) {   /* :98-102 */
            uint16_t length = (uint16_t)((i - start) + 1);      /* :104 */
            append_entry(out, start, length, EGSA);
            n_clust_out++;
            start = null;
        }
        e1 = e2; e2 = e3; ++i;                                  /* :112-114 */
        if (start == null && e2.lcp >= (uint32_t)k) start = i;  /* :116-120 */
    }
    if (start != null) {                                        /* :127-135 */
        uint16_t length = (uint16_t)((i - start) + 1);
        append_entry(out, start, length, EGSA);
        n_clust_out++;
    }
    return n_clust_out;
}

/* ebwt2clust.cpp:141-198 */
int orc_ebwt2clust(const char* input, const orc_e2c_opts* o, uint64_t* n_clust_out) {
    int lcp = o->lcp == 0 ? 1 : o->lcp;        /* :175-177 */
    int da  = o->da  == 0 ? 4 : o->da;
    int pos = o->pos == 0 ? 1 : o->pos;
    int k   = o->k   == 0 ? 16 : o->k;         /* :179 */
    int min_len = o->min_len == 0 ? 2 : o->min_len;   /* :180 */
    if (!input || !*input) return ORC_HELP;    /* :182 */
    if (!width_ok(lcp) || !width_ok(da) || !width_ok(pos)) return ORC_BAD_WIDTH;
    o_stream S;
    int rc = stream_open(&S, input);           /* :184 */
    if (rc) return rc;
    S.lcp_size = lcp; S.da_size = da; S.suff_size = pos;   /* :185 */
    char* op = path_cat(input, ".clusters");   /* :189-192 */
    o_sink sink; sink.out = fopen(op, "wb"); sink.min_len = min_len;
    free(op);
    if (!sink.out) { stream_close(&S); return ORC_IO; }
    uint64_t nc = cluster_lm(&S, &sink, k);    /* :194 */
    fclose(sink.out);
    stream_close(&S);
    if (n_clust_out) *n_clust_out = nc;
    return ORC_OK;
}

/* in-memory variant of cluster_lm with an explicit phantom (for cross-checking the
 * stencil / sharded formulations against the sequential semantics) */
size_t orc_cluster_lm_mem(const uint32_t* L, uint64_t n, uint32_t phantom, int k, int min_len,
                          uint64_t* starts, uint16_t* lens, size_t cap, uint64_t* n_closed) {
    const uint64_t null = ~(uint64_t)0;
    size_t nrec = 0; uint64_t closed = 0;
#define LX(j) ((j) < n ? L[(j)] : phantom)
#define EMIT(s, l) do { closed++; if ((int)(l) >= min_len) { if (nrec < cap) { starts[nrec] = (s); lens[nrec] = (l); } nrec++; } } while (0)
    uint64_t rd = 0;              /* number of read_el calls so far */
    uint32_t e1 = LX(rd); rd++;
    uint32_t e2 = LX(rd); rd++;
    uint32_t e3;
    uint64_t start = e1 >= (uint32_t)k ? 0 : e2 >= (uint32_t)k ? 1 : null;
    uint64_t i = 1;
    while (rd <= n) {             /* eof is raised by the first read past the end (index n) */
        e3 = LX(rd); rd++;
        if (start != null && ((e1 > e2 && e2 <= e3) || e3 < (uint32_t)k)) {
            uint16_t length = (uint16_t)((i - start) + 1);
            EMIT(start, length);
            start = null;
        }
        e1 = e2; e2 = e3; ++i;
        if (start == null && e2 >= (uint32_t)k) start = i;
    }
    if (start != null) { uint16_t length = (uint16_t)((i - start) + 1); EMIT(start, length); }
#undef LX
#undef EMIT
    if (n_closed) *n_closed = closed;
    return nrec;
}

/* The phantom LCP cluster_lm sees for an n-entry EGSA (slot model, n >= 1):
 * if the cluster closed at j = n-2 (append_entry ran in the last real iteration)
 * the slot holds that cluster's start; otherwise the low byte is the last BWT
 * byte and bytes 1..x-1 are the last LCP's.  Truncated to x bytes, then uint32_t. */
uint32_t orc_phantom_lcp_for_scan(const uint32_t* L, const uint8_t* bwt, uint64_t n, int k, int x) {
    uint64_t mask = x >= 8 ? ~(uint64_t)0 : (((uint64_t)1 << (8 * x)) - 1);
    uint64_t slot = n ? ((((uint64_t)L[n - 1]) & ~(uint64_t)0xFF) | bwt[n - 1]) : 0;
    if (n >= 3) {
        /* replay the sequential scan up to (not including) the phantom read to learn whether
         * the last real iteration appended, and with which start */
        const uint64_t null = ~(uint64_t)0;
        uint32_t e1 = L[0], e2 = L[1], e3;
        uint64_t start = e1 >= (uint32_t)k ? 0 : e2 >= (uint32_t)k ? 1 : null, i = 1;
        for (uint64_t rd = 2; rd < n; ++rd) {
            e3 = L[rd];
            int closed = 0; uint64_t cstart = 0;
            if (start != null && ((e1 > e2 && e2 <= e3) || e3 < (uint32_t)k)) { closed = 1; cstart = start; start = null; }
            e1 = e2; e2 = e3; ++i;
            if (start == null && e2 >= (uint32_t)k) start = i;
            if (rd == n - 1 && closed) slot = cstart;
        }
    }
    return (uint32_t)(slot & mask);
}

/* ------------------------------------------------------------------------- */
/* clust2snp                                                                  */
/* ------------------------------------------------------------------------- */

typedef struct { uint64_t* v; size_t n, cap; } vec64;
static void v_push(vec64* a, uint64_t x) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 16; a->v = (uint64_t*)realloc(a->v, a->cap * 8); }
    a->v[a->n++] = x;
}
static void v_free(vec64* a) { free(a->v); a->v = NULL; a->n = a->cap = 0; }

/* clust2snp.cpp:101-117 */
typedef struct {
    vec64 idx0, pos0, idx1, pos1;
    uint64_t right_idx, right_pos;
} o_cand;

typedef struct { o_cand* v; size_t n, cap; } cand_vec;
static void c_push(cand_vec* a, o_cand c) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 64; a->v = (o_cand*)realloc(a->v, a->cap * sizeof(o_cand)); }
    a->v[a->n++] = c;
}

typedef struct {
    uint64_t nr_reads1;
    int k_left, k_right, max_gap, consensus_reads, max_err, mcov_out, max_clust_length;
    double pval;
    int lcp, da, pos;
    uint64_t n_clust, n_bases;
} c2s_par;

/* clust2snp.cpp:367-500 */
static void find_variants(const c2s_par* P, const o_gsa* cl, size_t n, cand_vec* out) {
    unsigned int counts[2][4] = {{0,0,0,0},{0,0,0,0}};
    uint64_t max_lcp_val = 0, max_lcp_read_idx = 0, max_lcp_read_pos = 0;
    for (size_t i = 0; i < n; ++i) {                        /* :377-393 */
        o_gsa e = cl[i];
        if (e.lcp > max_lcp_val) { max_lcp_val = e.lcp; max_lcp_read_idx = e.text; max_lcp_read_pos = e.suff; }
        int sample = (uint64_t)e.text < P->nr_reads1 ? 0 : 1;
        counts[sample][base_to_int(e.bwt)]++;
    }
    if (max_lcp_val < (uint64_t)(int64_t)P->k_right) return;  /* :396  uint64_t < int */
    unsigned char f0[4], f1[4]; int n0 = 0, n1 = 0;
    for (int c = 0; c < 4; ++c) {                           /* :402-407  unsigned >= int */
        if (counts[0][c] >= (unsigned int)P->mcov_out) f0[n0++] = int_to_base(c);
        if (counts[1][c] >= (unsigned int)P->mcov_out) f1[n1++] = int_to_base(c);
    }
    /* :409-416: both lists are already ascending; union size */
    int present[256]; memset(present, 0, sizeof(present));
    for (int a = 0; a < n0; ++a) present[f0[a]] = 1;
    for (int a = 0; a < n1; ++a) present[f1[a]] = 1;
    int all = present['A'] + present['C'] + present['G'] + present['T'];
    int equal = (n0 == n1) && memcmp(f0, f1, (size_t)n0) == 0;
    if (n0 == 0 || n1 == 0 || n0 > 2 || n1 > 2 || equal || all > 3) return;   /* :419-429 */
    for (int a = 0; a < n0; ++a) for (int b = 0; b < n1; ++b) {              /* :431-435 */
        unsigned char c0 = f0[a], c1 = f1[b];
        if (c0 == c1) continue;
        o_cand cv; memset(&cv, 0, sizeof(cv));
        for (size_t i = 0; i < n; ++i) {                    /* :446-476 */
            o_gsa e = cl[i];
            int sample = (uint64_t)e.text < P->nr_reads1 ? 0 : 1;
            uint64_t prefix_len = e.suff, lcp = e.lcp;
            unsigned char ch = e.bwt;
            /* `prefix_len >= k_left`, `lcp >= k_right`: uint64_t vs int; `size() < consensus_reads`: size_t vs int */
            if (prefix_len >= (uint64_t)(int64_t)P->k_left && ch == c0 && sample == 0 &&
                lcp >= (uint64_t)(int64_t)P->k_right && (uint64_t)cv.idx0.n < (uint64_t)(int64_t)P->consensus_reads) {
                v_push(&cv.idx0, e.text);
                v_push(&cv.pos0, (uint32_t)(e.suff - (uint32_t)P->k_left));
            }
            if (prefix_len >= (uint64_t)(int64_t)P->k_left && ch == c1 && sample == 1 &&
                lcp >= (uint64_t)(int64_t)P->k_right && (uint64_t)cv.idx1.n < (uint64_t)(int64_t)P->consensus_reads) {
                v_push(&cv.idx1, e.text);
                v_push(&cv.pos1, (uint32_t)(e.suff - (uint32_t)P->k_left));
            }
        }
        if (cv.idx0.n > 0 && cv.idx1.n > 0) {               /* :478-490 */
            cv.right_idx = max_lcp_read_idx; cv.right_pos = max_lcp_read_pos;
            c_push(out, cv);
        } else { v_free(&cv.idx0); v_free(&cv.pos0); v_free(&cv.idx1); v_free(&cv.pos1); }
    }
}

/* clust2snp.cpp:877-966.  Returns ORC_OK or the crash the reference would take. */
static int statistics(c2s_par* P, const char* clusters_path, int verbose) {
    FILE* f = fopen(clusters_path, "rb");
    if (!f) return ORC_MISSING_CLUSTERS;
    const uint64_t MAX_C_LEN = (uint64_t)P->max_clust_length;   /* :882 */
    uint64_t* freq = (uint64_t*)calloc(MAX_C_LEN + 1, 8);
    uint64_t max_len = 0;
    int eof = 0, have = 0;
    uint64_t start = 0; uint16_t length = 0;      /* stale across iterations (H4) */
    while (!eof) {                                 /* :889 */
        uint8_t b[10];
        size_t got = fread(b, 1, 10, f);
        if (got == 10) { memcpy(&start, b, 8); memcpy(&length, b + 8, 2); have = 1; }
        else { eof = 1; if (!have) { free(freq); fclose(f); return ORC_REF_SIGFPE; } }
        if (length <= MAX_C_LEN) { freq[length]++; max_len = length > max_len ? length : max_len; }   /* :899-904 */
        P->n_clust++; P->n_bases += length;        /* :906-907 */
    }
    fclose(f);
    (void)start;
    uint64_t max = 0;
    for (uint64_t i = 1; i <= MAX_C_LEN; ++i) max = freq[i] * i > max ? freq[i] * i : max;   /* :912 */
    if (max == 0) { free(freq); return ORC_REF_SIGFPE; }      /* :921 divides by max */
    uint64_t cumulative = 0;
    if (verbose) {
        printf("\nDistribution of base coverage: \n");
        printf("\ncluster length\t# bases in a cluster with this length\t cumulative fraction (from 2m = %d)\n", 2 * P->mcov_out);
        for (uint64_t i = 0; i <= max_len; ++i) {
            printf("%llu\t", (unsigned long long)i);
            for (uint64_t j = 0; j < (100 * freq[i] * i) / max; ++j) putchar('-');
            printf("\t%llu", (unsigned long long)(freq[i] * i));
            if ((int64_t)i >= 2 * P->mcov_out) { cumulative += freq[i] * i; printf("\t%g", (double)cumulative / (double)P->n_bases); }
            printf("\n");
        }
    }
    /* :938-946.  -Ofast code generation of the oracle build: the first test divides,
     * the loop multiplies by the hoisted reciprocal 1.0/n_bases (objdump of statistics()). */
    int mcl = 2 * P->mcov_out;
    if (mcl < 0 || (uint64_t)mcl > MAX_C_LEN) { free(freq); return ORC_REF_SIGSEGV; }
    cumulative = freq[mcl] * (uint64_t)mcl;
    double nb = (double)P->n_bases;
    if ((double)cumulative / nb < P->pval) {
        double rnb = 1.0 / nb;
        uint64_t c = (uint64_t)mcl;
        while (c < MAX_C_LEN) {
            c++;
            cumulative += freq[c] * c;
            if (!((double)cumulative * rnb < P->pval)) break;
        }
        mcl = (int)c;
    }
    P->max_clust_length = mcl;
    if (verbose) printf("\nCluster sizes allowed: [%d,%d]\n", P->mcov_out * 2, P->max_clust_length);
    free(freq);
    return ORC_OK;
}

/* clust2snp.cpp:788-857 (the EGSA/.clusters lock-step part of find_events) */
static int find_events_scan(c2s_par* P, o_stream* EGSA, const char* clusters_path, cand_vec* cands) {
    FILE* cf = fopen(clusters_path, "rb");
    if (!cf) return ORC_MISSING_CLUSTERS;
    uint64_t i = 0;
    o_gsa e = read_el(EGSA);                       /* :796 */
    int rc = ORC_OK;
    int ceof = 0;
    uint64_t start = 0; uint16_t length = 0;       /* stale after the failed read (H4) */
    o_gsa* cl = NULL; size_t cl_cap = 0;
    if (P->n_clust - 1 == 0) { fclose(cf); return ORC_REF_SIGFPE; }   /* :849 */
    while (!ceof) {                                /* :806 */
        uint8_t b[10];
        size_t got = fread(b, 1, 10, cf);
        if (got == 10) { memcpy(&start, b, 8); memcpy(&length, b + 8, 2); } else ceof = 1;
        if ((int)length >= P->mcov_out * 2 && (int)length <= P->max_clust_length) {   /* :816 */
            while (i < start) { e = read_el(EGSA); ++i; }                   /* :818-823 */
            size_t n = 0, vcap = 0;            /* std::vector capacity of gsa_cluster: 0,1,2,4,... */
            while (i < start + length) {                                    /* :827-833 */
                if (n == cl_cap) { cl_cap = cl_cap ? cl_cap * 2 : 256; cl = (o_gsa*)realloc(cl, cl_cap * sizeof(o_gsa)); }
                int realloc_now = (n == vcap);
                if (realloc_now) vcap = vcap ? vcap * 2 : 1;
                cl[n++] = e;
                int was_eof = stream_eof(EGSA);
                e = read_el(EGSA);
                /* first read past the end right behind a reallocating push_back: slot clobbered */
                if (!was_eof && stream_eof(EGSA) && realloc_now && i + 1 < start + length) rc = ORC_REF_UB_ABORT;
                ++i;
            }
            if (rc) break;
            find_variants(P, cl, n, cands);                                 /* :840-843 */
        }
    }
    free(cl);
    fclose(cf);
    return rc;
}

/* ---- FASTA access: clust2snp.cpp:147-212 (get_reads) ---- */
typedef struct { char* s; size_t n, cap; } str;
static void s_clear(str* a) { a->n = 0; if (a->s) a->s[0] = 0; }
static void s_append(str* a, const char* b, size_t m) {
    if (a->n + m + 1 > a->cap) { a->cap = (a->n + m + 1) * 2; a->s = (char*)realloc(a->s, a->cap); }
    memcpy(a->s + a->n, b, m); a->n += m; a->s[a->n] = 0;
}
static void s_assign(str* a, const str* b) { s_clear(a); s_append(a, b->s ? b->s : "", b->n); }
static char s_first(const str* a) { return a->n ? a->s[0] : '\0'; }    /* std::string line[0] */

typedef struct { FILE* f; int eof, fail; } o_text;
/* std::getline(istream&, string&): untouched on a failed stream; otherwise erased first */
static void o_getline(o_text* t, str* line) {
    if (!t->f || t->fail) return;
    s_clear(line);
    size_t extracted = 0;
    for (;;) {
        int c = getc_unlocked(t->f);
        if (c == EOF) { t->eof = 1; if (!extracted) t->fail = 1; return; }
        extracted++;
        if (c == '\n') return;
        char ch = (char)c; s_append(line, &ch, 1);
    }
}

static int get_reads(const char* fasta_path, const vec64* ranks, str* out_dna /* ranks->n strings */, int verbose) {
    o_text F; F.f = fopen(fasta_path, "r"); F.eof = F.fail = 0;
    if (!F.f) F.fail = 1;
    uint64_t j = 0;
    str ID = {0,0,0}, DNA = {0,0,0}, line = {0,0,0};
    s_append(&ID, "", 0); s_append(&DNA, "", 0); s_append(&line, "", 0);
    o_getline(&F, &ID);                                        /* :159 */
    o_getline(&F, &line); s_assign(&DNA, &line);               /* :162-163 */
    while (!F.eof && s_first(&line) != '>') {                  /* :166-171 */
        o_getline(&F, &line);
        if (s_first(&line) != '>') s_append(&DNA, line.s, line.n);
    }
    if (verbose) printf("(2/4) Extracting reads from fasta file ...\n");
    if (ranks->n == 1) { if (F.f) fclose(F.f); free(ID.s); free(DNA.s); free(line.s); return ORC_REF_SIGFPE; }  /* :200 */
    int last_perc = 0;
    for (uint64_t i = 0; i < ranks->n; ++i) {                  /* :178-208 */
        while (j < ranks->v[i]) {
            s_assign(&ID, &line);
            o_getline(&F, &line); s_assign(&DNA, &line);
            while (!F.eof && s_first(&line) != '>') {
                o_getline(&F, &line);
                if (s_first(&line) != '>') s_append(&DNA, line.s, line.n);
            }
            ++j;
        }
        out_dna[i].s = NULL; out_dna[i].n = out_dna[i].cap = 0;
        s_append(&out_dna[i], DNA.s, DNA.n);
        int perc = (int)((i * 100) / (ranks->n - 1));
        if (perc >= last_perc + 10) { last_perc = perc; if (verbose) printf(" %d%% done.\n", perc); }
    }
    if (F.f) fclose(F.f);
    free(ID.s); free(DNA.s); free(line.s);
    return ORC_OK;
}

/* clust2snp.cpp:218-232 */
int orc_dH(const char* a, size_t la, const char* b, size_t lb) {
    int len = (int)(la < lb ? la : lb);
    int d = 0;
    for (int i = 0; i < len; ++i) d += a[la - (size_t)i - 1] != b[lb - (size_t)i - 1];
    return d;
}

/* clust2snp.cpp:254-302 */
static void distance_n(const char* a, size_t la, const char* b, size_t lb, int max_gap, int* D, int* L) {
    int dist_no_indel = orc_dH(a, la, b, lb);
    if (max_gap == 0) { *D = dist_no_indel; *L = 0; return; }       /* :262 */
    if (max_gap < 0) { *D = dist_no_indel; *L = 0; return; }        /* vector<int>(negative) throws in the reference */
    int* dab = (int*)calloc((size_t)max_gap, sizeof(int));
    int* dba = (int*)calloc((size_t)max_gap, sizeof(int));
    for (int i = 1; i < max_gap + 1; ++i) {                          /* :267-272  a.substr(0, a.length()-i) */
        size_t l1 = (size_t)i <= la ? la - (size_t)i : la;           /* size_t wrap -> whole string */
        dab[i - 1] = orc_dH(a, l1, b, lb) + i;
    }
    for (int i = 1; i < max_gap + 1; ++i) {                          /* :275-280 */
        size_t l1 = (size_t)i <= lb ? lb - (size_t)i : lb;
        dba[i - 1] = orc_dH(a, la, b, l1) + i;
    }
    int ia = 0, ib = 0;                                              /* :282-283 first minimum */
    for (int i = 1; i < max_gap; ++i) { if (dab[i] < dab[ia]) ia = i; if (dba[i] < dba[ib]) ib = i; }
    if (dist_no_indel < dab[ia] && dist_no_indel < dba[ib]) { *D = dist_no_indel; *L = 0; }      /* :285-288 */
    else if (dab[ia] < dba[ib]) { *D = dab[ia] - (ia + 1); *L = ia + 1; }                       /* :290-293 */
    else { *D = dba[ib] - (ib + 1); *L = -(ib + 1); }                                           /* :300 */
    free(dab); free(dba);
}
void orc_distance(const char* a, const char* b, int max_gap, int* d, int* l) {
    distance_n(a, strlen(a), b, strlen(b), max_gap, d, l);
}

/* include.hpp:334-371 */
typedef struct { char* C; int* counts; int size; } o_cons;
static void cons_init(o_cons* c, int size) {
    c->size = size;
    c->C = (char*)malloc((size_t)size + 1); memset(c->C, 'A', (size_t)size); c->C[size] = 0;   /* :341 */
    c->counts = (int*)calloc((size_t)size * 4, sizeof(int));                                   /* :340 */
}
static void cons_free(o_cons* c) { free(c->C); free(c->counts); }
static void cons_increment(o_cons* c, int i, unsigned char b) {     /* :349-358 */
    int b_i = base_to_int(b);
    c->counts[i * 4 + b_i]++;
    if (c->counts[i * 4 + b_i] > c->counts[i * 4 + base_to_int((unsigned char)c->C[i])]) c->C[i] = (char)b;
}
void orc_cons(const char* rows, int n, int k, char* out) {
    o_cons c; cons_init(&c, k);
    for (int j = 0; j < n; ++j) for (int i = 0; i < k; ++i) cons_increment(&c, i, (unsigned char)rows[(size_t)j * (size_t)k + (size_t)i]);
    memcpy(out, c.C, (size_t)k); out[k] = 0;
    cons_free(&c);
}

/* clust2snp.cpp:124-137 */
typedef struct { char *left0, *left1, *right; size_t rlen; int support_0, support_1; } o_variant;

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y;
}

/* std::string operator[]: position size() yields '\0'; beyond is UB in the reference */
static unsigned char rd_at(const str* r, uint64_t p) { return p < r->n ? (unsigned char)r->s[p] : 0; }

/* clust2snp.cpp:505-628 */
static int extract_variants(const c2s_par* P, const cand_vec* cands, const char* fasta_path,
                            o_variant** out_v, size_t* out_n, int verbose) {
    *out_v = NULL; *out_n = 0;
    vec64 ranks = {0,0,0};
    for (size_t c = 0; c < cands->n; ++c) {                       /* :512-518 */
        const o_cand* v = &cands->v[c];
        for (size_t t = 0; t < v->idx0.n; ++t) v_push(&ranks, v->idx0.v[t]);
        for (size_t t = 0; t < v->idx1.n; ++t) v_push(&ranks, v->idx1.v[t]);
        v_push(&ranks, v->right_idx);
    }
    if (ranks.n == 0) return ORC_REF_SIGSEGV;                     /* :531 on an empty vector */
    qsort(ranks.v, ranks.n, 8, cmp_u64);                          /* :521-523 */
    size_t u = 0;
    for (size_t t = 0; t < ranks.n; ++t) if (t == 0 || ranks.v[t] != ranks.v[t - 1]) ranks.v[u++] = ranks.v[t];
    ranks.n = u;
    str* reads = (str*)calloc(ranks.n, sizeof(str));
    int rc = get_reads(fasta_path, &ranks, reads, verbose);       /* :528 */
    if (rc) { free(reads); v_free(&ranks); return rc; }
    if (verbose) printf("(3/4) Filtering %zu candidates and computing consensus of left-contexts ... \n", cands->n);
    o_variant* vars = (o_variant*)calloc(cands->n ? cands->n : 1, sizeof(o_variant));
    size_t nv = 0;
    const int kl = P->k_left;
    for (size_t c = 0; c < cands->n; ++c) {                       /* :541-624 */
        const o_cand* v = &cands->v[c];
        o_cons left[2]; int supp[2] = {0, 0};
        const vec64* IDX[2] = { &v->idx0, &v->idx1 };
        const vec64* POS[2] = { &v->pos0, &v->pos1 };
        for (int s = 0; s < 2; ++s) {
            cons_init(&left[s], kl > 0 ? kl : 0);
            for (size_t j = 0; j < IDX[s]->n; ++j) {              /* :545-552 */
                uint64_t* hit = (uint64_t*)bsearch(&IDX[s]->v[j], ranks.v, ranks.n, 8, cmp_u64);
                size_t idx = (size_t)(hit - ranks.v);             /* read_ranks_inv */
                for (int i = 0; i < kl; ++i) cons_increment(&left[s], i, rd_at(&reads[idx], POS[s]->v[j] + (uint64_t)i));
            }
            for (size_t j = 0; j < IDX[s]->n; ++j) {              /* :556-567 */
                uint64_t* hit = (uint64_t*)bsearch(&IDX[s]->v[j], ranks.v, ranks.n, 8, cmp_u64);
                size_t idx = (size_t)(hit - ranks.v);
                int d_H = 0;
                for (int i = 0; i < kl; ++i) d_H += (unsigned char)left[s].C[i] != rd_at(&reads[idx], POS[s]->v[j] + (uint64_t)i);
                if (d_H <= P->max_err) supp[s]++;
            }
        }
        if (supp[0] > 0 && supp[1] > 0) {                         /* :595-611 */
            uint64_t* hit = (uint64_t*)bsearch(&v->right_idx, ranks.v, ranks.n, 8, cmp_u64);
            const str* r = &reads[(size_t)(hit - ranks.v)];
            o_variant ov;
            ov.left0 = strdup(left[0].C); ov.left1 = strdup(left[1].C);
            /* substr(pos, k_right): throws if pos > size (reference aborts) */
            if (v->right_pos > r->n) { cons_free(&left[0]); cons_free(&left[1]); free(ov.left0); free(ov.left1); rc = ORC_REF_UB_ABORT; break; }
            size_t avail = r->n - (size_t)v->right_pos;
            size_t want = P->k_right < 0 ? avail : (size_t)P->k_right;   /* size_t(npos-like) for negative */
            ov.rlen = want < avail ? want : avail;
            ov.right = (char*)malloc(ov.rlen + 1); memcpy(ov.right, r->s + v->right_pos, ov.rlen); ov.right[ov.rlen] = 0;
            ov.support_0 = supp[0]; ov.support_1 = supp[1];
            vars[nv++] = ov;
        }
        cons_free(&left[0]); cons_free(&left[1]);
    }
    for (size_t t = 0; t < ranks.n; ++t) free(reads[t].s);
    free(reads); v_free(&ranks);
    *out_v = vars; *out_n = nv;
    return rc;
}

/* clust2snp.cpp:633-780 */
static uint64_t to_file(const c2s_par* P, const o_variant* vars, size_t nv, const char* out_path, int verbose) {
    FILE* out = fopen(out_path, "w");
    uint64_t id_nr = 1;
    if (verbose) printf("(4/4) Computing edit distances and saving SNPs/indels to file ... \n");
    for (size_t t = 0; t < nv; ++t) {
        const o_variant* v = &vars[t];
        size_t l0 = strlen(v->left0), l1 = strlen(v->left1);
        int D, L; distance_n(v->left0, l0, v->left1, l1, P->max_gap, &D, &L);     /* :646 */
        if (D <= 3) {                                                             /* :648 max_snvs_def */
            const char* kind = L != 0 ? "INDEL" : "SNP";
            size_t snv_cap = l0 + l1 + 4;
            char* snv = (char*)malloc(snv_cap); snv[0] = 0;                       /* :671-689 */
            if (L == 0) snprintf(snv, snv_cap, "%c/%c", v->left0[l0 - 1], v->left1[l1 - 1]);
            else if (L > 0) snprintf(snv, snv_cap, "%s/", v->left0 + (l0 - (size_t)L));
            else snprintf(snv, snv_cap, "/%s", v->left1 + (l1 - (size_t)(-L)));
            fprintf(out, ">%s_higher_path_%llu|P_1:%zu_%s|%d|nb_pol_1\n", kind, (unsigned long long)id_nr, v->rlen, snv, v->support_0);
            fprintf(out, "%s%s\n", L < 0 ? v->left0 + (size_t)(-L) : v->left0, v->right);          /* :701-717 */
            fprintf(out, ">%s_lower_path_%llu|P_1:%zu_%s|%d|nb_pol_1\n", kind, (unsigned long long)id_nr, v->rlen, snv, v->support_1);
            fprintf(out, "%s%s\n", L > 0 ? v->left1 + (size_t)L : v->left1, v->right);              /* :747-762 */
            id_nr++;
            free(snv);
        }
    }
    fclose(out);
    return id_nr - 1;
}

static void par_defaults(c2s_par* P, const orc_c2s_opts* o) {          /* clust2snp.cpp:1032-1043 */
    memset(P, 0, sizeof(*P));
    P->nr_reads1 = (uint64_t)(int64_t)(int)o->nr_reads1;                /* atoi -> uint64_t (:988) */
    P->lcp = o->lcp == 0 ? 1 : o->lcp;
    P->da  = o->da  == 0 ? 4 : o->da;
    P->pos = o->pos == 0 ? 1 : o->pos;
    P->max_err = o->max_err == 0 ? 2 : o->max_err;
    P->consensus_reads = o->consensus_reads == 0 ? 20 : o->consensus_reads;
    P->max_gap = o->max_gap == 0 ? 10 : o->max_gap;
    P->max_clust_length = 150;                                          /* -M is not parseable (H5) */
    P->k_left = o->k_left == 0 ? 31 : o->k_left;
    P->k_right = o->k_right == 0 ? 30 : o->k_right;
    P->pval = o->pval == 0 ? 0.99 : o->pval;
    P->mcov_out = o->mcov_out == 0 ? 5 : o->mcov_out;
}

static void cands_free(cand_vec* c) {
    for (size_t t = 0; t < c->n; ++t) { v_free(&c->v[t].idx0); v_free(&c->v[t].pos0); v_free(&c->v[t].idx1); v_free(&c->v[t].pos1); }
    free(c->v); c->v = NULL; c->n = c->cap = 0;
}

static int run_scan(const char* input, const orc_c2s_opts* o, c2s_par* P, cand_vec* cands, int verbose) {
    par_defaults(P, o);
    if (!input || !*input || P->nr_reads1 == 0) return ORC_HELP;        /* :1045 */
    if (!width_ok(P->lcp) || !width_ok(P->da) || !width_ok(P->pos)) return ORC_BAD_WIDTH;
    o_stream S;
    int rc = stream_open(&S, input);                                    /* :1047 */
    if (rc) return rc;
    S.lcp_size = P->lcp; S.da_size = P->da; S.suff_size = P->pos;       /* :1048 */
    char* cp = path_cat(input, ".clusters");                            /* :1056-1072 */
    FILE* t = fopen(cp, "rb");
    if (!t) { free(cp); stream_close(&S); return ORC_MISSING_CLUSTERS; }
    fclose(t);
    rc = statistics(P, cp, verbose);                                    /* :1080 */
    if (!rc) {
        if (verbose) printf("(1/4) Filtering relevant clusters ... \n");
        rc = find_events_scan(P, &S, cp, cands);                        /* :1081 */
    }
    free(cp);
    stream_close(&S);
    return rc;
}

int orc_clust2snp(const char* input, const orc_c2s_opts* o, const char* out_path,
                  orc_c2s_stats* st, int verbose) {
    c2s_par P; cand_vec cands = {0,0,0};
    if (st) memset(st, 0, sizeof(*st));
    int rc = run_scan(input, o, &P, &cands, verbose);
    if (st) { st->n_clust = P.n_clust; st->n_bases = P.n_bases; st->max_clust_length = P.max_clust_length; st->n_candidates = cands.n; }
    if (rc) { cands_free(&cands); return rc; }
    if (verbose) printf("Done. %zu potential variants detected (some might be detected twice: on fw and rev strands)\n", cands.n);
    o_variant* vars; size_t nv;
    rc = extract_variants(&P, &cands, input, &vars, &nv, verbose);      /* :864 */
    cands_free(&cands);
    if (rc) { free(vars); return rc; }
    char* op;
    if (out_path) op = strdup(out_path);
    else {                                                              /* :1074-1076 */
        size_t cut = strlen(input);
        for (size_t p = strlen(input); p-- > 0;) if (strncmp(input + p, ".fast", 5) == 0) { cut = p; break; }
        op = (char*)malloc(cut + 5); memcpy(op, input, cut); memcpy(op + cut, ".snp", 5);
    }
    uint64_t ne = to_file(&P, vars, nv, op, verbose);                   /* :868 */
    free(op);
    if (st) { st->n_variants = nv; st->n_events = ne; }
    for (size_t t = 0; t < nv; ++t) { free(vars[t].left0); free(vars[t].left1); free(vars[t].right); }
    free(vars);
    return ORC_OK;
}

int orc_candidates(const char* input, const orc_c2s_opts* o, uint64_t* out, size_t cap,
                   size_t* n_words, uint64_t* n_cand, int* max_clust_length) {
    c2s_par P; cand_vec cands = {0,0,0};
    int rc = run_scan(input, o, &P, &cands, 0);
    size_t w = 0;
    for (size_t c = 0; c < cands.n; ++c) {
        const o_cand* v = &cands.v[c];
        uint64_t hdr[4] = { v->idx0.n, v->idx1.n, v->right_idx, v->right_pos };
        for (int t = 0; t < 4; ++t) { if (w < cap) out[w] = hdr[t]; w++; }
        const vec64* L[4] = { &v->idx0, &v->pos0, &v->idx1, &v->pos1 };
        for (int a = 0; a < 4; ++a) for (size_t t = 0; t < L[a]->n; ++t) { if (w < cap) out[w] = L[a]->v[t]; w++; }
    }
    if (n_words) *n_words = w;
    if (n_cand) *n_cand = cands.n;
    if (max_clust_length) *max_clust_length = P.max_clust_length;
    cands_free(&cands);
    return rc;
}
