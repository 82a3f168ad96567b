// e2s_plan.h -- the pure functions of the boundary stitch and of statistics(), written ONCE for the host (the C ABI's
// e2s_stitch_plan / e2s_hist_after_stitch / e2s_auto_max_clust_length) and for the device (k_job_plan: the resident job evaluates
// them on the GPU so that no host round trip separates the scan from the per-cluster pass).  Plain integer code and IEEE double
// arithmetic (no fast-math on either side): both compilations give the same results.
#pragma once
#include <stdint.h>
#include "../../include/e2s_hip.h"

#if defined(__HIPCC__)
#define E2S_PLAN_HD __host__ __device__ inline
#else
#define E2S_PLAN_HD inline
#endif

// start of the cluster that is open when rank r begins (walk back over pass-through shards)
E2S_PLAN_HD uint64_t e2s_plan_incoming_start(const e2s_boundary* all, int r) {
    for (int q = r - 1; q >= 0; --q) {
        if (all[q].open_start != E2S_NONE) return all[q].open_start;
        if (!all[q].pass_through) return E2S_NONE;
    }
    return E2S_NONE;
}

// e2s_stitch_plan (include/e2s_hip.h): what `rank` must do to its own records once it has everybody's descriptor.
// cluster_lm's handling of the array's end, ebwt2clust.cpp:83-84,98-106,116-118,127-131, with the EOF phantom (hazard H1).
E2S_PLAN_HD int e2s_plan_core(const e2s_boundary* all, int nranks, int rank, uint64_t n, int k, int lcp_bytes, e2s_stitch_plan_t* out) {
    e2s_stitch_plan_t z = {};
    *out = z;
    const e2s_boundary& me = all[rank];
    const uint32_t ku = (uint32_t)k;
    uint64_t inc = E2S_NONE;
    if (me.begins_inside) {
        inc = e2s_plan_incoming_start(all, rank);
        if (inc == E2S_NONE) return E2S_ERR_INPUT;               // descriptors do not fit together
        if (me.n_ends > 0) {
            out->patch_first = 1;
            out->first_start = inc;
            out->first_len = (uint16_t)((me.first_end_low16 - (uint32_t)(inc & 0xFFFF)) & 0xFFFF);
        }
    }
    if (rank != nranks - 1) return E2S_OK;
    // ---- last rank: the reference's EOF phantom (hazard H1; slot model, see oracle/e2s_oracle.c) ----
    const uint64_t xmask = lcp_bytes >= 8 ? ~0ull : ((1ull << (8 * lcp_bytes)) - 1ull);
    uint64_t slot = (((uint64_t)me.tail_lcp) & ~0xFFull) | (uint64_t)(me.tail_bwt & 0xFF);
    if (n >= 3 && me.end_before_last) {
        // append_entry ran right before the phantom read: the slot holds that cluster's start
        uint64_t s = me.last_end_start;
        if (s == E2S_NONE) s = inc;                              // it was this shard's only end and began earlier
        if (s == E2S_NONE) return E2S_ERR_INPUT;
        slot = s;
    }
    const uint32_t P = (uint32_t)(slot & xmask);
    out->phantom_lcp = P;
    const bool open_last = me.open_after != 0;                   // open(n-1)
    uint64_t cur = E2S_NONE;
    if (open_last) cur = (me.open_start != E2S_NONE) ? me.open_start : e2s_plan_incoming_start(all, rank);
    int nt = 0;
    if (n == 1) {
        // no loop iteration: start = ge(0) ? 0 : (P >= k ? 1 : null); tail flush with i = 1 (ebwt2clust.cpp:83-84,127-131)
        if (open_last) { out->tail_start[nt] = 0; out->tail_len[nt++] = 2; }
        else if (P >= ku) { out->tail_start[nt] = 1; out->tail_len[nt++] = 1; }
    } else if (open_last) {
        if (cur == E2S_NONE) return E2S_ERR_INPUT;
        const bool gt = me.tail_prev_lcp > me.tail_lcp;
        if ((gt && me.tail_lcp <= P) || P < ku) {                // the cluster ends at n-1 (ebwt2clust.cpp:98-106)
            out->tail_start[nt] = cur; out->tail_len[nt++] = (uint16_t)((n - 1 - cur + 1) & 0xFFFF);
            if (P >= ku) { out->tail_start[nt] = n; out->tail_len[nt++] = 1; }     // the phantom opens (n,1) (:116-118)
        } else {                                                 // still open after the loop: the flush includes index n
            out->tail_start[nt] = cur; out->tail_len[nt++] = (uint16_t)((n - cur + 1) & 0xFFFF);
        }
    } else if (P >= ku) { out->tail_start[nt] = n; out->tail_len[nt++] = 1; }
    out->n_tail = (uint32_t)nt;
    return E2S_OK;
}

// What e2s_hist_after_stitch changes in a rank's statistics, as <= 4 signed single-record updates (the histogram itself stays
// where it is): record 0's provisional length out and its stitched length in, the EOF tail records in; and the rank's
// (n_records, last_len) after them.
struct e2s_stat_delta { int32_t len[4]; int32_t sign[4]; int32_t n; uint64_t n_records, n_bases, last_len; };
E2S_PLAN_HD void e2s_hist_delta_core(const e2s_scan_stats* raw, const e2s_boundary* b, const e2s_stitch_plan_t* plan, e2s_stat_delta* d) {
    d->n = 0; d->n_records = raw->hist.n_records; d->n_bases = raw->hist.n_bases; d->last_len = raw->hist.last_len;
    const int min_len = raw->min_len;
    if (plan->patch_first) {                                     // record 0: provisional length out, stitched length in
        const int oldl = (int)(b->first_end_low16 & 0xFFFFu), newl = (int)plan->first_len;
        if (oldl >= min_len) { d->len[d->n] = oldl; d->sign[d->n++] = -1; d->n_records -= 1; d->n_bases -= (uint64_t)oldl; }
        if (newl >= min_len) { d->len[d->n] = newl; d->sign[d->n++] = +1; d->n_records += 1; d->n_bases += (uint64_t)newl; }
        if (raw->last_none || raw->last_is_first) d->last_len = newl >= min_len ? (uint64_t)newl : 0;
    }
    for (uint32_t t = 0; t < plan->n_tail; ++t) {                // the EOF tail records live in the stitch plan
        const int l = plan->tail_len[t];
        if (l >= min_len) { d->len[d->n] = l; d->sign[d->n++] = +1; d->n_records += 1; d->n_bases += (uint64_t)l; d->last_len = (uint64_t)l; }
    }
    if (!d->n_records) d->last_len = 0;
}

// statistics(), clust2snp.cpp:877-966, over the MERGED histogram (freq[0..150], records, bases, length of the last record):
// the H4 double count of the last record and the auto-detection loop.  The reference build (-Ofast) divides in the first test and
// multiplies by the hoisted reciprocal inside the loop; reproduced here (H19).  Returns 0, or which of the reference's crashes
// the input runs into: 1 no record (division by zero), 2 no length <= 150 (division by zero), 3 2 * mcov outside the histogram.
E2S_PLAN_HD int e2s_auto_len_core(uint64_t* freq /* [E2S_MAX_CLUST_LEN + 1], modified */, uint64_t nrec, uint64_t nb, uint64_t last_len, int mcov_out, double pval,
                                  int* max_clust_length, uint64_t* n_clust, uint64_t* n_bases) {
    if (nrec == 0) return 1;
    // H4: `while(not clusters.eof())` counts the last record twice
    if (last_len <= E2S_MAX_CLUST_LEN) freq[last_len]++;
    nrec += 1; nb += last_len;
    uint64_t mx = 0;
    for (uint64_t i = 1; i <= E2S_MAX_CLUST_LEN; ++i) { const uint64_t v = freq[i] * i; if (v > mx) mx = v; }
    if (mx == 0) return 2;
    int mcl = 2 * mcov_out;
    if (mcl < 0 || mcl > E2S_MAX_CLUST_LEN) return 3;
    uint64_t cum = freq[mcl] * (uint64_t)mcl;
    const double dnb = (double)nb;
    if ((double)cum / dnb < pval) {
        const double r = 1.0 / dnb;
        uint64_t c = (uint64_t)mcl;
        while (c < E2S_MAX_CLUST_LEN) {
            ++c;
            cum += freq[c] * c;
            if (!((double)cum * r < pval)) break;
        }
        mcl = (int)c;
    }
    *max_clust_length = mcl;
    if (n_clust) *n_clust = nrec;
    if (n_bases) *n_bases = nb;
    return 0;
}
