// e2s_tail.hip -- clust2snp's tail on the GPU: the reads resident in HBM, one wave per candidate.
//
//   K3a k_consensus      extract_variants: the <= c + c left contexts of a candidate gathered from the reads, the two
//                        consensus strings with cons' order-dependent tie-break (include.hpp:334-371), the supports
//                        (clust2snp.cpp:544-593), the right context, and distance() (clust2snp.cpp:254-302)
//   K3b k_tail_partials  exclusive scan of the per-group event counts / text lengths (one block)
//   K3c k_tail_write     to_file (clust2snp.cpp:633-780): the KisSNP2 text of every event, ids sequential, written into
//                        one image in HBM at offsets that follow from the scan
//
// Exactness: every step mirrors ebwt2snp_amd/host/snp_tail.cpp, which is pinned byte for byte to the reference's `.snp`;
// what the GPU form cannot reproduce (a read rank past the last read, where the reference repeats stale text; k_left outside
// 1..64; an irregular FASTA) is reported back and the caller takes the host tail instead.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "e2s_internal.h"

typedef unsigned int u32;
typedef unsigned long long u64;

namespace {

__device__ __forceinline__ u32 tl_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// include.hpp:265-279: A/a 0, C/c 1, G/g 2, T/t 3, everything else 0 ('N' is rand() % 4 in the reference: mapped to 0)
__device__ __forceinline__ u32 tl_b2i(u32 ch) { const u32 f = ch & 0xDFu; return f == 'C' ? 1u : f == 'G' ? 2u : f == 'T' ? 3u : 0u; }
__device__ __forceinline__ u32 tl_digits(u64 v) { u32 d = 1; while (v >= 10ull) { v /= 10ull; ++d; } return d; }
// sum of digits(v) over v in [0, x)
__device__ __forceinline__ u64 tl_digit_sum_below(u64 x) {
    u64 s = x, p = 10ull;
    for (int k = 1; k < 20; ++k) { if (x > p) s += x - p; else break; if (k < 19) p *= 10ull; }
    return s;
}
__device__ __forceinline__ u32 tl_pos(const uint8_t* p, u32 pw, u32 j) {
    return pw == 1u ? (u32)p[j] : pw == 2u ? (u32)p[2 * j] | ((u32)p[2 * j + 1] << 8) : ((const u32*)p)[j];
}

}  // namespace

// ---- K3a ----------------------------------------------------------------------------------------------------------------------
// One wave per candidate.  The work is a chain of dependent gathers (candidate record -> read index -> text), so nothing is
// fetched one item at a time: lane j fetches list entry j and its read's start / length (one round trip for the whole list),
// then the n x k_left characters of an allele's left contexts are gathered by all 64 lanes at once into LDS (rows of 64
// bytes, 64 / kw rows per wave instruction, kw = k_left rounded up to a power of two), and consensus and supports run on LDS.
// Dynamic LDS per wave: 2 x list_cap rows of 64 x 2 B (character | its base_to_int) + (8 + 4) B of row base / length each.
extern __shared__ __attribute__((aligned(16))) uint8_t s_tail[];
__global__ __launch_bounds__(256) void k_consensus(TailArgs a, u32 list_cap, u32 kw_shift) {
    __shared__ uint8_t s_ab[4][2][128];
    const u32 lane = tl_lane(), wave = threadIdx.x >> 6;
    const u64 c = (u64)blockIdx.x * (blockDim.x >> 6) + wave;      // (1, 2 or 4 waves per block: long lists need the LDS)
    if (c >= a.n_cand) return;
    const size_t per_wave = (size_t)2 * list_cap * (128 + 8 + 4);
    uint16_t* const ctx = (uint16_t*)(s_tail + wave * per_wave);    // [2 * list_cap][64] of (character | base_to_int << 8): allele 0's rows, then allele 1's
    u64* const rbase = (u64*)(s_tail + wave * per_wave + (size_t)2 * list_cap * 128);   // text offset of row j's first character
    u32* const rhave = (u32*)(rbase + 2 * list_cap);                // characters of row j that exist (the rest read as 0)
    const uint8_t* r = a.stream + (size_t)a.off8[c] * 8;
    const u32* w = (const u32*)r;
    const u32 n0 = w[0], n1 = w[1], right_idx = w[2], right_pos = w[3];
    const u32 kl = (u32)a.kl;
    uint8_t* const sa = s_ab[wave][0];
    uint8_t* const sb = s_ab[wave][1];
    u32 supp[2] = {0u, 0u};
    u32 bad = 0;
    const u32 kw = 1u << kw_shift, rows_per = 64u >> kw_shift, row_of = lane >> kw_shift, col = lane & (kw - 1u);
    // rows 0 .. n0-1: allele 0, rows n0 .. n0+n1-1: allele 1 -- both gathered in ONE sweep (the lists lie side by side in the record)
    const u32 m0 = min(n0, list_cap), m1 = min(n1, list_cap), nrows = m0 + m1;
    const u32* idx = w + 4;
    const uint8_t* pp = r + 16 + 4 * (size_t)(n0 + n1);
    for (u32 j = lane; j < nrows; j += 64u) {                       // row bases: one round trip for the lists, none for the reads' index when dense
        const u32 e = j < m0 ? j : n0 + (j - m0);                   // entry of the record (the lists are n0 / n1 long there)
        const u32 rd = idx[e];
        u64 base = 0; u32 have = 0;
        if ((u64)rd >= a.n_reads) bad |= E2S_TAILERR_RANK;
        else {
            const u32 len = a.rstart ? a.rlen[rd] : a.stride, p0 = tl_pos(pp, a.pos_pack, e);
            have = p0 < len ? min(len - p0, kl) : 0u;
            base = (a.rstart ? a.rstart[rd] : (u64)rd * a.stride) + p0;
        }
        rbase[j] = base; rhave[j] = have;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (u32 j0 = 0; j0 < nrows; j0 += rows_per) {                  // the characters: every load of the candidate in flight together
        const u32 j = j0 + row_of;
        if (j < nrows) {
            const u32 have = rhave[j];
            const u32 ch = col < have ? (u32)a.text[rbase[j] + col] : 0u;
            ctx[(size_t)j * 64 + col] = (uint16_t)(ch | (tl_b2i(ch) << 8));   // base_to_int once per character, by all lanes at once
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < 2; ++s) {
        const u32 n = s ? m1 : m0;
        const uint16_t* rows = ctx + (size_t)(s ? m0 : 0u) * 64;
        // cons (include.hpp:334-371), one lane per column: the letter changes only on a strictly larger count (H8).  The four counts
        // of a column live in one register, 8 bits each (a list holds at most 150 contexts).
        u32 C = 'A', bc8 = 0u, cnt = 0u;                            // bc8 = 8 * base_to_int(C)
        for (u32 j = 0; j < n; ++j) {
            const u32 v = lane < kl ? (u32)rows[(size_t)j * 64 + lane] : 0u;
            const u32 ch = v & 0xFFu, b8 = (v >> 8) << 3;
            cnt += 1u << b8;
            if (((cnt >> b8) & 0xFFu) > ((cnt >> bc8) & 0xFFu)) { C = ch; bc8 = b8; }
        }
        // support = contexts within max_err of the consensus (clust2snp.cpp:544-593)
        for (u32 j = 0; j < n; ++j) {
            const u32 ch = lane < kl ? ((u32)rows[(size_t)j * 64 + lane] & 0xFFu) : 0u;
            const int d = __popcll(__ballot(lane < kl && C != ch));
            if (d <= a.max_err) ++supp[s];
        }
        (s ? sb : sa)[lane] = (uint8_t)C;
        (s ? sb : sa)[64 + lane] = 0;
        if (lane < kl) a.cons[c * 128 + (u64)s * 64 + lane] = (uint8_t)C;
    }
    bad = (u32)(__ballot(bad & E2S_TAILERR_RANK) ? E2S_TAILERR_RANK : 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool keep = supp[0] > 0u && supp[1] > 0u;                // clust2snp.cpp:595-611
    // distance (clust2snp.cpp:254-302) of the two consensus strings, both k_left long
    const u32 ca = sa[lane], cb = sb[lane];
    const int none = __popcll(__ballot(lane < kl && ca != cb));
    int D = none, L = 0;
    if (a.max_gap > 0) {
        int best_ab = 0, best_ba = 0, ia = 0, ib = 0;
        for (int i = 1; i <= a.max_gap; ++i) {
            int mab, mba;
            if ((u32)i <= kl) {
                const u32 la = kl - (u32)i;                         // a.substr(0, len - i) against b, right-aligned: a[x] ~ b[x + i]
                const u32 xi = min(lane + (u32)i, 127u);
                mab = __popcll(__ballot(lane < la && ca != (u32)sb[xi]));
                mba = __popcll(__ballot(lane < la && (u32)sa[xi] != cb));
            } else mab = mba = none;                                // (size_t wrap in the reference: the whole string stays)
            const int dab = mab + i, dba = mba + i;
            if (i == 1 || dab < best_ab) { best_ab = dab; ia = i; }  // first minimum (std::min_element)
            if (i == 1 || dba < best_ba) { best_ba = dba; ib = i; }
        }
        if (!(none < best_ab && none < best_ba)) {                  // no indel only if strictly better than both (H9)
            if (best_ab < best_ba) { D = best_ab - ia; L = ia; } else { D = best_ba - ib; L = -ib; }
        }
    }
    // right context = read.substr(right_pos, k_right) (may be shorter, H10)
    u32 right_len = 0; u64 right_at = 0;
    if (keep) {
        if ((u64)right_idx >= a.n_reads) bad |= E2S_TAILERR_RANK;
        else {
            const u32 len = a.rstart ? a.rlen[right_idx] : a.stride;
            if (right_pos > len) bad |= E2S_TAILERR_CRASH;          // substr throws in the reference
            else {
                const u32 avail = len - right_pos;
                right_len = a.kr < 0 ? avail : min(avail, (u32)a.kr);
                right_at = (a.rstart ? a.rstart[right_idx] : (u64)right_idx * a.stride) + right_pos;
            }
        }
    }
    const bool emit = keep && D <= 3;                               // max_snvs_def: -v is parsed but unused (H5)
    if (lane == 0) {
        const u32 kind = L != 0 ? 5u : 3u, tlen = L == 0 ? 3u : (u32)(L < 0 ? -L : L) + 1u, drl = tl_digits(right_len);
        const u32 l1 = 1u + kind + 13u + 5u + drl + 1u + tlen + 1u + tl_digits(supp[0]) + 10u;
        const u32 l3 = 1u + kind + 12u + 5u + drl + 1u + tlen + 1u + tl_digits(supp[1]) + 10u;
        const u32 l2 = kl - (L < 0 ? (u32)(-L) : 0u) + right_len + 1u, l4 = kl - (L > 0 ? (u32)L : 0u) + right_len + 1u;
        TailRec t;
        t.emit = emit ? 1u : 0u; t.L = L; t.s0 = supp[0]; t.s1 = supp[1]; t.right_len = right_len; t.base_len = l1 + l2 + l3 + l4; t.right_at = right_at;
        a.rec[c] = t;
        // (per group of 64 candidates: one counter for the whole job saturates at ~90 atomics per microsecond -- 3 ms for 2.6x10^5 waves)
        if (keep) atomicAdd(&a.part[3 * (c >> 6) + 2], 1ull);
        if (emit) { atomicAdd(&a.part[3 * (c >> 6)], 1ull); atomicAdd(&a.part[3 * (c >> 6) + 1], (u64)t.base_len); }
        if (bad) atomicOr(a.err, bad);
    }
}

// ---- K3b: exclusive scan of the per-group (events, text bytes) pairs; totals behind the last group -------------------------
__global__ __launch_bounds__(1024) void k_tail_partials(const unsigned long long* part, unsigned long long* part_excl, u32 ng, u64 first_id,
                                                        unsigned long long* totals /* events, bytes, kept, error bits */, const unsigned long long* n_keep,
                                                        const u32* err) {
    __shared__ u64 s_e[1024], s_b[1024];
    __shared__ u64 run_e, run_b, run_k;
    const u32 tid = threadIdx.x;
    if (tid == 0) { run_e = 0; run_b = 0; run_k = 0; }
    __syncthreads();
    for (u32 g0 = 0; g0 < ng; g0 += 1024) {
        const u32 g = g0 + tid;
        const u64 e = g < ng ? part[3 * g] : 0ull, b = g < ng ? part[3 * g + 1] : 0ull;
        if (g < ng && part[3 * g + 2]) atomicAdd((unsigned long long*)&run_k, part[3 * g + 2]);
        s_e[tid] = e; s_b[tid] = b;
        __syncthreads();
        for (u32 d = 1; d < 1024; d <<= 1) {
            const u64 ve = tid >= d ? s_e[tid - d] : 0ull, vb = tid >= d ? s_b[tid - d] : 0ull;
            __syncthreads();
            s_e[tid] += ve; s_b[tid] += vb;
            __syncthreads();
        }
        if (g < ng) { part_excl[2 * g] = run_e + s_e[tid] - e; part_excl[2 * g + 1] = run_b + s_b[tid] - b; }
        __syncthreads();
        if (tid == 1023) { run_e += s_e[1023]; run_b += s_b[1023]; }
        __syncthreads();
    }
    if (tid == 0) {
        totals[0] = run_e;
        totals[1] = run_b + 2ull * (tl_digit_sum_below(first_id + run_e) - tl_digit_sum_below(first_id));
        totals[2] = run_k;
        totals[3] = (u64)*err;
    }
}

// ---- K3c: the text --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 tl_put(uint8_t* b, u32 at, const char* s) { while (*s) b[at++] = (uint8_t)*s++; return at; }
__device__ __forceinline__ u32 tl_put_num(uint8_t* b, u32 at, u64 v) {
    const u32 d = tl_digits(v);
    for (u32 i = 0; i < d; ++i) { b[at + d - 1 - i] = (uint8_t)('0' + (u32)(v % 10ull)); v /= 10ull; }
    return at + d;
}

// A block takes a group of 64 candidates (the unit of the scan of event counts and text lengths): each of its 16 waves ranks the
// group's events for itself (ballot + shuffles, no LDS, no barrier) and writes four of the candidates -- an event is a chain of
// dependent steps (consensus letters -> header text -> stores), and short chains on many waves are what hides their latency
// (four waves of 64 candidates each took 0.63 ms for 2.6x10^5 events, sixteen of 16 each 0.42 ms).
__global__ __launch_bounds__(1024) void k_tail_write(TailArgs a, const unsigned long long* part_excl, u64 first_id, uint8_t* out, u64 out_cap) {
    __shared__ uint8_t s_hdr[16][2][128];
    __shared__ uint8_t s_cab[16][2][64];
    const u32 lane = tl_lane(), wave = threadIdx.x >> 6;
    const u64 g = blockIdx.x;
    TailRec t; t.emit = 0; t.base_len = 0; t.L = 0; t.s0 = t.s1 = t.right_len = 0; t.right_at = 0;
    {
        const u64 c = g * 64 + lane;
        if (c < a.n_cand) t = a.rec[c];
    }
    // exclusive scan of (emit, base_len) inside the group
    const u64 em = __ballot(t.emit != 0u);
    const u32 e_before = (u32)__popcll(em & ((1ull << lane) - 1ull));
    const u32 bl = t.emit ? t.base_len : 0u;
    u32 incl = bl;
    for (u32 d = 1; d < 64; d <<= 1) { const u32 v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    const u64 my_rank = part_excl[2 * g] + e_before;                // events before this one
    const u64 my_off = part_excl[2 * g + 1] + (incl - bl) + 2ull * (tl_digit_sum_below(first_id + my_rank) - tl_digit_sum_below(first_id));
    const u32 kl = (u32)a.kl;
    uint8_t* const h1 = s_hdr[wave][0];
    uint8_t* const h3 = s_hdr[wave][1];
    uint8_t* const la = s_cab[wave][0];
    uint8_t* const lb = s_cab[wave][1];
    for (u32 k = 0; k < 4; ++k) {
        const int src = (int)(wave * 4 + k);
        if (!((em >> src) & 1ull)) continue;                        // (wave-uniform)
        const u64 c2 = g * 64 + (u64)src;
        const int L = __shfl(t.L, src);
        const u32 s0 = __shfl(t.s0, src), s1 = __shfl(t.s1, src), rl = __shfl(t.right_len, src);
        const u64 rat = __shfl(t.right_at, src), o = __shfl(my_off, src), id = first_id + __shfl(my_rank, src);
        const uint8_t* ca = a.cons + c2 * 128;
        const uint8_t* cbp = ca + 64;
        const u32 ach = lane < kl ? (u32)ca[lane] : 0u, bch = lane < kl ? (u32)cbp[lane] : 0u;   // the two consensus strings, one letter per lane
        la[lane] = (uint8_t)ach; lb[lane] = (uint8_t)bch;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32 n1 = 0, n3 = 0;
        if (lane == 0) {
            for (int line = 0; line < 2; ++line) {
                uint8_t* b = line ? h3 : h1;
                u32 at = 0;
                b[at++] = '>';
                at = tl_put(b, at, L != 0 ? "INDEL" : "SNP");
                at = tl_put(b, at, line ? "_lower_path_" : "_higher_path_");
                at = tl_put_num(b, at, id);
                at = tl_put(b, at, "|P_1:");
                at = tl_put_num(b, at, rl);
                b[at++] = '_';
                if (L == 0) { b[at++] = la[kl - 1]; b[at++] = '/'; b[at++] = lb[kl - 1]; }         // last consensus letters (H20)
                else if (L > 0) { for (u32 i = kl - (u32)L; i < kl; ++i) b[at++] = la[i]; b[at++] = '/'; }
                else { b[at++] = '/'; for (u32 i = kl - (u32)(-L); i < kl; ++i) b[at++] = lb[i]; }
                b[at++] = '|';
                at = tl_put_num(b, at, line ? s1 : s0);
                at = tl_put(b, at, "|nb_pol_1\n");
                if (line) n3 = at; else n1 = at;
            }
        }
        n1 = __shfl(n1, 0); n3 = __shfl(n3, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u32 skip0 = L < 0 ? (u32)(-L) : 0u, skip1 = L > 0 ? (u32)L : 0u;
        const u32 l2 = kl - skip0 + rl + 1u, l4 = kl - skip1 + rl + 1u;
        if (o + n1 + l2 + n3 + l4 <= out_cap) {
            uint8_t* p = out + o;
            for (u32 i = lane; i < n1; i += 64) p[i] = h1[i];
            p += n1;
            if (lane >= skip0 && lane < kl) p[lane - skip0] = (uint8_t)ach;
            for (u32 i = lane; i < rl; i += 64) p[kl - skip0 + i] = a.text[rat + i];
            if (lane == 0) p[kl - skip0 + rl] = '\n';
            p += l2;
            for (u32 i = lane; i < n3; i += 64) p[i] = h3[i];
            p += n3;
            if (lane >= skip1 && lane < kl) p[lane - skip1] = (uint8_t)bch;
            for (u32 i = lane; i < rl; i += 64) p[kl - skip1 + i] = a.text[rat + i];
            if (lane == 0) p[kl - skip1 + rl] = '\n';
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// reads of one length, gathered into a dense [n_reads][len] matrix (one wave per read): the tail then needs no index lookups
__global__ __launch_bounds__(256) void k_reads_compact(const uint8_t* text, const u64* rstart, u32 len, u64 n_reads, uint8_t* out) {
    const u32 lane = tl_lane();
    const u64 r = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const uint8_t* src = text + rstart[r];
    uint8_t* dst = out + r * len;
    for (u32 i = lane; i < len; i += 64u) dst[i] = src[i];
}
void e2s_launch_reads_compact(const uint8_t* text, const uint64_t* rstart, uint32_t len, uint64_t n_reads, uint8_t* out, hipStream_t st) {
    if (!n_reads) return;
    hipLaunchKernelGGL(k_reads_compact, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, st, text, (const u64*)rstart, len, n_reads, out);
}
void e2s_launch_consensus(const TailArgs& a, uint32_t list_cap, hipStream_t st) {
    if (!a.n_cand) return;
    uint32_t kw_shift = 0;
    while ((1u << kw_shift) < (uint32_t)a.kl) ++kw_shift;          // k_left <= 64
    if (list_cap < 1) list_cap = 1;
    const size_t per_wave = (size_t)2 * list_cap * (128 + 8 + 4);
    const unsigned wpb = 4 * per_wave <= 60 * 1024 ? 4u : 2 * per_wave <= 120 * 1024 ? 2u : 1u;
    const size_t lds = wpb * per_wave;
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) { (void)hipFuncSetAttribute((const void*)k_consensus, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; }
    hipLaunchKernelGGL(k_consensus, dim3((unsigned)((a.n_cand + wpb - 1) / wpb)), dim3(64 * wpb), lds, st, a, list_cap, kw_shift);
}
void e2s_launch_tail_partials(const unsigned long long* part, unsigned long long* part_excl, uint32_t ng, unsigned long long first_id,
                              unsigned long long* totals, const unsigned long long* n_keep, const uint32_t* err, hipStream_t st) {
    hipLaunchKernelGGL(k_tail_partials, dim3(1), dim3(1024), 0, st, part, part_excl, ng, first_id, totals, n_keep, err);
}
void e2s_launch_tail_write(const TailArgs& a, const unsigned long long* part_excl, unsigned long long first_id, uint8_t* out, unsigned long long out_cap,
                           hipStream_t st) {
    if (!a.n_cand) return;
    hipLaunchKernelGGL(k_tail_write, dim3((unsigned)((a.n_cand + 63) / 64)), dim3(1024), 0, st, a, part_excl, first_id, out, out_cap);
}
