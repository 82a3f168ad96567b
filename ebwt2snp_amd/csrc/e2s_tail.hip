// e2s_tail.hip -- clust2snp's tail on the GPU: the reads resident in HBM, a wave (k_left > 32) or a half-wave per candidate.
//
//   K3a k_consensus[_pair]  extract_variants: the <= c + c left contexts of a candidate gathered from the reads, the two
//                           consensus strings with cons' order-dependent tie-break (include.hpp:334-371), the supports
//                           (clust2snp.cpp:544-593), the right context, and distance() (clust2snp.cpp:254-302)
//                           (k_left <= 32: the left contexts from the reads' bit planes -- k_reads_pack2 -- where every read of the wave is "ACGT" only)
//   K3b k_tail_partials     exclusive scan of the per-group event counts / text lengths (one block)
//   K3c k_tail_write        to_file (clust2snp.cpp:633-780): the KisSNP2 text of every event, ids sequential, written into
//                           one image in HBM at offsets that follow from the scan
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
__device__ __forceinline__ u32 tl_digits32(u32 v) {                 // (no division: the usual values have one to three digits)
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u : v < 100000000u ? 8u
         : v < 1000000000u ? 9u : 10u;
}
// sum of digits(v) over v in [0, x)
__device__ __forceinline__ u64 tl_digit_sum_below(u64 x) {
    u64 s = x, p = 10ull;
    for (int k = 1; k < 20; ++k) { if (x > p) s += x - p; else break; if (k < 19) p *= 10ull; }
    return s;
}
__device__ __forceinline__ u32 tl_pos(const uint8_t* p, u32 pw, u32 j) {
    return pw == 1u ? (u32)p[j] : pw == 2u ? (u32)p[2 * j] | ((u32)p[2 * j + 1] << 8) : ((const u32*)p)[j];
}

// entry j of a packed record's position list (1, 2 or 4 bytes each, the list 4-byte aligned inside an 8-byte aligned, 8-byte padded record):
// the aligned dword that holds it, shifted -- one load whatever the width, no branch for the compiler to wait at
__device__ __forceinline__ u32 tl_pos_rec(const uint8_t* p, u32 pw, u32 j) {
    const u32 at = pw * j;
    const u32 v = *(const u32*)(p + (at & ~3u)) >> (8u * (at & 3u));
    return pw >= 4u ? v : v & ((1u << (8u * pw)) - 1u);
}

// base_to_int (include.hpp:265-279) of four characters at once: the class from bits 1-2 of the case-folded byte (A 0, C 1, G 3, T 2 ->
// 0 1 2 3), kept only where the byte IS that letter -- its low three bits pick the one letter it could be from an 8-entry v_perm table
__device__ __forceinline__ u32 tl_b2i4(u32 v) {
    const u32 f = v & 0xDFDFDFDFu;
    const u32 e = __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, f & 0x07070707u);   // [FF 'A' FF 'C' 'T' FF FF 'G']
    const u32 x = e ^ f;                                            // zero byte <=> a letter of ACGT
    const u32 nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    const u32 ok = (~nz & 0x80808080u) >> 7;                        // 0x01 per valid byte
    const u32 code = (f >> 1) & 0x03030303u;
    return (code ^ (code >> 1)) & (ok | (ok << 1));
}
// The left contexts of a candidate into its LDS rows of 4 * LPR entries: LPR lanes per row, an unaligned
// dword each, four rows per instruction and candidate (gl = the lane's number inside the candidate's 4 * LPR lanes), BATCH loads
// issued back to back before the first LDS write.  Byte loads (one lane per character) are bound by the rate of load instructions:
// 10^7 random 31-byte rows take 0.50 ms that way and 0.24 ms as dwords on an MI355X (tools/gather_probe.hip), 0.22 ms being what
// HBM gives; written as one load + one LDS store per iteration the compiler also waits for every load before issuing the next.
// A dword may reach up to 3 bytes past its context: inside the text that is the next read.  A row whose dwords could reach past the END
// of the text (the last read's last bytes) carries TL_ROW_NEAR_END in its length word: the dword pass leaves it alone and
// tl_gather_rows_bytes fetches it one character per lane afterwards (wave-uniform branch, almost never taken).
// An entry is (character | 8 * base_to_int << 8): the consensus loop shifts by the high byte as it is.
constexpr u32 TL_ROW_NEAR_END = 0x80000000u;
template <u32 LPR>
__device__ __forceinline__ void tl_gather_rows(const TailArgs& a, uint16_t* ctx, const u64* rbase, const u32* rhave, u32 nrows, u32 nr_max, u32 gl) {
    constexpr u32 BATCH = 8;
    const u32 q4 = (gl & (LPR - 1u)) * 4u, rsub = gl / LPR;
    const u32 last = nrows ? nrows - 1u : 0u;
    for (u32 j0 = 0; j0 < nr_max; j0 += 4u * BATCH) {
        u32 w4[BATCH];
#pragma unroll
        for (u32 u = 0; u < BATCH; ++u) {
            w4[u] = 0u;
            if (j0 + 4u * u >= nr_max) break;                       // (wave-uniform)
            const u32 j = j0 + 4u * u + rsub, jj = min(j, last);
            const u32 have = rhave[jj];                             // (unconditional LDS reads, no branch around the load: a lane
            const u64 base = rbase[jj];                             //  without characters reads the first dword of the text)
            const bool ok = (j < nrows) & ((int)q4 < (int)have);    // (a near-end row's length word is negative)
            const u32 v = *(const u32*)(a.text + (ok ? base + q4 : 0ull));
            const u32 keep = have - q4;                             // characters of this dword that exist (>= 1 where ok)
            const u32 mask = keep >= 4u ? ~0u : (1u << (8u * keep)) - 1u;
            w4[u] = ok ? v & mask : 0u;
        }
#pragma unroll
        for (u32 u = 0; u < BATCH; ++u) {
            if (j0 + 4u * u >= nr_max) break;
            const u32 j = j0 + 4u * u + rsub;
            if (j < nrows) {
                const u32 cls8 = tl_b2i4(w4[u]) << 3;
                uint2 e;
                e.x = __builtin_amdgcn_perm(cls8, w4[u], 0x05010400u);   // [ch0 cls0 ch1 cls1]
                e.y = __builtin_amdgcn_perm(cls8, w4[u], 0x07030602u);   // [ch2 cls2 ch3 cls3]
                *(uint2*)(ctx + (size_t)j * (4u * LPR) + q4) = e;
            }
        }
    }
}
template <u32 LPR>
__device__ __forceinline__ void tl_gather_rows_bytes(const TailArgs& a, uint16_t* ctx, const u64* rbase, const u32* rhave, u32 nrows, u32 nr_max, u32 gl) {
    const u32 last = nrows ? nrows - 1u : 0u;
    for (u32 j = 0; j < nr_max; ++j) {
        const u32 hv = rhave[min(j, last)];
        if (j < nrows && (hv & TL_ROW_NEAR_END)) {
            const u32 have = hv & ~TL_ROW_NEAR_END;
            const u32 ch = gl < have ? (u32)a.text[rbase[j] + gl] : 0u;
            ctx[(size_t)j * (4u * LPR) + gl] = (uint16_t)(ch | (tl_b2i(ch) << 11));
        }
    }
}

}  // namespace

// ---- K3a, k_left 33..64 --------------------------------------------------------------------------------------------------------
// One wave per candidate.  The work is a chain of dependent gathers (candidate record -> read index -> text), so nothing is
// fetched one item at a time: lane j fetches list entry j and its read's start / length (one round trip for the whole list),
// then the n x k_left characters of both alleles' left contexts are gathered by all 64 lanes at once into LDS rows of 64 entries
// (tl_gather_rows: 16 lanes per row, a dword each), and consensus and supports run on LDS.
// Dynamic LDS per wave: 2 x list_cap rows of 64 x 2 B (character | 8 * its base_to_int) + (8 + 4) B of row base / length each.
extern __shared__ __attribute__((aligned(16))) uint8_t s_tail[];
__global__ __launch_bounds__(256) void k_consensus(TailArgs a, u32 list_cap) {
    __shared__ uint8_t s_ab[4][2][128];
    const u32 lane = tl_lane(), wave = threadIdx.x >> 6;
    const u64 c = (a.slot_lo_dev ? (u64)*a.slot_lo_dev : 0ull) + (u64)blockIdx.x * (blockDim.x >> 6) + wave;      // (1, 2 or 4 waves per block: long lists need the LDS)
    if (a.n_cand_dev) a.n_cand = min((u64)a.n_cand, (u64)*a.n_cand_dev);
    if (c >= a.n_cand) return;
    const size_t per_wave = (size_t)2 * list_cap * (128 + 8 + 4);
    uint16_t* const ctx = (uint16_t*)(s_tail + wave * per_wave);    // [2 * list_cap][64] of (character | 8 * base_to_int << 8): allele 0's rows, then allele 1's
    u64* const rbase = (u64*)(s_tail + wave * per_wave + (size_t)2 * list_cap * 128);   // text offset of row j's first character
    u32* const rhave = (u32*)(rbase + 2 * list_cap);                // characters of row j that exist (the rest read as 0)
    const u32 o8 = a.off8[c];
    bool rec_ok = (u64)o8 + 2ull <= a.stream8_cap;
    const uint8_t* r = a.stream + (size_t)(rec_ok ? o8 : 0u) * 8;
    const u32* w = (const u32*)r;
    u32 n0 = rec_ok ? w[0] : 0u, n1 = rec_ok ? w[1] : 0u;
    const u32 right_idx = w[2], right_pos = w[3];
    if (!rec_ok || (u64)o8 + ((16ull + (4ull + a.pos_pack) * ((u64)n0 + n1) + 7ull) >> 3) > a.stream8_cap) n0 = n1 = 0u;   // (a record that was never written: no event)
    const u32 kl = (u32)a.kl;
    uint8_t* const sa = s_ab[wave][0];
    uint8_t* const sb = s_ab[wave][1];
    u32 supp[2] = {0u, 0u};
    u32 bad = 0;
    bool any_near = false;
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
        const bool near = have != 0u && base + 64ull > a.text_bytes;
        any_near |= near;
        rbase[j] = base; rhave[j] = have | (near ? TL_ROW_NEAR_END : 0u);
    }
    any_near = __ballot(any_near) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    tl_gather_rows<16>(a, ctx, rbase, rhave, nrows, nrows, lane);   // rows of 64 entries, 16 lanes per row
    if (any_near) tl_gather_rows_bytes<16>(a, ctx, rbase, rhave, nrows, nrows, lane);
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
            const u32 ch = v & 0xFFu, b8 = v >> 8;
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
    // A gap of i costs at least i, so gaps longer than the plain distance `none` can neither be the first minimum that is used nor change
    // the test below (it only asks whether a minimum is <= none): the loop ends at min(max_gap, none) -- and is skipped for equal strings,
    // where the reference finds no indel.  (It used to run max_gap = 10 times for every candidate: a quarter of this kernel's instructions;
    // a SNP's two consensus strings differ in their last letter only: none = 1.)
    const int gap_hi = min(a.max_gap, none);
    if (gap_hi > 0) {
        int best_ab = 0, best_ba = 0, ia = 0, ib = 0;
        for (int i = 1; i <= gap_hi; ++i) {
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
        const u32 kind = L != 0 ? 5u : 3u, tlen = L == 0 ? 3u : (u32)(L < 0 ? -L : L) + 1u, drl = tl_digits32(right_len);
        const u32 l1 = 1u + kind + 13u + 5u + drl + 1u + tlen + 1u + tl_digits32(supp[0]) + 10u;
        const u32 l3 = 1u + kind + 12u + 5u + drl + 1u + tlen + 1u + tl_digits32(supp[1]) + 10u;
        const u32 l2 = kl - (L < 0 ? (u32)(-L) : 0u) + right_len + 1u, l4 = kl - (L > 0 ? (u32)L : 0u) + right_len + 1u;
        TailRec t;
        t.emit = (emit ? 1u : 0u) | (keep ? 2u : 0u); t.L = L; t.s0 = supp[0]; t.s1 = supp[1]; t.right_len = right_len; t.base_len = l1 + l2 + l3 + l4; t.right_at = right_at;
        a.rec[c] = t;
        // (per group of 64 candidates: one counter for the whole job saturates at ~90 atomics per microsecond -- 3 ms for 2.6x10^5 waves)
        if (!a.by_slot) {                                           // (by slot: the groups are those of the ORDER, summed by k_tail_group_sums)
            if (keep) atomicAdd(&a.part[3 * (c >> 6) + 2], 1ull);
            if (emit) { atomicAdd(&a.part[3 * (c >> 6)], 1ull); atomicAdd(&a.part[3 * (c >> 6) + 1], (u64)t.base_len); }
        }
        if (bad) atomicOr(a.err, bad);
    }
}

// ---- K3a for k_left <= 32: TWO candidates per wave, one per half-wave ------------------------------------------------------------
// A left context fills at most 32 lanes, so in k_consensus half of every vector instruction is idle, and the kernel is bound by the
// issue of those instructions as much as by its chain of gathers.  Here lanes 0-31 serve candidate 2w, lanes 32-63 candidate 2w + 1: the
// same instruction stream, loops running to the longer of the two lists, ballots read per half.  Results are those of k_consensus.
//
// Round 4, second form -- the two per-row loops were ~31 vector instructions per row (two thirds of the kernel's):
//   * rows are BYTES (32 B per row: 8 waves per SIMD fit where the (character, class) pairs allowed 6);
//   * cons (include.hpp:334-371) is sequential by definition -- the letter changes when a count becomes STRICTLY larger than the leader's,
//     so among equal counts the class that reached the count first wins and class 0's letter ('A', but also 'N', lower case, the 0 behind a
//     read's end) is that of the row where class 0 last TOOK the lead.  But the outcome is decided by the four totals alone unless
//     (a) two classes share the largest total, or (b) class 0 wins, another class occurs, and the column holds a 0 -- so: one pass that only
//     counts (5 instructions per row: the class slot is (ch >> 1) & 3 for the characters {0, A, C, G, T}, which the gather checks), the
//     winner from the totals, and the sequential loop only for a wave that holds such a column (or any other character).  (b) is narrowed
//     by what is cheap to know: the other classes' totals are all <= 1 and row 0 is class 0 => class 0 never lost the lead => 'A';
//   * the supports (clust2snp.cpp:544-593) one ROW per lane, both alleles' rows in one sweep: 32 B of the row against the consensus of
//     its allele as 8 dwords, mismatching bytes counted by SWAR + v_bcnt -- ~70 instructions for up to 32 rows instead of ~14 per row.
// Dynamic LDS per wave: 2 candidates x (2 x list_cap rows of 32 B + (8 + 4) B of row base / length each + one spare row), to 32 B.
// (Round 3: the record's first 256 bytes fetched as ONE window -- a chain of five dependent round trips cut to three -- changed nothing.)
// LDS of ONE candidate of the half-wave kernel (kernel and launcher take it from here): 2 x list_cap rows of 32 B, one spare row (the counting
// loops read two rows per trip), (8 + 4) B of row base / length per row, to 32 B
__host__ __device__ constexpr size_t tl_pair_cand_lds(u32 list_cap) { return ((size_t)2 * list_cap * (32 + 8 + 4) + 32 + 31) & ~(size_t)31; }
__device__ __forceinline__ u32 tl_half_pop(u64 m, u32 half) { return (u32)__popc(half ? (u32)(m >> 32) : (u32)m); }
__device__ __forceinline__ u32 tl_nzb(u32 x, u32 acc) {            // acc + the number of nonzero bytes of x
    const u32 t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    return (u32)__popc(t) + acc;
}
// the left contexts as rows of 32 bytes: 8 lanes per row, an unaligned dword each (see tl_gather_rows); returns nonzero in a lane that saw
// a character outside {0, 'A', 'C', 'G', 'T'} (its low three bits pick the one it could be from an 8-entry v_perm table)
__device__ __forceinline__ u32 tl_gather_rows_b(const TailArgs& a, uint8_t* rows, const u64* rbase, const u32* rhave, u32 nrows, u32 nr_max, u32 gl) {
    constexpr u32 BATCH = 8;
    const u32 q4 = (gl & 7u) * 4u, rsub = gl >> 3;
    const u32 last = nrows ? nrows - 1u : 0u;
    u32 odd = 0u;
    for (u32 j0 = 0; j0 < nr_max; j0 += 4u * BATCH) {
        u32 w4[BATCH], mk[BATCH];                                   // the dword as loaded, and which of its characters exist
#pragma unroll
        for (u32 u = 0; u < BATCH; ++u) {                           // (loads only: nothing here may wait for one of them)
            w4[u] = 0u; mk[u] = 0u;
            if (j0 + 4u * u >= nr_max) break;                       // (wave-uniform)
            const u32 j = j0 + 4u * u + rsub, jj = min(j, last);
            const u32 have = rhave[jj];
            const u64 base = rbase[jj];
            const bool ok = (j < nrows) & ((int)q4 < (int)have);    // (a near-end row's length word is negative)
            w4[u] = *(const u32*)(a.text + (ok ? base + q4 : 0ull));
            const u32 keep = have - q4;
            mk[u] = !ok ? 0u : keep >= 4u ? ~0u : (1u << (8u * keep)) - 1u;
        }
#pragma unroll
        for (u32 u = 0; u < BATCH; ++u) {
            if (j0 + 4u * u >= nr_max) break;
            const u32 j = j0 + 4u * u + rsub;
            // (a character that exists must be a letter of ACGT -- a 0 byte IN the text is not one; the 0s behind a row's end are ours)
            odd |= (w4[u] ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, w4[u] & 0x07070707u)) & mk[u];   // [FF 'A' FF 'C' 'T' FF FF 'G']
            if (j < nrows) *(u32*)(rows + (size_t)j * 32u + q4) = w4[u] & mk[u];
        }
    }
    return odd;
}
__device__ __forceinline__ void tl_gather_rows_b_bytes(const TailArgs& a, uint8_t* rows, const u64* rbase, const u32* rhave, u32 nrows, u32 nr_max, u32 gl) {
    const u32 last = nrows ? nrows - 1u : 0u;
    for (u32 j = 0; j < nr_max; ++j) {
        const u32 hv = rhave[min(j, last)];
        if (j < nrows && (hv & TL_ROW_NEAR_END)) {
            const u32 have = hv & ~TL_ROW_NEAR_END;
            rows[(size_t)j * 32u + gl] = gl < have ? a.text[rbase[j] + gl] : (uint8_t)0;
        }
    }
}
__global__ __launch_bounds__(256) void k_consensus_pair(TailArgs a, u32 list_cap) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ab[4][2][2][64];
    __shared__ u32 s_minh[4][2][2];
    const u32 lane = tl_lane(), wave = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31u;
    const u64 c0 = (a.slot_lo_dev ? (u64)*a.slot_lo_dev : 0ull) + ((u64)blockIdx.x * (blockDim.x >> 6) + wave) * 2ull;
    if (a.n_cand_dev) a.n_cand = min((u64)a.n_cand, (u64)*a.n_cand_dev);
    if (c0 >= a.n_cand) return;
    const u64 c = c0 + half;
    const bool valid = c < a.n_cand;
    const size_t per_cand = tl_pair_cand_lds(list_cap);
    uint8_t* const mine = s_tail + ((size_t)wave * 2 + half) * per_cand;
    uint8_t* const ctx = mine;                                      // [2 * list_cap + 1][32] characters: allele 0's rows, then allele 1's
    u64* const rbase = (u64*)(mine + ((size_t)2 * list_cap + 1) * 32);
    u32* const rhave = (u32*)(rbase + 2 * list_cap);
    const u32 o8 = a.off8[valid ? c : c0];
    const bool rec_ok = (u64)o8 + 2ull <= a.stream8_cap;
    const uint8_t* r = a.stream + (size_t)(rec_ok ? o8 : 0u) * 8;
    const u32* w = (const u32*)r;
    u32 n0 = valid && rec_ok ? w[0] : 0u, n1 = valid && rec_ok ? w[1] : 0u;
    const u32 right_idx = w[2], right_pos = w[3];
    if ((u64)o8 + ((16ull + (4ull + a.pos_pack) * ((u64)n0 + n1) + 7ull) >> 3) > a.stream8_cap) n0 = n1 = 0u;      // (a record that was never written: no event)
    const u32 kl = (u32)a.kl;
    uint8_t* const sa = s_ab[wave][half][0];
    uint8_t* const sb = s_ab[wave][half][1];
    u32 bad = 0;
    bool any_near = false;
    const u32 m0 = min(n0, list_cap), m1 = min(n1, list_cap), nrows = m0 + m1;
    const u32 nr_max = max((u32)__builtin_amdgcn_readlane((int)nrows, 0), (u32)__builtin_amdgcn_readlane((int)nrows, 32));
    const u32* idx = w + 4;
    const uint8_t* pp = r + 16 + 4 * (size_t)(n0 + n1);
    const bool try2 = a.text2 != nullptr;                           // the reads as bit planes are there
    u32 imp = 0u;                                                   // a row of a read that holds a character outside "ACGT"
    uint2* const prow = (uint2*)ctx;                                // [2 * list_cap] rows as (plane of bit 0, plane of bit 1): where the byte rows go if they are needed
    if (hl < 2u) s_minh[wave][half][hl] = 32u;                      // the shortest row of either allele: columns at or behind it hold a 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (u32 j0 = 0; j0 < nr_max; j0 += 32u) {
        const u32 j = j0 + hl;
        if (j < nrows) {
            const u32 e = j < m0 ? j : n0 + (j - m0);
            // (read and position are both of the record: one round trip, not one behind the other -- and no branch on the read's rank, which
            //  the compiler would put the second load behind)
            const u32 rd = idx[e], p0 = tl_pos_rec(pp, a.pos_pack, e);
            const bool rok = (u64)rd < a.n_reads;
            const u32 rdc = rok ? rd : 0u;
            const u32 len = a.rstart ? a.rlen[rdc] : a.stride;
            const u32 have = rok && p0 < len ? min(len - p0, kl) : 0u;
            const u64 base = rok ? (a.rstart ? a.rstart[rdc] : (u64)rdc * a.stride) + p0 : 0ull;
            if (!rok) bad |= E2S_TAILERR_RANK;
            const bool near = have != 0u && base + 32ull > a.text_bytes;
            any_near |= near;
            rbase[j] = base; rhave[j] = have | (near ? TL_ROW_NEAR_END : 0u);
            atomicMin(&s_minh[wave][half][j < m0 ? 0 : 1], have);
            if (try2) {
                // the row from the bit planes: the two groups of 32 characters it lies in (16 bytes, 8-byte aligned: one line 7 times out of 8),
                // shifted to its first character, cut to the characters that exist -- the lane that knows the row fetches it
                const u32 iw = a.impure ? a.impure[rdc >> 5] : 0u;   // (nullptr: no read of the text is flagged -- nothing to look up)
                // (a position at or behind its read's end -- have == 0, e.g. a 4-byte position column that holds anything -- names no character:
                //  nothing is fetched from there, as in the byte path; a row that exists lies inside the text, so its two groups inside the planes)
                const u64 g = have ? base >> 5 : 0ull;
                const uint2 g0 = a.text2[g], g1 = a.text2[g + 1ull];
                const u32 sh = (u32)base & 31u, vm = have >= 32u ? ~0u : (1u << have) - 1u;
                prow[j] = make_uint2(__builtin_amdgcn_alignbit(g1.x, g0.x, sh) & vm, __builtin_amdgcn_alignbit(g1.y, g0.y, sh) & vm);
                imp |= rok ? (iw >> (rdc & 31u)) & 1u : 0u;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32 supp0 = 0u, supp1 = 0u;
    const bool colok = hl < kl;
    // ---- the usual case on the bit planes: none of the wave's reads holds a character outside "ACGT", and the column totals decide both
    // consensus strings.  Anything else -- a flagged read, a tie at the top, class 0 winning a column that holds a 0 -- takes the bytes below.
    bool fast2 = try2 && __ballot(imp != 0u) == 0ull;
    if (fast2) {
        u32 C2[2] = {'A', 'A'};
        for (int s = 0; s < 2 && fast2; ++s) {
            const u32 n = s ? m1 : m0;
            const u32 n_max = max((u32)__builtin_amdgcn_readlane((int)n, 0), (u32)__builtin_amdgcn_readlane((int)n, 32));
            const uint2* pr = prow + (s ? m0 : 0u);
            const u32 n_col = colok ? n : 0u;
            u32 cnt = 0u;
            for (u32 j = 0; j < n_max; j += 2u) {                   // (the row behind an odd list's last one lies inside the candidate's LDS area and is masked)
                uint2 ra = pr[j], rb = pr[j + 1u];
                asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(rb.x), "+v"(rb.y));
#pragma unroll
                for (u32 q = 0; q < 2u; ++q) {
                    const uint2 w = q ? rb : ra;
                    const u32 one = j + q < n_col ? 1u : 0u;
                    cnt += one << ((((w.x >> hl) & 1u) << 3) | (((w.y >> hl) & 1u) << 4));   // slot = the code: A 0, C 1, T 2, G 3
                }
            }
            const u32 t0 = cnt & 0xFFu, t1 = (cnt >> 8) & 0xFFu, t2 = (cnt >> 16) & 0xFFu, t3 = cnt >> 24;
            const u32 M = max(max(t0, t1), max(t2, t3));
            const u32 x = cnt ^ (M * 0x01010101u);
            const u32 z = ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x)) & 0x80808080u;   // 0x80 in the bytes that hold M (at least one)
            const u32 slot8 = (u32)__builtin_ctz(z) - 7u;
            C2[s] = (0x47544341u >> slot8) & 0xFFu;                 // "ACTG" by slot
            const bool tie = __popc(z) > 1 && M != 0u;
            const uint2 r0 = pr[0];
            const bool never_lost = ((((r0.x | r0.y) >> hl) & 1u) == 0u) & ((cnt & 0xFEFEFE00u) == 0u);   // row 0 is class 0 and no other class occurs twice
            const bool zero_col = hl >= s_minh[wave][half][s];
            const bool redo = n_col != 0u && (tie || (slot8 == 0u && zero_col && (cnt & 0xFFFFFF00u) != 0u && !never_lost));
            if (__ballot(redo) != 0ull) fast2 = false;
        }
        if (fast2) {
            sa[hl] = (uint8_t)(colok ? C2[0] : 0u); sa[32 + hl] = 0;
            sb[hl] = (uint8_t)(colok ? C2[1] : 0u); sb[32 + hl] = 0;
            if (valid && colok) { a.cons[c * 128 + hl] = (uint8_t)C2[0]; a.cons[c * 128 + 64 + hl] = (uint8_t)C2[1]; }
            // the consensus strings as planes (every letter is one of ACGT here), per half; then the supports one row per lane: a column of the
            // row mismatches where its code differs or where the row has ended (its character is 0, the consensus holds a letter)
            const u32 k0 = (C2[0] >> 1) & 3u, k1 = (C2[1] >> 1) & 3u;
            const u64 b0l = __ballot(colok && (k0 & 1u)), b0h = __ballot(colok && (k0 & 2u)), b1l = __ballot(colok && (k1 & 1u)), b1h = __ballot(colok && (k1 & 2u));
            const u32 c0l = half ? (u32)(b0l >> 32) : (u32)b0l, c0h = half ? (u32)(b0h >> 32) : (u32)b0h;
            const u32 c1l = half ? (u32)(b1l >> 32) : (u32)b1l, c1h = half ? (u32)(b1h >> 32) : (u32)b1h;
            const u32 colmask = kl >= 32u ? ~0u : (1u << kl) - 1u;
            for (u32 j0 = 0; j0 < nr_max; j0 += 32u) {
                const u32 j = j0 + hl;
                const bool in = j < nrows, second = j >= m0;
                const uint2 row = prow[in ? j : 0u];
                const u32 hv = rhave[in ? j : 0u] & ~TL_ROW_NEAR_END;
                const u32 vm = hv >= 32u ? ~0u : (1u << hv) - 1u;
                const u32 diff = (row.x ^ (second ? c1l : c0l)) | (row.y ^ (second ? c1h : c0h));
                const u32 d = (u32)__popc(((diff & vm) | ~vm) & colmask);
                const bool ok = in && (int)d <= a.max_err;
                supp0 += tl_half_pop(__ballot(ok && !second), half);
                supp1 += tl_half_pop(__ballot(ok && second), half);
            }
        }
    }
  if (!fast2) {
    any_near = __ballot(any_near) != 0ull;
    u32 odd = tl_gather_rows_b(a, ctx, rbase, rhave, nrows, nr_max, hl);   // rows of 32 characters, 8 lanes per row
    if (any_near) { tl_gather_rows_b_bytes(a, ctx, rbase, rhave, nrows, nr_max, hl); odd = 1u; }   // (the text's last bytes: not checked, the sequential loop)
    const bool plain = __ballot(odd != 0u) == 0ull;                 // every character of the wave's rows is one of {0, A, C, G, T}
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < 2; ++s) {
        const u32 n = s ? m1 : m0;
        const u32 n_max = max((u32)__builtin_amdgcn_readlane((int)n, 0), (u32)__builtin_amdgcn_readlane((int)n, 32));
        const uint8_t* rows = ctx + (size_t)(s ? m0 : 0u) * 32;
        const u32 n_col = colok ? n : 0u;                           // rows this lane takes part in: ONE compare per row says whether it counts
        u32 C = 'A';
        bool seq = !plain;
        if (plain) {
            // the four totals of the column, 8 bits each (a list holds at most 150 contexts).  Two rows per trip, both LDS reads in flight
            // together; the row behind the last one of an odd list lies inside the candidate's LDS area (the spare row) and is masked.
            u32 cnt = 0u;
            for (u32 j = 0; j < n_max; j += 2u) {
                u32 ra = (u32)rows[(size_t)j * 32 + hl], rb = (u32)rows[(size_t)(j + 1u) * 32 + hl];   // (read whether the lane has the row or not)
                asm volatile("" : "+v"(ra), "+v"(rb));
#pragma unroll
                for (u32 q = 0; q < 2u; ++q) {
                    const u32 one = j + q < n_col ? 1u : 0u;
                    cnt += one << (((q ? rb : ra) << 2) & 0x18u);   // slot (ch >> 1) & 3: {0, A} 0, C 1, T 2, G 3
                }
            }
            const u32 t0 = cnt & 0xFFu, t1 = (cnt >> 8) & 0xFFu, t2 = (cnt >> 16) & 0xFFu, t3 = cnt >> 24;
            const u32 M = max(max(t0, t1), max(t2, t3));
            const u32 x = cnt ^ (M * 0x01010101u);
            const u32 z = ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x)) & 0x80808080u;   // 0x80 in the bytes that hold M (at least one)
            const u32 slot8 = (u32)__builtin_ctz(z) - 7u;           // the lowest such slot, times 8 (all totals 0: slot 0, 'A', as cons starts)
            C = (0x47544341u >> slot8) & 0xFFu;                     // "ACTG" by slot
            const bool tie = __popc(z) > 1 && M != 0u;
            const u32 r0 = (u32)rows[hl];
            const bool never_lost = (((r0 << 2) & 0x18u) == 0u) & ((cnt & 0xFEFEFE00u) == 0u);   // row 0 is class 0 and no other class occurs twice
            const bool zero_col = hl >= s_minh[wave][half][s];
            const bool redo = n_col != 0u && (tie || (slot8 == 0u && zero_col && (cnt & 0xFFFFFF00u) != 0u && !never_lost));
            seq = __ballot(redo) != 0ull;
        }
        if (seq) {                                                  // cons as the reference runs it (wave-uniform branch)
            u32 bc8 = 0u, cnt = 0u;                                 // bc8 = 8 * base_to_int(C)
            C = 'A';
            for (u32 j = 0; j < n_max; j += 2u) {
                u32 ra = (u32)rows[(size_t)j * 32 + hl], rb = (u32)rows[(size_t)(j + 1u) * 32 + hl];
                asm volatile("" : "+v"(ra), "+v"(rb));
#pragma unroll
                for (u32 q = 0; q < 2u; ++q) {
                    const bool on = j + q < n_col;
                    const u32 ch = on ? (q ? rb : ra) : 0u;
                    const u32 f = ch & 0xDFu, code = (f >> 1) & 3u;  // base_to_int without a branch: the letter f could be, by its low three bits
                    const u32 e = __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, f & 7u) & 0xFFu;
                    const u32 b8 = e == f ? (code ^ (code >> 1)) << 3 : 0u;
                    cnt += on ? 1u << b8 : 0u;
                    // (no `on` here: C's count is the largest at all times, so a lane without a row -- class 0 -- never takes over)
                    if (((cnt >> b8) & 0xFFu) > ((cnt >> bc8) & 0xFFu)) { C = ch; bc8 = b8; }
                }
            }
        }
        uint8_t* const sx = s ? sb : sa;
        sx[hl] = (uint8_t)(colok ? C : 0u);                         // (0 behind k_left: the rows hold 0 there, too)
        sx[32 + hl] = 0;
        if (valid && colok) a.cons[c * 128 + (u64)s * 64 + hl] = (uint8_t)C;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // supports of BOTH alleles in one sweep: lane hl takes row j0 + hl of the candidate's m0 + m1 rows -- its 32 characters against the
    // consensus of its allele, 8 dwords
    for (u32 j0 = 0; j0 < nr_max; j0 += 32u) {
        const u32 j = j0 + hl;
        const bool in = j < nrows, second = j >= m0;
        const uint8_t* row = ctx + (size_t)(in ? j : 0u) * 32;
        const uint8_t* kx = second ? sb : sa;
        const uint4 x0 = *(const uint4*)row, x1 = *(const uint4*)(row + 16), k0 = *(const uint4*)kx, k1 = *(const uint4*)(kx + 16);
        u32 d = tl_nzb(x0.x ^ k0.x, 0u);
        d = tl_nzb(x0.y ^ k0.y, d); d = tl_nzb(x0.z ^ k0.z, d); d = tl_nzb(x0.w ^ k0.w, d);
        d = tl_nzb(x1.x ^ k1.x, d); d = tl_nzb(x1.y ^ k1.y, d); d = tl_nzb(x1.z ^ k1.z, d); d = tl_nzb(x1.w ^ k1.w, d);
        const bool ok = in && (int)d <= a.max_err;
        supp0 += tl_half_pop(__ballot(ok && !second), half);
        supp1 += tl_half_pop(__ballot(ok && second), half);
    }
  }
    bad = tl_half_pop(__ballot(bad & E2S_TAILERR_RANK), half) ? E2S_TAILERR_RANK : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool keep = supp0 > 0u && supp1 > 0u;
    const u32 ca = sa[hl], cb = sb[hl];
    const int none = (int)tl_half_pop(__ballot(colok && ca != cb), half);
    int D = none, L = 0;
    // (gaps longer than `none` cannot matter -- see k_consensus; the bound is the larger of the two halves': for the other half the extra
    //  rounds only see costs above its own `none`, which never become a minimum that is used)
    const int none_other = __shfl_xor(none, 32);
    const int gap_hi = min(a.max_gap, max(none, none_other));
    if (gap_hi > 0) {
        int best_ab = 0, best_ba = 0, ia = 0, ib = 0;
        for (int i = 1; i <= gap_hi; ++i) {
            int mab, mba;
            if ((u32)i <= kl) {
                const u32 la = kl - (u32)i;
                const u32 xi = min(hl + (u32)i, 63u);
                mab = (int)tl_half_pop(__ballot(hl < la && ca != (u32)sb[xi]), half);
                mba = (int)tl_half_pop(__ballot(hl < la && (u32)sa[xi] != cb), half);
            } else mab = mba = none;
            const int dab = mab + i, dba = mba + i;
            if (i == 1 || dab < best_ab) { best_ab = dab; ia = i; }
            if (i == 1 || dba < best_ba) { best_ba = dba; ib = i; }
        }
        if (!(none < best_ab && none < best_ba)) {
            if (best_ab < best_ba) { D = best_ab - ia; L = ia; } else { D = best_ba - ib; L = -ib; }
        }
    }
    u32 right_len = 0; u64 right_at = 0;
    if (keep) {
        if ((u64)right_idx >= a.n_reads) bad |= E2S_TAILERR_RANK;
        else {
            const u32 len = a.rstart ? a.rlen[right_idx] : a.stride;
            if (right_pos > len) bad |= E2S_TAILERR_CRASH;
            else {
                const u32 avail = len - right_pos;
                right_len = a.kr < 0 ? avail : min(avail, (u32)a.kr);
                right_at = (a.rstart ? a.rstart[right_idx] : (u64)right_idx * a.stride) + right_pos;
            }
        }
    }
    const bool emit = keep && D <= 3;
    if (hl == 0 && valid) {
        const u32 kind = L != 0 ? 5u : 3u, tlen = L == 0 ? 3u : (u32)(L < 0 ? -L : L) + 1u, drl = tl_digits32(right_len);
        const u32 l1 = 1u + kind + 13u + 5u + drl + 1u + tlen + 1u + tl_digits32(supp0) + 10u;
        const u32 l3 = 1u + kind + 12u + 5u + drl + 1u + tlen + 1u + tl_digits32(supp1) + 10u;
        const u32 l2 = kl - (L < 0 ? (u32)(-L) : 0u) + right_len + 1u, l4 = kl - (L > 0 ? (u32)L : 0u) + right_len + 1u;
        TailRec t;
        t.emit = (emit ? 1u : 0u) | (keep ? 2u : 0u); t.L = L; t.s0 = supp0; t.s1 = supp1; t.right_len = right_len; t.base_len = l1 + l2 + l3 + l4; t.right_at = right_at;
        a.rec[c] = t;
        if (!a.by_slot) {
            if (keep) atomicAdd(&a.part[3 * (c >> 6) + 2], 1ull);
            if (emit) { atomicAdd(&a.part[3 * (c >> 6)], 1ull); atomicAdd(&a.part[3 * (c >> 6) + 1], (u64)t.base_len); }
        }
        if (bad) atomicOr(a.err, bad);
    }
}

// ---- the group sums where the consensus ran by slot (beside the ordering pass): group g = the SA-ordered candidates 64 g .. 64 g + 63
__global__ __launch_bounds__(256) void k_tail_group_sums(TailArgs a, u32 ng) {
    const u32 lane = tl_lane();
    const u64 g = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ng) return;
    if (a.n_cand_dev) a.n_cand = min((u64)a.n_cand, (u64)*a.n_cand_dev);
    const u64 c = g * 64 + lane;
    u32 em = 0, bl = 0;
    if (c < a.n_cand) { const TailRec* t = a.rec + a.slot_of[c]; em = t->emit; bl = (em & 1u) ? t->base_len : 0u; }
    const u64 ev = __ballot((em & 1u) != 0u), kp = __ballot((em & 2u) != 0u);
    for (int o = 32; o > 0; o >>= 1) bl += __shfl_xor(bl, o);
    if (lane == 0) { a.part[3 * g] = (u64)__popcll(ev); a.part[3 * g + 1] = (u64)bl; a.part[3 * g + 2] = (u64)__popcll(kp); }
}
void e2s_launch_tail_group_sums(const TailArgs& a, uint32_t ng, hipStream_t st) {
    if (!ng) return;
    hipLaunchKernelGGL(k_tail_group_sums, dim3((ng + 3) / 4), dim3(256), 0, st, a, ng);
}

// ---- K3b: exclusive scan of the per-group (events, text bytes) pairs; totals behind the last group -------------------------
// One block: a thread owns a run of consecutive groups (loads in flight together), the 1024 partial sums are scanned by one wave
// (16 per lane + a wave scan).  (A Hillis-Steele scan over 1024-group slabs took 30 us for 4064 groups: 80 barriers.)
__device__ __forceinline__ u64 tl_wave_incl_scan64(u64 x, u32 lane) {
    for (u32 d = 1; d < 64; d <<= 1) { const u64 v = __shfl_up((unsigned long long)x, d); if (lane >= d) x += v; }
    return x;
}
// jd / blob2 (the resident job): the totals go into the job's device record as well, and (events, ok) into this rank's share of the
// second all-gather -- ok = no error bit of the tail and none in the words the per-cluster pass leaves (ok_words: its own and the pre-filter's)
__global__ __launch_bounds__(1024) void k_tail_partials(const unsigned long long* part, unsigned long long* part_excl, u32 ng, u64 first_id,
                                                        unsigned long long* totals /* events, bytes, kept, error bits */, const unsigned long long* n_keep,
                                                        const u32* err, JobDev* jd, unsigned long long* blob2, const u32* bad_a, const u32* bad_b, const u32* bad_c,
                                                        const unsigned long long* n_cand_dev) {
    __shared__ u64 s_e[1024], s_b[1024], s_k[16];
    const u32 tid = threadIdx.x, lane = tl_lane();
    if (n_cand_dev) ng = (u32)min((u64)ng, (*n_cand_dev + 63ull) / 64ull);     // (the job sizes the launch for the slots there are: only the groups in use)
    const u32 per = (ng + 1023u) / 1024u, lo = min(ng, tid * per), hi = min(ng, lo + per);
    u64 se = 0, sb = 0, sk = 0;
    for (u32 g = lo; g < hi; ++g) { se += part[3 * g]; sb += part[3 * g + 1]; sk += part[3 * g + 2]; }
    s_e[tid] = se; s_b[tid] = sb;
    for (int o = 32; o > 0; o >>= 1) sk += __shfl_xor((unsigned long long)sk, o);
    if (lane == 0) s_k[tid >> 6] = sk;
    __syncthreads();
    if (tid < 64) {
        u64 le = 0, lb = 0;
        for (int i = 0; i < 16; ++i) { le += s_e[tid * 16 + i]; lb += s_b[tid * 16 + i]; }
        u64 re = tl_wave_incl_scan64(le, lane) - le, rb = tl_wave_incl_scan64(lb, lane) - lb;
        for (int i = 0; i < 16; ++i) {
            const u64 ve = s_e[tid * 16 + i], vb = s_b[tid * 16 + i];
            s_e[tid * 16 + i] = re; s_b[tid * 16 + i] = rb;
            re += ve; rb += vb;
        }
        if (tid == 63) {
            u64 k = 0;
            for (int i = 0; i < 16; ++i) k += s_k[i];
            totals[0] = re;
            totals[1] = rb + 2ull * (tl_digit_sum_below(first_id + re) - tl_digit_sum_below(first_id));
            totals[2] = k;
            totals[3] = (u64)*err;
            if (jd) {
                jd->events = re; jd->snp_bytes = totals[1]; jd->n_keep = k; jd->tail_err = *err; jd->first_id = first_id; jd->events_before = 0; jd->peers_not_ok = 0;
                // (bad_b is the plan's stat_rc: its low bits say where the reference's statistics() crashes -- every rank sees the same
                //  merged statistics, so that is a result, not a reason for the ranks to redo the step)
                const u32 ok = (*err == 0u && *bad_a == 0u && (*bad_b & ~E2S_JOBERR_CRASH_MASK) == 0u && *bad_c == 0u) ? 1u : 0u;
                jd->my_ok = ok;
                if (blob2) { blob2[0] = re; blob2[1] = (u64)ok; }
            }
        }
    }
    __syncthreads();
    u64 re = s_e[tid], rb = s_b[tid];
    for (u32 g = lo; g < hi; ++g) {
        part_excl[2 * g] = re; part_excl[2 * g + 1] = rb;
        re += part[3 * g]; rb += part[3 * g + 1];
    }
}

// ---- K3c: the text --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 tl_put(uint8_t* b, u32 at, const char* s) { while (*s) b[at++] = (uint8_t)*s++; return at; }
__device__ __forceinline__ u32 tl_put_num(uint8_t* b, u32 at, u64 v) {
    const u32 d = tl_digits(v);
    for (u32 i = 0; i < d; ++i) { b[at + d - 1 - i] = (uint8_t)('0' + (u32)(v % 10ull)); v /= 10ull; }
    return at + d;
}

// A block takes a group of 64 candidates (the unit of the scan of event counts and text lengths).  The header lines are serial
// work (digits, pieces of varying length): ONE THREAD PER HEADER builds them -- wave 0 the 64 ">..._higher_path_..." lines, wave 1 the
// 64 "lower" ones, every lane its own candidate's -- into LDS; then each of the 16 waves copies four events out, lanes = characters.
// (First version: lane 0 of a wave built the two headers of one event while 63 lanes waited: ~700 vector instructions per event,
// and the kernel was bound by their issue -- 0.42 ms for 2.6x10^5 events on c3.)
constexpr u32 TL_HDR_STRIDE = 148;                                  // >= the longest header (141), 37 words: lanes hit different banks
// ids of the events of a rank: first_id = 1 + the events of the ranks before it (second all-gather: (events, ok) per rank); the size of
// its text follows in closed form (only the id digits depend on first_id)
// ids2 (the resident job on several ranks): the gathered (events, ok) pairs -- every block derives the rank's first id from them (a
// launch of its own for that was 4 us + a launch gap on the critical path), block 0 completes the job's record
// mail_src / mail_dst / mail_words (the resident job): block 0 also delivers the step's mail -- the scan's descriptor and statistics and the
// job's record, all written by EARLIER kernels of the stream (and, just above, by this block's first thread) -- into the host's pinned
// mailbox: the copy engine's turn behind the last kernel (a launch gap + ~4 us) is gone, and the host's one wait is for this kernel
// (launch bounds: 8 waves per SIMD = 64 VGPRs -- at 66 the compiler's answer was 7 waves per SIMD, i.e. ONE 16-wave block per CU instead of two: 97 -> 144 us on c3)
__global__ __launch_bounds__(1024, 8) void k_tail_write(TailArgs a, const unsigned long long* part_excl, u64 first_id, const unsigned long long* first_id_dev, uint8_t* out,
                                                     u64 out_cap, const unsigned long long* ids2, int world, int rank, JobDev* jd,
                                                     const unsigned long long* mail_src, unsigned long long* mail_dst, u32 mail_words) {
    if (first_id_dev) first_id = *first_id_dev;
    if (ids2) {
        u64 before = 0; u32 bad = 0;
        for (int r = 0; r < world; ++r) { if (r < rank) before += ids2[2 * r]; if (ids2[2 * r + 1] == 0ull) bad = 1u; }
        first_id = 1ull + before;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const u64 ne = jd->events, old_id = jd->first_id;
            const u64 text = jd->snp_bytes - 2ull * (tl_digit_sum_below(old_id + ne) - tl_digit_sum_below(old_id));
            jd->snp_bytes = text + 2ull * (tl_digit_sum_below(first_id + ne) - tl_digit_sum_below(first_id));
            jd->first_id = first_id; jd->events_before = before; jd->peers_not_ok = bad;
        }
    }
    if (mail_dst && blockIdx.x == 0) {
        __syncthreads();                                            // (thread 0's completion of the record, above)
        for (u32 w = threadIdx.x; w < mail_words; w += 1024u) mail_dst[w] = __hip_atomic_load(mail_src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.n_cand_dev) a.n_cand = min((u64)a.n_cand, (u64)*a.n_cand_dev);
    if ((u64)blockIdx.x * 64ull >= a.n_cand) return;
    __shared__ __attribute__((aligned(4))) uint8_t s_hdr[2][64][TL_HDR_STRIDE];
    __shared__ uint8_t s_hlen[2][64];
    const u32 lane = tl_lane(), wave = threadIdx.x >> 6;
    const u64 g = blockIdx.x;
    TailRec t; t.emit = 0; t.base_len = 0; t.L = 0; t.s0 = t.s1 = t.right_len = 0; t.right_at = 0;
    u64 slot = g * 64 + lane;                                       // where this lane's candidate has its record and consensus strings
    {
        const u64 c = g * 64 + lane;
        if (c < a.n_cand) { if (a.slot_of) slot = a.slot_of[c]; t = a.rec[slot]; }
        t.emit &= 1u;
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
    if (wave < 2 && t.emit) {
        const u32 line = wave;
        const uint8_t* ca = a.cons + slot * 128;                    // consensus of allele 0, of allele 1 at + 64
        uint8_t* b = s_hdr[line][lane];
        const int L = t.L;
        u32 at = 0;
        b[at++] = '>';
        // (every literal in a call of its own: a literal CHOSEN at run time is a pointer into constant memory, and tl_put then loads it
        //  one character at a time -- seventeen dependent loads per header line)
        if (L != 0) at = tl_put(b, at, "INDEL"); else at = tl_put(b, at, "SNP");
        if (line) at = tl_put(b, at, "_lower_path_"); else at = tl_put(b, at, "_higher_path_");
        at = tl_put_num(b, at, first_id + my_rank);
        at = tl_put(b, at, "|P_1:");
        at = tl_put_num(b, at, t.right_len);
        b[at++] = '_';
        if (L == 0) { b[at++] = ca[kl - 1]; b[at++] = '/'; b[at++] = ca[64 + kl - 1]; }          // last consensus letters (H20)
        else if (L > 0) { for (u32 i = kl - (u32)L; i < kl; ++i) b[at++] = ca[i]; b[at++] = '/'; }
        else { b[at++] = '/'; for (u32 i = kl - (u32)(-L); i < kl; ++i) b[at++] = ca[64 + i]; }
        b[at++] = '|';
        at = tl_put_num(b, at, line ? t.s1 : t.s0);
        at = tl_put(b, at, "|nb_pol_1\n");
        s_hlen[line][lane] = (uint8_t)at;
    }
    __syncthreads();
    // a wave copies four events: what they read from global memory (consensus strings, right context) is fetched for all four behind the
    // barrier, BEFORE the first is written (the loop used to pay one round trip per event, four in a row)
    u32 ach4[4], bch4[4], rch4[4];
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const int src = (int)(wave * 4 + k);
        const bool on = (em >> src) & 1ull;                         // (wave-uniform)
        const u64 c2 = __shfl((unsigned long long)slot, src);
        const u32 rl = __shfl(t.right_len, src);
        const u64 rat = __shfl((unsigned long long)t.right_at, src);
        const uint8_t* ca = a.cons + c2 * 128;
        ach4[k] = on && lane < kl ? (u32)ca[lane] : 0u;
        bch4[k] = on && lane < kl ? (u32)ca[64 + lane] : 0u;
        rch4[k] = on && lane < rl ? (u32)a.text[rat + lane] : 0u;
    }
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const int src = (int)(wave * 4 + k);
        if (!((em >> src) & 1ull)) continue;                        // (wave-uniform)
        const int L = __shfl(t.L, src);
        const u32 rl = __shfl(t.right_len, src);
        const u64 rat = __shfl((unsigned long long)t.right_at, src), o = __shfl((unsigned long long)my_off, src);
        const u32 ach = ach4[k], bch = bch4[k], rch = rch4[k];
        const u32 n1 = s_hlen[0][src], n3 = s_hlen[1][src];
        const u32 skip0 = L < 0 ? (u32)(-L) : 0u, skip1 = L > 0 ? (u32)L : 0u;
        const u32 l2 = kl - skip0 + rl + 1u, l4 = kl - skip1 + rl + 1u;
        if (o + n1 + l2 + n3 + l4 <= out_cap) {
            uint8_t* p = out + o;
            for (u32 i = lane; i < n1; i += 64) p[i] = s_hdr[0][src][i];
            p += n1;
            if (lane >= skip0 && lane < kl) p[lane - skip0] = (uint8_t)ach;
            if (lane < rl) p[kl - skip0 + lane] = (uint8_t)rch;
            for (u32 i = lane + 64u; i < rl; i += 64) p[kl - skip0 + i] = a.text[rat + i];
            if (lane == 0) p[kl - skip0 + rl] = '\n';
            p += l2;
            for (u32 i = lane; i < n3; i += 64) p[i] = s_hdr[1][src][i];
            p += n3;
            if (lane >= skip1 && lane < kl) p[lane - skip1] = (uint8_t)bch;
            if (lane < rl) p[kl - skip1 + lane] = (uint8_t)rch;
            for (u32 i = lane + 64u; i < rl; i += 64) p[kl - skip1 + i] = a.text[rat + i];
            if (lane == 0) p[kl - skip1 + rl] = '\n';
        }
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
__global__ __launch_bounds__(256) void k_reads_extent(const u64* rstart, const u32* rlen, u64 n_reads, unsigned long long* out) {
    u64 m = 0;
    for (u64 r = (u64)blockIdx.x * 256 + threadIdx.x; r < n_reads; r += (u64)gridDim.x * 256) m = max(m, rstart[r] + rlen[r]);
    for (int d = 32; d; d >>= 1) m = max(m, (u64)__shfl_xor((unsigned long long)m, d));
    if (tl_lane() == 0 && m) atomicMax(out, (unsigned long long)m);
}
// ---- the reads as two bit planes per 32 characters (TailArgs::text2): thread t takes characters 32 t .. 32 t + 31
__global__ __launch_bounds__(256) void k_reads_pack2(const uint8_t* text, u64 text_bytes, u32 stride, u64 n_reads, uint2* out, u32* impure) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x, at = t * 32ull;
    if (at >= text_bytes) return;
    u32 lo = 0u, hi = 0u;
    // (a group inside the text of a 16-byte aligned buffer -- every group but the last, usually -- comes as two 16-byte loads)
    const bool whole = at + 32ull <= text_bytes && (((uintptr_t)text) & 15u) == 0u;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (whole) { q0 = *(const uint4*)(text + at); q1 = *(const uint4*)(text + at + 16); }
#pragma unroll
    for (u32 k = 0; k < 8u; ++k) {
        const u64 p = at + 4ull * k;
        u32 w = 0u, have = 0u;                                      // (the last group of the text: byte by byte)
        if (whole) { w = k == 0 ? q0.x : k == 1 ? q0.y : k == 2 ? q0.z : k == 3 ? q0.w : k == 4 ? q1.x : k == 5 ? q1.y : k == 6 ? q1.z : q1.w; have = 4u; }
        else if (p + 4ull <= text_bytes) { w = *(const u32*)(text + p); have = 4u; }
        else for (; p + have < text_bytes; ++have) w |= (u32)text[p + have] << (8u * have);
        const u32 x = w ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, w & 0x07070707u);   // zero byte <=> one of "ACGT" (see tl_gather_rows_b)
        const u32 badm = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;               // 0x80 per byte that is not
        const u32 okb = ((badm ^ 0x80808080u) >> 7) * 0xFFu;                                  // 0xFF per byte that is
        const u32 c = (w >> 1) & 0x03030303u & okb;
        lo |= (((c & 0x01010101u) * 0x10204080u) >> 28) << (4u * k);
        hi |= ((((c >> 1) & 0x01010101u) * 0x10204080u) >> 28) << (4u * k);
        if (impure && stride) {                                     // (reads of one length: the read follows from the position)
            for (u32 i = 0; i < have; ++i) if ((badm >> (8u * i + 7u)) & 1u) {
                const u64 rd = (p + i) / stride;
                if (rd < n_reads) { atomicOr(&impure[rd >> 5], 1u << (rd & 31u)); impure[(n_reads + 31) / 32] = 1u; }   // (the word behind the bits: any at all)
            }
        }
    }
    out[t] = make_uint2(lo, hi);
}
// reads of their own lengths: which of them hold a character outside "ACGT" (a thread per read)
__global__ __launch_bounds__(256) void k_reads_impure(const uint8_t* text, const u64* rstart, const u32* rlen, u64 n_reads, u32* impure) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint8_t* p = text + rstart[r];
    const u32 n = rlen[r];
    bool bad = false;
    for (u32 i = 0; i < n; ++i) { const u32 ch = p[i]; bad |= !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
    if (bad) { atomicOr(&impure[r >> 5], 1u << (r & 31u)); impure[(n_reads + 31) / 32] = 1u; }
}
void e2s_launch_reads_pack2(const uint8_t* text, uint64_t text_bytes, const uint64_t* rstart, const uint32_t* rlen, uint32_t stride, uint64_t n_reads,
                            uint2* out, uint32_t* impure, hipStream_t st) {
    if (!text_bytes) return;
    const uint64_t groups = (text_bytes + 31) / 32;
    hipLaunchKernelGGL(k_reads_pack2, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, text, (u64)text_bytes, rstart ? 0u : stride, (u64)n_reads, out, impure);
    if (rstart && n_reads) hipLaunchKernelGGL(k_reads_impure, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, text, (const u64*)rstart, rlen, (u64)n_reads, impure);
}
void e2s_launch_reads_extent(const uint64_t* rstart, const uint32_t* rlen, uint64_t n_reads, unsigned long long* out, hipStream_t st) {
    if (!n_reads) return;
    const uint64_t nb = (n_reads + 255) / 256;
    hipLaunchKernelGGL(k_reads_extent, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, st, (const u64*)rstart, rlen, n_reads, out);
}
void e2s_launch_reads_compact(const uint8_t* text, const uint64_t* rstart, uint32_t len, uint64_t n_reads, uint8_t* out, hipStream_t st) {
    if (!n_reads) return;
    hipLaunchKernelGGL(k_reads_compact, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, st, text, (const u64*)rstart, len, n_reads, out);
}
// LDS of a CU on the current device (gfx950: 160 KB); the consensus kernels' waves per block follow from it, not from constants
static size_t tl_device_lds_bytes(int dev) {
    static size_t cache[E2S_MAX_DEVICES] = {};
    size_t& c = cache[dev % E2S_MAX_DEVICES];
    if (!c) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || v <= 0) v = 64 * 1024;
        c = (size_t)v;
    }
    return c;
}
static void tl_consensus_shape(int kl, uint32_t list_cap, size_t lds_cu, size_t* per_wave, unsigned* wpb) {
    if (list_cap < 1) list_cap = 1;
    *per_wave = kl <= 32 ? 2 * tl_pair_cand_lds(list_cap) : (size_t)2 * list_cap * (128 + 8 + 4);
    // (four waves per block while six such blocks fit a CU, two while three do: 60 / 120 KB of gfx950's 160)
    *wpb = 4 * *per_wave <= lds_cu * 3 / 8 ? 4u : 2 * *per_wave <= lds_cu * 3 / 4 ? 2u : 1u;
}
// can the consensus kernels hold -c lists of this length in the device's LDS?  (no: the host tail takes over)
bool e2s_consensus_fits(int kl, uint32_t list_cap) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t per_wave; unsigned wpb;
    tl_consensus_shape(kl, list_cap, tl_device_lds_bytes(dev), &per_wave, &wpb);
    return wpb * per_wave + 2048 <= tl_device_lds_bytes(dev);       // (+ the kernels' static LDS)
}
void e2s_launch_consensus(const TailArgs& a, uint32_t list_cap, hipStream_t st) {
    if (!a.n_cand) return;
    if (list_cap < 1) list_cap = 1;
    // (a kernel's dynamic-LDS limit is set per device: the cache of what has been set is per device, too)
    static size_t lds_set_dev[E2S_MAX_DEVICES][2] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t* const lds_set = lds_set_dev[dev % E2S_MAX_DEVICES];
    size_t per_wave; unsigned wpb;
    tl_consensus_shape(a.kl, list_cap, tl_device_lds_bytes(dev), &per_wave, &wpb);
    const size_t lds = wpb * per_wave;
    if (a.kl <= 32) {                                               // two candidates per wave
        if (lds > 48 * 1024 && lds > lds_set[1]) { (void)hipFuncSetAttribute((const void*)k_consensus_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set[1] = lds; }
        const unsigned long long waves = (a.n_cand + 1) / 2;
        hipLaunchKernelGGL(k_consensus_pair, dim3((unsigned)((waves + wpb - 1) / wpb)), dim3(64 * wpb), lds, st, a, list_cap);
        return;
    }
    if (lds > 48 * 1024 && lds > lds_set[0]) { (void)hipFuncSetAttribute((const void*)k_consensus, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set[0] = lds; }
    hipLaunchKernelGGL(k_consensus, dim3((unsigned)((a.n_cand + wpb - 1) / wpb)), dim3(64 * wpb), lds, st, a, list_cap);
}
void e2s_launch_tail_partials(const unsigned long long* part, unsigned long long* part_excl, uint32_t ng, unsigned long long first_id,
                              unsigned long long* totals, const unsigned long long* n_keep, const uint32_t* err, JobDev* jd, unsigned long long* blob2,
                              const uint32_t* bad_a, const uint32_t* bad_b, const uint32_t* bad_c, const unsigned long long* n_cand_dev, hipStream_t st) {
    hipLaunchKernelGGL(k_tail_partials, dim3(1), dim3(1024), 0, st, part, part_excl, ng, first_id, totals, n_keep, err, jd, blob2, bad_a, bad_b, bad_c, n_cand_dev);
}
void e2s_launch_tail_write(const TailArgs& a, const unsigned long long* part_excl, unsigned long long first_id, const unsigned long long* first_id_dev,
                           uint8_t* out, unsigned long long out_cap, hipStream_t st, const unsigned long long* ids2, int world, int rank, JobDev* jd,
                           const unsigned long long* mail_src, unsigned long long* mail_dst, uint32_t mail_words) {
    if (!a.n_cand && !mail_dst) return;
    const unsigned long long groups = (a.n_cand + 63) / 64;
    hipLaunchKernelGGL(k_tail_write, dim3((unsigned)(groups ? groups : 1)), dim3(1024), 0, st, a, part_excl, first_id, first_id_dev, out, out_cap, ids2, world, rank, jd,
                       mail_src, mail_dst, mail_words);
}
