// e2s_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ebwt2snp hot path.
//
//   K1  k_scan_seg<X,FUSED>      cluster-boundary scan over the LCP array (cluster_lm + append_entry, ebwt2clust.cpp:54-139); FUSED: with
//                                clust2snp's pre-filter in the same pass.  Every wave a worker over a contiguous range of 4-KiB tiles, records
//                                into per-worker segments
//       k_seg_fix                  one block after the scan: the shard's descriptor (ScanMail), the segments' offsets and the statistics
//       k_pack_seg_write           the `.clusters` image from the segments
//   K2a k_cluster_filter2 / k_record_prefilter   the pre-filter as a kernel of its own     (find_events, clust2snp.cpp:788-857)
//   K2b k_flagged_lists<DEF>     exact per-cluster test, lists, packed candidates          (find_variants, clust2snp.cpp:367-500)
//   KO  k_ko_*                   SA order of the candidates
//   KU  k_unpack_*               BCR pairSA / .gesa AoS -> SoA in HBM                       (read_el, include.hpp:120-199)
//
// Integer streaming only: no MFMA.  Layout, roofline and byte counts are in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "e2s_internal.h"
#include "e2s_plan.h"
#include <type_traits>
#include "e2s_swar.h"
#include "e2s_env.h"


// ----------------------------------------------------------------------------------------------
// wave64 helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// inclusive add-scan across the 64 lanes of a wave (DPP row shifts + row broadcasts)
// (bound_ctrl on the row shifts: lanes without a source add 0, and the compiler folds each step into ONE v_add_u32_dpp --
// with bound_ctrl off it emits v_mov / v_mov_dpp / v_add per step, three times the instructions)
__device__ __forceinline__ u32 wave_incl_scan(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return x;
}
__device__ __forceinline__ u32 wave_bcast_last(u32 x) { return (u32)__builtin_amdgcn_readlane((int)x, 63); }

// ----------------------------------------------------------------------------------------------
// K1 (scan_mode 0): segmented single-pass scan.  A persistent block owns a CONTIGUOUS range of tiles
// (one tile = 256 thread-rows = 4 KiB of LCP), walks it front to back carrying its running record count
// and its open cluster start, and writes the records that END in its range into its own output segment.
// No block ever waits for another one: the segments are tied together afterwards by k_seg_fix
// (<= ~1300 descriptors, one thread).  One barrier per tile; the next tile's loads are in flight meanwhile.
// ----------------------------------------------------------------------------------------------

constexpr int SEG_STAGE = 256;                              // records of one wave-tile staged in LDS (more: written directly)
constexpr int SPAN_WORDS = E2S_SPAN_BYTES / 4;               // 16 words = 64 B of LCP per thread and tile

#ifndef E2S_SCAN_PREFETCH
#define E2S_SCAN_PREFETCH 1
#endif
#ifndef E2S_SCAN_WAVES
#define E2S_SCAN_WAVES 1
#endif
// Every WAVE is its own worker: it owns a contiguous range of wave-tiles (64 lanes x 64 B = 4 KiB of LCP), carries
// its running record count and its open cluster start in registers and writes into its own output segment.  No
// LDS, no barrier, nothing shared between waves: the scheduler interleaves them freely.
//
// The kernel sits at the memory system's ceiling for its read / write mix (~5 TB/s whatever the
// write pattern; tools/stream_probe.hip, tools/write_probe.hip), with the instruction stream kept short enough to stay there:
//   * per word only the two predicates (L >= k, L[j] >= L[j-1]) are evaluated SWAR and packed into 64-bit planes
//     with v_dot4; the start/end stencil runs on the planes (e2s_swar.h, span_pred / plane_end / plane_start);
//     the one bit pair a lane needs from its left neighbour travels by DPP wave_shr:1;
//   * END bits and START bits are walked separately (ctz / clear-lowest loops dropping offsets into rank-addressed LDS slots);
//     a ballot finds the tile's last start -- no scan across the lanes on the usual path.
__device__ __forceinline__ u32 wave_incl_max(u32 x) {
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));   // row_shr:1
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));   // row_shr:2
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));   // row_shr:4
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));   // row_shr:8
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1,3
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2,3
    return x;
}
// value of lane-1 (lane 0 receives `first`)
__device__ __forceinline__ u32 wave_shr1(u32 x, u32 first) {
    return (u32)__builtin_amdgcn_update_dpp((int)first, (int)x, 0x138, 0xf, 0xf, false);    // wave_shr:1
}

// What the pre-filter inherits at the start of a worker's range (local entry r0 >= 4096): the class code of the entry before it
// and its view of a cluster that reaches into the range -- start, base counts, whether it holds a class change so far.  A
// cluster that matters is at most 150 entries long, so the 192 entries before the range are enough: three entries per lane,
// ballots and scalar population counts (a whole warm-up tile through the plane code cost 8 % of a 12-tile worker).
// Offsets count from the tile before the range (open1 = 1 + offset of the last start among the 192 entries, 0: none).
struct PfCarry { u32 open1, cin0, ccarry, cb; };
// Every byte is loaded FIRST, all sixteen loads in flight together, and only then looked at: written with && / || the compiler evaluated the
// rules lazily -- a load, a wait, a branch, the next load: thirteen memory round trips one after the other at the head of every range,
// 10-28 us at the loaded memory system's latency (profiles/round5_scan_ranges.txt).  cb: what range_carry() returns for the same range
// (the entries r0 - 2 .. r0 are among the bytes read here).
__device__ __forceinline__ PfCarry prefilter_carry_192(const uint8_t* lcp, const uint8_t* bwt, const int64_t r0, const u32 k_u32, const u32 hge, const u32 lane) {
    auto cls = [](u32 x) -> u32 { x &= 0xDFu; return x == 'C' ? 1u : x == 'G' ? 3u : x == 'T' ? 2u : 0u; };   // bit 0 = C|G, bit 1 = G|T (span_bwt_code_planes)
    u64 Sm[3], Cm[3], Gm[3], Tm[3], Fm[3];
    PfCarry r{0u, 0u, 0u, 0u};
    u32 L2[3], L1[3], L0[3], Bc[3], Bp[3];
    const uint8_t* const pl = lcp + (r0 - 192 + (int64_t)lane);
    const uint8_t* const pb = bwt + (r0 - 192 + (int64_t)lane);
#pragma unroll
    for (int p = 0; p < 3; ++p) { L2[p] = pl[64 * p - 2]; L1[p] = pl[64 * p - 1]; L0[p] = pl[64 * p]; Bc[p] = pb[64 * p]; Bp[p] = pb[64 * p - 1]; }
    u32 lr0 = lcp[r0];                                          // (the range's first entry, the same in every lane)
    asm volatile("" : "+v"(L2[0]), "+v"(L1[0]), "+v"(L0[0]), "+v"(Bc[0]), "+v"(Bp[0]), "+v"(L2[1]), "+v"(L1[1]), "+v"(L0[1]), "+v"(Bc[1]), "+v"(Bp[1]),
                      "+v"(L2[2]), "+v"(L1[2]), "+v"(L0[2]), "+v"(Bc[2]), "+v"(Bp[2]), "+v"(lr0));
    const bool never = hge == 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const u32 lm2 = L2[p], lm1 = L1[p], l0 = L0[p];
        const bool gm1 = !never & (lm1 >= k_u32), g0 = !never & (l0 >= k_u32);
        const bool endm1 = gm1 & (((lm2 > lm1) & (lm1 <= l0)) | !g0);
        const u32 c = cls(Bc[p]), cp = cls(Bp[p]);
        Sm[p] = __ballot(g0 & !(gm1 & !endm1));
        Cm[p] = __ballot(c == 1u); Gm[p] = __ballot(c == 3u); Tm[p] = __ballot(c == 2u);
        Fm[p] = __ballot(c != cp);
        if (p == 2) r.ccarry = (u32)__builtin_amdgcn_readlane((int)c, 63);      // C = 1, G = 3, T = 2: bit 0 = C|G, bit 1 = G|T
    }
    {   // range_carry's value: GE / end of entry r0 - 1 (lane 63 of the last pass holds entries r0 - 3 .. r0 - 1)
        const u32 lm2 = L1[2], lm1 = L0[2], l0 = lr0;
        const bool gm1 = !never & (lm1 >= k_u32), g0 = !never & (l0 >= k_u32);
        const bool em1 = gm1 & (((lm2 > lm1) & (lm1 <= l0)) | !g0);
        r.cb = (u32)__builtin_amdgcn_readlane((int)((gm1 ? 1u : 0u) | (em1 ? 2u : 0u)), 63);
    }
    int sp = -1;                                                // the last start among the 192 entries
#pragma unroll
    for (int p = 2; p >= 0; --p) if (sp < 0 && Sm[p]) sp = 64 * p + 63 - __clzll(Sm[p]);
    if (sp >= 0) {
        u32 cC = 0, cG = 0, cT = 0; bool fl = false;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int lo = sp - 64 * p;                                 // entries of this pass at or after the start
            const u64 m = lo >= 64 ? 0ull : (lo <= 0 ? ~0ull : (~0ull << lo));
            const u64 mf = lo >= 63 ? 0ull : (lo < 0 ? ~0ull : (~0ull << (lo + 1)));   // strictly after the start
            cC += (u32)__popcll(Cm[p] & m); cG += (u32)__popcll(Gm[p] & m); cT += (u32)__popcll(Tm[p] & m);
            fl = fl || (Fm[p] & mf) != 0ull;
        }
        r.open1 = 1u + (u32)(4096 - 192 + sp);
        r.cin0 = cC | (cG << 8) | (cT << 16) | (fl ? (1u << 25) : 0u);
    }
    return r;
}

// The scan's own look at the 192 entries before a range (the kernel without the pre-filter; any entry width): x = GE / end of the entry
// before the range (range_carry's value), y = 1 + the offset of the LAST cluster start among the 192, counted from the first entry of the
// tile before the range (0: none).  All ten loads first, as above.
template <int X>
__device__ __forceinline__ uint2 scan_lookback_192(const uint8_t* base, const int64_t r0, const u32 k_u32, const u32 hge, const u32 lane, const u32 tile_entries) {
    auto ld = [&](int64_t i) -> u32 { return X == 1 ? (u32)base[i] : X == 2 ? (u32)((const uint16_t*)base)[i] : ((const u32*)base)[i]; };
    u32 L2[3], L1[3], L0[3];
    const int64_t e0 = r0 - 192 + (int64_t)lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) { L2[p] = ld(e0 + 64 * p - 2); L1[p] = ld(e0 + 64 * p - 1); L0[p] = ld(e0 + 64 * p); }
    u32 lr0 = ld(r0);
    asm volatile("" : "+v"(L2[0]), "+v"(L1[0]), "+v"(L0[0]), "+v"(L2[1]), "+v"(L1[1]), "+v"(L0[1]), "+v"(L2[2]), "+v"(L1[2]), "+v"(L0[2]), "+v"(lr0));
    const bool never = hge == 0;
    u64 Sm[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const u32 lm2 = L2[p], lm1 = L1[p], l0 = L0[p];
        const bool gm1 = !never & (lm1 >= k_u32), g0 = !never & (l0 >= k_u32);
        const bool endm1 = gm1 & (((lm2 > lm1) & (lm1 <= l0)) | !g0);
        Sm[p] = __ballot(g0 & !(gm1 & !endm1));
    }
    uint2 r = make_uint2(0u, 0u);
    {
        const u32 lm2 = L1[2], lm1 = L0[2], l0 = lr0;
        const bool gm1 = !never & (lm1 >= k_u32), g0 = !never & (l0 >= k_u32);
        const bool em1 = gm1 & (((lm2 > lm1) & (lm1 <= l0)) | !g0);
        r.x = (u32)__builtin_amdgcn_readlane((int)((gm1 ? 1u : 0u) | (em1 ? 2u : 0u)), 63);
    }
    int sp = -1;
#pragma unroll
    for (int p = 2; p >= 0; --p) if (sp < 0 && Sm[p]) sp = 64 * p + 63 - __clzll(Sm[p]);
    if (sp >= 0) r.y = 1u + (tile_entries - 192u + (u32)sp);
    return r;
}

constexpr int F2_Q = 64;                     // flagged clusters a wave queues in LDS between global reservations (pre-filter)
// packed base counts C | G << 8 | T << 16 (added) and flags << 24 (or-ed)
__device__ __forceinline__ u32 pf_merge(u32 x, u32 y) { return (((x & 0xFFFFFFu) + (y & 0xFFFFFFu)) & 0xFFFFFFu) | ((x | y) & 0xFF000000u); }

// FUSED (X == 1 only, e2s_prefilter_arm): the pre-filter of the per-cluster pass (k_cluster_filter2 below, which
// documents the method) runs inside the scan on the same GE / NLE / start / end planes -- the LCP is read once, the
// stencil evaluated once; the worker additionally reads the BWT, and before its range it looks at the 192 entries in front of
// it (start_carry below) for the pre-filter's view of a cluster that reaches into the range.  f is unused otherwise.
typedef u64 __attribute__((aligned(2))) u64_a2;           // a record's start field: 2-byte aligned in the packed image (one global_store_dwordx2)
__device__ __forceinline__ u64 wave_first64(u64 v) {
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}
// (The scan that wrote the packed image itself -- tickets, per-chunk status words, a decoupled look-back -- was built and measured in round 3 and
//  lost by 2x to segments + one layout kernel (profiles/round3_scan_image_variant.txt); it lives in the history: commit b90733f.)
#ifdef E2S_DIAG_STAMPS
// diagnostic build only (tools/scan_stamps.py): when every range of the scan started and ended (s_memrealtime, 100 MHz), and where it ran.
// The stamps go to a buffer nothing else reads; no result depends on them.
__device__ unsigned long long g_e2s_stamps[32768 * 8];
#endif
template <int X, bool FUSED>
__global__ __launch_bounds__(64, E2S_SCAN_WAVES) void k_scan_seg(SegArgs a, FilterArgs f) {
    static_assert(!FUSED || (X == 1 && SPAN_WORDS == 16), "the fused pre-filter works on 1-byte LCPs, 64 entries per lane");
    constexpr int NR = SPAN_WORDS;
    constexpr int ES = 4 * NR / X;                           // entries per thread span
    constexpr int TILE = 64 * ES;                            // entries per wave-tile
    constexpr int EPW = Swar<X>::EPW;
    constexpr int NW = NR + 2;                               // [word before][span][word after]
    constexpr u32 LM = (X == 4) ? 0xFFFFFFFFu : ((1u << (8 * (X & 3))) - 1u);
#ifdef E2S_DIAG_STAMPS
    const u64 diag_t0 = __builtin_amdgcn_s_memrealtime();
    u64 diag_tc = 0, diag_tq0 = 0, diag_tq1 = 0;
#endif
    // A block is ONE wave: the dispatcher places it wherever a wave slot is free (a four-wave block had to find a slot on every SIMD of one
    // CU at once -- with it 7 % of the slots stood empty between blocks, profiles/round5_scan_ranges.txt -- and its waves, which end at
    // different times, shared a histogram and so an ending).
    const u32 lane = threadIdx.x & 63u;
    // the wave's histogram replicas (4: lengths concentrate on a few values)
    __shared__ u32 s_hist[4][E2S_MAX_CLUST_LEN_K + 2];
    for (u32 i = lane; i < 4u * (E2S_MAX_CLUST_LEN_K + 2); i += 64u) (&s_hist[0][0])[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the lane's share of the statistics: records kept and their base count (lengths of disjoint clusters inside a range of < 2^32 entries)
    u32 l_kept = 0, l_nb32 = 0;
    u32 qn = 0;                                                 // (FUSED) flagged clusters queued in LDS: wave state, kept from range to range
    __shared__ u64 s_q[FUSED ? F2_Q : 1];
    u64* const q = s_q;
    // Every WAVE takes ONE range of tiles (= segment), and there are several times as many ranges as waves fit on the chip: the hardware's
    // dispatcher is the work queue -- the next wave starts where a slot comes free.  Measured with as many ranges as resident waves, one each
    // (profiles/round5_scan_worker_ends.txt): the four waves of a SIMD do not share its issue slots evenly -- the oldest ended at 0.55 of
    // the kernel's length, the youngest at 0.92, the slowest few at 1.0 -- and 27 % of the wave slots' time stood empty.
    const u32 b = (u32)__builtin_amdgcn_readfirstlane((int)blockIdx.x);   // range (= segment) index
    const bool working = b < a.n_blocks;
    if (working) {
    const u64 t0 = e2s_range_tile0(a, b);
    const u64 t1 = min(e2s_range_tile0(a, b + 1u), a.n_tiles);
    const uint8_t* base = (const uint8_t*)a.lcp;
    const int64_t n_prov = (int64_t)a.n_prov;
    const int64_t own_lo = (int64_t)a.own_lo, end_hi = (int64_t)a.end_hi, start_hi = (int64_t)a.start_hi;
    const u64 torg = t0 > 0 ? t0 - 1 : t0;                      // the tile before the range: offsets count from it (seg_range_base)
    const u64 g_range = a.prov_lo + torg * TILE;                // SA position of offset 0
    const u64 seg = (u64)b * a.seg_cap;
    // a segment holds (start - first entry of the worker's range : u32, length : u16) -- 6 bytes per record instead of the image's 10:
    // the segments are an intermediate, read once more by the kernel that lays the image out
    u32* const seg_starts = a.starts32 + seg;                   // worker-uniform bases: stores use a 32-bit record offset
    uint16_t* const seg_lens = a.lens + seg;
    // A record's start is stored as its offset from the first entry of the tile BEFORE the range: the start of a cluster that reaches into
    // the range from the 192 entries before it is found by the wave itself (start_carry), so that record 0 of a range is as finished as
    // the others -- what k_seg_fix had to do for four ranges in five (start, length, kept count, statistics) is left for the cluster that
    // reaches back further.
    const u32 shift1 = 1u;
    // what append_entry keeps (ebwt2clust.cpp:54-63) and statistics() counts (clust2snp.cpp:889-909), tallied as the records leave
    u32* const hrow = s_hist[lane & 3u];
    // (which record is the LAST one kept -- the statistics count it twice, H4 -- is found by k_seg_fix in the segment itself: tracked here
    //  it cost two VGPRs for the whole kernel, the two that stood between the pinned class constants and a fourth wave per SIMD)
    auto tally = [&](const u32 len16) {
        if ((int)len16 >= a.min_len) {
            ++l_kept; l_nb32 += len16;
            if (len16 <= (u32)E2S_MAX_CLUST_LEN_K) atomicAdd(&hrow[len16], 1u);
        }
    };

    // a lane reads 64 contiguous bytes as 4 x 16 B (neighbouring lanes 64 B apart: the four loads of the wave
    // cover one 4 KiB stretch between them) plus the words just before and after its span
    // interior tile: all of its words, the word before and the word after lie inside the provided range
    auto load_fast_w = [&](u64 tile, u32* W) __attribute__((always_inline)) {
        const int64_t tl = (int64_t)tile * TILE;
#ifdef E2S_ABL_NOLOAD                                            // diagnostic build: no memory reads, synthetic words (results are wrong)
        {
            u32 x = (u32)tile * 2654435761u + lane * 40503u;
#pragma unroll
            for (int w = 0; w < NW; ++w) { x = x * 1664525u + 1013904223u; W[w] = (x >> 3) & 0x1F1F1F1Fu; }
            return;
        }
#endif
        const uint8_t* p = base + tl * X;                           // scalar base + 32-bit lane offset
        const u32 o = lane * (u32)(4 * NR);
#pragma unroll
        for (int q = 0; q < NR / 4; ++q) {
            const uint4 v = *(const uint4*)(p + o + 16 * q);
            W[1 + 4 * q + 0] = v.x; W[1 + 4 * q + 1] = v.y; W[1 + 4 * q + 2] = v.z; W[1 + 4 * q + 3] = v.w;
        }
        W[0] = *(const u32*)(p + o - 4);
        W[NR + 1] = *(const u32*)(p + o + 4 * NR);
    };
    auto load_fast_b = [&](u64 tile, u32* B) __attribute__((always_inline)) {
        if constexpr (FUSED) {
#ifdef E2S_ABL_NOLOAD
            return;
#endif
            const int64_t tl = (int64_t)tile * TILE;
            const u32 o = lane * (u32)(4 * NR);
#pragma unroll
            for (int q = 0; q < NR / 4; ++q) {
                const uint4 w = *(const uint4*)(f.bwt + tl + o + 16 * q);
                B[4 * q + 0] = w.x; B[4 * q + 1] = w.y; B[4 * q + 2] = w.z; B[4 * q + 3] = w.w;
            }
        }
    };
    auto load_fast = [&](u64 tile, u32* W, u32* B) __attribute__((always_inline)) { load_fast_w(tile, W); load_fast_b(tile, B); };
    // the first tile of the shard and the tiles around its end: byte by byte, zero outside the provided range
    auto load_slow = [&](u64 tile, u32* W, u32* B) __attribute__((always_inline)) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)lane * ES;
#pragma unroll
        for (int w = 0; w < NW; ++w) {                                  // (static indices: the array stays in registers)
            u32 v = 0, vb = 0;
#pragma unroll
            for (int bt = 0; bt < 4; ++bt) {
                const int64_t byte = i0 * X + (int64_t)(w - 1) * 4 + bt;
                if (byte >= 0 && byte < n_prov * X) {
                    v |= (u32)base[byte] << (8 * bt);
                    if constexpr (FUSED) vb |= (u32)f.bwt[byte] << (8 * bt);
                }
            }
            W[w] = v;
            if constexpr (FUSED) { if (w >= 1 && w <= NR) B[w - 1] = vb; }
        }
    };
    // one entry (0 outside the provided range), the same value in every lane
    auto entry_at = [&](int64_t i) -> u32 {
        u32 v = 0;
        if (i >= 0 && i < n_prov) {
#pragma unroll
            for (int bt = 0; bt < X; ++bt) v |= (u32)base[i * X + bt] << (8 * bt);
        }
        return v;
    };

    // records of a tile are staged in LDS by the lanes that own their ENDs and written out in rank order, 64 per
    // store: a sparse store per record costs ~55 SIMD cycles on this chip whatever the number of active lanes (the
    // per-record stores were 16 of the kernel's 68 us)
    __shared__ uint64_t s_stage[SEG_STAGE + 1];                 // (+1: the start of a cluster still open at the tile's end)
    uint64_t* const st_rec = s_stage;
    // pre-filter state (FUSED): see k_cluster_filter2
    u32 pf_open1 = 0, cin0 = 0, ccarry = 0;
    auto queue_flush = [&]() {
        if (qn) {
            u64 gbase = 0;
            if (lane == 0) gbase = atomicAdd(f.fl_count, (unsigned long long)qn);
            gbase = __shfl(gbase, 0);
            if (lane < qn) {
                const u64 v = q[lane];
                if (gbase + lane < f.fl_cap) { f.fl_starts[gbase + lane] = v & 0xFFFFFFFFFFFFull; f.fl_lens[gbase + lane] = (uint16_t)(v >> 48); }
            }
            __builtin_amdgcn_wave_barrier();
        }
        qn = 0;
    };

    u32 run = 0;                                                // records written so far (wave-uniform)
    u32 open1 = 0;                                              // 1 + offset of the last start seen in the range (0: none yet)
    u32 state = 0;                                              // last event of the range so far: 0 none, 1 an end, 2 a start
    u32 in0 = 0, unknown = 0;
    u64 first_end = ~0ull;
    // GE / end of the entry just before the range (bit 0 / bit 1): what lane 0 of the first tile inherits.  Evaluated
    // after the first tile's loads have been issued, so that a worker pays one memory latency at its start, not two.
    u32 cb = 0;
    auto range_carry = [&]() -> u32 {
        const int64_t r0 = (int64_t)t0 * TILE;
        const u32 lm2 = entry_at(r0 - 2), lm1 = entry_at(r0 - 1), l0 = entry_at(r0);
        const bool gm1 = a.hge != 0 && lm1 >= a.k_u32, g0 = a.hge != 0 && l0 >= a.k_u32;
        const bool em1 = gm1 && ((lm2 > lm1 && lm1 <= l0) || !g0);
        return (u32)__builtin_amdgcn_readfirstlane((int)((gm1 ? 1u : 0u) | (em1 ? 2u : 0u)));
    };
    // W / B: the words of the tile being worked on; the next tile's are asked for into the same registers (see below).
    // ONE set of registers for a tile's words: the NEXT tile's are asked for into the same registers as soon as the planes are built from them --
    // a fifth of the way into the tile -- and waited for before the record stores.  (Rounds 2-5 kept two sets and asked a whole tile ahead: 34
    // registers and half of the kernel's code for a distance the other waves of the SIMD cover anyway; without them a FIFTH wave fits.)
    u32 WA[NW], BA[FUSED ? NR : 1];
    ClassConsts kc = class_consts();                            // (FUSED) held in VGPRs: see e2s_swar.h
    // (The six class constants pinned into VGPRs: v_bitop3 / v_perm issue faster with register operands -- round 2.  Round 4 made room
    //  for them AND a fourth wave per SIMD: 138 -> 123 VGPRs (the statistics' tally on two 32-bit counters, the range's first end and its
    //  flag wave-uniform, the last kept record found by k_seg_fix).  -DE2S_KC_UNPINNED leaves them to the compiler: 4 waves as well.)
#ifndef E2S_KC_UNPINNED
    if constexpr (FUSED) asm volatile("" : "+v"(kc.m07), "+v"(kc.ehi), "+v"(kc.elo), "+v"(kc.c20), "+v"(kc.one), "+v"(kc.h));
#endif
    auto body = [&](const u64 tile, u32* W, u32* B, auto prefetch) __attribute__((always_inline)) {
        constexpr bool PF = decltype(prefetch)::value;
        u32* const Wn = W; u32* const Bn = B;                   // (the next tile's words: the same registers)
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)lane * ES;
        const u32 w1_first = W[1];                              // (the range's first entry, for `in0` below: W is reloaded before that)
        // ---- start / end planes of the span ----
        u64 S, E, GE;                                           // bit e: a cluster starts / ends at entry e of the span
        {
            u64 NLE; u32 after;
            span_pred<X, NR>(W, a.kb, a.hge, GE, NLE, after);
            E = plane_end<ES>(GE, NLE, after);
            const u32 top = (u32)((GE >> (ES - 1)) & 1ull) | ((u32)((E >> (ES - 1)) & 1ull) << 1);
            const u32 before = wave_shr1(top, cb);
            S = plane_start<ES>(GE, E, before);
            cb = (u32)__builtin_amdgcn_readlane((int)top, 63);
        }
        if (a.prov_is_global0 && i0 == 0) {
            // the j = 0,1 rules of ebwt2clust.cpp:83-84: with L[0] >= k the cluster opened at 0 swallows entry 1
            // whatever L[1] is, and j = 0 is never tested
            const u32 e0v = W[1] & LM;
            const u32 e1v = (X == 4) ? W[2] : ((W[1] >> (8 * (X & 3))) & LM);
            const u32 e2v = (X == 4) ? W[3] : (X == 2 ? (W[2] & LM) : ((W[1] >> 16) & LM));
            E &= ~1ull;
            if (a.hge != 0 && e0v >= a.k_u32) {
                const bool end1 = ((e0v > e1v) && (e1v <= e2v)) || (e2v < a.k_u32);
                S &= ~2ull;
                E = end1 ? (E | 2ull) : (E & ~2ull);
                if (e2v >= a.k_u32) S = end1 ? (S | 4ull) : (S & ~4ull);
                GE |= 2ull;                                     // entry 1 belongs to the cluster opened at 0 whatever L[1] is (FUSED: "in a cluster")
            }
        }
        if constexpr (!FUSED) { if (PF) load_fast_w(tile + 1, Wn); }   // the planes are built: the next tile's words into the same registers
        // ---- pre-filter, part 1: class planes and the lane-to-lane carry of packed base counts (k_cluster_filter2) ----
        const u64 Sa = S, Ea = E;                               // before the window mask: the carries follow every cluster
        u64 C = 0, G = 0, T = 0, F = 0; u32 cin = 0;
        if constexpr (FUSED) {
            u64 c0, c1;                                             // code planes: c0 = C | G, c1 = G | T
            span_bwt_code_planes<NR>(B, kc, c0, c1);
            if (PF) { load_fast_w(tile + 1, Wn); load_fast_b(tile + 1, Bn); }   // (both sets of planes are built: the registers the class planes' temporaries had are free again)
            C = c0 & ~c1; G = c0 & c1; T = ~c0 & c1;
            const u32 ctop = (u32)(c0 >> 63) | ((u32)(c1 >> 63) << 1);
            const u32 cprev = wave_shr1(ctop, ccarry);
            ccarry = (u32)__builtin_amdgcn_readlane((int)ctop, 63);
            F = ((c0 ^ ((c0 << 1) | (u64)(cprev & 1u))) | (c1 ^ ((c1 << 1) | (u64)((cprev >> 1) & 1u)))) & GE & ~Sa;
            const bool hasS = Sa != 0ull;
            const int msbS = 63 - __clzll(Sa | 1ull);
            const u64 tm = hasS ? ~((1ull << msbS) - 1ull) : ~0ull;
            const u32 local = (u32)__popcll(C & tm) | ((u32)__popcll(G & tm) << 8) | ((u32)__popcll(T & tm) << 16) | ((F & tm) ? (1u << 25) : 0u);
            u32 out = local;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const u32 in = wave_shr1(out, cin0);
                out = hasS ? local : pf_merge(local, in);
            }
            cin = wave_shr1(out, cin0);                         // carry into this lane
            cin0 = (u32)__builtin_amdgcn_readlane((int)out, 63);
        }
        // restrict to the owned window: only the tiles that straddle its edges (tile-uniform test, a scalar branch --
        // left per-thread the compiler if-converts the 64-bit compares into ~300 always-executed instructions)
        const int64_t tl = (int64_t)tile * TILE;
        const bool masked = !(tl >= own_lo && tl + TILE <= end_hi && tl + TILE <= start_hi);
        if (masked) {
            u64 ks = 0, ke = 0;
#pragma unroll 1
            for (int f = 0; f < ES; ++f) {
                const int64_t e = i0 + f;
                if (e >= own_lo && e < start_hi) ks |= 1ull << f;
                if (e >= own_lo && e < end_hi) ke |= 1ull << f;
            }
            S &= ks; E &= ke;
        }
        if (tile == t0) {
            // does the range begin inside a cluster?  open(first) && !start(first)   (lane 0 holds the first entry)
            const bool ge0 = a.hge != 0 && (w1_first & LM) >= a.k_u32;
            in0 = (u32)__builtin_amdgcn_readfirstlane((ge0 && !(S & 1ull)) ? 1 : 0);
        }
        const u32 cnt = (u32)__popcll(E);
        const u32 inc = wave_incl_scan(cnt);
        const u32 tile_total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        const u32 base_off = (u32)((tile - torg) * TILE) + lane * ES; // offset from the first tile processed
        // 1 + offset of the lane's last start (meaningless where the lane has none); the last one of the tile is what later tiles
        // inherit: the lane is found with a ballot, no scan across the lanes
        const u32 ls1 = base_off + (u32)(64 - __clzll(S | 1ull));
        const u64 sm = __ballot(S != 0ull);
        u32 tile_last1 = 0;
        if (sm) tile_last1 = (u32)__builtin_amdgcn_readlane((int)ls1, 63 - __clzll(sm));
        // the last event of the tile decides what the range hands on
        const u64 evm = sm | __ballot(E != 0ull);
        if (evm) {
            const int ll = 63 - __clzll(evm);
            const u32 slo = (u32)__builtin_amdgcn_readlane((int)(u32)S, ll), shi = (u32)__builtin_amdgcn_readlane((int)(u32)(S >> 32), ll);
            const u32 elo = (u32)__builtin_amdgcn_readlane((int)(u32)E, ll), ehi = (u32)__builtin_amdgcn_readlane((int)(u32)(E >> 32), ll);
            const u64 sS = ((u64)shi << 32) | slo, sE = ((u64)ehi << 32) | elo;
            const int ms = sS ? 63 - __clzll(sS) : -1, me = sE ? 63 - __clzll(sE) : -1;   // a start precedes an end at one position
            state = ms > me ? 2u : 1u;
        }
        // the last start strictly below each lane (0: none in this tile) -- only the per-record path and the pre-filter's walk need it
        auto last_start_below = [&](const u64 Sx) -> u32 {
            const u32 m = wave_incl_max(Sx ? base_off + (u32)(64 - __clzll(Sx)) : 0u);
            return wave_shr1(m, 0u);
        };
        // pre-filter: what it inherits from tile to tile is taken over ALL starts (in an edge tile the scan's are window-masked)
        u32 pf_prev = pf_open1;
        if constexpr (FUSED) {
            if (!masked) pf_open1 = max(pf_open1, tile_last1);
            else {
                const u64 sm2 = __ballot(Sa != 0ull);
                if (sm2) pf_open1 = max(pf_open1, (u32)__builtin_amdgcn_readlane((int)(base_off + (u32)(64 - __clzll(Sa | 1ull))), 63 - __clzll(sm2)));
            }
        }
        // Wait for the next tile's words NOW, before the record stores: vmcnt counts loads and stores together and in
        // order, so waiting for the next tile's loads after a data-dependent number of stores would mean waiting for
        // every store to land (vmcnt(0)) once per tile.  The empty asm "uses and redefines" the registers: the compiler
        // places the wait here and the stores drain under the next tile's stencil.  No instruction is emitted.
        static_assert(NW == 18, "operand list below");
        if (PF)
        asm volatile("" : "+v"(Wn[0]), "+v"(Wn[1]), "+v"(Wn[2]), "+v"(Wn[3]), "+v"(Wn[4]), "+v"(Wn[5]), "+v"(Wn[6]), "+v"(Wn[7]), "+v"(Wn[8]),
                          "+v"(Wn[9]), "+v"(Wn[10]), "+v"(Wn[11]), "+v"(Wn[12]), "+v"(Wn[13]), "+v"(Wn[14]), "+v"(Wn[15]), "+v"(Wn[16]), "+v"(Wn[17]));
        if constexpr (FUSED && PF)
        asm volatile("" : "+v"(Bn[0]), "+v"(Bn[1]), "+v"(Bn[2]), "+v"(Bn[3]), "+v"(Bn[4]), "+v"(Bn[5]), "+v"(Bn[6]), "+v"(Bn[7]),
                          "+v"(Bn[8]), "+v"(Bn[9]), "+v"(Bn[10]), "+v"(Bn[11]), "+v"(Bn[12]), "+v"(Bn[13]), "+v"(Bn[14]), "+v"(Bn[15]));
        if (run == 0 && tile_total != 0) {
            // the range's first end (once per worker), and whether its cluster STARTED in the range: both wave-uniform (scalar registers) --
            // as per-lane variables set by "whichever lane wrote record 0" they cost three VGPRs for the whole kernel
            const int fl = __ffsll((long long)__ballot(cnt != 0)) - 1;                          // the lane that holds the first end
            const u32 epos = (u32)__builtin_amdgcn_readlane((int)(u32)(__ffsll((long long)E) - 1), fl);
            first_end = g_range + (u32)__builtin_amdgcn_readlane((int)base_off, fl) + epos;
            // a start at or before that end, in this tile (a lane before fl, or lane fl up to the end's position) or earlier in the range
            const u64 sb = __ballot(S != 0ull) & ((1ull << fl) - 1ull);
            const u32 slo = (u32)__builtin_amdgcn_readlane((int)(u32)S, fl), shi = (u32)__builtin_amdgcn_readlane((int)(u32)(S >> 32), fl);
            const u64 sfl = (((u64)shi << 32) | slo) & ((2ull << epos) - 1ull);
            if (open1 == 0u && sb == 0ull && sfl == 0ull) unknown = 1u;
        }
        // emit.  Starts and ends alternate along the array (S0 <= E0 < S1 <= E1 ...), so the cluster that starts at p is the
        // tile's record number "ends strictly before p", and the one that ends at q is record number "ends strictly before q":
        // every lane walks the END bits and the START bits of its plane SEPARATELY, each a bare ctz / clear-lowest loop that
        // drops a 32-bit offset into the record's slot in LDS -- slot r = {1 + offset of the start, offset of the end} -- at
        // consecutive ranks from the lane's exclusive END count.  Slot 0 is preset with the cluster that is open when the tile
        // begins (0 = began before the range: the record is finished by k_seg_fix / e2s_stitch).  The records then leave in
        // rank order, 64 per store.  Almost every tile takes this path: no cluster can be 65536 long (every start used is at or
        // after the last one seen before the tile), the segment and the staging buffer have room, the tile lies inside the owned
        // window -- tile-uniform tests instead of per-record ones.
        {
            const u32 rk0 = run + inc - cnt;
            const u32 tile_end1 = (u32)((tile - torg + 1) * TILE);
            const bool plain = run + tile_total <= a.seg_cap && tile_end1 + 2u - open1 < 65536u && tile_total <= (u32)SEG_STAGE && !masked;
#ifdef E2S_ABL_NOEMIT                                           // diagnostic build: time the kernel without the emit loops
            asm volatile("" :: "v"(rk0));
            if (rk0 == 0xFFFFFFFFu)
#endif
            if (plain) {
                u32* const st32 = (u32*)st_rec;
                if (lane == 0) st32[0] = open1;
                {
                    u32* pe = st32 + 2u * (inc - cnt) + 1u;
                    u32 h = (u32)E;
                    while (h) { *pe = base_off + (u32)__builtin_ctz(h); h &= h - 1u; pe += 2; }
                    h = (u32)(E >> 32);
                    while (h) { *pe = base_off + 32u + (u32)__builtin_ctz(h); h &= h - 1u; pe += 2; }
                }
                {
                    const u64 below = (S & (0ull - S)) - 1ull;          // the entries before the lane's first start
                    u32* ps = st32 + 2u * (inc - cnt + ((E & below) != 0ull ? 1u : 0u));
                    u32 h = (u32)S;
                    while (h) { *ps = base_off + 1u + (u32)__builtin_ctz(h); h &= h - 1u; ps += 2; }
                    h = (u32)(S >> 32);
                    while (h) { *ps = base_off + 33u + (u32)__builtin_ctz(h); h &= h - 1u; ps += 2; }
                }
#ifndef E2S_ABL_NOSTORE
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (open1 != 0u) {
                    for (u32 r = lane; r < tile_total; r += 64u) {          // dense, rank-ordered stores
                        const u64 v = st_rec[r];
                        const u32 s1 = (u32)v, e = (u32)(v >> 32);
                        seg_starts[run + r] = s1 - shift1;
                        seg_lens[run + r] = (uint16_t)(e + 2u - s1);
                        tally(e + 2u - s1);
                    }
                } else {                                                    // no start seen in the range before this tile
                    for (u32 r = lane; r < tile_total; r += 64u) {
                        const u64 v = st_rec[r];
                        const u32 s1 = (u32)v, e = (u32)(v >> 32);
                        if (s1 != 0u) {
                            seg_starts[run + r] = s1 - shift1;
                            seg_lens[run + r] = (uint16_t)(e + 2u - s1);
                            tally(e + 2u - s1);
                        } else seg_lens[run + r] = (uint16_t)((g_range + e + 1) & 0xFFFFull);   // finished by k_seg_fix / e2s_stitch (`unknown`, above)
                    }
                }
                __builtin_amdgcn_wave_barrier();
#endif
            } else {
                // the rare tiles: one record per iteration with every check (segment full, start unknown, length >= 65536), written
                // where it belongs by the lane that owns its END
                u32 cur1 = max(last_start_below(S), open1);
                u32 rk = rk0;
                auto emit_half = [&](u32 Eh, const u32 Sh, const u32 hbase) {
                    u32 done = 0;                               // entries of the half up to and including the previous end
                    while (Eh) {
                        const u32 e = (u32)__builtin_ctz(Eh);
                        const u32 upto = (2u << e) - 1u;
                        const u32 sc = Sh & upto & ~done;
                        done = upto;
                        Eh &= ~upto;
                        const u32 off = hbase + e;
                        const u32 st1 = sc ? hbase + (32u - (u32)__builtin_clz(sc)) : cur1;      // 1 + offset of the start
                        if (rk < a.seg_cap) {
                            if (st1 != 0u) {
                                const u32 len = off + 2u - st1;
                                seg_starts[rk] = st1 - shift1;
                                seg_lens[rk] = (uint16_t)len;
                                tally(len & 0xFFFFu);
                                if (len >= 65536u) { const u32 i = atomicAdd(a.wrap_count, 1u); if (i < E2S_WRAP_CAP) a.wrap_list[i] = seg + rk; }
                            } else seg_lens[rk] = (uint16_t)((g_range + off + 1) & 0xFFFFull);       // finished by k_seg_fix / e2s_stitch (`unknown`, above)
                        }
                        ++rk;
                    }
                    // a start after the half's last end is what a later half's first end belongs to
                    const u32 pend = Sh & ~done;
                    if (pend) cur1 = hbase + (32u - (u32)__builtin_clz(pend));
                };
                emit_half((u32)E, (u32)S, base_off);
                if (ES > 32) emit_half((u32)(E >> 32), (u32)(S >> 32), base_off + 32u);
            }
        }
        run += tile_total;
        open1 = max(open1, tile_last1);
        // ---- pre-filter, part 2: walk the owned ENDs of clusters that hold two different bases at all ----
        if constexpr (FUSED) {
            const u64 Fin = F | ((((cin >> 25) & 1u) && (GE & ~Sa & 1ull)) ? 1ull : 0ull);
            const u64 R = GE & ~Ea;
            u64 Ei = E & (((R + (Fin & R)) & Ea) | (Fin & Ea));     // E: the window-masked ends (own_lo <= end < end_hi)
            if (qn > (u32)(F2_Q / 2)) queue_flush();
            u32 pf_cur1 = 0;                                        // the last start before each lane, over ALL starts
            if (__ballot(Ei != 0ull)) pf_cur1 = max(last_start_below(Sa), pf_prev);
            while (__ballot(Ei != 0ull)) {                          // wave-uniform: the queue position is wave state
                const bool act = Ei != 0ull;
                const int e = act ? __ffsll((long long)Ei) - 1 : 0;
                Ei &= Ei - 1;
                const u64 upto = (2ull << e) - 1ull;
                const u64 sc = Sa & upto;
                const bool have = sc != 0ull;
                const int ms = 63 - __clzll(sc | 1ull);
                const u64 rm = have ? (upto & ~((1ull << ms) - 1ull)) : upto;
                u32 pk = (u32)__popcll(C & rm) | ((u32)__popcll(G & rm) << 8) | ((u32)__popcll(T & rm) << 16);
                if (!have) pk = pf_merge(pk, cin) & 0xFFFFFFu;
                const u32 st1 = have ? base_off + (u32)ms + 1u : pf_cur1;
                const u32 len = base_off + (u32)e + 2u - st1;
                const u32 cC = pk & 0xFFu, cG = (pk >> 8) & 0xFFu, cT = (pk >> 16) & 0xFFu, cA = len - cC - cG - cT;
                const u32 nfreq = (cA >= f.mcov_u ? 1u : 0u) + (cC >= f.mcov_u ? 1u : 0u) + (cG >= f.mcov_u ? 1u : 0u) + (cT >= f.mcov_u ? 1u : 0u);
                const bool pass = act && st1 != 0u && (int)len >= f.len_lo && (int)len <= f.len_hi && nfreq >= 2u;
                const u64 pm = __ballot(pass);
                if (pass) {
                    const u32 idx = qn + (u32)__popcll(pm & ((1ull << lane) - 1ull));
                    if (idx < (u32)F2_Q) q[idx] = (g_range + (u64)(st1 - 1u)) | ((u64)len << 48);
                    else atomicOr(f.err, E2S_DEVERR_DENSE);
                }
                qn = min(qn + (u32)__popcll(pm), (u32)F2_Q);
            }
        }
    };
    // The hot loop runs over interior tiles only and contains nothing but the
    // fast loader: with the byte-wise loader of the edge tiles inlined next to it the compiler's wait-count
    // analysis gave up at the join and waited for the prefetch it had just issued (vmcnt(0)) on every other tile.
    {
        const std::true_type P1{}; const std::false_type P0{};
        u64 tile = t0;
        // interior tiles are [1, tfe): tile 0 has no word before it, the last ones no word after
        const u64 tfe = n_prov >= (int64_t)(TILE + EPW) ? (u64)((n_prov - EPW) / TILE) : 0ull;
        const u64 tf = min(t1, tfe);
        auto slow_run = [&](u64 upto) __attribute__((always_inline)) { for (; tile < upto; ++tile) { load_slow(tile, WA, BA); body(tile, WA, BA, P0); } };
        bool have_cb = false;
        // what the range inherits: the scan's GE / end of the entry before it, and (FUSED) the pre-filter's carries
        auto start_carry = [&]() __attribute__((always_inline)) {
            u32 back1 = 0;                                      // 1 + offset of the last start among the 192 entries before the range (0: none)
            if (t0 > 0 && t0 < t1) {                            // (t0 >= 1: the 192 entries before the range lie inside the provided range)
                if constexpr (FUSED) {
                    const PfCarry pc = prefilter_carry_192(base, f.bwt, (int64_t)t0 * TILE, a.k_u32, a.hge, lane);
                    pf_open1 = pc.open1; cin0 = pc.cin0; ccarry = pc.ccarry; cb = pc.cb; back1 = pc.open1;
                } else {
                    const uint2 lb = scan_lookback_192<X>(base, (int64_t)t0 * TILE, a.k_u32, a.hge, lane, (u32)TILE);
                    cb = lb.x; back1 = lb.y;
                }
            } else cb = range_carry();
            // (a start before the owned window is not this shard's to use: its record is the stitch's, e2s_stitch_plan)
            if (back1 != 0u && (int64_t)(torg * TILE) + (int64_t)(back1 - 1u) >= own_lo) open1 = back1;
#ifdef E2S_DIAG_STAMPS
            asm volatile("" : "+v"(cb) : "v"(cin0), "v"(ccarry), "v"(pf_open1));
            diag_tc = __builtin_amdgcn_s_memrealtime();
#endif
        };
        if constexpr (FUSED) tile = t0;
        if (!have_cb && tile < min(t1, (u64)1)) { start_carry(); have_cb = true; }
        slow_run(min(t1, (u64)1));
        if (tile < tf) {
            load_fast(tile, WA, BA);
            if (!have_cb) { start_carry(); have_cb = true; }
            for (; tile + 1 < tf; ++tile) body(tile, WA, BA, P1);
            body(tile, WA, BA, P0); tile += 1;
        }
        if (!have_cb) start_carry();
        slow_run(t1);
#ifdef E2S_DIAG_STAMPS
        diag_tq0 = __builtin_amdgcn_s_memrealtime();
#endif
        if constexpr (FUSED) queue_flush();
#ifdef E2S_DIAG_STAMPS
        diag_tq1 = __builtin_amdgcn_s_memrealtime();
#endif
    }
    // worker descriptor (`unknown` / `first_end`: wave-uniform, set with the range's first end)
    u32 kept_w = l_kept;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept_w += (u32)__shfl_xor((int)kept_w, o);
    if (lane == 0) {
        SegDesc d;
        d.count = min(run, a.seg_cap);
        d.overflow = run > a.seg_cap ? 1u : 0u;
        d.first_unknown = unknown;
        d.first_end_pos = first_end;
        if (state == 0u) { d.open_after = (t0 < t1) ? in0 : 0u; d.open_start = ~0ull; }         // no event at all
        else if (state == 1u) { d.open_after = 0; d.open_start = ~0ull; }
        else { d.open_after = 1; d.open_start = g_range + (open1 - 1u); }
        d.kept = kept_w; d.first_start = ~0ull;
        {
            // the segment's tail read back (past the L1: this wave's stores went through it): eight lengths from the end, the last
            // record's start, record 0's length -- all in flight together; nearly every record is kept, so the first eight answer
            const volatile uint16_t* vl = seg_lens; const volatile u32* vs = seg_starts;
            const u32 cnt = d.count, lo = unknown ? 1u : 0u;
            u32 lk = 0u, lklen = 0u;
            u32 v[8];
#pragma unroll
            for (u32 q = 0; q < 8u; ++q) v[q] = (q + lo < cnt) ? (u32)vl[cnt - 1u - q] : 0u;
            const u32 ls32 = cnt ? vs[cnt - 1u] : 0u, fl16 = cnt ? (u32)vl[0] : 0u;
#pragma unroll
            for (u32 q = 0; q < 8u; ++q) if (!lk && q + lo < cnt && (int)v[q] >= a.min_len) { lk = cnt - q; lklen = v[q]; }
            if (!lk) for (u32 hi = cnt > 8u ? cnt - 8u : 0u; hi > lo && !lk; --hi) { const u32 x = (u32)vl[hi - 1u]; if ((int)x >= a.min_len) { lk = hi; lklen = x; } }
            d.last_kept1 = lk; d.last_kept_len = lklen; d.last_start32 = ls32; d.first_len16 = fl16; d.reserved = 0u;
        }
        a.desc[b] = d;
        a.hot[b] = make_uint2(d.count | (d.overflow << 31), kept_w | (unknown << 31) | (d.open_after << 30));   // what k_seg_fix reads of every range
#ifdef E2S_DIAG_STAMPS
        if (b < 32768u) {
            g_e2s_stamps[8 * b + 0] = diag_t0;
            g_e2s_stamps[8 * b + 1] = __builtin_amdgcn_s_memrealtime();
            g_e2s_stamps[8 * b + 2] = ((u64)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (u64)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID | HW_ID
            g_e2s_stamps[8 * b + 3] = ((u64)run << 32) | (u64)(t1 - t0);
            g_e2s_stamps[8 * b + 4] = diag_tc; g_e2s_stamps[8 * b + 5] = diag_tq0; g_e2s_stamps[8 * b + 6] = diag_tq1;
        }
#endif
    }
    }
    // the wave's share of the statistics: one flush of its histogram replicas and the two sums, into one of E2S_STATS_REPL copies of the
    // statistics (k_seg_fix adds them up): tens of thousands of waves adding to the same 150 words would queue on them
    unsigned long long* const gstats = a.stats + (size_t)(b % (u32)E2S_STATS_REPL) * E2S_STATS_STRIDE;
    u64 l_nrec = (u64)l_kept, l_nb = (u64)l_nb32;              // (a wave is one range: the records it kept are the records the statistics count)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l_nrec += __shfl_xor((unsigned long long)l_nrec, o); l_nb += __shfl_xor((unsigned long long)l_nb, o); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (u32 i = lane; i <= (u32)E2S_MAX_CLUST_LEN_K; i += 64u) {
        const u32 v = s_hist[0][i] + s_hist[1][i] + s_hist[2][i] + s_hist[3][i];
        if (v) atomicAdd(&gstats[i], (unsigned long long)v);
    }
    if (lane == 0 && l_nrec) { atomicAdd(&gstats[E2S_MAX_CLUST_LEN_K + 1], (unsigned long long)l_nrec); atomicAdd(&gstats[E2S_MAX_CLUST_LEN_K + 2], (unsigned long long)l_nb); }
#ifdef E2S_DIAG_STAMPS
    if (lane == 0 && b < 32768u && working) g_e2s_stamps[8 * b + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

// shard descriptors: begins_inside / open_after / end_before_last and the two last LCP values, evaluated scalar-ly on the
// boundary entries (one thread).  The nine entries the rules can look at -- three before the first owned entry to one after it,
// three before the last owned entry to one after it -- are loaded up front, all in flight together: evaluated lazily the rules
// were a chain of ~10 dependent global loads (~20 us with every other thread of the block waiting at the barrier).
__device__ void shard_flags_eval(const void* lcp, u64 n_prov, u64 prov_lo, u64 own_lo, u64 start_hi, u64 end_hi, u32 k, u32 hge, u32 xb,
                                 u32* flags3, u32* tailv2) {
    auto ld = [&](int64_t i) -> u32 {           // local index -> LCP, 0 outside the provided range
        if (i < 0 || i >= (int64_t)n_prov) return 0u;
        const uint8_t* p = (const uint8_t*)lcp + i * xb;
        return xb == 1 ? (u32)p[0] : xb == 2 ? (u32)*(const uint16_t*)p : *(const u32*)p;
    };
    const int64_t lo = (int64_t)own_lo, last = (int64_t)start_hi - 1;   // first / last owned entry
    u32 A[5], B[5];                             // A[j] = L(lo - 3 + j), B[j] = L(last - 3 + j)
#pragma unroll
    for (int j = 0; j < 5; ++j) { A[j] = ld(lo - 3 + j); B[j] = ld(last - 3 + j); }
    auto L = [&](int64_t i) -> u32 {
        if (i >= lo - 3 && i <= lo + 1) return A[i - (lo - 3)];
        if (i >= last - 3 && i <= last + 1) return B[i - (last - 3)];
        return ld(i);                           // (never: the rules below reach two entries back and one ahead)
    };
    const bool never = hge == 0;
    auto ge = [&](int64_t i) { return !never && L(i) >= k; };
    auto gidx = [&](int64_t i) { return (int64_t)prov_lo + i; };
    auto open = [&](int64_t i) -> bool {       // ebwt2clust.cpp:83-84,116
        const int64_t g = gidx(i);
        if (g < 0) return false;
        if (g == 1) return ge(i - 1) || ge(i);
        return ge(i);
    };
    auto endf = [&](int64_t i) -> bool {       // ebwt2clust.cpp:98-102 (never evaluated for j = 0)
        const int64_t g = gidx(i);
        if (g <= 0) return false;
        return open(i) && ((L(i - 1) > L(i) && L(i) <= L(i + 1)) || !ge(i + 1));
    };
    auto startf = [&](int64_t i) -> bool {
        const int64_t g = gidx(i);
        if (!open(i)) return false;
        if (g == 0) return true;
        return !open(i - 1) || endf(i - 1);
    };
    u32 begins_inside = 0, open_after = 0, end_before_last = 0;
    if (start_hi > own_lo) {
        begins_inside = (open(lo) && !startf(lo)) ? 1u : 0u;
        // the end test of the very last entry of the EGSA needs the EOF phantom: left to e2s_stitch
        const bool last_is_eof = (last >= (int64_t)end_hi);
        open_after = (open(last) && (last_is_eof || !endf(last))) ? 1u : 0u;
        // did a cluster close at n_total-2, i.e. did append_entry run in the reference's last real iteration?
        if (last_is_eof) end_before_last = endf(last - 1) ? 1u : 0u;
    }
    flags3[E2S_FLAG_BEGINS_INSIDE] = begins_inside;
    flags3[E2S_FLAG_OPEN_AFTER] = open_after;
    flags3[E2S_FLAG_END_BEFORE_LAST] = end_before_last;
    tailv2[0] = L(last);
    tailv2[1] = L(last - 1);
}

// One block ties the segments together: offsets of the records in rank order and of the KEPT records in the image (exclusive scans of
// the per-worker counts), the records whose start lies in an earlier block -- with their share of the kept counts and the
// statistics, which the scan could not tally without their length --, the statistics themselves (the scan's block-wise sums plus
// those shares), and the ScanMail the host downloads.
constexpr int SEGFIX_MAX = E2S_SEGFIX_MAX;   // segments (= ranges of the scan) one launch ties together: 32 per thread at most
__device__ __forceinline__ u64 seg_range_base(const SegArgs& a, u32 b, u32 row_entries) {      // SA position of the first entry of worker b's range
    const u64 t0 = e2s_range_tile0(a, b);
    return a.prov_lo + (t0 > 0 ? t0 - 1 : t0) * (u64)row_entries;     // (offsets count from the tile before the range: k_scan_seg)
}
// blob (the resident job): this rank's share of the first all-gather -- e2s_boundary + e2s_scan_stats exactly as e2s_boundary_desc /
// e2s_scan_stats_desc hand them out (tail_bwt_arg >= 0: the BWT byte of the shard's last entry given by the caller, e2s_upload_lcp).
//
// Up to 32768 segments by ONE block, in a few microseconds: it reads 8 bytes per segment (SegArgs::hot: count, kept count, three flags),
// coalesced -- wave w owns a contiguous run of segments and takes them 64 at a time, segment j * 64 + lane of its run in lane `lane` --, and
// writes the two exclusive sums the same way: a wave scan per 64 segments, a carry from one 64 to the next, the sixteen waves' totals
// through LDS.  (Its first scalable form gave every thread a contiguous run and walked the 56-byte descriptors three times: 247 us for
// 32768 segments, one CU's worth of uncoalesced loads behind one another.)
// The one thing a segment may still need from its predecessors -- the start of its record 0 where the cluster reaches back further than
// the 192 entries the scan looks at itself (k_scan_seg, start_carry) -- is rare: the lane walks back through the descriptors until one
// says where the cluster began.
__global__ __launch_bounds__(1024) void k_seg_fix(SegArgs a, u32* flags, u32* tailv, u64* seg_off, u64* kept_off, u32* err,
                                                  const uint8_t* bwt, const void* pos, const void* da, int xb, int yb, int zb, ScanMail* out,
                                                  u64* stats_out, uint8_t* blob, int is_last_rank, int tail_bwt_arg) {
    constexpr int JMAX = SEGFIX_MAX / 1024;  // 64-segment groups per wave at most
    static_assert(JMAX <= 32, "a lane's groups: one bit each in adjmask");
    __shared__ u64 s_wcnt[16], s_wkept[16];
    __shared__ u32 s_first, s_last, s_err, s_lastk;
    __shared__ u32 s_hadj[E2S_MAX_CLUST_LEN_K + 2];
    __shared__ u64 s_nadj[2];
    const u64 NONE = ~0ull, PASS = ~0ull - 1;
    const u32 tid = threadIdx.x, nb = a.n_blocks, lane = tid & 63u, wave = tid >> 6;
    const u32 row_entries = 64u * (64u / (u32)xb);
    __shared__ u32 s_flags[4], s_tailv[2];
    if (tid == 0) { s_first = 0xFFFFFFFFu; s_last = 0; s_err = 0; s_lastk = 0; s_nadj[0] = 0; s_nadj[1] = 0; }
    for (u32 i = tid; i < (u32)(E2S_MAX_CLUST_LEN_K + 2); i += 1024u) s_hadj[i] = 0u;
    // the shard's boundary flags (an otherwise idle thread, while the others fetch the hot words)
    if (tid == 1023) shard_flags_eval(a.lcp, a.n_prov, a.prov_lo, a.own_lo, a.start_hi, a.end_hi, a.k_u32, a.hge, (u32)xb, s_flags, s_tailv);
    const u32 jn = (nb + 1023u) / 1024u;                        // groups per wave
    const u32 w0 = wave * jn * 64u;                             // the wave's first segment
    // what the cluster that is open behind segment q began at: its start | PASS (q lies inside one cluster throughout) | NONE (closed)
    auto open_of = [&](u32 q) -> u64 {
        const uint2 hq = a.hot[q];
        if (!((hq.y >> 30) & 1u)) return NONE;
        const u64 os = a.desc[q].open_start;
        return os != NONE ? os : PASS;
    };
    auto carry_into = [&](u32 b) -> u64 {                       // the nearest earlier segment that is not inside one cluster throughout
        for (u32 q = b; q-- > 0;) { const u64 v = open_of(q); if (v != PASS) return v; }
        return NONE;
    };
    // the scan's statistics (eight copies): asked for now, by the threads that will add them up after the sums -- a round trip of its own otherwise
    u64 sr[E2S_STATS_REPL];
#pragma unroll
    for (int r = 0; r < E2S_STATS_REPL; ++r) sr[r] = tid <= (u32)E2S_MAX_CLUST_LEN_K ? a.stats[(size_t)r * E2S_STATS_STRIDE + tid] : 0ull;
    // pass 1 over the hot words: the lane's sums, and which of its segments need their record 0 finished here
    u32 unkmask = 0u, ferr = 0u;
    u64 lsum = 0, lksum = 0;
    const u32 nbm1 = nb ? nb - 1u : 0u;
    for (u32 j0 = 0; j0 < (u32)JMAX && j0 < jn; j0 += 16u) {     // (sixteen groups at a time; a short list -- 4096 segments: four groups -- takes one trip)
#pragma unroll
        for (u32 jj = 0; jj < 16u; ++jj) {                      // (unconditional loads at clamped addresses: sixteen in flight together)
            const u32 j = j0 + jj;
            const u32 b = w0 + j * 64u + lane;
            uint2 hv = a.hot[min(b, nbm1)];
            if (j >= jn || b >= nb) hv = make_uint2(0u, 0u);
            if (hv.x >> 31) ferr = E2S_DEVERR_SEGFULL;           // (its segment overflowed)
            if ((hv.x & 0x7FFFFFFFu) && (hv.y >> 31)) unkmask |= 1u << j;
            lsum += hv.x & 0x7FFFFFFFu; lksum += hv.y & 0x3FFFFFFFu;
        }
    }
    __syncthreads();                                            // (s_first / s_last / s_err / s_hadj are set)
    // record 0 of a segment whose cluster began before the 192 entries its wave looked at: start and length -- and whether append_entry
    // keeps it.  The one that began in an earlier SHARD stays provisional (e2s_stitch finishes it) and keeps a slot in the image whatever
    // its length.  (Rare; a loop over the set bits.)
    u64 t_nadj = 0, t_badj = 0;
    u32 adjmask = 0u;
    for (u32 m = unkmask; m; m &= m - 1u) {
        const u32 j = (u32)__ffs((int)m) - 1u;
        const u32 b = w0 + j * 64u + lane;
        const u64 c = carry_into(b);
        const u64 seg = (u64)b * a.seg_cap;
        u32 len16; bool slot;
        if (c == NONE) { len16 = (u32)a.lens[seg]; slot = true; }          // begins in an earlier shard: e2s_stitch
        else {
            const u64 true_len = a.desc[b].first_end_pos - c + 1;
            len16 = (u32)(true_len & 0xFFFFull);
            a.desc[b].first_start = c;
            a.lens[seg] = (uint16_t)len16;
            if (true_len >= 65536ull) { const u32 i = atomicAdd(a.wrap_count, 1u); if (i < E2S_WRAP_CAP) a.wrap_list[i] = seg; }
            slot = (int)len16 >= a.min_len;
        }
        if ((int)len16 >= a.min_len) {                          // what the statistics count: the length as stored
            ++t_nadj; t_badj += (u64)len16;
            if (len16 <= (u32)E2S_MAX_CLUST_LEN_K) atomicAdd(&s_hadj[len16], 1u);
        }
        if (slot) adjmask |= 1u << j;
    }
    lksum += (u64)__popc(adjmask);
    // wave-level reductions first: one LDS atomic per wave instead of one per segment on a single address
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ferr |= (u32)__shfl_xor((int)ferr, o);
        t_nadj += __shfl_xor((unsigned long long)t_nadj, o); t_badj += __shfl_xor((unsigned long long)t_badj, o);
        lsum += __shfl_xor((unsigned long long)lsum, o); lksum += __shfl_xor((unsigned long long)lksum, o);
    }
    if (lane == 0) {
        if (ferr) atomicOr(&s_err, ferr);
        if (t_nadj) { atomicAdd((unsigned long long*)&s_nadj[0], (unsigned long long)t_nadj); atomicAdd((unsigned long long*)&s_nadj[1], (unsigned long long)t_badj); }
        s_wcnt[wave] = lsum; s_wkept[wave] = lksum;
    }
    __syncthreads();
    {
        // pass 2: exclusive sums of the raw and the kept counts, in segment order: the waves before, the groups before, the lanes before
        u64 run = 0, krun = 0;
        for (u32 w = 0; w < wave; ++w) { run += s_wcnt[w]; krun += s_wkept[w]; }
        u32 fmin = 0xFFFFFFFFu, fmax = 0u, lastk = 0u;
        for (u32 j0 = 0; j0 < (u32)JMAX && j0 < jn; j0 += 8u) {  // (jn is the same in every wave: the trips hold wave-level operations)
#pragma unroll
            for (u32 jj = 0; jj < 8u; ++jj) {
                const u32 j = j0 + jj;
                const u32 b = w0 + j * 64u + lane;
                uint2 hv = a.hot[min(b, nbm1)];                  // (again: out of the caches; unconditional, eight in flight)
                if (j < jn) {
                    if (b >= nb) hv = make_uint2(0u, 0u);
                    const u32 cnt = hv.x & 0x7FFFFFFFu, kept = (hv.y & 0x3FFFFFFFu) + ((adjmask >> j) & 1u);
                    const u32 inc = wave_incl_scan(cnt), kinc = wave_incl_scan(kept);
                    if (b < nb) { seg_off[b] = run + (inc - cnt); kept_off[b] = krun + (kinc - kept); }
                    run += wave_bcast_last(inc); krun += wave_bcast_last(kinc);
                    if (cnt) { fmin = min(fmin, b); fmax = max(fmax, b); }
                    if (kept) lastk = max(lastk, b + 1u);
                }
            }
        }
        if (wave == 15u && lane == 0) { seg_off[nb] = run; kept_off[nb] = krun; }      // (the last wave's run ends at or behind the last segment: the totals)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            fmin = min(fmin, (u32)__shfl_xor((int)fmin, o)); fmax = max(fmax, (u32)__shfl_xor((int)fmax, o)); lastk = max(lastk, (u32)__shfl_xor((int)lastk, o));
        }
        if (lane == 0) {
            if (fmin != 0xFFFFFFFFu) { atomicMin(&s_first, fmin); atomicMax(&s_last, fmax); }
            if (lastk) atomicMax(&s_lastk, lastk);
        }
    }
    const u64 carry_end_of_all = tid == 0 ? carry_into(nb) : NONE;  // what a segment behind the last one would inherit (the shard's open cluster)
    // the statistics: the scan's sums plus the records finished above
    if (tid <= (u32)E2S_MAX_CLUST_LEN_K) {
        u64 v = (u64)s_hadj[tid];
#pragma unroll
        for (int r = 0; r < E2S_STATS_REPL; ++r) v += sr[r];
        stats_out[tid] = v;
        if (blob) ((e2s_scan_stats*)(blob + sizeof(e2s_boundary)))->hist.freq[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        ScanMail m;
        const bool any = s_first != 0xFFFFFFFFu;
        const u64 first_addr = any ? (u64)s_first * a.seg_cap : NONE;
        // every global value the mail needs is loaded unconditionally (clamped addresses) so that the loads are all
        // in flight together instead of queueing behind one another's branches
        const u64 li = a.n_prov - 1;
        const u64 total = nb ? seg_off[nb] : 0;                      // written above by the thread that owns the last run
        const u64 ktotal = nb ? kept_off[nb] : 0;
        // (what other threads of this block wrote above is read back past the L1: volatile)
        const volatile SegDesc* vdesc = a.desc; const volatile uint16_t* vlens = a.lens; const volatile u64* vkept = kept_off;
        SegDesc dl = a.desc[any ? s_last : 0];
        dl.first_start = vdesc[any ? s_last : 0].first_start;
        const u64 last_addr = any ? (u64)s_last * a.seg_cap + dl.count - 1 : NONE;
        const SegDesc dk = a.desc[s_lastk ? s_lastk - 1u : 0u];
        const SegDesc df = a.desc[any ? s_first : 0];
        const u64 v_first_end = df.first_end_pos;
        const u32 v_last_st32 = dl.last_start32;
        // (record 0 of the first segment: its length as its wave wrote it -- unless it was finished above, by another thread of this block)
        const u32 v_first_len = df.first_unknown ? (u32)vlens[any ? first_addr : 0] : df.first_len16;
        const u32 v_text = da ? (yb == 1 ? (u32)((const uint8_t*)da)[li] : yb == 2 ? (u32)((const uint16_t*)da)[li] : ((const u32*)da)[li]) : 0u;
        const u32 v_suff = pos ? (zb == 1 ? (u32)((const uint8_t*)pos)[li] : zb == 2 ? (u32)((const uint16_t*)pos)[li] : ((const u32*)pos)[li]) : 0u;
        const u32 v_lcp = xb == 1 ? (u32)((const uint8_t*)a.lcp)[li] : xb == 2 ? (u32)((const uint16_t*)a.lcp)[li] : ((const u32*)a.lcp)[li];
        const u32 v_bwt = bwt ? (u32)bwt[li] : 0u;
        const u32 v_wrap = *a.wrap_count, v_err = *err;
        const u32 f0 = s_flags[E2S_FLAG_BEGINS_INSIDE], f1 = s_flags[E2S_FLAG_OPEN_AFTER], f2 = s_flags[E2S_FLAG_END_BEFORE_LAST];
        const u32 tv0 = s_tailv[0], tv1 = s_tailv[1];
        flags[E2S_FLAG_BEGINS_INSIDE] = f0; flags[E2S_FLAG_OPEN_AFTER] = f1; flags[E2S_FLAG_END_BEFORE_LAST] = f2;
        tailv[0] = tv0; tailv[1] = tv1;
        m.total_ends = total;
        m.first_end_pos = any ? v_first_end : NONE;
        m.first_rec_addr = first_addr; m.last_rec_addr = last_addr;
        m.begins_inside = f0; m.open_after = f1; m.end_before_last = f2;
        m.err = s_err | v_err;
        m.wrap_count = v_wrap;
        m.tail_lcp = tv0; m.tail_prev_lcp = tv1;
        m.open_start = (f1 && carry_end_of_all != NONE) ? carry_end_of_all : NONE;
        // the start of the shard's last record: in its block's range, or -- record 0 of that block -- what was found above
        const bool last_is_r0 = any && dl.count == 1u && dl.first_unknown;
        const u64 v_last_start = last_is_r0 ? dl.first_start : seg_range_base(a, s_last, row_entries) + v_last_st32;
        m.last_end_start = (total > 0 && !(total == 1 && f0)) ? v_last_start : NONE;
        m.first_len_prov = total > 0 ? v_first_len : 0u;
        m.last_text = v_text; m.last_suff = v_suff; m.last_lcp = v_lcp; m.last_bwt = v_bwt;
        m.img_records = ktotal;
        *out = m;
        // the two sums, and the last record the statistics count (the reference counts it twice, H4) with the place it lives at
        u64 sum_rec = s_nadj[0], sum_bases = s_nadj[1];
#pragma unroll
        for (int r = 0; r < E2S_STATS_REPL; ++r) { sum_rec += a.stats[(size_t)r * E2S_STATS_STRIDE + E2S_MAX_CLUST_LEN_K + 1]; sum_bases += a.stats[(size_t)r * E2S_STATS_STRIDE + E2S_MAX_CLUST_LEN_K + 2]; }
        stats_out[E2S_MAX_CLUST_LEN_K + 1] = sum_rec;
        stats_out[E2S_MAX_CLUST_LEN_K + 2] = sum_bases;
        u64 last_len = 0, at = NONE;
        if (s_lastk) {
            const u32 bk = s_lastk - 1u;
            const u32 lk = dk.last_kept1;
            if (lk) { at = (u64)bk * a.seg_cap + lk - 1u; last_len = dk.last_kept_len; }
            else { last_len = vlens[(u64)bk * a.seg_cap]; at = (u64)bk * a.seg_cap; }       // only its record 0, finished above
            // (a provisional record 0 that holds a slot without reaching min_len is not what the statistics end with)
            if ((int)last_len < a.min_len) {
                last_len = 0; at = NONE;
                for (u32 b2 = bk; b2-- > 0;) {
                    const SegDesc d2 = a.desc[b2];
                    const u32 lk2 = d2.last_kept1;
                    if (lk2) { at = (u64)b2 * a.seg_cap + lk2 - 1u; last_len = d2.last_kept_len; break; }
                    if (vkept[b2 + 1u] != vkept[b2] && (int)vlens[(u64)b2 * a.seg_cap] >= a.min_len) { last_len = vlens[(u64)b2 * a.seg_cap]; at = (u64)b2 * a.seg_cap; break; }
                }
            }
        }
        stats_out[E2S_MAX_CLUST_LEN_K + 3] = last_len;
        stats_out[E2S_MAX_CLUST_LEN_K + 4] = at;
        if (blob) {
            e2s_boundary b = {};
            b.n_ends = m.total_ends; b.open_start = m.open_start; b.last_end_start = m.last_end_start;
            b.begins_inside = m.begins_inside; b.open_after = m.open_after;
            b.pass_through = (m.total_ends == 0 && m.begins_inside && m.open_after) ? 1u : 0u;
            b.end_before_last = m.end_before_last; b.first_end_low16 = m.first_len_prov & 0xFFFFu;
            b.tail_lcp = m.tail_lcp; b.tail_prev_lcp = m.tail_prev_lcp;
            b.tail_bwt = is_last_rank ? (tail_bwt_arg >= 0 ? (u32)tail_bwt_arg : (m.last_bwt & 0xFFu)) : 0u;
            *(e2s_boundary*)blob = b;
            e2s_scan_stats* st = (e2s_scan_stats*)(blob + sizeof(e2s_boundary));
            st->hist.n_records = stats_out[E2S_MAX_CLUST_LEN_K + 1]; st->hist.n_bases = stats_out[E2S_MAX_CLUST_LEN_K + 2];
            st->last_none = at == NONE ? 1u : 0u;
            st->last_is_first = (at != NONE && at == m.first_rec_addr) ? 1u : 0u;
            st->hist.last_len = at == NONE ? 0ull : last_len;
            st->min_len = a.min_len; st->reserved = 0;
        }
    }
}

// ---- the resident job's plan, on the device (e2s_resident_job): every rank's stitch plan and statistics after the stitch, the merged
// histogram, statistics()'s auto-detection of max_clust_length -- the pure functions of e2s_plan.h, the same code the host entry points
// run -- and what follows from THIS rank's plan: record 0 finished in its segment, the seed clusters (EOF tail records, wrapped
// lengths) at the head of the flagged list, the per-cluster pass's counters cleared.  One block.
__global__ __launch_bounds__(256) void k_job_plan(JobPlanArgs a) {
    // the gathered blobs and the scan's mail are staged in LDS by all threads at once (ONE global round trip); everything below reads LDS.
    // Dynamic LDS, sized for the ranks there are (with room for 64 the block waited for a CU with 100 KB free while the image's layout
    // kernel ran beside it: 65 us instead of 15)
    extern __shared__ __attribute__((aligned(16))) uint8_t s_plan_lds[];
    __shared__ uint64_t s_freq[E2S_MAX_CLUST_LEN_K + 1];
    __shared__ ScanMail s_mail;
    const u32 tid = threadIdx.x;
    const int W = a.world;
    uint8_t* const s_blob = s_plan_lds;
    e2s_boundary* const s_all = (e2s_boundary*)(s_blob + (size_t)W * E2S_JOB_BLOB);
    e2s_stitch_plan_t* const s_plan = (e2s_stitch_plan_t*)(s_all + W);
    e2s_stat_delta* const s_delta = (e2s_stat_delta*)(s_plan + W);
    int* const s_rc = (int*)(s_delta + W);
    static_assert(E2S_JOB_BLOB % 16 == 0 && sizeof(ScanMail) % 8 == 0, "staging by words");
    for (u32 i = tid; i < (u32)W * (E2S_JOB_BLOB / 16u); i += 256u) ((uint4*)s_blob)[i] = ((const uint4*)a.blobs)[i];
    if (tid < sizeof(ScanMail) / 8) ((u64*)&s_mail)[tid] = ((const u64*)a.mail)[tid];
    // the per-cluster pass's counters (k_cand_init's part; the ordering pass's bucket counts were left clean by k_cand_offsets)
    if (tid == 255) { *a.cand_count = 0; *a.stream8_count = 0; *a.cand_err = 0; }
    if (a.zero8 && tid >= 240 && tid < 248) a.zero8[tid - 240] = 0ull;
    if (a.n_pieces && tid >= 200 && tid <= 200 + a.n_pieces) {  // (the flagged list is complete: the scan kernel wrote it)
        const u32 c = tid - 200;
        const u64 nfl = *a.fl_count;
        a.piece_bounds[c] = c == a.n_pieces ? nfl : nfl / a.n_pieces * c;
        a.piece_snap[c] = 0ull;                                  // ([0] stays 0; [c + 1] is written by piece c's last block)
        if (c < a.n_pieces) a.piece_done[c] = 0u;
    }      // the tail's totals, kept count and error word (a memset of its own cost a launch gap before the consensus)
    __syncthreads();
    for (u32 i = tid; i < (u32)W * (sizeof(e2s_boundary) / 8); i += 256u) {
        const u32 r = i / (sizeof(e2s_boundary) / 8), w = i % (sizeof(e2s_boundary) / 8);
        ((u64*)&s_all[r])[w] = ((const u64*)(s_blob + (size_t)r * E2S_JOB_BLOB))[w];
    }
    __syncthreads();
    auto stats_of = [&](int r) { return (const e2s_scan_stats*)(s_blob + (size_t)r * E2S_JOB_BLOB + sizeof(e2s_boundary)); };
    if ((int)tid < W) {
        s_rc[tid] = e2s_plan_core(s_all, W, (int)tid, a.n_total, a.k, a.lcp_bytes, &s_plan[tid]);
        e2s_hist_delta_core(stats_of((int)tid), &s_all[tid], &s_plan[tid], &s_delta[tid]);
    }
    if (tid >= 64u && tid - 64u <= (u32)E2S_MAX_CLUST_LEN_K) {                       // (the waves after the first: the first one is busy with the plans)
        u64 f = 0;
        for (int r = 0; r < W; ++r) f += stats_of(r)->hist.freq[tid - 64u];
        s_freq[tid - 64u] = f;
    }
    __syncthreads();
    if (tid >= 64u) return;
    u32 rc = 0;
    u64 nrec = 0, nb = 0, last_len = 0;
    if (tid == 0) {
        for (int r = 0; r < W; ++r) {
            if (s_rc[r]) rc |= E2S_JOBERR_PLAN;
            const e2s_stat_delta& d = s_delta[r];
            for (int i = 0; i < d.n; ++i) if (d.len[i] <= E2S_MAX_CLUST_LEN_K) s_freq[d.len[i]] += (u64)(int64_t)d.sign[i];
            nrec += d.n_records; nb += d.n_bases;
            if (d.n_records) last_len = d.last_len;
        }
        if (nrec && last_len <= (u64)E2S_MAX_CLUST_LEN_K) s_freq[last_len]++;        // H4: the last record counted twice (e2s_auto_len_core)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // statistics()'s auto-detection (e2s_auto_len_core, e2s_plan.h) with the histogram's 151 terms spread over the wave -- three bins a
    // lane, exact integer prefix sums, the same two floating-point tests ((double)cum / n_bases < pval once, then (double)cum * (1 /
    // n_bases) < pval per length: the -Ofast arithmetic of the reference build, H19); serially, one thread spent ~20 us in LDS latency
    nrec = wave_first64(nrec); nb = wave_first64(nb); last_len = wave_first64(last_len);
    int mcl = 0; uint64_t n_clust = 0, n_bases = 0;
    {
        u32 arc = 0;
        if (nrec == 0) arc = 1;
        else {
            n_clust = nrec + 1; n_bases = nb + last_len;
            u64 term[3], incl[3];
            u64 run = 0, tmax = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32 i = 3u * tid + (u32)q;
                term[q] = i <= (u32)E2S_MAX_CLUST_LEN_K ? s_freq[i] * (u64)i : 0ull;
                run += term[q]; incl[q] = run;
                if (i >= 1u) tmax = max(tmax, term[q]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tmax = max(tmax, (u64)__shfl_xor((unsigned long long)tmax, o));
            u64 scan = run;                                     // inclusive scan of the lanes' sums
            for (u32 dd = 1; dd < 64u; dd <<= 1) { const u64 v = __shfl_up((unsigned long long)scan, dd); if (tid >= dd) scan += v; }
            const u64 base = scan - run;
            const int mcl0 = 2 * a.mcov;
            if (tmax == 0) arc = 2;
            else if (mcl0 < 0 || mcl0 > E2S_MAX_CLUST_LEN_K) arc = 3;
            else {
                // cum(c) = sum of the terms mcl0 .. c = P(c) - P(mcl0 - 1)
                const u32 lm = (u32)mcl0 / 3u, qm = (u32)mcl0 % 3u;
                u64 pm = 0, tm = 0;                              // P(mcl0), term(mcl0) from their lane
                { const u64 cand = base + (qm == 0 ? incl[0] : qm == 1 ? incl[1] : incl[2]), ct = qm == 0 ? term[0] : qm == 1 ? term[1] : term[2];
                  pm = __shfl((unsigned long long)cand, (int)lm); tm = __shfl((unsigned long long)ct, (int)lm); }
                const u64 pbefore = pm - tm;
                const double dnb = (double)n_bases;
                mcl = mcl0;
                if ((double)tm / dnb < a.pval) {
                    const double rr = 1.0 / dnb;
                    u32 first = 0xFFFFFFFFu;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const u32 c = 3u * tid + (u32)q;
                        const bool in = c > (u32)mcl0 && c <= (u32)E2S_MAX_CLUST_LEN_K;
                        const u64 cum = base + incl[q] - pbefore;
                        const bool brk = in && !((double)cum * rr < a.pval);
                        if (brk) first = min(first, c);
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) first = min(first, (u32)__shfl_xor((int)first, o));
                    mcl = first == 0xFFFFFFFFu ? E2S_MAX_CLUST_LEN_K : (int)first;
                }
            }
        }
        rc |= arc;
    }
    if (tid != 0) return;
    JobDev jd = {};
    const e2s_stitch_plan_t pl = s_plan[a.rank];
    const ScanMail& m = s_mail;
    jd.plan = pl; jd.mcl = mcl; jd.n_clust = n_clust; jd.n_bases = n_bases;
    jd.len_hi = mcl < E2S_MAX_CLUST_LEN_K ? mcl : E2S_MAX_CLUST_LEN_K;
    // this rank's record 0, finished where it lives (e2s_stitch: k_seg_patch)
    if (pl.patch_first && m.first_rec_addr != ~0ull) {
        SegDesc* d = a.seg.desc + m.first_rec_addr / a.seg.seg_cap;
        d->first_start = pl.first_start; d->first_unknown = 1u; a.seg.lens[m.first_rec_addr] = pl.first_len;
    }
    jd.img_skip = (pl.patch_first && (int)pl.first_len < a.min_len) ? 1u : 0u;
    u32 nt = 0;
    for (u32 t = 0; t < pl.n_tail; ++t) if ((int)pl.tail_len[t] >= a.min_len) ++nt;
    jd.n_tail_kept = nt;
    jd.img_records = m.img_records - jd.img_skip + nt;
    jd.first_wrapped = (pl.patch_first && m.first_end_pos != ~0ull && m.first_end_pos - pl.first_start + 1 >= 65536ull) ? 1u : 0u;
    // seed clusters: what the streaming pre-filter cannot see (cand_seeds on the host path)
    u32 ns = 0;
    auto seed = [&](u64 st, u32 ln) {
        if ((int)ln >= a.len_lo && (int)ln <= jd.len_hi) { if (ns < 64u) { a.fl_starts[ns] = st; a.fl_lens[ns] = (uint16_t)ln; } ++ns; }
    };
    for (u32 t = 0; t < pl.n_tail; ++t) seed(pl.tail_start[t], pl.tail_len[t]);
    if (jd.first_wrapped) seed(pl.first_start, pl.first_len);
    const u32 nw = m.wrap_count < (u32)E2S_WRAP_CAP ? m.wrap_count : (u32)E2S_WRAP_CAP;
    for (u32 i = 0; i < nw; ++i) {
        const u64 r = a.seg.wrap_list[i];
        const u32 b = (u32)(r / a.seg.seg_cap);
        const SegDesc d = a.seg.desc[b];
        const u64 st = (r % a.seg.seg_cap == 0 && d.first_unknown) ? d.first_start : seg_range_base(a.seg, b, a.row_entries) + a.seg.starts32[r];
        seed(st, a.seg.lens[r]);
    }
    if (ns > 64u || m.wrap_count > (u32)E2S_WRAP_CAP) rc |= E2S_JOBERR_SEEDS;
    if (m.img_records + (u64)nt > a.img_cap) rc |= E2S_JOBERR_IMAGE;    // (the layout kernel clips silently: the verdict must travel with the rank's ok)
    jd.stat_rc = rc;
    *a.out = jd;
}
void e2s_launch_job_plan(const JobPlanArgs& a, hipStream_t st) {
    const size_t lds = (size_t)a.world * (E2S_JOB_BLOB + sizeof(e2s_boundary) + sizeof(e2s_stitch_plan_t) + sizeof(e2s_stat_delta) + sizeof(int)) + 64;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_job_plan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_job_plan, dim3(1), dim3(256), lds, st, a);
}
// the image finished in place from the device's plan (e2s_stitch on a pre-stitch image: k_img_patch with host arguments)
__global__ void k_img_patch_dev(uint8_t* img, u64 cap, const JobDev* jd, const ScanMail* mail, int min_len) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const e2s_stitch_plan_t pl = jd->plan;
    auto put = [&](u64 idx, u64 st, uint16_t ln) { if (idx < cap) { uint8_t* o = img + idx * 10ull; *(u64_a2*)o = st; *(uint16_t*)(o + 8) = ln; } };
    if (pl.patch_first && !jd->img_skip) put(0, pl.first_start, pl.first_len);
    u64 at = mail->img_records;
    for (u32 t = 0; t < pl.n_tail; ++t) if ((int)pl.tail_len[t] >= min_len) put(at++, pl.tail_start[t], pl.tail_len[t]);
}
void e2s_launch_img_patch_dev(uint8_t* img, unsigned long long cap, const JobDev* jd, const ScanMail* mail, int min_len, hipStream_t st) {
    hipLaunchKernelGGL(k_img_patch_dev, dim3(1), dim3(64), 0, st, img, (u64)cap, jd, mail, min_len);
}

// e2s_stitch on the image: record 0 of a shard that began inside a cluster, the records the EOF rule adds behind the last one
__global__ void k_img_patch(uint8_t* img, u32 patch_first, u64 first_start, uint16_t first_len, u64 n_at, u32 n_tail, u64 t0, uint16_t l0, u64 t1, uint16_t l1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto put = [&](u64 idx, u64 st, uint16_t ln) { uint8_t* o = img + idx * 10ull; *(u64_a2*)o = st; *(uint16_t*)(o + 8) = ln; };
    if (patch_first) put(0, first_start, first_len);
    if (n_tail > 0) put(n_at, t0, l0);
    if (n_tail > 1) put(n_at + 1, t1, l1);
}
void e2s_launch_img_patch(uint8_t* img, uint32_t patch_first, unsigned long long first_start, uint16_t first_len, unsigned long long n_at, uint32_t n_tail,
                          unsigned long long t0, uint16_t l0, unsigned long long t1, uint16_t l1, hipStream_t st) {
    hipLaunchKernelGGL(k_img_patch, dim3(1), dim3(64), 0, st, img, patch_first, (u64)first_start, first_len, (u64)n_at, n_tail, (u64)t0, l0, (u64)t1, l1);
}

// segments -> contiguous record arrays (only for consumers that index records by rank)
__global__ __launch_bounds__(256) void k_seg_compact(SegArgs a, const u64* seg_off, u32 row_entries, u64* out_starts, uint16_t* out_lens) {
    const u32 b = blockIdx.x;
    const SegDesc d = a.desc[b];
    const u32 n = d.count;
    const u64 src = (u64)b * a.seg_cap, dst = seg_off[b], rb = seg_range_base(a, b, row_entries);
    for (u32 i = threadIdx.x; i < n; i += 256) {
        out_starts[dst + i] = (i == 0u && d.first_unknown) ? d.first_start : rb + a.starts32[src + i];
        out_lens[dst + i] = a.lens[src + i];
    }
}

// one thread per tile: finish the record whose start lies in an earlier tile.  Records whose true length
// reaches 65536 (the reference's uint16_t length wraps, ebwt2clust.cpp:104) are listed for the per-cluster pass.
__global__ __launch_bounds__(1024) void k_tile_scan(const u32* cnt, u64* prefix, u64 n) {
    __shared__ u64 s_part[1024];
    const u32 tid = threadIdx.x;
    const u64 per = (n + 1023) / 1024;
    const u64 lo = min(n, per * tid), hi = min(n, lo + per);
    u64 s = 0;
    for (u64 i = lo; i < hi; ++i) s += cnt[i];
    s_part[tid] = s;
    __syncthreads();
    if (tid == 0) { u64 run = 0; for (int i = 0; i < 1024; ++i) { const u64 v = s_part[i]; s_part[i] = run; run += v; } }
    __syncthreads();
    u64 run = s_part[tid];
    for (u64 i = lo; i < hi; ++i) { prefix[i] = run; run += cnt[i]; }
}

// ---- `.clusters` file image on the device (append_entry, ebwt2clust.cpp:54-63): the records with length >= min_len, packed
// [start:u64][length:u16] 10 bytes each, in order.  Count per 4096 records, k_tile_scan, then write through LDS.
constexpr int PK_PER = 16, PK_BLOCK = 256 * PK_PER;
__global__ __launch_bounds__(256) void k_pack_count(const uint16_t* lens, u64 n, int min_len, u32* cnt) {
    __shared__ u32 s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const u64 i0 = (u64)blockIdx.x * PK_BLOCK + (u64)threadIdx.x * PK_PER;
    u32 c = 0;
    for (int j = 0; j < PK_PER; ++j) if (i0 + j < n && (int)lens[i0 + j] >= min_len) ++c;
    for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_c;
}
__global__ __launch_bounds__(256) void k_pack_write(const u64* starts, const uint16_t* lens, u64 n, int min_len, const u64* prefix, const u32* cnt,
                                                    u32 n_blocks, uint16_t* out, u64* total) {
    __shared__ uint16_t s_img[PK_BLOCK * 5];
    __shared__ u32 s_w[4];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u64 i0 = (u64)blockIdx.x * PK_BLOCK + (u64)tid * PK_PER;
    u32 c = 0;
    for (int j = 0; j < PK_PER; ++j) if (i0 + j < n && (int)lens[i0 + j] >= min_len) ++c;
    u32 inc = c;                                                // inclusive scan over the block
    for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if (lane >= (u32)o) inc += v; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    u32 r = inc - c;
    for (u32 w = 0; w < wave; ++w) r += s_w[w];
    for (int j = 0; j < PK_PER; ++j) {
        if (i0 + j < n) {
            const uint16_t l = lens[i0 + j];
            if ((int)l >= min_len) {
                const u64 st = starts[i0 + j];
                uint16_t* o = s_img + (size_t)r * 5;
                o[0] = (uint16_t)st; o[1] = (uint16_t)(st >> 16); o[2] = (uint16_t)(st >> 32); o[3] = (uint16_t)(st >> 48); o[4] = l;
                ++r;
            }
        }
    }
    __syncthreads();
    const u32 m = cnt[blockIdx.x] * 5u;
    uint16_t* dst = out + prefix[blockIdx.x] * 5ull;
    for (u32 i = tid; i < m; i += 256u) dst[i] = s_img[i];
    if (blockIdx.x == n_blocks - 1 && tid == 0) *total = prefix[blockIdx.x] + cnt[blockIdx.x];
}

// ---- the same image straight from the scan's segments: the scan has counted what append_entry keeps per segment and k_seg_fix has
// scanned those counts, so ONE kernel lays the image out -- every segment packs itself at its offset (6-byte segment records in,
// 10-byte file records out); the EOF tail records of the stitch plan follow the last one.
// One block per segment, 1024 records per trip.  Ranks come from ballots (a lane looks at ONE record, so the loads are
// coalesced); the records are laid out in LDS at the image's own alignment (LDS offset = byte offset in the image mod 16), so the
// image leaves in whole 16-byte units -- the first version wrote it 2 bytes per lane and took 4 ms for 1.07 GB.
// slot0_b: the segment whose record 0 holds slot 0 of the image whatever its length (a shard that begins inside a cluster; ~0: none).
constexpr int PKS_REC = 1024;                                // records per trip (4 waves x 4 x 64)
// mail (the resident job: the host has not seen the scan's mail yet): slot0_b is worked out from it here.  out_cap: records the image
// has room for -- a trip that would reach past it is dropped (the host sees the size in the mail and reports the overflow).
__global__ __launch_bounds__(256) void k_pack_seg_write(SegArgs a, u32 row_entries, u32 slot0_b, const u64* off, uint8_t* out, u64 out_cap, const ScanMail* mail,
                                                        u32 n_tail, u64 t0, uint16_t l0, u64 t1, uint16_t l1) {
    if (mail) slot0_b = (mail->begins_inside && mail->total_ends && mail->first_rec_addr != ~0ull) ? (u32)(mail->first_rec_addr / a.seg_cap) : 0xFFFFFFFFu;
    __shared__ __attribute__((aligned(16))) uint8_t s_img[PKS_REC * 10 + 32];
    __shared__ u32 s_w[4];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 n_seg = a.n_blocks; const int min_len = a.min_len;
    for (u32 b = blockIdx.x; b < n_seg; b += gridDim.x) {        // (a grid smaller than n_seg: the throttled form, see the launcher)
    const SegDesc d = a.desc[b];
    const u32 n = d.count;
    const u32* ps = a.starts32 + (u64)b * a.seg_cap;
    const uint16_t* pl = a.lens + (u64)b * a.seg_cap;
    const u64 rb = seg_range_base(a, b, row_entries);
    u64 done = off[b];                                          // records of the image before the trip at hand
    for (u32 c0 = 0; c0 < n; c0 += PKS_REC) {
        // thread t looks at the four records c0 + 4 t .. + 3: ONE 16-byte and ONE 8-byte load (the segment arrays are 16-byte aligned, c0 a
        // multiple of 1024; the arrays are seg_cap long, a multiple of 8, so a whole load never leaves the segment)
        const u32 i0 = c0 + 4u * tid;
        uint4 s4 = make_uint4(0, 0, 0, 0); uint2 l2 = make_uint2(0, 0);
        if (i0 < n) { s4 = *(const uint4*)(ps + i0); l2 = *(const uint2*)(pl + i0); }
        const u32 sv[4] = {s4.x, s4.y, s4.z, s4.w};
        const u32 lv[4] = {l2.x & 0xFFFFu, l2.x >> 16, l2.y & 0xFFFFu, l2.y >> 16};
        u64 st[4]; u32 ln[4]; u32 keep[4]; u32 mine = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 i = i0 + (u32)q;
            ln[q] = i < n ? lv[q] : 0u;
            st[q] = (i == 0u && d.first_unknown) ? d.first_start : rb + (u64)sv[q];
            keep[q] = (i < n && ((int)ln[q] >= min_len || (i == 0u && b == slot0_b))) ? 1u : 0u;
            mine += keep[q];
        }
        // rank of the thread's first kept record: a wave scan (DPP) of the per-thread counts + the waves before
        u32 incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o); if (lane >= (u32)o) incl += v; }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        u32 r = incl - mine;
        for (u32 w = 0; w < wave; ++w) r += s_w[w];
        const u32 m = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const u64 D = done * 10ull;                              // byte offset of the trip's first record in the image
        const u32 sh = (u32)(D & 15ull);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (keep[q]) {
                uint16_t* o = (uint16_t*)(s_img + sh + (size_t)r * 10);
                o[0] = (uint16_t)st[q]; o[1] = (uint16_t)(st[q] >> 16); o[2] = (uint16_t)(st[q] >> 32); o[3] = (uint16_t)(st[q] >> 48); o[4] = (uint16_t)ln[q];
                ++r;
            }
        }
        __syncthreads();
        // the 16-byte units [D & ~15, ...) that the trip's bytes [D, D + 10 m) touch: whole units as one store, the two partial ones by 2 bytes
        const u32 nbytes = m * 10u, units = (sh + nbytes + 15u) >> 4;
        uint8_t* const g0 = out + (D & ~15ull);
        if (done + m <= out_cap)
        for (u32 u = tid; u < units; u += 256u) {
            const u32 lo = u << 4, hi = lo + 16u;
            if (lo >= sh && hi <= sh + nbytes) *(uint4*)(g0 + lo) = *(const uint4*)(s_img + lo);
            else for (u32 x = max(lo, sh); x < min(hi, sh + nbytes); x += 2u) *(uint16_t*)(g0 + x) = *(const uint16_t*)(s_img + x);
        }
        done += m;
        __syncthreads();
    }
    }
    if (blockIdx.x == (n_seg - 1) % gridDim.x && tid == 0 && off[n_seg] + n_tail <= out_cap) {     // the records the EOF rule adds (e2s_stitch_plan)
        u64 at = off[n_seg];
        const u64 ts[2] = {t0, t1}; const uint16_t tl[2] = {l0, l1};
        for (u32 t = 0; t < n_tail; ++t) if ((int)tl[t] >= min_len) {
            uint16_t* o = (uint16_t*)(out + at * 10ull);
            o[0] = (uint16_t)ts[t]; o[1] = (uint16_t)(ts[t] >> 16); o[2] = (uint16_t)(ts[t] >> 32); o[3] = (uint16_t)(ts[t] >> 48); o[4] = tl[t];
            ++at;
        }
    }
}
void e2s_launch_pack_segments(const SegArgs& a, u32 row_entries, u32 slot0_b, const u64* kept_off, uint8_t* out, u64 out_cap, const ScanMail* mail, u32 n_tail, u64 t0,
                              uint16_t l0, u64 t1, uint16_t l1, hipStream_t st, u32 max_blocks) {
    // max_blocks (0: one block per segment): an experiment knob -- the pack running beside the per-cluster pass throttled to a few blocks
    // per CU, so that it does not saturate HBM under that latency-bound kernel (no net gain measured: e2s_capi.hip, pack_blocks_beside)
    const u32 n_seg = a.n_blocks;
    const u32 grid = max_blocks && max_blocks < n_seg ? max_blocks : n_seg;
    hipLaunchKernelGGL(k_pack_seg_write, dim3(grid), dim3(256), 0, st, a, row_entries, slot0_b, kept_off, out, out_cap, mail, n_tail, t0, l0, t1, l1);
}
// e2s_stitch on the segments: record 0 of a shard that began inside a cluster gets its start and length from the stitch plan
__global__ void k_seg_patch(SegDesc* d, uint16_t* len0, u64 first_start, uint16_t first_len) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    d->first_start = first_start; d->first_unknown = 1u; *len0 = first_len;
}
void e2s_launch_seg_patch(SegDesc* d, uint16_t* len0, unsigned long long first_start, uint16_t first_len, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_patch, dim3(1), dim3(64), 0, st, d, len0, (u64)first_start, first_len);
}

__global__ void k_apply_plan(u64* starts, uint16_t* lens, u32 patch_first, u64 first_start, uint16_t first_len, u64 n_ends, u32 n_tail,
                             u64 t0, uint16_t l0, u64 t1, uint16_t l1) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (patch_first) { starts[0] = first_start; lens[0] = first_len; }
    if (n_tail > 0) { starts[n_ends] = t0; lens[n_ends] = l0; }
    if (n_tail > 1) { starts[n_ends + 1] = t1; lens[n_ends + 1] = l1; }
}

// length of the last record that survives the min_len filter (the reference counts it twice, H4)
__global__ void k_hist_tail(const uint16_t* lens, u64 n, int min_len, u64* out_last_len) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u64 v = 0;
    for (u64 i = n; i-- > 0;) if ((int)lens[i] >= min_len) { v = lens[i]; break; }
    *out_last_len = v;
}

// ----------------------------------------------------------------------------------------------
// KH: histogram of cluster lengths (statistics(), clust2snp.cpp:889-909; H4 applied on the host)
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hist(const uint16_t* lens, u64 n, int min_len, u64* out /*151 freq, n_rec, n_bases*/) {
    // 4 histogram replicas per wave (lengths concentrate on a few values: one copy serialises on the hot bins -- 21 ms for
    // 3x10^6 records), 8 lengths per lane and load
    constexpr int REP = 4, NB = E2S_MAX_CLUST_LEN_K + 1, NWV = 4;
    __shared__ u32 s_freq[NWV * REP][NB + 1];
    __shared__ u64 s_acc[2];
    for (int i = threadIdx.x; i < NWV * REP * (NB + 1); i += 256) (&s_freq[0][0])[i] = 0;
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    __syncthreads();
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32* const my = s_freq[wave * REP + (lane & (REP - 1))];
    u64 nrec = 0, nb = 0;
    auto take = [&](u32 l) { if ((int)l >= min_len) { nrec++; nb += (u64)l; if (l <= (u32)E2S_MAX_CLUST_LEN_K) atomicAdd(&my[l], 1u); } };
    const u64 nv = n >> 3;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nv; i += (u64)gridDim.x * 256) {
        const uint4 v = *(const uint4*)(lens + 8 * i);
        take(v.x & 0xFFFFu); take(v.x >> 16); take(v.y & 0xFFFFu); take(v.y >> 16);
        take(v.z & 0xFFFFu); take(v.z >> 16); take(v.w & 0xFFFFu); take(v.w >> 16);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7ull)) take((u32)lens[(nv << 3) + threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nrec += __shfl_xor(nrec, o); nb += __shfl_xor(nb, o); }
    if (lane == 0) { atomicAdd((u64*)&s_acc[0], nrec); atomicAdd((u64*)&s_acc[1], nb); }
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += 256) {
        u32 v = 0;
#pragma unroll
        for (int r = 0; r < NWV * REP; ++r) v += s_freq[r][i];
        if (v) atomicAdd(&out[i], (u64)v);
    }
    if (threadIdx.x == 0 && s_acc[0]) { atomicAdd(&out[E2S_MAX_CLUST_LEN_K + 1], s_acc[0]); atomicAdd(&out[E2S_MAX_CLUST_LEN_K + 2], s_acc[1]); }
}

// .clusters file -> record arrays: packed {u64 start; u16 length} records as they are on disk; also checks that they are ascending
// and non-overlapping (H16) -- err gets bit 0 otherwise
__global__ void k_unpack_clusters(const uint8_t* packed, u64 n, u64* starts, uint16_t* lens, u32* err) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto rd = [&](u64 r, u64& st, u32& ln) {
        const uint8_t* p = packed + r * 10;
        u64 v = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) v |= (u64)p[b] << (8 * b);
        st = v; ln = (u32)p[8] | ((u32)p[9] << 8);
    };
    u64 st; u32 ln;
    rd(i, st, ln);
    starts[i] = st; lens[i] = (uint16_t)ln;
    if (i) { u64 ps; u32 pl; rd(i - 1, ps, pl); if (st < ps + pl) atomicOr(err, 1u); }
}
void e2s_launch_unpack_clusters(const uint8_t* packed, u64 n, u64* starts, uint16_t* lens, u32* err, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_unpack_clusters, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, packed, n, starts, lens, err);
}

// ----------------------------------------------------------------------------------------------
// K2: per-cluster pass.  One wave per cluster record, lane i <-> entry start+i (clusters are
// at most 150 entries: three 64-entry chunks).  find_variants, clust2snp.cpp:367-500.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 load_w(const void* p, int w, int64_t i) {
    return w == 1 ? (u32)((const uint8_t*)p)[i] : w == 2 ? (u32)((const uint16_t*)p)[i] : ((const u32*)p)[i];
}

// include.hpp:265-279 (N/n: the reference draws rand()%4; we map to 0 deterministically)
__device__ __forceinline__ u32 base_to_int(u32 c) {
    c &= ~0x20u;                                   // fold case: 'a'..'t' -> 'A'..'T'
    return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
}

constexpr int MAXCH = 3;                           // ceil(150/64)

// One cluster record [s, s+len) handled by one wave: lane i <-> entry s+i (three 64-entry chunks cover
// the 150-entry maximum).  `alloc(lane)` hands out the memory of the next candidate (wave-uniform).
// the four columns of a cluster's entries, lane i <-> entry s + i per 64-entry chunk: LOADS ONLY, so that the caller can issue them for the
// cluster of its next round before it works on the current one (len 0: nothing is loaded)
template <int NCH> struct ClusterColsN { u32 text[NCH], suff[NCH], lcp[NCH], bwt[NCH]; };   // NCH chunks of 64 entries, an entry per lane
typedef ClusterColsN<MAXCH> ClusterCols;
// chunks C0 .. C0 + NCH - 1 of the cluster [s, s + len) into k's slots 0 .. NCH - 1
template <bool DEF, int NCH, int C0 = 0>
__device__ __forceinline__ void cluster_load(const CandArgs& a, const u64 s, const int len, const u32 lane, ClusterColsN<NCH>& k) {
    const int nch = (len + 63) >> 6;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        k.text[c] = k.suff[c] = k.lcp[c] = k.bwt[c] = 0;
        if (C0 + c < nch) {
            const int o = (C0 + c) * 64 + (int)lane;
            if (o < len) {
                const u64 g = s + (u64)o;
                if (g == a.n_total) { k.text[c] = a.ph_text; k.suff[c] = a.ph_suff; k.lcp[c] = a.ph_lcp; k.bwt[c] = a.ph_bwt; }
                else {
                    const int64_t i = (int64_t)(g - a.prov_lo);
                    k.bwt[c] = a.bwt[i];
                    if (DEF) { k.lcp[c] = ((const uint8_t*)a.lcp)[i]; k.suff[c] = ((const uint8_t*)a.pos)[i]; k.text[c] = ((const u32*)a.da)[i]; }
                    else { k.lcp[c] = load_w(a.lcp, a.xb, i); k.suff[c] = load_w(a.pos, a.zb, i); k.text[c] = load_w(a.da, a.yb, i); }
                }
            }
        }
    }
}
// MAXCH: the chunks the cluster may have (1 for the clusters of <= 64 entries -- most of them: the mode of the lengths is the coverage --,
// whose code then holds one chunk's registers and none of the chunk loops' branches)
template <bool DEF, int MAXCH, typename Alloc>     // DEF: widths (x,y,z) = (1,4,1)
__device__ __forceinline__ void cluster_record(const CandArgs& a, const ClusterColsN<MAXCH>& k, const int len, const u64 key, const u32 lane, Alloc alloc) {
    const int nch = (len + 63) >> 6;
    const u32 (&text)[MAXCH] = k.text; const u32 (&suff)[MAXCH] = k.suff; const u32 (&lcp)[MAXCH] = k.lcp; const u32 (&bwt)[MAXCH] = k.bwt;
    bool valid[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) valid[c] = c < nch && c * 64 + (int)lane < len;
    // counts[sample][base] and the first entry with the strictly largest LCP (clust2snp.cpp:377-393).  The eight counts: every lane adds
    // 1 << (8 x base) into the packed word of its sample (a cluster has <= 192 entries: four 8-bit fields hold the sums), two DPP scans
    // sum them over the wave -- eight ballots per chunk, each with its AND / count / add on the scalar unit, were a third of the kernel's
    // scalar instructions.  The largest LCP by a DPP scan, too (six ds_bpermute round trips before).
    u32 pk0 = 0u, pk1 = 0u;
    u32 best_lcp = 0, best_text = 0, best_suff = 0;
    u32 smp[MAXCH], bas[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        if (c < nch) {
            smp[c] = ((u64)text[c] < a.nr_reads1) ? 0u : 1u;            // uint32 text vs uint64 nr_reads1
            bas[c] = base_to_int(bwt[c]);
            const u32 inc = valid[c] ? 1u << (8u * bas[c]) : 0u;
            pk0 += smp[c] ? 0u : inc; pk1 += smp[c] ? inc : 0u;
            const u32 m = wave_bcast_last(wave_incl_max(valid[c] ? lcp[c] : 0u));
            if (m > best_lcp) {                                          // strictly greater: earlier chunk wins ties
                const u64 who = __ballot(valid[c] && lcp[c] == m);
                const int src = __ffsll((long long)who) - 1;
                best_lcp = m;
                best_text = (u32)__shfl((int)text[c], src);
                best_suff = (u32)__shfl((int)suff[c], src);
            }
        }
    }
    if ((u64)best_lcp < a.k_right_u) return;                            // clust2snp.cpp:396
    const u32 t0 = wave_bcast_last(wave_incl_scan(pk0)), t1 = wave_bcast_last(wave_incl_scan(pk1));
    u32 f0 = 0, f1 = 0;                                                  // frequent characters, :402-407
#pragma unroll
    for (int b = 0; b < 4; ++b) { if (((t0 >> (8 * b)) & 0xFFu) >= a.mcov_u) f0 |= 1u << b; if (((t1 >> (8 * b)) & 0xFFu) >= a.mcov_u) f1 |= 1u << b; }
    if (f0 == 0 || f1 == 0 || __popc(f0) > 2 || __popc(f1) > 2 || f0 == f1 || __popc(f0 | f1) > 3) return;   // :419-429
    u32 pair = 0;
    // (the pairs (b0, b1) of frequent characters, b0 != b1, in the reference's order: only the set bits are walked)
    for (u32 q0 = f0; q0; q0 &= q0 - 1u) {
        const int b0 = __ffs((int)q0) - 1;
        for (u32 q1 = f1 & ~(1u << b0); q1; q1 &= q1 - 1u) {
            const int b1 = __ffs((int)q1) - 1;
            // int_to_base, raw compare ch == c (:455,466) -- from a packed constant: indexing the string literal with a run-time index is a
            // load from constant memory on every pair's critical path
            const u32 c0 = (0x54474341u >> (8 * b0)) & 0xFFu, c1 = (0x54474341u >> (8 * b1)) & 0xFFu;
            // membership and rank in SA order (:446-476)
            u32 n0 = 0, n1 = 0, rk0[MAXCH], rk1[MAXCH];
            bool m0[MAXCH], m1[MAXCH];
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                m0[c] = m1[c] = false; rk0[c] = rk1[c] = 0;
                if (c < nch) {
                    const bool okc = valid[c] && (u64)suff[c] >= a.k_left_u && (u64)lcp[c] >= a.k_right_u;
                    m0[c] = okc && bwt[c] == c0 && smp[c] == 0;
                    m1[c] = okc && bwt[c] == c1 && smp[c] == 1;
                    const u64 b0m = __ballot(m0[c]), b1m = __ballot(m1[c]);
                    const u64 lt = (1ull << lane) - 1ull;
                    rk0[c] = n0 + (u32)__popcll(b0m & lt);
                    rk1[c] = n1 + (u32)__popcll(b1m & lt);
                    n0 += (u32)__popcll(b0m); n1 += (u32)__popcll(b1m);
                }
            }
            const u32 e0 = min(n0, a.list_cap), e1 = min(n1, a.list_cap);  // size() < consensus_reads (:458,469)
            if (e0 > 0 && e1 > 0) {                                        // :478
                // the candidate in its packed form (include/e2s_hip.h, e2s_candidates_packed), written where `alloc` says:
                //   u32 n0, n1, right_idx, right_pos | idx0[n0] | idx1[n1] | pos0[n0] | pos1[n1] (pos_pack bytes each) | zero padding to 8 bytes
                const u32 size8 = (16u + (4u + a.pos_pack) * (e0 + e1) + 7u) >> 3;
                u32* w = alloc(lane, size8, (key << 2) | (u64)(pair & 3u));
                if (w) {
                    if (lane == 0) {
                        ((u64*)w)[size8 - 1u] = 0ull;                          // (the padding; list entries that share the unit follow in program order)
                        w[0] = e0; w[1] = e1; w[2] = best_text; w[3] = best_suff;
                    }
                    __builtin_amdgcn_wave_barrier();
                    uint8_t* const pb = (uint8_t*)(w + 4u + e0 + e1);
                    auto put_pos = [&](u32 at, u32 v) {
                        if (a.pos_pack == 1u) pb[at] = (uint8_t)v;
                        else if (a.pos_pack == 2u) { pb[2u * at] = (uint8_t)v; pb[2u * at + 1u] = (uint8_t)(v >> 8); }
                        else { pb[4u * at] = (uint8_t)v; pb[4u * at + 1u] = (uint8_t)(v >> 8); pb[4u * at + 2u] = (uint8_t)(v >> 16); pb[4u * at + 3u] = (uint8_t)(v >> 24); }
                    };
#pragma unroll
                    for (int c = 0; c < MAXCH; ++c) {
                        if (c < nch) {
                            if (m0[c] && rk0[c] < a.list_cap) { w[4u + rk0[c]] = text[c]; put_pos(rk0[c], suff[c] - a.k_left_lo); }
                            if (m1[c] && rk1[c] < a.list_cap) { w[4u + e0 + rk1[c]] = text[c]; put_pos(e0 + rk1[c], suff[c] - a.k_left_lo); }
                        }
                    }
                }
            }
            ++pair;
        }
    }
}

constexpr int FW_STAGE = 256;                // candidates a block stages in LDS (packed) before one global reservation
// a block's staging area for packed candidates: [s_used, 8-byte units) of the dynamic buffer; one LDS atomic per candidate
struct StageAlloc {
    uint8_t* buf; u32 cap8; u32* used8; u32* n; u64* keys; u32* offs;
    __device__ __forceinline__ u32* operator()(u32 lane, u32 size8, u64 key) const {
        u32 off = 0xFFFFFFFFu;
        if (lane == 0) {
            const u32 slot = atomicAdd(n, 1u);
            const u32 o = atomicAdd(used8, size8);
            if (slot < (u32)FW_STAGE && o + size8 <= cap8) { keys[slot] = key; offs[slot] = o; off = o; }
        }
        off = (u32)__shfl((int)off, 0);
        return off != 0xFFFFFFFFu ? (u32*)(buf + (size_t)off * 8) : nullptr;   // (never null: the kernel flushes while a round's worst case still fits)
    }
};

extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];   // packed candidates of a block (k_flagged_lists)

// ----------------------------------------------------------------------------------------------
// K2a, second formulation: the same pre-filter built like the scan kernel.  Every wave is a worker over a
// contiguous range of 4096-entry tiles (a lane owns 64 entries); everything is a 64-bit plane per lane: cluster
// starts / ends (the scan's stencil), BWT = C / G / T.  A cluster is evaluated by the lane that
// holds its END: its own piece [start or span begin, end] by population counts of the class planes, plus -- if
// the cluster began in an earlier lane -- a carry (packed counts + flags) handed from lane to lane with DPP
// wave_shr:1 (three rounds cover every cluster of <= 150 entries) and from tile to tile in an SGPR.  No LDS
// slots, no atomics, no barrier.  Only ENDs of clusters that contain two different bases at all are walked
// (bit-parallel: a carry-propagating add over the interior of the runs); a worker first runs the tile before
// its range without evaluating anything, to pick up the cluster that reaches into its range.
// ----------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_cluster_filter2(FilterArgs a) {
    constexpr int NR = 16, ES = 64, TILE = 4096, NW = NR + 2;
    __shared__ u64 s_q[4][F2_Q];
    u64* const q = s_q[threadIdx.x >> 6];
    const u32 lane = threadIdx.x & 63u;
    const u32 b = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    if (b >= a.w_blocks) return;
    const u64 t0 = a.w_tile0 + (u64)b * a.w_per + min(b, a.w_rem);
    const u64 t1 = min(t0 + (u64)a.w_per + (b < a.w_rem ? 1u : 0u), a.w_tile0 + a.w_tiles);
    const u64 torg = t0 > 0 ? t0 - 1 : t0;                      // offsets count from the tile before the range (the carries' 192 entries lie in it)
    const uint8_t* base = a.lcp;
    const uint8_t* bwtb = a.bwt;
    const int64_t n_prov = (int64_t)a.n_prov;
    const int64_t e_lo = (int64_t)a.own_lo, e_hi = (int64_t)a.end_hi;
    const u64 g_org = a.prov_lo + torg * TILE;                  // SA position of offset 0

    auto load_fast_w = [&](u64 tile, u32* W) __attribute__((always_inline)) {
        const uint8_t* p = base + (int64_t)tile * TILE;
        const u32 o = lane * 64u;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const uint4 v = *(const uint4*)(p + o + 16 * qd);
            W[1 + 4 * qd + 0] = v.x; W[1 + 4 * qd + 1] = v.y; W[1 + 4 * qd + 2] = v.z; W[1 + 4 * qd + 3] = v.w;
        }
        W[0] = *(const u32*)(p + o - 4);
        W[NR + 1] = *(const u32*)(p + o + 64);
    };
    auto load_fast_b = [&](u64 tile, u32* B) __attribute__((always_inline)) {
        const uint8_t* pb = bwtb + (int64_t)tile * TILE;
        const u32 o = lane * 64u;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const uint4 w = *(const uint4*)(pb + o + 16 * qd);
            B[4 * qd + 0] = w.x; B[4 * qd + 1] = w.y; B[4 * qd + 2] = w.z; B[4 * qd + 3] = w.w;
        }
    };
    auto load_fast = [&](u64 tile, u32* W, u32* B) __attribute__((always_inline)) { load_fast_w(tile, W); load_fast_b(tile, B); };
    auto load_slow = [&](u64 tile, u32* W, u32* B) __attribute__((always_inline)) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)lane * ES;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            u32 v = 0, vb = 0;
#pragma unroll
            for (int bt = 0; bt < 4; ++bt) {
                const int64_t e = i0 + (int64_t)(w - 1) * 4 + bt;
                if (e >= 0 && e < n_prov) { v |= (u32)base[e] << (8 * bt); vb |= (u32)bwtb[e] << (8 * bt); }
            }
            W[w] = v;
            if (w >= 1 && w <= NR) B[w - 1] = vb;
        }
    };
    auto entry_at = [&](int64_t i) -> u32 { return (i >= 0 && i < n_prov) ? (u32)base[i] : 0u; };

    u32 open1 = 0;                                              // 1 + offset of the last start seen (0: none yet)
    u32 cin0 = 0;                                               // carry into lane 0 of the next tile: counts | flags << 24
    u32 ccarry = 0;                                             // class code of the entry before the tile (2 bits)
    u32 qn = 0;                                                 // queued flagged clusters (wave-uniform)
    u32 cb = 0;                                                 // GE / end of the entry before the first tile
    auto range_carry = [&]() -> u32 {                           // (evaluated after the first tile's loads have been issued)
        const int64_t r0 = (int64_t)t0 * TILE;
        const u32 lm2 = entry_at(r0 - 2), lm1 = entry_at(r0 - 1), l0 = entry_at(r0);
        const bool gm1 = a.hge != 0 && lm1 >= a.k_u32, g0 = a.hge != 0 && l0 >= a.k_u32;
        const bool em1 = gm1 && ((lm2 > lm1 && lm1 <= l0) || !g0);
        return (u32)__builtin_amdgcn_readfirstlane((int)((gm1 ? 1u : 0u) | (em1 ? 2u : 0u)));
        // (the class of the entry before the first tile only matters for a cluster that is rejected for its length)
    };
    auto queue_flush = [&]() {
        if (qn) {
            u64 gbase = 0;
            if (lane == 0) gbase = atomicAdd(a.fl_count, (unsigned long long)qn);
            gbase = __shfl(gbase, 0);
            if (lane < qn) {
                const u64 v = q[lane];
                if (gbase + lane < a.fl_cap) { a.fl_starts[gbase + lane] = v & 0xFFFFFFFFFFFFull; a.fl_lens[gbase + lane] = (uint16_t)(v >> 48); }
            }
            __builtin_amdgcn_wave_barrier();
        }
        qn = 0;
    };

    u32 WA[NW], BA[NR];                                         // ONE register set: the next tile's words are asked for into it once the planes are built (k_scan_seg)
    ClassConsts kc = class_consts();                            // held in VGPRs: see e2s_swar.h
    asm volatile("" : "+v"(kc.m07), "+v"(kc.ehi), "+v"(kc.elo), "+v"(kc.c20), "+v"(kc.one), "+v"(kc.h));
    auto body = [&](const u64 tile, u32* W, u32* B, auto prefetch) __attribute__((always_inline)) {
        constexpr bool PF = decltype(prefetch)::value;
        u32* const Wn = W; u32* const Bn = B;
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)lane * ES;
        u64 S, E, GE;
        {
            u64 NLE; u32 after;
            span_pred<1, NR>(W, a.kb, a.hge, GE, NLE, after);
            E = plane_end<ES>(GE, NLE, after);
            const u32 top = (u32)((GE >> (ES - 1)) & 1ull) | ((u32)((E >> (ES - 1)) & 1ull) << 1);
            const u32 before = wave_shr1(top, cb);
            S = plane_start<ES>(GE, E, before);
            cb = (u32)__builtin_amdgcn_readlane((int)top, 63);
        }
        u64 IN = GE;                                            // entry lies in a cluster
        if (a.prov_is_global0 && i0 == 0) {                     // the j = 0,1 rules (see k_scan_seg)
            const u32 e0v = W[1] & 0xFFu, e1v = (W[1] >> 8) & 0xFFu, e2v = (W[1] >> 16) & 0xFFu;
            E &= ~1ull;
            if (a.hge != 0 && e0v >= a.k_u32) {
                const bool end1 = ((e0v > e1v) && (e1v <= e2v)) || (e2v < a.k_u32);
                S &= ~2ull;
                E = end1 ? (E | 2ull) : (E & ~2ull);
                if (e2v >= a.k_u32) S = end1 ? (S | 4ull) : (S & ~4ull);
                IN |= 2ull;                                     // entry 1 belongs to the cluster opened at 0 whatever L[1] is
            }
        }
        u64 c0, c1;                                             // code planes: c0 = C | G, c1 = G | T (e2s_swar.h)
        span_bwt_code_planes<NR>(B, kc, c0, c1);
        if (PF) { load_fast_w(tile + 1, Wn); load_fast_b(tile + 1, Bn); }
        const u64 C = c0 & ~c1, G = c0 & c1, T = ~c0 & c1;
        // (LCP >= k_right somewhere in the cluster, clust2snp.cpp:396, is left to K2b's exact test: dropping the plane here lets
        // through only the rare cluster whose LCPs all lie in [k, k_right) and saves 7 % of this kernel's instructions)
        // class changes inside clusters: F(e) = class(e) != class(e-1), e in a cluster and not its first entry
        const u32 ctop = (u32)(c0 >> 63) | ((u32)(c1 >> 63) << 1);
        const u32 cprev = wave_shr1(ctop, ccarry);
        ccarry = (u32)__builtin_amdgcn_readlane((int)ctop, 63);
        const u64 F = ((c0 ^ ((c0 << 1) | (u64)(cprev & 1u))) | (c1 ^ ((c1 << 1) | (u64)((cprev >> 1) & 1u)))) & IN & ~S;
        // what the lane hands on: counts and flags from its last start (or, without a start, its whole span) to its end
        const bool hasS = S != 0ull;
        const int msbS = 63 - __clzll(S | 1ull);
        const u64 tm = hasS ? ~((1ull << msbS) - 1ull) : ~0ull;
        auto packed = [&](u64 m) -> u32 {
            return (u32)__popcll(C & m) | ((u32)__popcll(G & m) << 8) | ((u32)__popcll(T & m) << 16) | ((F & m) ? (1u << 25) : 0u);
        };
        const u32 local = packed(tm);
        auto merge = [](u32 x, u32 y) -> u32 { return (((x & 0xFFFFFFu) + (y & 0xFFFFFFu)) & 0xFFFFFFu) | ((x | y) & 0xFF000000u); };
        u32 out = local;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const u32 in = wave_shr1(out, cin0);
            out = hasS ? local : merge(local, in);
        }
        const u32 cin = wave_shr1(out, cin0);                   // carry into this lane
        cin0 = (u32)__builtin_amdgcn_readlane((int)out, 63);
        // last start at or below each lane (as in the scan)
        const u32 base_off = (u32)((tile - torg) * TILE) + lane * ES;
        const u32 ls1 = hasS ? base_off + (u32)msbS + 1u : 0u;
        const u32 mx = wave_incl_max(ls1);
        const u32 cur1 = max(wave_shr1(mx, 0u), open1);
        open1 = max(open1, (u32)__builtin_amdgcn_readlane((int)mx, 63));
        if (PF) {
            static_assert(NW == 18 && NR == 16, "operand lists below");
            asm volatile("" : "+v"(Wn[0]), "+v"(Wn[1]), "+v"(Wn[2]), "+v"(Wn[3]), "+v"(Wn[4]), "+v"(Wn[5]), "+v"(Wn[6]), "+v"(Wn[7]), "+v"(Wn[8]),
                              "+v"(Wn[9]), "+v"(Wn[10]), "+v"(Wn[11]), "+v"(Wn[12]), "+v"(Wn[13]), "+v"(Wn[14]), "+v"(Wn[15]), "+v"(Wn[16]), "+v"(Wn[17]));
            asm volatile("" : "+v"(Bn[0]), "+v"(Bn[1]), "+v"(Bn[2]), "+v"(Bn[3]), "+v"(Bn[4]), "+v"(Bn[5]), "+v"(Bn[6]), "+v"(Bn[7]),
                              "+v"(Bn[8]), "+v"(Bn[9]), "+v"(Bn[10]), "+v"(Bn[11]), "+v"(Bn[12]), "+v"(Bn[13]), "+v"(Bn[14]), "+v"(Bn[15]));
        }
        // ENDs to evaluate: owned, and of clusters with a class change somewhere (incoming flag: bit 25 of the carry)
        u64 Ev = E;
        const int64_t tl = (int64_t)tile * TILE;
        if (!(tl >= e_lo && tl + TILE <= e_hi)) {
            u64 ke = 0;
#pragma unroll 1
            for (int f = 0; f < ES; ++f) { const int64_t e = i0 + f; if (e >= e_lo && e < e_hi) ke |= 1ull << f; }
            Ev &= ke;
        }
        const u64 Fin = F | ((((cin >> 25) & 1u) && (IN & ~S & 1ull)) ? 1ull : 0ull);
        const u64 R = IN & ~E;
        u64 Ei = Ev & (((R + (Fin & R)) & E) | (Fin & E));
        if (qn > (u32)(F2_Q / 2)) queue_flush();
        // wave-uniform loop (every lane stays until the last one is done): the queue position is wave state
        while (__ballot(Ei != 0ull)) {
            const bool act = Ei != 0ull;
            const int e = act ? __ffsll((long long)Ei) - 1 : 0;
            Ei &= Ei - 1;
            const u64 upto = (2ull << e) - 1ull;
            const u64 sc = S & upto;
            const bool have = sc != 0ull;
            const int ms = 63 - __clzll(sc | 1ull);
            const u64 rm = have ? (upto & ~((1ull << ms) - 1ull)) : upto;
            u32 pk = (u32)__popcll(C & rm) | ((u32)__popcll(G & rm) << 8) | ((u32)__popcll(T & rm) << 16);
            if (!have) pk = merge(pk, cin) & 0xFFFFFFu;
            const u32 st1 = have ? base_off + (u32)ms + 1u : cur1;
            const u32 len = base_off + (u32)e + 2u - st1;
            const u32 cC = pk & 0xFFu, cG = (pk >> 8) & 0xFFu, cT = (pk >> 16) & 0xFFu, cA = len - cC - cG - cT;
            const u32 nfreq = (cA >= a.mcov_u ? 1u : 0u) + (cC >= a.mcov_u ? 1u : 0u) + (cG >= a.mcov_u ? 1u : 0u) + (cT >= a.mcov_u ? 1u : 0u);
            // st1 == 0: the cluster began more than 192 entries before the range, i.e. it fails the length test
            const bool pass = act && st1 != 0u && (int)len >= a.len_lo && (int)len <= a.len_hi && nfreq >= 2u;
            const u64 pm = __ballot(pass);
            if (pass) {
                const u32 idx = qn + (u32)__popcll(pm & ((1ull << lane) - 1ull));
                if (idx < (u32)F2_Q) q[idx] = (g_org + (u64)(st1 - 1u)) | ((u64)len << 48);
                else atomicOr(a.err, E2S_DEVERR_DENSE);
            }
            qn = min(qn + (u32)__popcll(pm), (u32)F2_Q);
        }
    };
    {
        const std::true_type Y{}; const std::false_type N{};
        u64 tile = torg;
        const u64 tfe = n_prov >= (int64_t)(TILE + 4) ? (u64)((n_prov - 4) / TILE) : 0ull;   // interior tiles are [1, tfe)
        const u64 tf = min(t1, tfe);
        // what the range inherits: GE / end of the entry before it, and the carries of a cluster that reaches into it (from the 192
        // entries before the range; offsets count from the tile before the range)
        auto start_carry = [&]() __attribute__((always_inline)) {
            cb = range_carry();
            if (t0 > 0 && t0 < t1) {
                const PfCarry pc = prefilter_carry_192(base, bwtb, (int64_t)t0 * TILE, a.k_u32, a.hge, lane);
                open1 = pc.open1; cin0 = pc.cin0; ccarry = pc.ccarry;
            }
        };
        tile = t0;
        bool have_cb = false;
        auto slow_run = [&](u64 upto) __attribute__((always_inline)) { for (; tile < upto; ++tile) { load_slow(tile, WA, BA); body(tile, WA, BA, N); } };
        if (!have_cb && tile < min(t1, (u64)1)) { start_carry(); have_cb = true; }
        slow_run(min(t1, (u64)1));
        if (tile < tf) {
            load_fast(tile, WA, BA);
            if (!have_cb) { start_carry(); have_cb = true; }
            for (; tile + 1 < tf; ++tile) body(tile, WA, BA, Y);
            body(tile, WA, BA, N); tile += 1;
        }
        if (!have_cb) start_carry();
        slow_run(t1);
    }
    queue_flush();
}

// ---- packed candidate record (what crosses PCIe; include/e2s_hip.h, e2s_candidates_packed):
//   u32 n0, n1, right_idx, right_pos | u32 idx0[n0] | u32 idx1[n1] | pos0[n0] | pos1[n1] (pw bytes each) | pad to 8 bytes
// built from a fixed-stride record [CandHeader][idx0[cap]][pos0[cap]][idx1[cap]][pos1[cap]] by the 64 lanes of a wave.
__device__ __forceinline__ u32 packed_size8(u32 n0, u32 n1, u32 pw) { return (16u + (4u + pw) * (n0 + n1) + 7u) >> 3; }
// K2b: lists and candidate records (clust2snp.cpp:431-496) for the few clusters K2a let through.  Persistent
// blocks, one wave per cluster, candidates staged in LDS so that a block reserves output slots ~once per 32.
#ifndef K2B_WAVES
#define K2B_WAVES 1
#endif
template <bool DEF>
__global__ __launch_bounds__(256, K2B_WAVES) void k_flagged_lists(CandArgs a, const unsigned long long* lo_dev, const unsigned long long* n_dev, u32 cap8) {
    __shared__ u32 s_nstage, s_used8;
    __shared__ u64 s_gbase, s_sbase;
    __shared__ u64 s_keys[FW_STAGE];
    __shared__ u32 s_offs[FW_STAGE];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) { s_nstage = 0; s_used8 = 0; }
    __syncthreads();
    // a round adds at most 4 waves x 4 pairs of the largest record: flush while that still fits
    const u32 max8 = (16u + (4u + a.pos_pack) * 2u * a.list_cap + 7u) >> 3;
    auto stage_flush = [&]() {
        const u32 n = min(s_nstage, (u32)FW_STAGE), used = min(s_used8, cap8);
        if (n) {
            // ONE reservation of slots and of stream space per flush (global atomics on one word saturate at ~90 per microsecond:
            // with 32 fixed-stride records per flush they were half of this kernel's time), then the block's chunk leaves in 8-byte units
            if (tid == 0) {
                s_gbase = atomicAdd(a.cand_count, (unsigned long long)n);
                s_sbase = atomicAdd(a.stream8_count, (unsigned long long)used);
            }
            __syncthreads();
            const u64 gbase = s_gbase, sbase = s_sbase;
            for (u32 i = tid; i < n; i += 256u) {
                const u64 slot = gbase + i;
                if (slot < a.cand_cap) { a.slot_off8[slot] = (u32)(sbase + s_offs[i]); a.cand_keys[slot] = s_keys[i]; }
            }
            if (gbase + n <= a.cand_cap && sbase + used <= a.stream8_cap) {
                uint2* dst = (uint2*)(a.host_stream + sbase * 8);
                const uint2* src = (const uint2*)s_stage;
                for (u32 w = tid; w < used; w += 256u) dst[w] = src[w];
            }
        }
        __syncthreads();
        if (tid == 0) { s_nstage = 0; s_used8 = 0; }
        __syncthreads();
    };
    if (a.len_hi_dev) a.len_hi = *a.len_hi_dev;                  // (the resident job: max_clust_length was worked out on the device)
    const u64 n = min((u64)*n_dev, a.n_rec);                     // this launch handles list entries [*lo_dev, *n_dev)
    const u64 first = lo_dev ? (u64)*lo_dev : 0ull;
    const u64 per_round = (u64)gridDim.x * 4;
    // A round is a chain of dependent memory latencies (entry -> its EGSA columns -> the test), and the block's four waves meet at the
    // barrier below every round: a wave keeps TWO rounds ahead in flight -- the list entry of round r + 2 and the four columns of round
    // r + 1's cluster are fetched before it works on round r's (registers only).
    // which entries are worked on at all: length 0 = an unused reserved slot; the exact length bounds (the pre-filter may have run with
    // the absolute maximum, before max_clust_length was known); entries must be resident (index n_total being the EOF phantom): a
    // record from a broken list must not fault
    auto entry_kind = [&](u64 s, int len) -> int {                  // 0: skip, 1: work on it, 2: outside the shard (error)
        if (len == 0 || len < a.len_lo || len > a.len_hi) return 0;
        if (s < a.prov_lo || s + (u64)len > a.prov_hi + (a.prov_hi == a.n_total ? 1ull : 0ull) || len < 1 || len > 3 * 64) return 2;
        return 1;
    };
    u64 s_n2 = 0, k_n2 = 0; int len_n2 = 0;                        // round r + 2's entry
    u64 s_n1 = 0, k_n1 = 0; int len_n1 = 0, kind_n1 = 0;          // round r + 1's entry, its columns on their way
    // Only a cluster's FIRST chunk (its first 64 entries) is fetched ahead: most flagged clusters have no more (the mode of the lengths is the
    // coverage), the registers of two more chunks per cluster in flight bought nothing for them, and a longer cluster fetches its other
    // chunks when its round comes (one exposed round trip for the few).
    ClusterColsN<1> c_n1;
    {
        const u64 r0 = first + (u64)blockIdx.x * 4 + wave;
        if (r0 < n) { s_n1 = a.starts[r0]; len_n1 = (int)a.lens[r0]; k_n1 = a.keys ? a.keys[r0] : s_n1; }
        const u64 r1 = r0 + per_round;
        if (r1 < n) { s_n2 = a.starts[r1]; len_n2 = (int)a.lens[r1]; k_n2 = a.keys ? a.keys[r1] : s_n2; }
        kind_n1 = r0 < n ? entry_kind(s_n1, len_n1) : 0;
        cluster_load<DEF, 1>(a, s_n1, kind_n1 == 1 ? len_n1 : 0, lane, c_n1);
    }
    for (u64 base = first + (u64)blockIdx.x * 4; base < n; base += per_round) {
        // (flushing earlier to start the PCIe stream sooner was tried: chunks of ~8 records ran 25 % slower)
        if (s_nstage > (u32)FW_STAGE - 16u || s_used8 + 16u * max8 > cap8) stage_flush();
        const u64 r = base + wave;
        const u64 key = k_n1, s_cur = s_n1; const int len = len_n1, kind = kind_n1;
        const ClusterColsN<1> cur = c_n1;
        // round r + 1: its entry arrived a round ago -- its columns leave now; round r + 2: its entry leaves now
        s_n1 = s_n2; len_n1 = len_n2; k_n1 = k_n2;
        kind_n1 = r + per_round < n ? entry_kind(s_n1, len_n1) : 0;
        cluster_load<DEF, 1>(a, s_n1, kind_n1 == 1 ? len_n1 : 0, lane, c_n1);
        { const u64 rn = r + 2 * per_round; if (rn < n) { s_n2 = a.starts[rn]; len_n2 = (int)a.lens[rn]; k_n2 = a.keys ? a.keys[rn] : s_n2; } }
        if (r < n) {
            if (kind == 2) { if (lane == 0) atomicOr(a.err, E2S_DEVERR_REMOTE); }
            else if (kind == 1) {
                const StageAlloc alloc{s_stage, cap8, &s_used8, &s_nstage, s_keys, s_offs};
                if (len <= 64) cluster_record<DEF, 1>(a, cur, len, key, lane, alloc);
                else {
                    ClusterColsN<MAXCH - 1> rest;
                    cluster_load<DEF, MAXCH - 1, 1>(a, s_cur, len, lane, rest);
                    ClusterCols all;
                    all.text[0] = cur.text[0]; all.suff[0] = cur.suff[0]; all.lcp[0] = cur.lcp[0]; all.bwt[0] = cur.bwt[0];
#pragma unroll
                    for (int c = 1; c < MAXCH; ++c) { all.text[c] = rest.text[c - 1]; all.suff[c] = rest.suff[c - 1]; all.lcp[c] = rest.lcp[c - 1]; all.bwt[c] = rest.bwt[c - 1]; }
                    cluster_record<DEF, MAXCH>(a, all, len, key, lane, alloc);
                }
            }
        }
        __syncthreads();
    }
    stage_flush();
    if (a.snap_out && tid == 0) {                                // (stage_flush ended in a barrier: the block's reservations are all made)
        __threadfence();
        if (atomicAdd(a.done_ctr, 1u) == gridDim.x - 1u) *a.snap_out = atomicAdd(a.cand_count, 0ull);
    }
}

// ----------------------------------------------------------------------------------------------
// KO: SA order of the candidates.  Output slots are claimed with atomics, so the order of the buffer is
// arbitrary; the reference emits candidates in SA order (cluster, then the (c0,c1) loop order).  Keys are
// unique and spread over the shard's SA range, so they are ranked by bucket: count per bucket of the key
// range (~2 keys per bucket), exclusive scan of the counts, group the keys by bucket, rank inside the
// bucket.  O(n) work for evenly spread keys (the first version ranked by counting all pairs: 54 us for
// 1.7x10^4 keys and quadratic beyond).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 ko_bucket(u64 key, u64 key_lo, u32 shift, u32 nb) {
    const u64 d = key >= key_lo ? key - key_lo : 0ull;
    const u64 b = d >> shift;
    return b < nb ? (u32)b : nb - 1u;
}
// n_dev (optional): the count lives on the device (the launch is sized for the capacity n)
__device__ __forceinline__ u64 ko_n(u64 n, const unsigned long long* n_dev) { return n_dev ? min(n, (u64)*n_dev) : n; }
// The scan for 1024 .. 16384 buckets (what e2s_cand_order_buckets gives up to 5x10^5 candidates) by ONE SMALL block: four waves, a
// thread's <= 64 counts in its registers between the two halves of the scan (one round trip), the waves' totals through four words of LDS.
// The 1024-thread form below took 30 us whatever the number of buckets -- not for its work: the ordering pass runs beside the consensus
// kernel, which fills every wave slot of the chip, and a block of sixteen waves waits until one CU has FOUR free slots on each SIMD at
// once, about a consensus wave's lifetime; one wave per SIMD finds room as soon as anything ends.  (With the consensus kernel at 42 us for
// a rank of eight, the ordering pass beside it had become the longer of the two.)
__device__ __forceinline__ void ko_scan256(u32* cnt, u32 nb, u32* s_w /* [4], shared */) {
    const u32 tid = threadIdx.x, per = nb / 256u, lo = tid * per;  // per: 4, 8, 16, 32 or 64
    uint4 v[16];
#pragma unroll
    for (u32 q = 0; q < 16u; ++q) v[q] = 4u * q < per ? *(const uint4*)(cnt + lo + 4u * q) : make_uint4(0, 0, 0, 0);
    u32 sum = 0;
#pragma unroll
    for (u32 q = 0; q < 16u; ++q) sum += v[q].x + v[q].y + v[q].z + v[q].w;
    const u32 inc = wave_incl_scan(sum);
    if ((tid & 63u) == 63u) s_w[tid >> 6] = inc;
    __syncthreads();
    u32 run = inc - sum;
    for (u32 w = 0; w < (tid >> 6); ++w) run += s_w[w];
#pragma unroll
    for (u32 q = 0; q < 16u; ++q) {
        if (4u * q < per) {
            uint4 o;
            o.x = run; run += v[q].x; o.y = run; run += v[q].y; o.z = run; run += v[q].z; o.w = run; run += v[q].w;
            *(uint4*)(cnt + lo + 4u * q) = o;
        }
    }
    if (tid == 255u) cnt[nb] = run;
}
__global__ void k_ko_count(const u64* keys, u64 n, const unsigned long long* n_dev, u64 key_lo, u32 shift, u32 nb, u32* cnt, u32* slot) {
    n = ko_n(n, n_dev);
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slot[i] = atomicAdd(&cnt[ko_bucket(keys[i], key_lo, shift, nb)], 1u);
}
// (Tried: the count kernel's LAST block -- a ticket -- doing the scan, one launch less on the chain: 77 us instead of 9 + 15 -- two thousand
//  blocks' tickets on one word, and a device-scope fence per block that empties the L1 under the consensus kernel beside it: 39 -> 92 us.)
__global__ __launch_bounds__(256) void k_ko_scan256(u32* cnt, u32 nb) {
    __shared__ u32 s_w[4];
    ko_scan256(cnt, nb, s_w);
}
// exclusive scan of cnt[0..nb) in place, total at cnt[nb] (one block; nb a power of two >= 256, or a multiple of 4096).  A thread owns `per`
// consecutive counts and moves them 16 at a time (four 16-byte loads in flight), not one global round trip each.
__global__ __launch_bounds__(1024) void k_ko_scan(u32* cnt, u32 nb) {
    __shared__ u32 s_part[1024];
    const u32 tid = threadIdx.x;
    const u32 per = nb >= 1024u ? nb / 1024u : 1u;             // power of two
    const u32 lo = tid * per;
    const bool act = lo < nb;
    u32 sum = 0;
    if (act) {
        if (per >= 4u) {
            for (u32 i = 0; i < per; i += 16u) {
                uint4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (i + 4u * q < per) ? *(const uint4*)(cnt + lo + i + 4u * q) : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) sum += v[q].x + v[q].y + v[q].z + v[q].w;
            }
        } else for (u32 i = 0; i < per; ++i) sum += cnt[lo + i];
    }
    s_part[tid] = sum;
    __syncthreads();
    if (tid < 64) {                           // 1024 partials: 16 per lane, wave scan
        u32 loc = 0;
        for (int i = 0; i < 16; ++i) loc += s_part[tid * 16 + i];
        const u32 inc = wave_incl_scan(loc);
        u32 run = inc - loc;
        for (int i = 0; i < 16; ++i) { const u32 v = s_part[tid * 16 + i]; s_part[tid * 16 + i] = run; run += v; }
        if (tid == 63) cnt[nb] = run;
    }
    __syncthreads();
    if (!act) return;
    u32 run = s_part[tid];
    if (per >= 4u) {
        for (u32 i = 0; i < per; i += 16u) {
            uint4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (i + 4u * q < per) ? *(const uint4*)(cnt + lo + i + 4u * q) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i + 4u * q < per) {
                    uint4 o;
                    o.x = run; run += v[q].x; o.y = run; run += v[q].y; o.z = run; run += v[q].z; o.w = run; run += v[q].w;
                    *(uint4*)(cnt + lo + i + 4u * q) = o;
                }
            }
        }
    } else for (u32 i = 0; i < per; ++i) { const u32 c = cnt[lo + i]; cnt[lo + i] = run; run += c; }
}
__global__ void k_ko_place(const u64* keys, u64 n, const unsigned long long* n_dev, u64 key_lo, u32 shift, u32 nb, const u32* off, const u32* slot, u64* gkeys, u32* gidx) {
    n = ko_n(n, n_dev);
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = keys[i];
    const u32 at = off[ko_bucket(k, key_lo, shift, nb)] + slot[i];
    gkeys[at] = k; gidx[at] = (u32)i;
}
__global__ void k_ko_rank(const u64* gkeys, const u32* gidx, u64 n, const unsigned long long* n_dev, u64 key_lo, u32 shift, u32 nb, const u32* off, u32* order) {
    n = ko_n(n, n_dev);
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u64 k = gkeys[j];
    const u32 b = ko_bucket(k, key_lo, shift, nb), lo = off[b], hi = off[b + 1];
    u32 r = lo;
    for (u32 t = lo; t < hi; ++t) r += gkeys[t] < k ? 1u : 0u;
    order[r] = gidx[j];
}
// bad (the resident job): 1 where the pass overflowed a buffer or left an error bit -- what the host checks in the mail, for the tail's kernels
__global__ void k_cand_offsets(const u32* order, const u32* slot_off8, u64 n, const unsigned long long* n_dev, const unsigned long long* stream8_count, u32* off_out,
                               const unsigned long long* fl_count, const u32* err_a, const u32* err_b, unsigned long long* mail, u32* bad, u64 fl_cap, u64 stream8_cap,
                               unsigned long long* zero, u64 n_zero, u32* ko_cnt, u32 n_ko) {
    const u64 cap = n;
    n = ko_n(n, n_dev);
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 z = i; z < n_zero; z += (u64)gridDim.x * blockDim.x) zero[z] = 0ull;     // (the resident job: the tail's sums, cleared here instead of by a memset)
    for (u64 z = i; z < n_ko; z += (u64)gridDim.x * blockDim.x) ko_cnt[z] = 0u;       // the ordering pass's bucket counts: left clean for the next pass
    if (i < n) off_out[i] = slot_off8[order[i]];
    else if (i == n) {
        off_out[n] = (u32)*stream8_count;
        if (mail) { mail[0] = *n_dev; mail[1] = *fl_count; mail[2] = (unsigned long long)*err_a | ((unsigned long long)*err_b << 32); mail[3] = *stream8_count; }
        if (bad) *bad = (*n_dev > cap || *fl_count > fl_cap || *stream8_count > stream8_cap || *err_a != 0u || *err_b != 0u) ? 1u : 0u;
    }
}
void e2s_launch_cand_offsets(const u32* order, const u32* slot_off8, u64 n, const unsigned long long* n_dev, const unsigned long long* stream8_count, u32* off_out,
                             const unsigned long long* fl_count, const u32* err_a, const u32* err_b, unsigned long long* mail, u32* bad, u64 fl_cap, u64 stream8_cap,
                             unsigned long long* zero, u64 n_zero, u32* ko_cnt, u32 n_ko, hipStream_t st) {
    hipLaunchKernelGGL(k_cand_offsets, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, order, slot_off8, n, n_dev, stream8_count, off_out,
                       fl_count, err_a, err_b, mail, bad, fl_cap, stream8_cap, zero, n_zero, ko_cnt, n_ko);
}
size_t e2s_cand_order_scratch_bytes(u64 n) { return ((size_t)E2S_KO_MAX_BUCKETS + 2 + 2 * n + 2) * 4 + n * 8; }
// ~2 keys per bucket up to 2^14 buckets, then up to 32 per bucket before the table grows again: the single-block scan of the counts
// is a chain of global round trips (2^16 buckets: 50-70 us on c3, more than the three other ordering kernels together), while a key
// ranks itself against its bucket out of L2
uint32_t e2s_cand_order_buckets(unsigned long long n) {
    u32 nb = 256;
    while (nb < 16384u && (u64)nb * 2 < n) nb <<= 1;
    while (nb < 65536u && nb < E2S_KO_MAX_BUCKETS && (u64)nb * 32 < n) nb <<= 1;
    return nb;
}
// counts_cleared: the caller has zeroed the first e2s_cand_order_buckets(n) + 1 words of `scratch` already (k_cand_init)
void e2s_launch_cand_order(const u64* keys, u64 n, const unsigned long long* n_dev, u64 key_lo, u64 key_span, void* scratch, u32* order, hipStream_t st,
                           bool counts_cleared) {
    if (!n) return;
    const u32 nb = e2s_cand_order_buckets(n);
    u32 shift = 0;
    while (shift < 63 && (key_span >> shift) >= (u64)nb) ++shift;           // (key - key_lo) >> shift < nb for key - key_lo < key_span
    u32* cnt = (u32*)scratch;
    u32* slot = cnt + E2S_KO_MAX_BUCKETS + 2;
    u32* gidx = slot + n;
    u64* gkeys = (u64*)(((uintptr_t)(gidx + n) + 7) & ~(uintptr_t)7);
    if (!counts_cleared) (void)hipMemsetAsync(cnt, 0, ((size_t)nb + 1) * 4, st);   // (a failure stays in hipGetLastError, which the caller reads after the launches)
    const dim3 g((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_ko_count, g, blk, 0, st, keys, n, n_dev, key_lo, shift, nb, cnt, slot);
    if (nb >= 1024u && nb <= 16384u) hipLaunchKernelGGL(k_ko_scan256, dim3(1), dim3(256), 0, st, cnt, nb);
    else hipLaunchKernelGGL(k_ko_scan, dim3(1), dim3(1024), 0, st, cnt, nb);
    hipLaunchKernelGGL(k_ko_place, g, blk, 0, st, keys, n, n_dev, key_lo, shift, nb, (const u32*)cnt, (const u32*)slot, gkeys, gidx);
    hipLaunchKernelGGL(k_ko_rank, g, blk, 0, st, (const u64*)gkeys, (const u32*)gidx, n, n_dev, key_lo, shift, nb, (const u32*)cnt, order);
}

// ----------------------------------------------------------------------------------------------
// KU: file layouts -> struct-of-arrays in HBM (8-byte fields truncated to 32 bits like t_GSA)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 rd_le(const uint8_t* p, int w) {
    u32 v = 0;
    const int m = w < 4 ? w : 4;
    for (int i = 0; i < m; ++i) v |= (u32)p[i] << (8 * i);
    return v;
}
__device__ __forceinline__ void wr_le(void* base, int w, u64 i, u32 v) {
    if (w == 1) ((uint8_t*)base)[i] = (uint8_t)v; else if (w == 2) ((uint16_t*)base)[i] = (uint16_t)v; else ((u32*)base)[i] = v;
}

// pairSA = [pos:z][da:y] per entry (include.hpp:159-175)
__global__ void k_unpack_pairsa(const uint8_t* src, u64 n, int zb_file, int yb_file, void* pos, int zb, void* da, int yb) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = src + i * (u64)(zb_file + yb_file);
    wr_le(pos, zb, i, rd_le(p, zb_file));
    wr_le(da, yb, i, rd_le(p + zb_file, yb_file));
}
// narrow an 8-byte little-endian column to 4 bytes (LCP file with -x 8)
__global__ void k_narrow8(const uint8_t* src, u64 n, u32* dst) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = rd_le(src + i * 8, 4);
}
// .gesa = [da:y][pos:z][lcp:x][bwt:1] per entry (include.hpp:126-155)
__global__ void k_unpack_gesa(const uint8_t* src, u64 n, int yb_file, int zb_file, int xb_file,
                              uint8_t* bwt, void* lcp, int xb, void* pos, int zb, void* da, int yb) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = src + i * (u64)(yb_file + zb_file + xb_file + 1);
    wr_le(da, yb, i, rd_le(p, yb_file));
    wr_le(pos, zb, i, rd_le(p + yb_file, zb_file));
    wr_le(lcp, xb, i, rd_le(p + yb_file + zb_file, xb_file));
    bwt[i] = p[yb_file + zb_file + xb_file];
}

// ----------------------------------------------------------------------------------------------
// launchers (called from e2s_capi.cpp through e2s_internal.h)
// ----------------------------------------------------------------------------------------------
static int resident_blocks(const void* fn, int threads, size_t dyn_lds);
int e2s_scan_seg_resident_blocks(int xb) {                   // resident WORKERS (waves)
    const void* fn = xb == 1 ? (const void*)k_scan_seg<1, false> : xb == 2 ? (const void*)k_scan_seg<2, false> : (const void*)k_scan_seg<4, false>;
    // (the kernels are bound by their vector instructions: 3, 4 or 5 workers per SIMD run at the same speed -- profiles/round4_scan_workers.txt:
    //  0.766-0.769 ms for K1 on c3 -- 2 cost 8 %, and the 7 that K1's 66 VGPRs would allow cost 5 %: more workers, each with its start-up,
    //  its descriptor and its share of the histogram's flush.  At most five per SIMD.)
    int per_cu = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    per_cu = resident_blocks(fn, 64, 0) / (cus > 0 ? cus : 1);     // one-wave blocks
    return (per_cu < 20 ? per_cu : 20) * cus;
}
#ifdef E2S_DIAG_STAMPS
extern "C" int e2s_diag_scan_stamps(void* out, size_t bytes) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_e2s_stamps), bytes < sizeof(g_e2s_stamps) ? bytes : sizeof(g_e2s_stamps));
}
#endif
int e2s_scan_seg_fused_resident_blocks() { return resident_blocks((const void*)k_scan_seg<1, true>, 64, 0); }
void e2s_launch_scan_seg(const SegArgs& a, int xb, hipStream_t st) {
    const dim3 grid(a.n_blocks), blk(64);                       // a range = a wave = a block
    const FilterArgs f{};
    if (xb == 1) hipLaunchKernelGGL((k_scan_seg<1, false>), grid, blk, 0, st, a, f);
    else if (xb == 2) hipLaunchKernelGGL((k_scan_seg<2, false>), grid, blk, 0, st, a, f);
    else hipLaunchKernelGGL((k_scan_seg<4, false>), grid, blk, 0, st, a, f);
}
// the scan with the per-cluster pass's pre-filter inside (1-byte LCPs, 16-byte aligned LCP and BWT)
void e2s_launch_scan_seg_fused(const SegArgs& a, const FilterArgs& f, hipStream_t st) {
    hipLaunchKernelGGL((k_scan_seg<1, true>), dim3(a.n_blocks), dim3(64), 0, st, a, f);
}
// the scan that writes the image itself: g.n_waves waves (a whole number of blocks) taking tickets
void e2s_launch_seg_fix(const SegArgs& a, u32* flags, u32* tailv, u64* seg_off, u64* kept_off, u32* err, const uint8_t* bwt, const void* pos,
                        const void* da, int xb, int yb, int zb, ScanMail* out, u64* stats_out, uint8_t* blob, int is_last_rank, int tail_bwt_arg, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_fix, dim3(1), dim3(1024), 0, st, a, flags, tailv, seg_off, kept_off, err, bwt, pos, da, xb, yb, zb, out, stats_out, blob, is_last_rank,
                       tail_bwt_arg);
}
void e2s_launch_seg_compact(const SegArgs& a, const u64* seg_off, u32 row_entries, u64* out_starts, uint16_t* out_lens, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_compact, dim3(a.n_blocks), dim3(256), 0, st, a, seg_off, row_entries, out_starts, out_lens);
}
u64 e2s_pack_clusters_blocks(u64 n) { return (n + PK_BLOCK - 1) / PK_BLOCK; }
// scratch: cnt[blocks] (u32) and prefix[blocks] (u64); out: 10 bytes per record; total: records written
void e2s_launch_pack_clusters(const u64* starts, const uint16_t* lens, u64 n, int min_len, u32* cnt, u64* prefix, uint8_t* out, u64* total,
                              hipStream_t st) {
    const u64 nb = e2s_pack_clusters_blocks(n);
    if (nb == 0) { (void)hipMemsetAsync(total, 0, 8, st); return; }
    hipLaunchKernelGGL(k_pack_count, dim3((unsigned)nb), dim3(256), 0, st, lens, n, min_len, cnt);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, cnt, prefix, nb);
    hipLaunchKernelGGL(k_pack_write, dim3((unsigned)nb), dim3(256), 0, st, starts, lens, n, min_len, prefix, cnt, (u32)nb, (uint16_t*)out, total);
}
void e2s_launch_tile_scan(const u32* cnt, u64* prefix, u64 n, hipStream_t st) {
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, cnt, prefix, n);
}
void e2s_launch_apply_plan(u64* starts, uint16_t* lens, u32 patch_first, u64 first_start, uint16_t first_len, u64 n_ends, u32 n_tail,
                           u64 t0, uint16_t l0, u64 t1, uint16_t l1, hipStream_t st) {
    hipLaunchKernelGGL(k_apply_plan, dim3(1), dim3(64), 0, st, starts, lens, patch_first, first_start, first_len, n_ends, n_tail, t0, l0, t1, l1);
}
void e2s_launch_hist_tail(const uint16_t* lens, u64 n, int min_len, u64* out_last_len, hipStream_t st) {
    hipLaunchKernelGGL(k_hist_tail, dim3(1), dim3(64), 0, st, lens, n, min_len, out_last_len);
}
void e2s_launch_hist(const uint16_t* lens, u64 n, int min_len, u64* out, hipStream_t st) {
    u64 blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_hist, dim3((unsigned)blocks), dim3(256), 0, st, lens, n, min_len, out);
}
// persistent grids: exactly the blocks that can be resident (a larger static-stride grid would run in two rounds)
static int resident_blocks(const void* fn, int threads, size_t dyn_lds) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 2;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return per_cu * cus;
}
void e2s_launch_cluster_filter2(FilterArgs a, unsigned long long tile_lo, unsigned long long tile_hi, int max_workers, hipStream_t st) {
    static int resident = 0;
    if (!resident) {
        // three workers per SIMD: the kernel is VALU bound (the same speed with 2..4), and this leaves registers for the
        // waves of k_flagged_lists, which runs beside the next chunk of this pass
        resident = 3 * 1024;
        const int fit = 4 * resident_blocks((const void*)k_cluster_filter2, 256, 0);
        if (fit < resident) resident = fit;
        if (const char* ev = e2s_env_dev("E2S_FILTER_WORKERS")) { const int cap = atoi(ev); if (cap > 0 && cap < fit) resident = cap; }
    }
    if (tile_hi <= tile_lo) return;
    a.w_tile0 = tile_lo; a.w_tiles = tile_hi - tile_lo;
    const uint64_t cap = (max_workers > 0 && max_workers < resident) ? (uint64_t)max_workers : (uint64_t)resident;
    // As in the scan (k_scan_seg): several times as many ranges as waves are resident, handed out by the dispatcher as slots come free -- the
    // SIMD's arbiter serves its oldest wave first, and with one static range per wave the youngest ran on alone for the last fifth of the
    // kernel.  Ranges of at least 16 tiles (a range's start costs about two), whole rounds, sixteen at most.
    static const int rounds_env = e2s_env_dev("E2S_FILTER_ROUNDS") ? atoi(e2s_env_dev("E2S_FILTER_ROUNDS")) : 16;
    uint64_t nb = a.w_tiles < cap ? a.w_tiles : cap;
    if (a.w_tiles > cap && max_workers <= 0) {
        const uint64_t rounds = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(rounds_env, 1), a.w_tiles / (cap * 16)));
        nb = cap * rounds;
    }
    a.w_per = (uint32_t)(a.w_tiles / nb); a.w_rem = (uint32_t)(a.w_tiles % nb); a.w_blocks = (uint32_t)nb;
    hipLaunchKernelGGL(k_cluster_filter2, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, st, a);
}
// LCP columns wider than a byte whose values all fit one (short reads: LCP < read length) are kept as bytes: the plane kernels
// and the streaming per-cluster pass then apply.  Max reduction, then the narrowing copy.
template <typename T>
__global__ __launch_bounds__(256) void k_lcp_max(const T* lcp, u64 n, u32* out) {
    u32 m = 0;
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u) m = max(m, (u32)lcp[i]);
    for (int o = 32; o; o >>= 1) m = max(m, (u32)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63u) == 0 && m) atomicMax(out, m);
}
template <typename T>
__global__ __launch_bounds__(256) void k_lcp_narrow(const T* lcp, u64 n, uint8_t* out) {
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u) out[i] = (uint8_t)lcp[i];
}
void e2s_launch_lcp_max(const void* lcp, int xb, unsigned long long n, uint32_t* out, hipStream_t st) {
    const unsigned g = (unsigned)std::min<u64>((n + 255) / 256, 8192);
    if (!g) return;
    if (xb == 2) hipLaunchKernelGGL((k_lcp_max<uint16_t>), dim3(g), dim3(256), 0, st, (const uint16_t*)lcp, (u64)n, out);
    else hipLaunchKernelGGL((k_lcp_max<u32>), dim3(g), dim3(256), 0, st, (const u32*)lcp, (u64)n, out);
}
void e2s_launch_lcp_narrow(const void* lcp, int xb, unsigned long long n, uint8_t* out, hipStream_t st) {
    const unsigned g = (unsigned)std::min<u64>((n + 255) / 256, 8192);
    if (!g) return;
    if (xb == 2) hipLaunchKernelGGL((k_lcp_narrow<uint16_t>), dim3(g), dim3(256), 0, st, (const uint16_t*)lcp, (u64)n, out);
    else hipLaunchKernelGGL((k_lcp_narrow<u32>), dim3(g), dim3(256), 0, st, (const u32*)lcp, (u64)n, out);
}
// Pre-filter over cluster RECORDS as they are (a `.clusters` file of unknown k, wide columns, ...): one thread per record
// counts the bases of its stretch of the BWT; the records that hold at least two different bases with >= mcov occurrences
// (necessary for clust2snp.cpp:419-429, see k_cluster_filter2) go to the flagged list for k_flagged_lists.  A record that
// reaches the EOF phantom or outside the resident shard is passed on as it is (k_flagged_lists models / reports it).
__global__ __launch_bounds__(256) void k_record_prefilter(const u64* starts, const uint16_t* lens, u64 n_rec, const uint8_t* bwt, u64 prov_lo, u64 prov_hi,
                                                          u32 mcov, int len_lo, int len_hi, u64* fl_starts, uint16_t* fl_lens, u64 fl_cap,
                                                          unsigned long long* fl_count) {
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n_rec; i += (u64)gridDim.x * 256u) {
        const u32 len = lens[i];
        if ((int)len < len_lo || (int)len > len_hi) continue;   // clust2snp.cpp:816
        const u64 s = starts[i];
        bool pass = s < prov_lo || s + len > prov_hi;           // not (entirely) real resident entries
        if (!pass) {
            const uint8_t* p = bwt + (s - prov_lo);
            u32 cC = 0, cG = 0, cT = 0;
            for (u32 j = 0; j < len; ++j) {
                const u32 c = (u32)p[j] & 0xDFu;                // fold case
                cC += c == 'C' ? 1u : 0u; cG += c == 'G' ? 1u : 0u; cT += c == 'T' ? 1u : 0u;
            }
            const u32 cA = len - cC - cG - cT;                  // everything else counts as A (include.hpp:265-279)
            pass = (cA >= mcov ? 1u : 0u) + (cC >= mcov ? 1u : 0u) + (cG >= mcov ? 1u : 0u) + (cT >= mcov ? 1u : 0u) >= 2u;
        }
        if (pass) {
            const u64 at = atomicAdd(fl_count, 1ull);
            if (at < fl_cap) { fl_starts[at] = s; fl_lens[at] = (uint16_t)len; }
        }
    }
}
void e2s_launch_record_prefilter(const unsigned long long* starts, const uint16_t* lens, unsigned long long n_rec, const uint8_t* bwt,
                                 unsigned long long prov_lo, unsigned long long prov_hi, uint32_t mcov, int len_lo, int len_hi,
                                 unsigned long long* fl_starts, uint16_t* fl_lens, unsigned long long fl_cap, unsigned long long* fl_count, hipStream_t st) {
    const unsigned g = (unsigned)std::min<u64>((n_rec + 255) / 256, 65536);
    if (!g) return;
    hipLaunchKernelGGL(k_record_prefilter, dim3(g), dim3(256), 0, st, (const u64*)starts, lens, (u64)n_rec, bwt, (u64)prov_lo, (u64)prov_hi, mcov, len_lo,
                       len_hi, (u64*)fl_starts, fl_lens, (u64)fl_cap, fl_count);
}
// one launch instead of a handful of tiny memsets / copies before a phase (each costs a few microseconds of stream time)
__global__ void k_phase_init(u64* count, u64 count_value, uint16_t* lens, u32 n_lens, u32* w0, u32 n0, u32* w1, u32 n1, u64* z, u32 nz) {
    const u32 t = threadIdx.x;
    if (t == 0 && count) *count = count_value;
    if (lens && t < n_lens) lens[t] = 0;
    if (w0 && t < n0) w0[t] = 0;
    if (w1 && t < n1) w1[t] = 0;
    if (z) for (u32 i = t; i < nz; i += 256u) z[i] = 0ull;
}
// the per-cluster pass's counters cleared and the seed clusters (read from the pinned mailbox) placed into the reserved head
// of the flagged list, in one launch
__global__ void k_cand_init(u64* cand_count, u64* stream8_count, u32* err, const u64* seed_starts, const uint16_t* seed_lens, u32 n_seed_slots,
                            u64* fl_starts, uint16_t* fl_lens, u64* fl_count, u32 set_fl_count, u32* ko_cnt, u32 n_ko) {
    const u32 t = threadIdx.x;
    if (t == 0) { *cand_count = 0; if (stream8_count) *stream8_count = 0; *err = 0; if (set_fl_count) *fl_count = n_seed_slots; }
    // (seed_starts == nullptr: no seeds -- the usual case; reading the empty slots from host memory would cost a PCIe round trip)
    if (fl_starts && t < n_seed_slots) { fl_starts[t] = seed_starts ? seed_starts[t] : 0ull; fl_lens[t] = seed_starts ? seed_lens[t] : (uint16_t)0; }
    if (ko_cnt) for (u32 i = t; i < n_ko; i += 256u) ko_cnt[i] = 0u;     // the ordering pass's bucket counts
}
void e2s_launch_cand_init(unsigned long long* cand_count, unsigned long long* stream8_count, uint32_t* err, const unsigned long long* seed_starts,
                          const uint16_t* seed_lens, uint32_t n_seed_slots, unsigned long long* fl_starts, uint16_t* fl_lens,
                          unsigned long long* fl_count, bool set_fl_count, uint32_t* ko_cnt, uint32_t n_ko, hipStream_t st) {
    hipLaunchKernelGGL(k_cand_init, dim3(1), dim3(256), 0, st, (u64*)cand_count, (u64*)stream8_count, err, (const u64*)seed_starts, seed_lens,
                       n_seed_slots, (u64*)fl_starts, fl_lens, (u64*)fl_count, set_fl_count ? 1u : 0u, ko_cnt, n_ko);
}
void e2s_launch_phase_init(unsigned long long* count, unsigned long long count_value, uint16_t* lens, uint32_t n_lens, uint32_t* w0, uint32_t n0,
                           uint32_t* w1, uint32_t n1, unsigned long long* z, uint32_t nz, hipStream_t st) {
    hipLaunchKernelGGL(k_phase_init, dim3(1), dim3(256), 0, st, (u64*)count, (u64)count_value, lens, n_lens, w0, n0, w1, n1, (u64*)z, nz);
}
__global__ void k_snapshot(const unsigned long long* src, unsigned long long* dst) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = *src; }
void e2s_launch_snapshot(const unsigned long long* src, unsigned long long* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_snapshot, dim3(1), dim3(64), 0, st, src, dst);
}
void e2s_launch_flagged_lists(const CandArgs& a, const unsigned long long* lo_dev, const unsigned long long* n_dev, bool def_widths, hipStream_t st) {
    // room for two rounds' worst case and, at usual list lengths, ~200 packed candidates between two global reservations
    const uint32_t max8 = (16u + (4u + a.pos_pack) * 2u * a.list_cap + 7u) >> 3;
    const uint32_t cap8 = std::max<uint32_t>(3072u, 2u * 16u * max8);
    const size_t dyn = (size_t)cap8 * 8;
    // c2 (2x10^4 flagged clusters): 512 / 1024 / 2048 / 4096 blocks -> 0.124 / 0.117 / 0.131 / 0.177 ms for the phase; c3 (3x10^5): 1024 / 4096 ->
    // 0.889 / 0.845 ms: the caller asks for more blocks when it expects many clusters
    const unsigned blocks = a.blocks_hint ? a.blocks_hint : 1024u;
    if (dyn > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)k_flagged_lists<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        (void)hipFuncSetAttribute((const void*)k_flagged_lists<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    }
    if (def_widths) hipLaunchKernelGGL((k_flagged_lists<true>), dim3(blocks), dim3(256), dyn, st, a, lo_dev, n_dev, cap8);
    else hipLaunchKernelGGL((k_flagged_lists<false>), dim3(blocks), dim3(256), dyn, st, a, lo_dev, n_dev, cap8);
}
void e2s_launch_unpack_pairsa(const uint8_t* src, u64 n, int zf, int yf, void* pos, int zb, void* da, int yb, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_unpack_pairsa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, n, zf, yf, pos, zb, da, yb);
}
void e2s_launch_narrow8(const uint8_t* src, u64 n, u32* dst, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_narrow8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, n, dst);
}
void e2s_launch_unpack_gesa(const uint8_t* src, u64 n, int yf, int zf, int xf, uint8_t* bwt, void* lcp, int xb, void* pos, int zb, void* da, int yb, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_unpack_gesa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, n, yf, zf, xf, bwt, lcp, xb, pos, zb, da, yb);
}
