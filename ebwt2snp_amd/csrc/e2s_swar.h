// e2s_swar.h -- the SWAR stencil of the cluster-boundary scan (cluster_lm, ebwt2clust.cpp:68-139).
// Pure 32-bit integer code shared by the gfx950 kernel (e2s_kernels.hip) and a host-compiled
// check (tests/swar_check.cpp) that compares it with the sequential definition.
#pragma once
#include <stdint.h>

typedef unsigned int u32;
typedef unsigned long long u64;

#if defined(__HIPCC__)
#define E2S_HD __host__ __device__ __forceinline__
#else
#define E2S_HD inline
#endif
// ({hi,lo} >> 8*sh) & 0xffffffff, sh in 0..3 -- v_alignbyte_b32 on the GPU
E2S_HD u32 alignbyte(u32 hi, u32 lo, u32 sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> (8 * (sh & 3)));
#endif
}
// (m & a) | (~m & b)
E2S_HD u32 bfi(u32 m, u32 a, u32 b) { return (m & a) | (~m & b); }

// ----------------------------------------------------------------------------------------------
// K1: boundary scan.  One thread-row = 16 bytes of LCP = EPT entries; a tile = ROWS x 256 thread-rows,
// row-major (thread t of row r holds bytes [r*4096 + t*16, +16) of the tile), so every wave
// instruction loads 1 KiB contiguous.  Flags are computed SWAR in "lane-MSB form": one flag per
// X-byte lane of a 32-bit word, held in that lane's top bit.
// ----------------------------------------------------------------------------------------------
template <int X> struct Swar;
template <> struct Swar<1> { static constexpr u32 H = 0x80808080u; static constexpr int EPW = 4; static constexpr int LB = 3; };
template <> struct Swar<2> { static constexpr u32 H = 0x80008000u; static constexpr int EPW = 2; static constexpr int LB = 4; };
template <> struct Swar<4> { static constexpr u32 H = 0x80000000u; static constexpr int EPW = 1; static constexpr int LB = 5; };

// previous-entry / next-entry views of a word stream
template <int X> E2S_HD u32 prev_elems(u32 cur, u32 before) {
    if (X == 4) return before;
    return alignbyte(cur, before, 4 - X);            // ({cur,before} >> 8*(4-X))
}
template <int X> E2S_HD u32 next_elems(u32 after, u32 cur) {
    if (X == 4) return after;
    return alignbyte(after, cur, X);                 // ({after,cur} >> 8*X)
}

// W[0..PRE-1]: words before the span, W[PRE..PRE+NR-1]: the span (NR words), W[PRE+NR]: the word after.
// Returns END/START flags (lane-MSB form) for the span, before range masking.
template <int X, int NR>
E2S_HD void span_flags(const u32* W, u32 kb, u32 hge, u32* end, u32* start) {
    constexpr u32 H = Swar<X>::H;
    constexpr int PRE = (X == 4) ? 2 : 1;
    constexpr int NW = PRE + NR + 1;
    u32 ge[NW], nle[NW];                       // ge: L>=k ; nle: (L[j] >= L[j-1]) i.e. NOT gt(j)
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u32 x = W[w];
        const u32 xh = x | H;
        // lane-wise x >= k: where the top bits of x and k differ x's top bit decides, else the
        // borrow-free subtract of the low bits does
        const u32 t = xh - (kb & ~H);
        ge[w] = bfi(x ^ kb, x, t) & hge;
        const u32 y = (w == 0) ? prev_elems<X>(x, 0u) : prev_elems<X>(x, W[w - 1]);
        const u32 t2 = xh - (y & ~H);
        nle[w] = bfi(x ^ y, x, t2);           // top bit: x >= y  (garbage elsewhere, masked at the end)
    }
    // end(j)   = ge(j) & ((gt(j) & !gt(j+1)) | !ge(j+1))          ebwt2clust.cpp:98-102
    // start(j) = ge(j) & (!ge(j-1) | end(j-1))                    ebwt2clust.cpp:108,116-120
    u32 endw[NR + 1];
#pragma unroll
    for (int i = 0; i < NR + 1; ++i) {
        const int w = PRE - 1 + i;             // the word before the span, then the span
        const u32 nle_n = next_elems<X>(nle[w + 1], nle[w]);
        const u32 ge_n = next_elems<X>(ge[w + 1], ge[w]);
        const u32 locmin = ~nle[w] & nle_n;    // gt(j) & !gt(j+1)
        const u32 t = ~locmin & ge_n;
        endw[i] = ~t & ge[w] & H;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int w = PRE + i;
        const u32 ge_p = prev_elems<X>(ge[w], ge[w - 1]);
        const u32 end_p = prev_elems<X>(endw[i + 1], endw[i]);
        end[i] = endw[i + 1];
        start[i] = ge[w] & ~(ge_p & ~end_p) & H;
    }
}

struct RowFlags { u32 end[4]; u32 start[4]; };
template <int X> E2S_HD void row_flags(const u32* W, u32 kb, u32 hge, RowFlags& f) { span_flags<X, 4>(W, kb, hge, f.end, f.start); }

// ---- compact event stream of a span: 2 bits per entry (bit 2e = a cluster starts at entry e, bit 2e+1 = one ends
// there), 16 entries per 32-bit word.  Same stencil as span_flags, but streamed word by word so that only the
// running state (~10 registers) is live, and packed on the fly.
template <int X> E2S_HD u32 word_events(u32 endm, u32 startm) {
    const u32 c = endm | (startm >> 1);                     // per lane: top bit = end, next bit = start
    if (X == 1) {                                           // 4 entries -> 8 bits
        const u32 f = (c >> 6) & 0x03030303u;
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_udot4(f, 0x40100401u, 0u, false);          // v_dot4_u32_u8: sum of field_i * 4^i
#else
        return (f & 3u) | (((f >> 8) & 3u) << 2) | (((f >> 16) & 3u) << 4) | (((f >> 24) & 3u) << 6);
#endif
    }
    if (X == 2) return ((c >> 14) & 3u) | ((c >> 28) & 0xCu);                // 2 entries -> 4 bits
    return c >> 30;                                                          // 1 entry  -> 2 bits
}

template <int X, int NR>
E2S_HD void span_events(const u32* W, u32 kb, u32 hge, u32* EV) {
    constexpr u32 H = Swar<X>::H;
    constexpr int PRE = (X == 4) ? 2 : 1;
    constexpr int NW = PRE + NR + 1;
    constexpr int FB = 2 * Swar<X>::EPW;                    // event bits per span word
    constexpr int NEV = (NR * FB + 31) / 32;
#pragma unroll
    for (int g = 0; g < NEV; ++g) EV[g] = 0;
    u32 ge1 = 0, ge2 = 0, nle1 = 0, end2 = 0;               // ge/nle of word w-1, ge/end of word w-2
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u32 x = W[w];
        const u32 xh = x | H;
        const u32 ge0 = bfi(x ^ kb, x, xh - (kb & ~H)) & hge;                              // L >= k
        const u32 y = (w == 0) ? prev_elems<X>(x, 0u) : prev_elems<X>(x, W[w - 1]);
        const u32 nle0 = bfi(x ^ y, x, xh - (y & ~H));                                     // top bit: L[j] >= L[j-1]
        if (w >= 1) {
            // end(j) = ge(j) & ((gt(j) & !gt(j+1)) | !ge(j+1)) for the entries of word w-1       (ebwt2clust.cpp:98-102)
            const u32 locmin = ~nle1 & next_elems<X>(nle0, nle1);
            const u32 end1 = ~(~locmin & next_elems<X>(ge0, ge1)) & ge1 & H;
            if (w - 1 >= PRE) {
                // start(j) = ge(j) & (!ge(j-1) | end(j-1))                                       (:108,116-120)
                const u32 start1 = ge1 & ~(prev_elems<X>(ge1, ge2) & ~prev_elems<X>(end1, end2)) & H;
                const int i = w - 1 - PRE;
                EV[(i * FB) / 32] |= word_events<X>(end1, start1) << ((i * FB) % 32);
            }
            end2 = end1;
        }
        ge2 = ge1; ge1 = ge0; nle1 = nle0;
    }
}

// ---- start / end planes of a span: bit e of S = a cluster starts at entry e, bit e of E = one ends there
// (4*NR/X entries, at most 64).  Same streamed stencil; the clean lane-MSB masks are packed with v_dot4_u32_u8:
// bytes are 0x80 or 0, so dot4 with weights 1,2,4,8 | 16,32,64,128 over two words gives 128 * (8 mask bits).
E2S_HD u32 e2s_udot4(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    return c + (a & 0xFF) * (b & 0xFF) + ((a >> 8) & 0xFF) * ((b >> 8) & 0xFF) + ((a >> 16) & 0xFF) * ((b >> 16) & 0xFF) + (a >> 24) * (b >> 24);
#endif
}

template <int X, int NR>
E2S_HD void span_planes(const u32* W, u32 kb, u32 hge, u64& S, u64& E) {
    constexpr u32 H = Swar<X>::H;
    constexpr int PRE = (X == 4) ? 2 : 1;
    constexpr int NW = PRE + NR + 1;
    constexpr int EPW = Swar<X>::EPW;
    static_assert(NR * EPW <= 64 && NR % 2 == 0, "span too long for one 64-bit plane");
    u32 sp[NR / 2], ep[NR / 2];                              // X = 1: 128 * (8 bits) per word pair; else bits of a word pair
    u32 ge1 = 0, ge2 = 0, nle1 = 0, end2 = 0;               // ge/nle of word w-1, ge/end of word w-2
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u32 x = W[w];
        const u32 xh = x | H;
        const u32 ge0 = bfi(x ^ kb, x, xh - (kb & ~H)) & hge;                              // L >= k
        const u32 y = (w == 0) ? prev_elems<X>(x, 0u) : prev_elems<X>(x, W[w - 1]);
        const u32 nle0 = bfi(x ^ y, x, xh - (y & ~H));                                     // top bit: L[j] >= L[j-1]
        if (w >= 1) {
            const u32 locmin = ~nle1 & next_elems<X>(nle0, nle1);                          // ebwt2clust.cpp:98-102
            const u32 end1 = ~(~locmin & next_elems<X>(ge0, ge1)) & ge1 & H;
            if (w - 1 >= PRE) {
                const u32 start1 = ge1 & ~(prev_elems<X>(ge1, ge2) & ~prev_elems<X>(end1, end2)) & H;   // :108,116-120
                const int i = w - 1 - PRE;
                if (X == 1) {
                    if (i % 2 == 0) { sp[i / 2] = e2s_udot4(start1, 0x08040201u, 0u); ep[i / 2] = e2s_udot4(end1, 0x08040201u, 0u); }
                    else { sp[i / 2] = e2s_udot4(start1, 0x80402010u, sp[i / 2]); ep[i / 2] = e2s_udot4(end1, 0x80402010u, ep[i / 2]); }
                } else if (X == 2) {
                    const u32 sb = ((start1 >> 15) & 1u) | ((start1 >> 30) & 2u), eb = ((end1 >> 15) & 1u) | ((end1 >> 30) & 2u);
                    if (i % 2 == 0) { sp[i / 2] = sb; ep[i / 2] = eb; } else { sp[i / 2] |= sb << 2; ep[i / 2] |= eb << 2; }
                } else {
                    if (i % 2 == 0) { sp[i / 2] = start1 >> 31; ep[i / 2] = end1 >> 31; } else { sp[i / 2] |= (start1 >> 31) << 1; ep[i / 2] |= (end1 >> 31) << 1; }
                }
            }
            end2 = end1;
        }
        ge2 = ge1; ge1 = ge0; nle1 = nle0;
    }
    // assemble: word pair j holds 2*EPW entries
    constexpr int PB = 2 * EPW;                              // bits per pair
    u64 s = 0, e = 0;
#pragma unroll
    for (int j = 0; j < NR / 2; ++j) {
        const u64 sv = (X == 1) ? (u64)(sp[j] >> 7) : (u64)sp[j];
        const u64 ev = (X == 1) ? (u64)(ep[j] >> 7) : (u64)ep[j];
        s |= sv << (j * PB); e |= ev << (j * PB);
    }
    S = s; E = e;
}

// ---- second formulation (K1 from round 1b on): per word only the two PREDICATES are evaluated SWAR and packed
// into bit planes -- GE: L[e] >= k, NLE: L[e] >= L[e-1] -- and the stencil itself runs on the 64-bit planes
// (a handful of instructions per span instead of ~10 per word):
//     end(e)   = GE(e) & ((~NLE(e) & NLE(e+1)) | ~GE(e+1))                       ebwt2clust.cpp:98-102
//     start(e) = GE(e) & ~(GE(e-1) & ~end(e-1))                                  ebwt2clust.cpp:108,116-120
// W2 = [word before the span][NR span words][word after the span].  `after`: bit 0 = GE, bit 1 = NLE of the first
// entry after the span.  `before` (plane_start): bit 0 = GE(-1), bit 1 = end(-1), i.e. the top GE / E bits of
// the span before this one (the previous lane on the GPU).
// per byte (a + b + (c & 1)) >> 1 -- v_lerp_u8.  With b = ~y and c = 1 the top bit of each byte is a >= y:
// (a - y + 256) >> 1 >= 128  <=>  a >= y.  One instruction per unsigned byte compare of four entries.
E2S_HD u32 e2s_lerp_u8(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_lerp(a, b, c);
#else
    u32 r = 0;
    for (int i = 0; i < 4; ++i) {
        const u32 v = (((a >> (8 * i)) & 0xFFu) + ((b >> (8 * i)) & 0xFFu) + ((c >> (8 * i)) & 1u)) >> 1;
        r |= (v & 0xFFu) << (8 * i);
    }
    return r;
#endif
}

template <int X, int NR>
E2S_HD void span_pred(const u32* W2, u32 kb, u32 hge, u64& GE, u64& NLE, u32& after) {
    constexpr u32 H = Swar<X>::H;
    constexpr int EPW = Swar<X>::EPW;
    static_assert(NR * EPW <= 64 && NR % 2 == 0, "span too long for one 64-bit plane");
    const u32 kl = kb & ~H;
    u32 gs[NR / 2], ns[NR / 2];                               // X = 1: 128 * (8 plane bits) per word pair
    u64 g64 = 0, n64 = 0;
#pragma unroll
    for (int i = 0; i <= NR; ++i) {                           // i = NR: the word after the span (first entry only)
        const u32 x = W2[1 + i];
        u32 ge, nle;                                          // clean lane-MSB masks
        if (X == 1) {
            const u32 ny = alignbyte(~x, ~W2[i], 3);          // ~(previous entries)
            ge = e2s_lerp_u8(x, ~kb, 0x01010101u) & hge;
            nle = e2s_lerp_u8(x, ny, 0x01010101u) & H;
        } else {
            const u32 xh = x | H;
            const u32 y = prev_elems<X>(x, W2[i]);
            ge = bfi(x ^ kb, x, xh - kl) & hge;
            nle = bfi(x ^ y, x, xh - (y & ~H)) & H;
        }
        if (i == NR) { after = ((ge >> (8 * X - 1)) & 1u) | (((nle >> (8 * X - 1)) & 1u) << 1); break; }
        if (X == 1) {
            if (i % 2 == 0) { gs[i / 2] = e2s_udot4(ge, 0x08040201u, 0u); ns[i / 2] = e2s_udot4(nle, 0x08040201u, 0u); }
            else { gs[i / 2] = e2s_udot4(ge, 0x80402010u, gs[i / 2]); ns[i / 2] = e2s_udot4(nle, 0x80402010u, ns[i / 2]); }
        } else if (X == 2) {
            g64 |= (u64)(((ge >> 15) & 1u) | (ge >> 30)) << (2 * i);
            n64 |= (u64)(((nle >> 15) & 1u) | (nle >> 30)) << (2 * i);
        } else {
            g64 |= (u64)(ge >> 31) << i;
            n64 |= (u64)(nle >> 31) << i;
        }
    }
    if (X == 1) {
        // s = 128 * v (v = 8 plane bits): v << 8j = s << (8j - 7)
        u32 gw[2] = {0, 0}, nw[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < NR / 2; ++j) {
            const int h = j / 4, q = j % 4;
            if (q == 0) { gw[h] = gs[j] >> 7; nw[h] = ns[j] >> 7; }
            else { gw[h] |= gs[j] << (8 * q - 7); nw[h] |= ns[j] << (8 * q - 7); }
        }
        g64 = (u64)gw[0] | ((u64)gw[1] << 32);
        n64 = (u64)nw[0] | ((u64)nw[1] << 32);
    }
    GE = g64; NLE = n64;
}
template <int ES> E2S_HD u64 plane_end(u64 GE, u64 NLE, u32 after) {
    const u64 gn = (GE >> 1) | ((u64)(after & 1u) << (ES - 1));
    const u64 nn = (NLE >> 1) | ((u64)((after >> 1) & 1u) << (ES - 1));
    return GE & ((~NLE & nn) | ~gn);
}
template <int ES> E2S_HD u64 plane_start(u64 GE, u64 E, u32 before) {
    const u64 gp = (GE << 1) | (u64)(before & 1u);
    const u64 ep = (E << 1) | (u64)((before >> 1) & 1u);
    const u64 s = GE & ~(gp & ~ep);
    return ES == 64 ? s : (s & ((1ull << (ES & 63)) - 1ull));
}

// ---- K2a planes (1-byte entries): which entries of a 64-entry span hold C / G / T (case folded, include.hpp:265-279;
// every other byte counts as A), and which have LCP >= k_right.  byte == K  <=>  byte >= K and not byte >= K+1:
// two v_lerp_u8 and one bit-op per word and class.
E2S_HD u32 e2s_pack8(u32 lo4, u32 hi4) { return e2s_udot4(hi4, 0x80402010u, e2s_udot4(lo4, 0x08040201u, 0u)); }   // 128 * (8 bits)
template <int NR>
E2S_HD void span_assemble(const u32* s, u64& out) {          // s[j] = 128 * (plane bits 8j .. 8j+7)
    u32 w[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < NR / 2; ++j) {
        const int h = j / 4, q = j % 4;
        if (q == 0) w[h] = s[j] >> 7; else w[h] |= s[j] << (8 * q - 7);
    }
    out = (u64)w[0] | ((u64)w[1] << 32);
}
template <int NR>
E2S_HD void span_bwt_classes(const u32* B, u64& C, u64& G, u64& T) {
    constexpr u32 H = 0x80808080u, ONE = 0x01010101u;
    static_assert(NR == 16, "one 64-bit plane per class");
    u32 sc[NR / 2], sg[NR / 2], st[NR / 2];
#pragma unroll
    for (int j = 0; j < NR / 2; ++j) {
        u32 mc[2], mg[2], mt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32 x = B[2 * j + i] & 0xDFDFDFDFu;         // fold case
            mc[i] = e2s_lerp_u8(x, ~0x43434343u, ONE) & ~e2s_lerp_u8(x, ~0x44444444u, ONE) & H;
            mg[i] = e2s_lerp_u8(x, ~0x47474747u, ONE) & ~e2s_lerp_u8(x, ~0x48484848u, ONE) & H;
            mt[i] = e2s_lerp_u8(x, ~0x54545454u, ONE) & ~e2s_lerp_u8(x, ~0x55555555u, ONE) & H;
        }
        sc[j] = e2s_pack8(mc[0], mc[1]); sg[j] = e2s_pack8(mg[0], mg[1]); st[j] = e2s_pack8(mt[0], mt[1]);
    }
    span_assemble<NR>(sc, C); span_assemble<NR>(sg, G); span_assemble<NR>(st, T);
}
// ---- K2a class planes, second formulation: TWO planes that encode the class of every entry -- P1 = C | G, P2 = G | T
// (C = P1 & ~P2, G = P1 & P2, T = ~P1 & P2, everything else counts as A: include.hpp:250-279) -- from the two bits in which
// the letters differ, (byte >> 1) & 3 = A 0, C 1, G 3, T 2, gated by an EXACT test "the case-folded byte is C, G or T":
// the low three bits of the byte select the one letter it could be from an 8-entry table (v_perm_b32), one XOR-compare
// and one v_lerp_u8 zero test decide.  Per word 4 + 4 + 2 instructions instead of 16, two plane assemblies instead of three.
// The constants travel in a struct so that the kernel can hold them in VGPRs (v_bitop3_b32 issues twice as fast with
// register operands as with an SGPR or a literal: tools/valu_rate.hip).
struct ClassConsts { u32 m07, ehi, elo, c20, one, h; };
E2S_HD ClassConsts class_consts() { return ClassConsts{0x07070707u, 0x47FFFF54u, 0x43FFFFFFu, 0x20202020u, 0x01010101u, 0x80808080u}; }
// byte i of the result = byte sel.byte[i] (0..7) of the eight bytes {hi, lo} -- v_perm_b32 on the GPU
E2S_HD u32 e2s_perm8(u32 hi, u32 lo, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const u64 t = ((u64)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; ++i) r |= (u32)((t >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
    return r;
#endif
}
template <int NR>
E2S_HD void span_bwt_code_planes(const u32* B, const ClassConsts& k, u64& P1, u64& P2) {
    static_assert(NR == 16, "one 64-bit plane per code bit");
    u32 s1[NR / 2], s2[NR / 2];
#pragma unroll
    for (int j = 0; j < NR / 2; ++j) {
        u32 m1[2], m2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32 x = B[2 * j + i];
            const u32 e = e2s_perm8(k.ehi, k.elo, x & k.m07);      // the letter with these low three bits (0xFF: none)
            const u32 nd = ~(x ^ e) | k.c20;                       // 0xFF  <=>  the byte is that letter, in either case
            const u32 v = e2s_lerp_u8(nd, 0u, k.one);              // top bit: nd == 0xFF
            m1[i] = (x << 6) & v & k.h;                            // bit 1 of the byte: C, G
            m2[i] = (x << 5) & v & k.h;                            // bit 2: G, T
        }
        s1[j] = e2s_pack8(m1[0], m1[1]); s2[j] = e2s_pack8(m2[0], m2[1]);
    }
    span_assemble<NR>(s1, P1); span_assemble<NR>(s2, P2);
}
template <int NR>
E2S_HD u64 span_ge_plane(const u32* W /* the NR span words */, u32 kb, u32 hge) {
    static_assert(NR == 16, "one 64-bit plane");
    u32 sk[NR / 2];
#pragma unroll
    for (int j = 0; j < NR / 2; ++j)
        sk[j] = e2s_pack8(e2s_lerp_u8(W[2 * j], ~kb, 0x01010101u) & hge, e2s_lerp_u8(W[2 * j + 1], ~kb, 0x01010101u) & hge);
    u64 out; span_assemble<NR>(sk, out);
    return out;
}
