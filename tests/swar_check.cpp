// tests/swar_check.cpp -- host-compiled check of the SWAR stencil the gfx950 scan kernel uses
// (ebwt2snp_amd/csrc/e2s_swar.h) against the per-entry definition of cluster_lm's tests
// (ebwt2clust.cpp:98-102,116).  Built and run by tests/test_swar_host.py; no GPU needed.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../ebwt2snp_amd/csrc/e2s_swar.h"

template <int X, int NR> static int run(unsigned seed, int iters) {
    constexpr int PRE = (X == 4) ? 2 : 1;
    constexpr int EPT = 4 * NR / X;
    constexpr int EPW = 4 / X;
    constexpr u32 H = Swar<X>::H;
    const u64 lane_max = X == 1 ? 0xFFull : X == 2 ? 0xFFFFull : 0xFFFFFFFFull;
    srand(seed);
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        // entries -PRE*EPW .. EPT+EPW-1 around one thread-row
        const int lo = -PRE * EPW, hi = EPT + EPW;
        std::vector<u64> L(hi - lo);
        const u64 kchoices[] = {16, 1, 0, 127, 128, 129, 200, 255, 256, 40000, 65535, 65536, 0x80000000ull, 0xFFFFFFFFull};
        u64 k = kchoices[rand() % 14];
        const int mode = rand() % 4;
        for (auto& v : L) {
            u64 r;
            if (mode == 0) r = (u64)(rand() % 4) + (k > 2 ? k - 2 : 0);
            else if (mode == 1) r = ((u64)rand() << 16 ^ (u64)rand());
            else if (mode == 2) r = (u64)(rand() % 300);
            else r = (rand() & 1) ? lane_max - (u64)(rand() % 3) : (u64)(rand() % 3);
            v = r & lane_max;
        }
        u32 W[PRE + NR + 1];
        memset(W, 0, sizeof(W));
        for (int e = lo; e < hi; ++e) {
            const int idx = e - lo;                       // entry idx in stream
            const int w = idx / EPW, l = idx % EPW;
            W[w] |= (u32)(L[idx] << (8 * X * l) & 0xFFFFFFFFull) ;
        }
        const u32 k32 = (u32)k;
        const u32 hge = (k > lane_max) ? 0u : H;
        const u32 kb = X == 1 ? (k32 & 0xFF) * 0x01010101u : X == 2 ? (k32 & 0xFFFF) * 0x00010001u : k32;
        struct { u32 end[NR]; u32 start[NR]; } f;
        span_flags<X, NR>(W, kb, hge, f.end, f.start);
        auto Lx = [&](int e) -> u64 { return L[e - lo]; };
        auto ge = [&](int e) { return hge != 0 && Lx(e) >= k; };
        auto endf = [&](int e) { return ge(e) && ((Lx(e - 1) > Lx(e) && Lx(e) <= Lx(e + 1)) || !ge(e + 1)); };
        for (int e = 0; e < EPT; ++e) {
            const bool want_end = endf(e);
            const bool want_start = ge(e) && (!ge(e - 1) || endf(e - 1));
            const int w = e / EPW, l = e % EPW;
            const u32 bit = 1u << (8 * X * l + 8 * X - 1);
            const bool got_end = (f.end[w] & bit) != 0, got_start = (f.start[w] & bit) != 0;
            if (want_end != got_end || want_start != got_start) {
                if (bad < 5) fprintf(stderr, "X=%d it=%d e=%d k=%llu end %d/%d start %d/%d\n", X, it, e, (unsigned long long)k, want_end, got_end, want_start, got_start);
                ++bad;
            }
        }
        // the streamed, packed form must say the same (bit 2e = start, bit 2e+1 = end, 16 entries per word)
        {
            u32 EV[(NR * 2 * EPW + 31) / 32];
            span_events<X, NR>(W, kb, hge, EV);
            for (int e = 0; e < EPT; ++e) {
                const bool gs = (EV[e / 16] >> (2 * (e % 16))) & 1u, gen = (EV[e / 16] >> (2 * (e % 16) + 1)) & 1u;
                const bool want_end = endf(e);
                const bool want_start = ge(e) && (!ge(e - 1) || endf(e - 1));
                if (gs != want_start || gen != want_end) { if (bad < 5) fprintf(stderr, "span_events X=%d e=%d\n", X, e); ++bad; }
            }
        }
        // and the start / end planes
        {
            u64 S, E;
            span_planes<X, NR>(W, kb, hge, S, E);
            for (int e = 0; e < EPT; ++e) {
                const bool want_end = endf(e);
                const bool want_start = ge(e) && (!ge(e - 1) || endf(e - 1));
                if ((bool)((S >> e) & 1) != want_start || (bool)((E >> e) & 1) != want_end) { if (bad < 5) fprintf(stderr, "span_planes X=%d e=%d\n", X, e); ++bad; }
            }
            if (EPT < 64 && ((S | E) >> (EPT & 63))) { if (bad < 5) fprintf(stderr, "span_planes stray X=%d\n", X); ++bad; }
        }
        // second formulation: predicate planes + plane-level stencil
        {
            u32 W2[NR + 2];
            for (int w = 0; w < NR + 2; ++w) W2[w] = W[PRE - 1 + w];
            u64 GE, NLE, S, E; u32 after;
            span_pred<X, NR>(W2, kb, hge, GE, NLE, after);
            E = plane_end<EPT>(GE, NLE, after);
            const u32 before = (ge(-1) ? 1u : 0u) | (endf(-1) ? 2u : 0u);
            S = plane_start<EPT>(GE, E, before);
            for (int e = 0; e < EPT; ++e) {
                const bool want_end = endf(e);
                const bool want_start = ge(e) && (!ge(e - 1) || endf(e - 1));
                if ((bool)((GE >> e) & 1) != ge(e) || (bool)((NLE >> e) & 1) != (Lx(e) >= Lx(e - 1)) ||
                    (bool)((S >> e) & 1) != want_start || (bool)((E >> e) & 1) != want_end) { if (bad < 5) fprintf(stderr, "span_pred X=%d NR=%d e=%d k=%llu\n", X, NR, e, (unsigned long long)k); ++bad; }
            }
            if (after != ((ge(EPT) ? 1u : 0u) | (Lx(EPT) >= Lx(EPT - 1) ? 2u : 0u))) { if (bad < 5) fprintf(stderr, "span_pred after X=%d\n", X); ++bad; }
            if (EPT < 64 && ((S | E | GE | NLE) >> (EPT & 63))) { if (bad < 5) fprintf(stderr, "span_pred stray X=%d\n", X); ++bad; }
        }
        // no stray bits outside the lane-MSB positions
        for (int w = 0; w < NR; ++w) if ((f.end[w] | f.start[w]) & ~H) { if (bad < 5) fprintf(stderr, "stray bits X=%d\n", X); ++bad; }
    }
    return bad;
}

// K2a planes: classes of BWT bytes and LCP >= k_right over a 64-entry span
static int run_classes(unsigned seed, int iters) {
    srand(seed);
    int bad = 0;
    const unsigned char pool[] = {'A', 'C', 'G', 'T', 'a', 'c', 'g', 't', 'N', '$', 'n', 0x42, 0x44, 0x46, 0x48, 0x53, 0x55, 0x63 ^ 0x80, 0xC3, 0xFF, 0x00, 0x23, 0x67 | 0x80};
    for (int it = 0; it < iters; ++it) {
        unsigned char b[64], l[64];
        for (int i = 0; i < 64; ++i) { b[i] = it < 4 ? (unsigned char)(64 * it + i) : (rand() % 3) ? pool[rand() % sizeof(pool)] : (unsigned char)rand(); l[i] = (unsigned char)rand(); }   // (the first four rounds: every byte value)
        u32 B[16], W[16];
        memcpy(B, b, 64); memcpy(W, l, 64);
        u64 C, G, T, P1, P2;
        span_bwt_classes<16>(B, C, G, T);
        span_bwt_code_planes<16>(B, class_consts(), P1, P2);
        if ((P1 & ~P2) != C || (P1 & P2) != G || (~P1 & P2) != T) { if (bad < 5) fprintf(stderr, "code planes differ from the class planes\n"); ++bad; }
        const u64 kr = (u64)(rand() % 300);
        const u32 hge = kr > 0xFF ? 0u : 0x80808080u, kb = (u32)(kr & 0xFF) * 0x01010101u;
        const u64 K = span_ge_plane<16>(W, kb, hge);
        for (int i = 0; i < 64; ++i) {
            const unsigned char f = b[i] & 0xDF;
            const bool wc = f == 'C', wg = f == 'G', wt = f == 'T', wk = hge != 0 && (u64)l[i] >= kr;
            if ((bool)((C >> i) & 1) != wc || (bool)((G >> i) & 1) != wg || (bool)((T >> i) & 1) != wt || (bool)((K >> i) & 1) != wk) {
                if (bad < 5) fprintf(stderr, "classes i=%d byte=%02x\n", i, b[i]);
                ++bad;
            }
        }
    }
    return bad;
}

int main() {
    int bad = run<1, 4>(1, 200000) + run<2, 4>(2, 200000) + run<4, 4>(3, 200000) +
              run<1, 16>(4, 60000) + run<2, 16>(5, 60000) + run<4, 16>(6, 60000) + run_classes(7, 40000);
    printf("swar_check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad ? 1 : 0;
}
