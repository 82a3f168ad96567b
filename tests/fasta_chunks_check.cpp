// host check (no GPU): the chunked FASTA index of the streamed upload (e2s_fasta_scan_chunk + e2s_fasta_join_chunks) against the index over
// the whole file (e2s_index_fasta) -- random files of the plain shape cut into chunks of every small size, with and without a final newline,
// and the shapes the streamed path must hand back (multi-line records, two headers in a row, a trailing header, lines longer than the read-ahead).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include "../ebwt2snp_amd/host/fasta_index.hpp"

static bool chunked(const std::string& f, size_t ch, size_t over, FastaJoined& ix) {
    const uint64_t n = f.size();
    std::vector<FastaChunk> cs((size_t)((n + ch - 1) / ch));
    for (size_t c = 0; c < cs.size(); ++c) {
        const uint64_t a = c * ch, b = std::min<uint64_t>(a + ch, n);
        const uint64_t lo = a ? a - 1 : 0, hi = std::min<uint64_t>(b + over, n);
        std::string buf = f.substr((size_t)lo, (size_t)(hi - lo));     // (a copy: reading outside [lo, hi) would show under the sanitizers)
        e2s_fasta_scan_chunk((const uint8_t*)buf.data(), lo, buf.size(), a, b, n, cs[c]);
    }
    return e2s_fasta_join_chunks(cs, ix);
}
int main() {
    std::mt19937_64 rng(7);
    int cases = 0, handed_back = 0;
    for (int it = 0; it < 400; ++it) {
        const int shape = it % 8;                                    // 0..4 plain; 5 multi-line; 6 two headers in a row; 7 trailing header
        const int nrec = 1 + (int)(rng() % 40);
        const bool same_len = rng() % 2;
        const int L0 = (int)(rng() % 30);
        std::string f;
        for (int r = 0; r < nrec; ++r) {
            f += ">r" + std::to_string(rng() % 100000) + "\n";
            if (shape == 6 && r == nrec / 2) f += ">extra\n";
            const int L = same_len ? L0 : (int)(rng() % 30);
            std::string d; for (int i = 0; i < L; ++i) d += "ACGTN"[rng() % 5];
            if (shape == 5 && r == nrec / 2 && L >= 2) f += d.substr(0, L / 2) + "\n" + d.substr(L / 2) + "\n";
            else f += d + "\n";
        }
        if (shape == 7) f += ">last\n";
        if (rng() % 2 && shape != 7) f.pop_back();                   // no newline at the end of the file
        FastaIndex want;
        const bool ok_want = e2s_index_fasta((const uint8_t*)f.data(), f.size(), 1 + (unsigned)(rng() % 3), want);
        const bool plain_want = ok_want && want.joined.empty() && want.text == (const uint8_t*)f.data();
        for (size_t ch : {1u, 2u, 3u, 5u, 7u, 16u, 64u, 1000u}) {
            for (size_t over : {64u, 8u}) {
                FastaJoined got;
                const bool ok = chunked(f, ch, over, got);
                ++cases;
                if (!ok) {                                           // handed back: always allowed for what is not plain; for a plain file only when a line outruns the read-ahead
                    ++handed_back;
                    if (plain_want && over >= 64) { printf("FAIL: a plain file was handed back (it %d chunk %zu)\n", it, ch); return 1; }
                    continue;
                }
                if (!plain_want) { printf("FAIL: a file that is not plain was indexed (it %d shape %d chunk %zu)\n", it, shape, ch); return 1; }
                if (got.n != want.start.size() || memcmp(got.start.get(), want.start.data(), 8 * got.n) || memcmp(got.len.get(), want.len.data(), 4 * got.n) || got.max_len != want.max_len) { printf("FAIL: index differs (it %d chunk %zu over %zu)\n", it, ch, over); return 1; }
                bool u = want.max_len > 0; for (uint32_t l : want.len) u = u && l == want.max_len;
                if (u != got.uniform) { printf("FAIL: uniform flag (it %d)\n", it); return 1; }
            }
        }
    }
    printf("fasta chunks: %d cases, %d handed back to the whole-file index, all as expected\n", cases, handed_back);
    return 0;
}
