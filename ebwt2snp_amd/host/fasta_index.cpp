// fasta_index.cpp -- see fasta_index.hpp.  The file is cut into pieces at header lines; every piece is scanned by a thread
// of its own (memchr per line), twice: once to count, once to fill.
#include "fasta_index.hpp"

#include <algorithm>
#include <cstring>
#include <thread>

namespace {
template <class F> void run_threads(unsigned T, F f) {
    if (T <= 1) { f(0u); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&f, t]() { f(t); });
    f(0u);
    for (auto& x : th) x.join();
}
struct Piece { uint64_t headers = 0, dna_bytes = 0; bool multi = false, hh = false, last_hdr = false; };
}  // namespace

bool e2s_index_fasta(const uint8_t* p, size_t n, unsigned threads, FastaIndex& out) {
    out = FastaIndex();
    if (n == 0 || p[0] != '>') return false;
    const unsigned T = std::max(1u, std::min<unsigned>(threads, (unsigned)(n / (4u << 20)) + 1u));
    // piece t starts at the first header line at or after n*t/T
    std::vector<size_t> cut(T + 1);
    cut[0] = 0; cut[T] = n;
    for (unsigned t = 1; t < T; ++t) {
        size_t c = (size_t)((unsigned __int128)n * t / T);
        for (;;) {
            const uint8_t* e = c ? (const uint8_t*)memchr(p + c - 1, '\n', n - (c - 1)) : nullptr;
            if (c && !e) { c = n; break; }
            if (c) c = (size_t)(e - p) + 1;
            if (c >= n || p[c] == '>') break;
            ++c;
        }
        cut[t] = std::min(c, n);
    }
    for (unsigned t = 1; t <= T; ++t) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    std::vector<Piece> pc(T);
    run_threads(T, [&](unsigned t) {
        Piece q;
        size_t at = cut[t];
        bool prev_hdr = false; int lines = 0;
        while (at < cut[t + 1]) {
            const bool hdr = p[at] == '>';
            const uint8_t* e = (const uint8_t*)memchr(p + at, '\n', n - at);
            const size_t le = e ? (size_t)(e - p) : n;
            if (hdr) { ++q.headers; if (prev_hdr) q.hh = true; lines = 0; }
            else { q.dna_bytes += le - at; if (++lines > 1) q.multi = true; }
            prev_hdr = hdr;
            at = e ? le + 1 : n;
        }
        q.last_hdr = prev_hdr;
        pc[t] = q;
    });
    uint64_t total = 0, dna = 0; bool multi = false;
    std::vector<uint64_t> base(T + 1, 0), dbase(T + 1, 0);
    for (unsigned t = 0; t < T; ++t) {
        base[t] = total; dbase[t] = dna;
        total += pc[t].headers; dna += pc[t].dna_bytes;
        if (pc[t].hh || pc[t].last_hdr) return false;       // (a piece ends where the next header begins: a trailing header has no DNA line)
        multi = multi || pc[t].multi;
    }
    if (total == 0) return false;
    out.start.resize(total); out.len.resize(total);
    if (multi) { out.joined.resize(dna); out.text = out.joined.data(); out.text_bytes = dna; }
    else { out.text = p; out.text_bytes = n; }
    std::vector<uint32_t> mx(T, 0);
    bool too_long = false;
    run_threads(T, [&](unsigned t) {
        size_t at = cut[t];
        uint64_t j = base[t], w = dbase[t];
        uint64_t cur_start = 0, cur_len = 0; bool open = false;
        auto close = [&]() {
            if (!open) return;
            if (cur_len > 0xFFFFFFFFull) too_long = true;
            out.start[j - 1] = cur_start; out.len[j - 1] = (uint32_t)cur_len;
            mx[t] = std::max<uint32_t>(mx[t], (uint32_t)std::min<uint64_t>(cur_len, 0xFFFFFFFFull));
        };
        while (at < cut[t + 1]) {
            const bool hdr = p[at] == '>';
            const uint8_t* e = (const uint8_t*)memchr(p + at, '\n', n - at);
            const size_t le = e ? (size_t)(e - p) : n;
            if (hdr) { close(); ++j; open = true; cur_len = 0; cur_start = multi ? w : 0; }
            else {
                if (multi) { memcpy(out.joined.data() + w, p + at, le - at); w += le - at; }
                else if (cur_len == 0) cur_start = at;
                cur_len += le - at;
            }
            at = e ? le + 1 : n;
        }
        close();
    });
    if (too_long) return false;
    for (unsigned t = 0; t < T; ++t) out.max_len = std::max(out.max_len, mx[t]);
    return true;
}
