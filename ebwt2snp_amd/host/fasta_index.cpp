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

void e2s_fasta_scan_chunk(const uint8_t* buf, uint64_t buf_off, size_t buf_n, uint64_t a, uint64_t b, uint64_t file_n, FastaChunk& out) {
    out = FastaChunk();
    const uint64_t buf_end = buf_off + buf_n;
    uint64_t pos = a;
    if (a > 0 && buf[a - 1 - buf_off] != '\n') {                // the chunk begins inside a line that started before it
        const uint8_t* e = (const uint8_t*)memchr(buf + (a - buf_off), '\n', (size_t)(buf_end - a));
        if (!e) return;                                          // ... and that line covers the whole buffer: none starts here
        pos = (uint64_t)(e - buf) + buf_off + 1;
    }
    bool reserved = false;
    while (pos < b && pos < file_n) {
        const bool hdr = buf[pos - buf_off] == '>';
        const uint8_t* e = (const uint8_t*)memchr(buf + (pos - buf_off), '\n', (size_t)(buf_end - pos));
        if (!e && buf_end < file_n) { out.plain = false; return; }   // the line runs past the read-ahead
        const uint64_t le = e ? (uint64_t)(e - buf) + buf_off : file_n;
        const int kind = hdr ? 0 : 1;
        if (out.first_kind < 0) out.first_kind = kind;
        if (hdr) { ++out.headers; if (out.last_kind == 0) out.plain = false; }
        else {
            if (out.last_kind == 1) out.plain = false;
            const uint64_t l = le - pos;
            if (l > 0xFFFFFFFFull) { out.plain = false; return; }
            if (!reserved) {                                     // (records about as long as this line and a short header: one allocation per chunk)
                reserved = true;
                const size_t guess = (size_t)((b - pos) / (l + 8) + 16);
                out.start.reserve(guess); out.len.reserve(guess);
            }
            out.start.push_back(pos); out.len.push_back((uint32_t)l);
            out.max_len = std::max(out.max_len, (uint32_t)l); out.min_len = std::min(out.min_len, (uint32_t)l);
        }
        out.last_kind = kind;
        pos = e ? le + 1 : file_n;
    }
}

bool e2s_fasta_join_chunks(std::vector<FastaChunk>& chunks, FastaJoined& out) {
    out = FastaJoined();
    int last = -1; uint64_t total = 0, headers = 0; uint32_t mn = 0xFFFFFFFFu, mx = 0;
    for (const FastaChunk& c : chunks) {
        if (!c.plain) return false;
        if (c.first_kind < 0) continue;
        if (last < 0 ? c.first_kind != 0 : c.first_kind == last) return false;     // the file begins with a header; header and DNA line alternate
        last = c.last_kind;
        total += c.start.size(); headers += c.headers;
        mn = std::min(mn, c.min_len); mx = std::max(mx, c.max_len);
    }
    if (last != 1 || total == 0 || total != headers) return false;                // (a trailing header has no DNA line)
    out.start.reset(new uint64_t[total]); out.len.reset(new uint32_t[total]);
    std::vector<uint64_t> base(chunks.size() + 1, 0);
    for (size_t i = 0; i < chunks.size(); ++i) base[i + 1] = base[i] + chunks[i].start.size();
    const unsigned T = (unsigned)std::min<size_t>(8, std::max<size_t>(1, chunks.size()));
    run_threads(T, [&](unsigned t) {
        for (size_t i = chunks.size() * t / T; i < chunks.size() * (t + 1) / T; ++i) {
            FastaChunk& c = chunks[i];
            if (c.start.empty()) continue;
            memcpy(out.start.get() + base[i], c.start.data(), c.start.size() * 8);
            memcpy(out.len.get() + base[i], c.len.data(), c.len.size() * 4);
            std::vector<uint64_t>().swap(c.start); std::vector<uint32_t>().swap(c.len);
        }
    });
    out.n = total; out.max_len = mx; out.uniform = mx > 0 && mn == mx;
    return true;
}
