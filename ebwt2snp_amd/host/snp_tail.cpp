// snp_tail.cpp -- host tail of clust2snp: candidates -> KisSNP2 records.
//
// Replaces extract_variants + get_reads + cons + distance + to_file
// (clust2snp.cpp:147-212, 218-302, 505-780; include.hpp:334-371).  Candidates are rare
// (about one per 12k EGSA entries), so this stays on the host; the FASTA is memory-mapped and
// walked once up to the last read any candidate needs.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <thread>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <ctime>
#include <cstdlib>

#include "../../include/e2s_hip.h"
#include "../csrc/e2s_env.h"

namespace {

// include.hpp:265-279.  'N'/'n' is rand()%4 in the reference (seeded with time()): not reproducible
// there; mapped to 0 here ("non-determinism of the reference").
inline int base_to_int(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
    }
    return 0;
}

struct Mapped {
    const char* p = nullptr; size_t n = 0; int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char*)m;
        madvise(m, n, MADV_SEQUENTIAL);
        return true;
    }
    ~Mapped() { if (p && n) munmap((void*)p, n); if (fd >= 0) ::close(fd); }
};

// std::getline over the mapped file with ifstream's eof/fail semantics
struct LineReader {
    const char* p; size_t n, at = 0; bool eof = false, failed = false;
    LineReader(const char* p_, size_t n_) : p(p_), n(n_) {}
    // returns false (line untouched) once the stream has failed
    bool next(const char*& b, size_t& len) {
        if (failed) return false;
        if (at >= n) { eof = true; failed = true; b = p + n; len = 0; return true; }   // erased, nothing extracted
        const char* s = p + at;
        const char* e = (const char*)memchr(s, '\n', n - at);
        if (e) { b = s; len = (size_t)(e - s); at = (size_t)(e - p) + 1; }
        else { b = s; len = n - at; at = n; eof = true; }
        return true;
    }
};

// E2S_CLI_TIMING=1: phase times of the tail on stderr
struct TailClock {
    bool on = getenv("E2S_CLI_TIMING") != nullptr;
    double t0 = now();
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    void mark(const char* what) { if (!on) return; const double t = now(); fprintf(stderr, "[timing]   tail: %s %.2f ms\n", what, t - t0); t0 = t; }
};

// Worker threads of the tail (the reference is single-threaded; every result below is independent of the thread count).
// E2S_TAIL_THREADS fixes the number; otherwise one thread per `grain` items, at most 16 / the machine's cores.
unsigned tail_threads(uint64_t items, uint64_t grain) {
    const int env = getenv("E2S_TAIL_THREADS") ? atoi(getenv("E2S_TAIL_THREADS")) : 0;       // (read per call: tests switch it)
    if (env > 0) return (unsigned)env;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(16u, hw), items / std::max<uint64_t>(grain, 1)));
}
template <class F>
void run_threads(unsigned T, F f) {                             // f(t) for t in [0, T)
    if (T <= 1) { f(0u); return; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&f, t]() { f(t); });
    f(0u);
    for (auto& x : th) x.join();
}

// get_reads on a regular FASTA, in parallel.  Regular: the first line is a header ('>'), every header line is followed by
// at least one line that is no header, and every wanted rank exists -- then record j is the j-th header line and its DNA
// the lines up to the next header, which is all get_reads' line juggling amounts to.  The file is cut into one piece per
// thread at line starts; a thread counts the headers of its piece, and after a prefix sum extracts the wanted records
// that BEGIN in its piece (reading on past the piece's end for the last one).  Anything irregular: returns false and the
// sequential restatement below decides.
// The reads the tail works on: views into the mapped FASTA where a record's DNA is one line (nothing is copied), a string
// of its own where it spans lines (or where the sequential restatement produced it).
struct ReadRef { const char* p; size_t n; size_t size() const { return n; } };
struct ReadStore {
    Mapped f;
    std::vector<ReadRef> ref;                                   // one per wanted rank
    std::vector<std::deque<std::string>> joined;                // multi-line records, per worker thread (stable addresses)
    std::vector<std::string> seq;                               // the sequential path's copies
};

bool fetch_reads_parallel(ReadStore& S, const std::vector<uint64_t>& ranks, unsigned T) {
    const Mapped& f = S.f;
    std::vector<ReadRef>& out = S.ref;
    const char* p = f.p; const size_t n = f.n;
    if (n == 0 || p[0] != '>' || ranks.empty()) return false;
    std::vector<size_t> cut(T + 1);
    cut[0] = 0; cut[T] = n;
    for (unsigned t = 1; t < T; ++t) {                          // first line start at or after n*t/T
        size_t c = (size_t)((unsigned __int128)n * t / T);
        if (c == 0) { cut[t] = 0; continue; }
        const char* e = (const char*)memchr(p + c - 1, '\n', n - (c - 1));
        cut[t] = e ? (size_t)(e - p) + 1 : n;
    }
    for (unsigned t = 1; t <= T; ++t) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    TailClock fc;
    struct Piece { uint64_t headers = 0; bool any = false, first_hdr = false, last_hdr = false, hh = false; };
    std::vector<Piece> pc(T);
    run_threads(T, [&](unsigned t) {
        Piece q;
        size_t at = cut[t];
        bool prev_hdr = false;
        while (at < cut[t + 1]) {
            const bool hdr = p[at] == '>';
            if (!q.any) { q.any = true; q.first_hdr = hdr; }
            if (hdr) { ++q.headers; if (prev_hdr) q.hh = true; }
            prev_hdr = hdr;
            const char* e = (const char*)memchr(p + at, '\n', n - at);
            at = e ? (size_t)(e - p) + 1 : n;
        }
        q.last_hdr = prev_hdr;
        pc[t] = q;
    });
    fc.mark("  fasta: headers counted per piece");
    uint64_t total = 0; bool prev_hdr = false, seen = false;
    std::vector<uint64_t> base(T + 1, 0);
    for (unsigned t = 0; t < T; ++t) {
        base[t] = total; total += pc[t].headers;
        if (pc[t].hh) return false;
        if (pc[t].any) { if (seen && prev_hdr && pc[t].first_hdr) return false; prev_hdr = pc[t].last_hdr; seen = true; }
    }
    base[T] = total;
    if (prev_hdr) return false;                                 // the file ends with a header line
    if (ranks.back() >= total) return false;                    // a rank past the last read: the reference repeats stale text
    out.assign(ranks.size(), ReadRef{"", 0});
    S.joined.assign(T, std::deque<std::string>());
    fc.mark("  fasta: output slots");
    run_threads(T, [&](unsigned t) {
        if (!pc[t].headers) return;
        size_t w = (size_t)(std::lower_bound(ranks.begin(), ranks.end(), base[t]) - ranks.begin());
        const size_t w_end = (size_t)(std::lower_bound(ranks.begin(), ranks.end(), base[t + 1]) - ranks.begin());
        if (w == w_end) return;
        size_t at = cut[t];
        uint64_t j = base[t];                                   // rank of the next header met
        bool taking = false;
        int lines = 0;                                          // DNA lines of the record being taken
        std::string* own = nullptr;
        while (at < n) {
            const bool hdr = p[at] == '>';
            const char* e = (const char*)memchr(p + at, '\n', n - at);
            const size_t le = e ? (size_t)(e - p) : n;
            if (hdr) {
                if (taking) { ++w; taking = false; if (w == w_end) break; }
                if (at >= cut[t + 1]) break;                    // (only reached while finishing a record of this piece)
                taking = (j == ranks[w]);
                lines = 0; own = nullptr;
                ++j;
            } else if (taking) {
                if (lines == 0) out[w] = ReadRef{p + at, le - at};
                else {
                    if (!own) { S.joined[t].emplace_back(out[w].p, out[w].n); own = &S.joined[t].back(); }
                    own->append(p + at, le - at);
                    out[w] = ReadRef{own->data(), own->size()};
                }
                ++lines;
            }
            at = e ? le + 1 : n;
        }
    });
    fc.mark("  fasta: wanted reads copied");
    return true;
}

// get_reads, clust2snp.cpp:147-212: the DNA of the reads with the given ascending ranks
bool fetch_reads(const char* path, const std::vector<uint64_t>& ranks, ReadStore& S, unsigned* pieces = nullptr) {
    if (pieces) *pieces = 1;
    Mapped& f = S.f;
    std::vector<std::string>& out = S.seq;
    LineReader R(nullptr, 0);
    const bool ok = f.open(path);
    if (ok) {
        const long par_min = e2s_env_dev("E2S_TAIL_PAR_MIN_BYTES") ? atol(e2s_env_dev("E2S_TAIL_PAR_MIN_BYTES")) : (8L << 20);
        const unsigned T = tail_threads(f.n, (uint64_t)std::max(par_min, 1L));
        if (T > 1 && f.n >= (size_t)par_min && fetch_reads_parallel(S, ranks, T)) { if (pieces) *pieces = T; return true; }
    }
    if (ok) R = LineReader(f.p, f.n); else R.failed = true;
    const char* lb = ""; size_t ll = 0;          // `line`
    std::string dna;
    auto first_char = [&]() { return ll ? lb[0] : '\0'; };
    const char* b; size_t l;
    if (R.next(b, l)) { /* ID */ }
    if (R.next(b, l)) { lb = b; ll = l; }
    dna.assign(lb, ll);
    while (!R.eof && first_char() != '>') {
        if (R.next(b, l)) { lb = b; ll = l; }
        if (first_char() != '>') dna.append(lb, ll);
    }
    uint64_t j = 0;
    out.clear();
    out.reserve(ranks.size());
    for (uint64_t want : ranks) {
        while (j < want) {
            if (R.next(b, l)) { lb = b; ll = l; }
            dna.assign(lb, ll);
            while (!R.eof && first_char() != '>') {
                if (R.next(b, l)) { lb = b; ll = l; }
                if (first_char() != '>') dna.append(lb, ll);
            }
            ++j;
        }
        out.push_back(dna);
    }
    S.ref.clear();
    S.ref.reserve(out.size());
    for (const std::string& r : out) S.ref.push_back(ReadRef{r.data(), r.size()});
    return ok;
}

// dH, clust2snp.cpp:218-232: Hamming distance of the right-aligned overlap
inline int dH(const char* a, size_t la, const char* b, size_t lb) {
    const size_t len = la < lb ? la : lb;
    int d = 0;
    for (size_t i = 1; i <= len; ++i) d += a[la - i] != b[lb - i];
    return d;
}

// distance, clust2snp.cpp:254-302
void distance(const std::string& a, const std::string& b, int max_gap, int& D, int& L) {
    const int none = dH(a.data(), a.size(), b.data(), b.size());
    D = none; L = 0;
    if (max_gap <= 0) return;
    int best_ab = 0, best_ba = 0, ia = 0, ib = 0;
    for (int i = 1; i <= max_gap; ++i) {
        // a.substr(0, a.length()-i): a size_t wrap for i > length keeps the whole string
        const size_t la = (size_t)i <= a.size() ? a.size() - (size_t)i : a.size();
        const size_t lb = (size_t)i <= b.size() ? b.size() - (size_t)i : b.size();
        const int dab = dH(a.data(), la, b.data(), b.size()) + i;
        const int dba = dH(a.data(), a.size(), b.data(), lb) + i;
        if (i == 1 || dab < best_ab) { best_ab = dab; ia = i; }       // first minimum (std::min_element)
        if (i == 1 || dba < best_ba) { best_ba = dba; ib = i; }
    }
    if (none < best_ab && none < best_ba) return;                     // strictly better than both (H9)
    if (best_ab < best_ba) { D = best_ab - ia; L = ia; }
    else { D = best_ba - ib; L = -ib; }
}

struct Variant { std::string left0, left1, right; int s0, s1; };

}  // namespace

// one candidate, wherever it is stored
struct CandView {
    uint32_t n0, n1, right_idx, right_pos;
    const uint32_t* idx0; const uint32_t* idx1;
    const uint8_t* pos0; const uint8_t* pos1; uint32_t pw;       // positions: pw bytes each, little endian
    uint32_t pos(int s, uint32_t j) const {
        const uint8_t* p = s ? pos1 : pos0;
        return pw == 1 ? (uint32_t)p[j] : pw == 2 ? (uint32_t)p[2 * j] | ((uint32_t)p[2 * j + 1] << 8)
                                                   : (uint32_t)p[4 * j] | ((uint32_t)p[4 * j + 1] << 8) | ((uint32_t)p[4 * j + 2] << 16) | ((uint32_t)p[4 * j + 3] << 24);
    }
};

template <typename Get>
static int snp_core(uint64_t n_cand, uint32_t list_cap, Get get, const e2s_tail_params* p, const char* fasta_path, const char* out_path,
                    uint64_t* n_variants, uint64_t* n_events) {
    if (n_variants) *n_variants = 0;
    if (n_events) *n_events = 0;
    // zero candidates: the reference indexes an empty vector (clust2snp.cpp:531) and dies before creating the file
    if (n_cand == 0) return E2S_ERR_REF_CRASH;

    // all read ranks any candidate touches, ascending and unique (clust2snp.cpp:512-523): a bitmap over the rank range instead
    // of sorting ~40 ranks per candidate; its per-word prefix counts then find a read's slot in O(1)
    TailClock tc;
    uint32_t max_rank = 0;
    for (uint64_t c = 0; c < n_cand; ++c) {
        const CandView v = get(c);
        if (v.n0 > list_cap || v.n1 > list_cap) return E2S_ERR_ARG;
        for (uint32_t t = 0; t < v.n0; ++t) max_rank = std::max(max_rank, v.idx0[t]);
        for (uint32_t t = 0; t < v.n1; ++t) max_rank = std::max(max_rank, v.idx1[t]);
        max_rank = std::max(max_rank, v.right_idx);
    }
    const size_t n_words = ((size_t)max_rank >> 6) + 1;
    std::vector<uint64_t> bits(n_words, 0);
    auto mark = [&](uint32_t r) { bits[r >> 6] |= 1ull << (r & 63u); };
    for (uint64_t c = 0; c < n_cand; ++c) {
        const CandView v = get(c);
        for (uint32_t t = 0; t < v.n0; ++t) mark(v.idx0[t]);
        for (uint32_t t = 0; t < v.n1; ++t) mark(v.idx1[t]);
        mark(v.right_idx);
    }
    std::vector<uint32_t> before(n_words);                      // set bits in the words before this one
    uint64_t n_ranks = 0;
    for (size_t w = 0; w < n_words; ++w) { before[w] = (uint32_t)n_ranks; n_ranks += (uint64_t)__builtin_popcountll(bits[w]); }
    if (n_ranks >= (1ull << 32)) return E2S_ERR_ARG;
    std::vector<uint64_t> ranks;
    ranks.reserve((size_t)n_ranks);
    for (size_t w = 0; w < n_words; ++w)
        for (uint64_t m = bits[w]; m; m &= m - 1) ranks.push_back(((uint64_t)w << 6) + (uint64_t)__builtin_ctzll(m));
    if (ranks.size() == 1) return E2S_ERR_REF_CRASH;       // (i*100)/(size-1), clust2snp.cpp:200
    ReadStore store;
    tc.mark("read ranks collected");
    unsigned pieces = 1;
    if (!fetch_reads(fasta_path, ranks, store, &pieces)) return E2S_ERR_IO;
    const std::vector<ReadRef>& reads = store.ref;
    tc.mark(pieces > 1 ? "reads fetched from the FASTA (in parallel pieces)" : "reads fetched from the FASTA (sequential)");
    auto read_of = [&](uint32_t rank) -> const ReadRef& {
        return reads[(size_t)before[rank >> 6] + (size_t)__builtin_popcountll(bits[rank >> 6] & ((1ull << (rank & 63u)) - 1ull))];
    };

    const int kl = p->k_left;
    if (kl < 0) return E2S_ERR_REF_CRASH;                  // cons(size_t(-n)) throws in the reference
    unsigned char b2i[256];
    for (int ch = 0; ch < 256; ++ch) b2i[ch] = (unsigned char)base_to_int((unsigned char)ch);
    // candidates are independent: worker threads take blocks of them; the variants are collected in candidate order afterwards
    std::vector<Variant> slot((size_t)n_cand);
    std::vector<unsigned char> keep((size_t)n_cand, 0);
    std::atomic<uint64_t> next_block{0};
    std::atomic<int> crashed{0};
    constexpr uint64_t BLOCK = 256;
    const unsigned T = tail_threads(n_cand, 2048);
    run_threads(T, [&](unsigned) {
        std::vector<int> cnt((size_t)kl * 4);
        std::vector<unsigned char> ctx;
        for (;;) {
            const uint64_t c0 = next_block.fetch_add(BLOCK), c1 = std::min<uint64_t>(n_cand, c0 + BLOCK);
            if (c0 >= n_cand) break;
            for (uint64_t c = c0; c < c1; ++c) {
                const CandView cv = get(c);
                const CandView* h = &cv;
                std::string cons[2];
                int supp[2] = {0, 0};
                for (int s = 0; s < 2; ++s) {
                    const uint32_t n = s ? h->n1 : h->n0;
                    const uint32_t* idx = s ? cv.idx1 : cv.idx0;
                    // cons, include.hpp:334-371: the letter changes only on a strictly larger count (H8)
                    std::string& C = cons[s];
                    C.assign((size_t)kl, 'A');
                    std::fill(cnt.begin(), cnt.end(), 0);
                    // the left contexts of the allele, fetched once: ctx[j][i] = at(read_j, pos_j + i) (0 past the read's end)
                    ctx.resize((size_t)n * (size_t)kl);
                    for (uint32_t j = 0; j < n; ++j) {
                        const ReadRef& r = read_of(idx[j]);
                        unsigned char* row = ctx.data() + (size_t)j * (size_t)kl;
                        const uint64_t p0 = cv.pos(s, j), len = r.size();
                        const uint64_t have = p0 < len ? std::min<uint64_t>(len - p0, (uint64_t)kl) : 0;
                        memcpy(row, r.p + (have ? p0 : 0), (size_t)have);
                        memset(row + have, 0, (size_t)kl - (size_t)have);
                    }
                    for (uint32_t j = 0; j < n; ++j) {
                        const unsigned char* row = ctx.data() + (size_t)j * (size_t)kl;
                        for (int i = 0; i < kl; ++i) {
                            const unsigned char ch = row[i];
                            const int b = b2i[ch];
                            if (++cnt[(size_t)i * 4 + b] > cnt[(size_t)i * 4 + b2i[(unsigned char)C[i]]]) C[i] = (char)ch;
                        }
                    }
                    for (uint32_t j = 0; j < n; ++j) {             // support = contexts within max_err of the consensus
                        const unsigned char* row = ctx.data() + (size_t)j * (size_t)kl;
                        int d = 0;
                        for (int i = 0; i < kl; ++i) d += (unsigned char)C[i] != row[i];
                        if (d <= p->max_err) supp[s]++;
                    }
                }
                if (supp[0] > 0 && supp[1] > 0) {                  // clust2snp.cpp:595-611
                    const ReadRef& r = read_of(h->right_idx);
                    if (h->right_pos > r.size()) { crashed.store(1); continue; }      // substr throws in the reference
                    Variant& v = slot[(size_t)c];
                    v.left0 = cons[0]; v.left1 = cons[1];
                    const size_t avail = r.n - (size_t)h->right_pos;
                    v.right.assign(r.p + h->right_pos, p->k_right < 0 ? avail : std::min<size_t>(avail, (size_t)p->k_right));
                    v.s0 = supp[0]; v.s1 = supp[1];
                    keep[(size_t)c] = 1;
                }
            }
        }
    });
    if (crashed.load()) return E2S_ERR_REF_CRASH;
    std::vector<Variant> vars;
    for (uint64_t c = 0; c < n_cand; ++c) if (keep[(size_t)c]) vars.push_back(std::move(slot[(size_t)c]));
    if (n_variants) *n_variants = vars.size();

    tc.mark("consensus");
    // to_file, clust2snp.cpp:633-780
    FILE* out = fopen(out_path, "w");
    if (!out) return E2S_ERR_IO;
    uint64_t id = 1;
    std::string line;
    for (const Variant& v : vars) {
        int D, L;
        distance(v.left0, v.left1, p->max_gap, D, L);
        if (D > 3) continue;                               // max_snvs_def: -v is parsed but unused (H5)
        const char* kind = L != 0 ? "INDEL" : "SNP";
        std::string type;
        if (L == 0) { type += v.left0.back(); type += '/'; type += v.left1.back(); }     // last consensus letters (H20)
        else if (L > 0) { type = v.left0.substr(v.left0.size() - (size_t)L); type += '/'; }
        else { type = "/"; type += v.left1.substr(v.left1.size() - (size_t)(-L)); }
        const std::string rl = std::to_string(v.right.size());
        const std::string ids = std::to_string(id);
        line.clear();
        line += ">"; line += kind; line += "_higher_path_"; line += ids; line += "|P_1:"; line += rl; line += "_"; line += type;
        line += "|"; line += std::to_string(v.s0); line += "|nb_pol_1\n";
        line += (L < 0 ? v.left0.substr((size_t)(-L)) : v.left0); line += v.right; line += "\n";
        line += ">"; line += kind; line += "_lower_path_"; line += ids; line += "|P_1:"; line += rl; line += "_"; line += type;
        line += "|"; line += std::to_string(v.s1); line += "|nb_pol_1\n";
        line += (L > 0 ? v.left1.substr((size_t)L) : v.left1); line += v.right; line += "\n";
        fwrite(line.data(), 1, line.size(), out);
        ++id;
    }
    fclose(out);
    tc.mark("distance + text");
    if (n_events) *n_events = id - 1;
    return E2S_OK;
}

extern "C" int e2s_snp_from_candidates(const void* cands, const uint32_t* order, uint64_t n_cand, uint32_t list_cap,
                                       const e2s_tail_params* p, const char* fasta_path, const char* out_path,
                                       uint64_t* n_variants, uint64_t* n_events) {
    if (!p || !fasta_path || !out_path || (!cands && n_cand)) return E2S_ERR_ARG;
    const size_t stride = sizeof(e2s_cand_header) + (size_t)4 * list_cap * sizeof(uint32_t);
    const uint8_t* base = (const uint8_t*)cands;
    auto get = [&](uint64_t c) {                                      // SA order through `order`
        const uint8_t* r = base + (size_t)(order ? order[c] : c) * stride;
        const e2s_cand_header* h = (const e2s_cand_header*)r;
        const uint32_t* l = (const uint32_t*)(r + sizeof(e2s_cand_header));
        return CandView{h->n0, h->n1, h->right_idx, h->right_pos, l, l + 2 * (size_t)list_cap,
                        (const uint8_t*)(l + list_cap), (const uint8_t*)(l + 3 * (size_t)list_cap), 4u};
    };
    return snp_core(n_cand, list_cap, get, p, fasta_path, out_path, n_variants, n_events);
}

extern "C" int e2s_snp_from_packed(const void* stream, const uint32_t* offsets8, uint64_t n_cand, uint32_t pos_bytes, const e2s_tail_params* p,
                                   const char* fasta_path, const char* out_path, uint64_t* n_variants, uint64_t* n_events) {
    if (!p || !fasta_path || !out_path || ((!stream || !offsets8) && n_cand)) return E2S_ERR_ARG;
    if (pos_bytes != 1 && pos_bytes != 2 && pos_bytes != 4) return E2S_ERR_ARG;
    const uint8_t* base = (const uint8_t*)stream;
    auto get = [&](uint64_t c) {
        const uint8_t* r = base + (size_t)offsets8[c] * 8;
        const uint32_t* w = (const uint32_t*)r;
        const uint32_t n0 = w[0], n1 = w[1];
        const uint8_t* p0 = r + 16 + 4 * (size_t)(n0 + n1);
        return CandView{n0, n1, w[2], w[3], w + 4, w + 4 + n0, p0, p0 + (size_t)pos_bytes * n0, pos_bytes};
    };
    return snp_core(n_cand, 0xFFFFFFFFu, get, p, fasta_path, out_path, n_variants, n_events);
}
