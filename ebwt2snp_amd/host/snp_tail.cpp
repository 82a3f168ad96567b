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
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <ctime>
#include <cstdlib>

#include "../../include/e2s_hip.h"

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

// get_reads, clust2snp.cpp:147-212: the DNA of the reads with the given ascending ranks
bool fetch_reads(const char* path, const std::vector<uint64_t>& ranks, std::vector<std::string>& out) {
    Mapped f;
    LineReader R(nullptr, 0);
    const bool ok = f.open(path);
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

// E2S_CLI_TIMING=1: phase times of the tail on stderr
struct TailClock {
    bool on = getenv("E2S_CLI_TIMING") != nullptr;
    double t0 = now();
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    void mark(const char* what) { if (!on) return; const double t = now(); fprintf(stderr, "[timing]   tail: %s %.2f ms\n", what, t - t0); t0 = t; }
};

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

    // all read ranks any candidate touches, ascending and unique (clust2snp.cpp:512-523)
    std::vector<uint64_t> ranks;
    ranks.reserve(n_cand * 8);
    for (uint64_t c = 0; c < n_cand; ++c) {
        const CandView v = get(c);
        if (v.n0 > list_cap || v.n1 > list_cap) return E2S_ERR_ARG;
        for (uint32_t t = 0; t < v.n0; ++t) ranks.push_back(v.idx0[t]);
        for (uint32_t t = 0; t < v.n1; ++t) ranks.push_back(v.idx1[t]);
        ranks.push_back(v.right_idx);
    }
    TailClock tc;
    std::sort(ranks.begin(), ranks.end());
    ranks.erase(std::unique(ranks.begin(), ranks.end()), ranks.end());
    if (ranks.size() == 1) return E2S_ERR_REF_CRASH;       // (i*100)/(size-1), clust2snp.cpp:200
    std::vector<std::string> reads;
    tc.mark("read ranks sorted");
    if (!fetch_reads(fasta_path, ranks, reads)) return E2S_ERR_IO;
    tc.mark("reads fetched from the FASTA");
    auto read_of = [&](uint32_t rank) -> const std::string& {
        return reads[(size_t)(std::lower_bound(ranks.begin(), ranks.end(), (uint64_t)rank) - ranks.begin())];
    };

    const int kl = p->k_left;
    if (kl < 0) return E2S_ERR_REF_CRASH;                  // cons(size_t(-n)) throws in the reference
    std::vector<Variant> vars;
    std::vector<int> cnt((size_t)kl * 4);
    std::vector<unsigned char> ctx;
    unsigned char b2i[256];
    for (int ch = 0; ch < 256; ++ch) b2i[ch] = (unsigned char)base_to_int((unsigned char)ch);
    for (uint64_t c = 0; c < n_cand; ++c) {
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
                const std::string& r = read_of(idx[j]);
                unsigned char* row = ctx.data() + (size_t)j * (size_t)kl;
                const uint64_t p0 = cv.pos(s, j), len = r.size();
                const uint64_t have = p0 < len ? std::min<uint64_t>(len - p0, (uint64_t)kl) : 0;
                memcpy(row, r.data() + (have ? p0 : 0), (size_t)have);
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
            const std::string& r = read_of(h->right_idx);
            if (h->right_pos > r.size()) return E2S_ERR_REF_CRASH;      // substr throws
            Variant v;
            v.left0 = cons[0]; v.left1 = cons[1];
            v.right = p->k_right < 0 ? r.substr(h->right_pos) : r.substr(h->right_pos, (size_t)p->k_right);
            v.s0 = supp[0]; v.s1 = supp[1];
            vars.push_back(std::move(v));
        }
    }
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
