// san_driver.cpp -- TEST INFRASTRUCTURE (make sanitize): the host code that walks pointers, maps files and runs worker threads --
// the host tail (host/snp_tail.cpp), the FASTA index (host/fasta_index.cpp) -- and the oracle's C restatement (oracle/e2s_oracle.c),
// built with -fsanitize=address,undefined or -fsanitize=thread and run over one case:
//   san_driver <input fasta path (EGSA files beside it)> <n1> k m x y z L R g c e mcov pval
// The oracle writes <input>.clusters and its `.snp`; the product's host tail is fed with the oracle's candidates (fixed-stride layout
// and packed stream) and must write the same bytes.  Exit 0: identical and no sanitizer report (a report ends the process non-zero).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/e2s_hip.h"
extern "C" {
#include "../oracle/e2s_oracle.h"
}

static std::vector<unsigned char> slurp(const std::string& p) {
    std::vector<unsigned char> v;
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 15) { fprintf(stderr, "usage: san_driver input n1 k m x y z L R g c e mcov pval\n"); return 2; }
    const std::string in = argv[1];
    orc_e2c_opts eo{atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7])};
    orc_c2s_opts co{};
    co.nr_reads1 = atoll(argv[2]); co.k_left = atoi(argv[8]); co.k_right = atoi(argv[9]); co.max_gap = atoi(argv[10]); co.consensus_reads = atoi(argv[11]);
    co.max_err = atoi(argv[12]); co.mcov_out = atoi(argv[13]); co.pval = atof(argv[14]); co.lcp = eo.lcp; co.da = eo.da; co.pos = eo.pos;
    uint64_t nprinted = 0;
    if (orc_ebwt2clust(in.c_str(), &eo, &nprinted) != ORC_OK) { fprintf(stderr, "oracle ebwt2clust failed\n"); return 3; }
    const std::string want_path = in + ".oracle.snp";
    orc_c2s_stats st{};
    if (orc_clust2snp(in.c_str(), &co, want_path.c_str(), &st, 0) != ORC_OK) { fprintf(stderr, "oracle clust2snp failed\n"); return 3; }
    size_t nw = 0; uint64_t nc = 0; int mcl = 0;
    orc_candidates(in.c_str(), &co, nullptr, 0, &nw, &nc, &mcl);
    std::vector<uint64_t> w(nw + 1);
    if (orc_candidates(in.c_str(), &co, w.data(), w.size(), &nw, &nc, &mcl) != ORC_OK) { fprintf(stderr, "oracle candidates failed\n"); return 3; }
    const uint32_t cap = (uint32_t)(co.consensus_reads ? co.consensus_reads : 20);
    const size_t stride = sizeof(e2s_cand_header) + (size_t)4 * cap * 4;
    std::vector<unsigned char> fixed(nc * stride, 0), packed; std::vector<uint32_t> off8;
    size_t at = 0;
    for (uint64_t c = 0; c < nc; ++c) {
        const uint64_t n0 = w[at], n1 = w[at + 1], ri = w[at + 2], rp = w[at + 3];
        const uint64_t* i0 = &w[at + 4]; const uint64_t* p0 = i0 + n0; const uint64_t* i1 = p0 + n0; const uint64_t* p1 = i1 + n1;
        at += 4 + 2 * n0 + 2 * n1;
        e2s_cand_header hd{}; hd.record_index = c; hd.n0 = (uint32_t)n0; hd.n1 = (uint32_t)n1; hd.right_idx = (uint32_t)ri; hd.right_pos = (uint32_t)rp;
        unsigned char* o = fixed.data() + c * stride;
        memcpy(o, &hd, sizeof(hd));
        uint32_t* l = (uint32_t*)(o + sizeof(hd));
        for (uint64_t j = 0; j < n0; ++j) { l[j] = (uint32_t)i0[j]; l[cap + j] = (uint32_t)p0[j]; }
        for (uint64_t j = 0; j < n1; ++j) { l[2 * cap + j] = (uint32_t)i1[j]; l[3 * cap + j] = (uint32_t)p1[j]; }
        // the packed stream, 4-byte positions
        off8.push_back((uint32_t)(packed.size() / 8));
        const uint32_t h4[4] = {(uint32_t)n0, (uint32_t)n1, (uint32_t)ri, (uint32_t)rp};
        packed.insert(packed.end(), (const unsigned char*)h4, (const unsigned char*)h4 + 16);
        auto put = [&](const uint64_t* v, uint64_t n) { for (uint64_t j = 0; j < n; ++j) { const uint32_t x = (uint32_t)v[j]; packed.insert(packed.end(), (const unsigned char*)&x, (const unsigned char*)&x + 4); } };
        put(i0, n0); put(i1, n1); put(p0, n0); put(p1, n1);
        while (packed.size() % 8) packed.push_back(0);
    }
    off8.push_back((uint32_t)(packed.size() / 8));
    e2s_tail_params tp{co.k_left ? co.k_left : 31, co.k_right ? co.k_right : 30, co.max_gap ? co.max_gap : 10, co.max_err ? co.max_err : 2};
    const std::vector<unsigned char> want = slurp(want_path);
    uint64_t nv = 0, ne = 0;
    const std::string o1 = in + ".host_fixed.snp", o2 = in + ".host_packed.snp";
    int rc = e2s_snp_from_candidates(fixed.data(), nullptr, nc, cap, &tp, in.c_str(), o1.c_str(), &nv, &ne);
    if (rc || slurp(o1) != want) { fprintf(stderr, "host tail (fixed stride): rc %d, bytes %s\n", rc, slurp(o1) == want ? "equal" : "DIFFER"); return 4; }
    rc = e2s_snp_from_packed(packed.data(), off8.data(), nc, 4, &tp, in.c_str(), o2.c_str(), &nv, &ne);
    if (rc || slurp(o2) != want) { fprintf(stderr, "host tail (packed): rc %d\n", rc); return 4; }
    printf("ok: %llu clusters printed, %llu candidates, %llu events, %zu bytes of .snp, host tail identical to the oracle\n", (unsigned long long)nprinted,
           (unsigned long long)nc, (unsigned long long)ne, want.size());
    return 0;
}
