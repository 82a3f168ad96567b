// reads_upload.cpp -- host side of the GPU tail: the FASTA indexed (fasta_index.cpp) and made resident in HBM, and the one call
// the clust2snp executable makes after e2s_find_candidates: GPU tail where it applies, host tail (snp_tail.cpp) otherwise.
// Replaces get_reads (clust2snp.cpp:147-212) and the call sequence extract_variants -> to_file (clust2snp.cpp:864-868).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../include/e2s_hip.h"
#include "fasta_index.hpp"

extern "C" int e2s_upload_reads_fasta(e2s_handle h, const char* fasta_path, uint64_t* n_reads) {
    if (!h || !fasta_path) return E2S_ERR_ARG;
    if (n_reads) *n_reads = 0;
    const int fd = ::open(fasta_path, O_RDONLY);
    if (fd < 0) return E2S_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) { ::close(fd); return st.st_size == 0 ? E2S_ERR_INPUT : E2S_ERR_IO; }
    const size_t n = (size_t)st.st_size;
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { ::close(fd); return E2S_ERR_IO; }
    madvise(m, n, MADV_SEQUENTIAL);
    const int env = getenv("E2S_TAIL_THREADS") ? atoi(getenv("E2S_TAIL_THREADS")) : 0;
    const unsigned T = env > 0 ? (unsigned)env : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    FastaIndex ix;
    int rc = E2S_ERR_INPUT;                                      // irregular line structure: the host tail decides
    if (e2s_index_fasta((const uint8_t*)m, n, T, ix))
        rc = e2s_upload_reads(h, ix.text, ix.text_bytes, ix.start.data(), ix.len.data(), ix.start.size(), ix.max_len);
    if (rc == E2S_OK && n_reads) *n_reads = ix.start.size();
    munmap(m, n);
    ::close(fd);
    return rc;
}

extern "C" int e2s_write_snp(e2s_handle h, const e2s_tail_params* p, const char* fasta_path, const char* out_path, uint64_t* n_variants,
                             uint64_t* n_events) {
    if (!h || !p || !fasta_path || !out_path) return E2S_ERR_ARG;
    uint64_t nv = 0, ne = 0, nb = 0;
    int rc = e2s_call_variants(h, p, 1, &nv, &ne, &nb);
    if (rc == E2S_OK) {
        FILE* f = fopen(out_path, "w");
        if (!f) return E2S_ERR_IO;
        auto sink = [](void* user, const void* data, size_t bytes) -> int { return fwrite(data, 1, bytes, (FILE*)user) == bytes ? 0 : 1; };
        rc = e2s_stream_snp(h, sink, f);
        fclose(f);
        if (n_variants) *n_variants = nv;
        if (n_events) *n_events = ne;
        return rc;
    }
    if (rc != E2S_ERR_STATE) return rc;                          // (E2S_ERR_REF_CRASH included: nothing is written)
    // the host tail over the packed candidates: reads not resident, or an input only the sequential restatement reproduces
    const void* stream = nullptr; const uint32_t* off8 = nullptr; uint64_t n_cand = 0, bytes = 0; uint32_t pw = 4;
    rc = e2s_candidates_packed(h, &stream, &off8, &n_cand, &bytes, &pw);
    if (rc) return rc;
    return e2s_snp_from_packed(stream, off8, n_cand, pw, p, fasta_path, out_path, n_variants, n_events);
}
