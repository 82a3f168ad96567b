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
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/e2s_hip.h"
#include "fasta_index.hpp"
#include "../csrc/e2s_env.h"

struct e2s_fasta_index {                                          // the mapped file and its index (e2s_fasta_index_open)
    int fd = -1; void* map = nullptr; size_t n = 0;
    FastaIndex ix;
    FastaJoined jx;                                               // (a file of the plain shape: indexed chunk by chunk in ONE pass over the mapping)
    ~e2s_fasta_index() { if (map) munmap(map, n); if (fd >= 0) ::close(fd); }
};
// One pass over the mapping for a file of the plain shape (header, one line of DNA, ...): chunks of 8 MB handed to the threads as they come
// free, every chunk scanned once (e2s_fasta_scan_chunk) -- the two-pass index (count, then fill) touched every page twice.  false: another shape.
static bool index_plain_one_pass(const uint8_t* p, size_t n, unsigned T, FastaJoined& out) {
    const size_t CH = (size_t)8 << 20;
    std::vector<FastaChunk> chunks((n + CH - 1) / CH);
    std::atomic<size_t> next{0};
    std::atomic<bool> plain{true};
    auto work = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= chunks.size() || !plain.load(std::memory_order_relaxed)) return;
            e2s_fasta_scan_chunk(p, 0, n, (uint64_t)c * CH, std::min<uint64_t>((uint64_t)(c + 1) * CH, n), n, chunks[c]);
            if (!chunks[c].plain) plain.store(false, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    return plain.load() && e2s_fasta_join_chunks(chunks, out);
}
extern "C" void e2s_internal_set_error(e2s_handle h, const char* msg);

extern "C" int e2s_fasta_index_open(const char* fasta_path, e2s_fasta_index** out, const void** text, uint64_t* text_bytes, const uint64_t** start,
                                    const uint32_t** len, uint64_t* n_reads, uint32_t* max_len) {
    if (!fasta_path || !out) return E2S_ERR_ARG;
    *out = nullptr;
    e2s_fasta_index* f = new e2s_fasta_index();
    f->fd = ::open(fasta_path, O_RDONLY);
    if (f->fd < 0) { delete f; return E2S_ERR_IO; }
    struct stat st{};
    if (fstat(f->fd, &st) != 0) { delete f; return E2S_ERR_IO; }
    if (st.st_size == 0) { delete f; return E2S_ERR_INPUT; }      // (an empty file is an input error, not an I/O one)
    f->n = (size_t)st.st_size;
    f->map = mmap(nullptr, f->n, PROT_READ, MAP_PRIVATE, f->fd, 0);
    if (f->map == MAP_FAILED) { f->map = nullptr; delete f; return E2S_ERR_IO; }
    madvise(f->map, f->n, MADV_SEQUENTIAL);
    const int env = getenv("E2S_TAIL_THREADS") ? atoi(getenv("E2S_TAIL_THREADS")) : 0;
    const unsigned T = env > 0 ? (unsigned)env : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (((const uint8_t*)f->map)[0] == '>' && !getenv("E2S_FASTA_TWO_PASS") && index_plain_one_pass((const uint8_t*)f->map, f->n, T, f->jx)) {
        if (text) *text = f->map;
        if (text_bytes) *text_bytes = f->n;
        if (start) *start = f->jx.start.get();
        if (len) *len = f->jx.len.get();
        if (n_reads) *n_reads = f->jx.n;
        if (max_len) *max_len = f->jx.max_len;
        *out = f;
        return E2S_OK;
    }
    if (!e2s_index_fasta((const uint8_t*)f->map, f->n, T, f->ix)) { delete f; return E2S_ERR_INPUT; }      // irregular line structure: the host tail decides
    if (text) *text = f->ix.text;
    if (text_bytes) *text_bytes = f->ix.text_bytes;
    if (start) *start = f->ix.start.data();
    if (len) *len = f->ix.len.data();
    if (n_reads) *n_reads = f->ix.start.size();
    if (max_len) *max_len = f->ix.max_len;
    *out = f;
    return E2S_OK;
}
extern "C" void e2s_fasta_index_close(e2s_fasta_index* ix) { delete ix; }

extern "C" int e2s_internal_upload_reads_fd(e2s_handle h, int fd, uint64_t file_bytes, uint64_t* n_reads);
extern "C" int e2s_upload_reads_fasta(e2s_handle h, const char* fasta_path, uint64_t* n_reads) {
    if (!h || !fasta_path) return E2S_ERR_ARG;
    if (n_reads) *n_reads = 0;
    // a file of the plain shape (header, one line of DNA, ...) streams to the GPU in one pass, indexed on the way (no mapping); E2S_READS_MAPPED=1
    // keeps to the mapping.  Anything else -- and any file the streamed pass hands back -- is mapped and indexed whole, as before.
    if (!getenv("E2S_READS_MAPPED")) {
        const int fd = ::open(fasta_path, O_RDONLY);
        if (fd < 0) return E2S_ERR_IO;
        struct stat st{};
        if (fstat(fd, &st) != 0) { ::close(fd); return E2S_ERR_IO; }
        if (st.st_size > 0) {
            uint64_t nr = 0;
            const int rc = e2s_internal_upload_reads_fd(h, fd, (uint64_t)st.st_size, &nr);
            ::close(fd);
            if (rc != 1) { if (rc == E2S_OK && n_reads) *n_reads = nr; return rc; }
            // (development tier, the tests: a file the streamed pass hands back is an error instead of being mapped -- shows which path ran)
            if (e2s_env_dev("E2S_READS_STREAM_ONLY")) { e2s_internal_set_error(h, "the streamed pass handed the FASTA back (not of the plain shape)"); return E2S_ERR_STATE; }
        } else ::close(fd);
    }
    e2s_fasta_index* f = nullptr;
    const void* text = nullptr; uint64_t tb = 0, nr = 0; const uint64_t* st = nullptr; const uint32_t* ln = nullptr; uint32_t ml = 0;
    int rc = e2s_fasta_index_open(fasta_path, &f, &text, &tb, &st, &ln, &nr, &ml);
    if (rc) return rc;
    rc = e2s_upload_reads(h, text, tb, st, ln, nr, ml);
    if (rc == E2S_OK && n_reads) *n_reads = nr;
    e2s_fasta_index_close(f);
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
    rc = e2s_snp_from_packed(stream, off8, n_cand, pw, p, fasta_path, out_path, n_variants, n_events);
    // (the host tail has no handle and no text: the handle's message says what ITS code means, not what some earlier call left there)
    if (rc == E2S_ERR_REF_CRASH) e2s_internal_set_error(h, "host tail: the reference crashes on this input (no candidate, one read rank in all, a negative -L or a right context beyond its read); nothing written");
    else if (rc == E2S_ERR_IO) e2s_internal_set_error(h, "host tail: the FASTA cannot be read or the output cannot be written");
    else if (rc) e2s_internal_set_error(h, "host tail failed");
    return rc;
}
