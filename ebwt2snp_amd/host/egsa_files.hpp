// egsa_files.hpp -- file side of the drop-in executables: locate and map the EGSA index files the way
// egsa_stream's constructor does (include.hpp:42-81): <input>.gesa first, else the BCR triple
// <input>.out / .out.lcp / .out.pairSA, else "Error: missing index files." and exit(1).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <string>

#include "e2s_hip.h"

struct MappedFile {
    const uint8_t* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = (const uint8_t*)""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const uint8_t*)m;
        return true;
    }
    ~MappedFile() { if (p && n) munmap((void*)p, n); if (fd >= 0) ::close(fd); }
};

struct EgsaFiles {
    bool gesa = false;
    MappedFile aos, bwt, lcp, pairsa;
    uint64_t n = 0;
    // exits like the reference when nothing is found.  The reference derives everything from the LCP stream's EOF
    // (include.hpp:92-107) and keeps reading stale values when another file is shorter; here a width outside 1/2/4/8 or a
    // file too short for the entry count is refused before anything is mapped onto the GPU.
    void open_or_die(const std::string& input, int x, int y, int z) {
        auto width_ok = [](int w) { return w == 1 || w == 2 || w == 4 || w == 8; };
        if (!width_ok(x) || !width_ok(y) || !width_ok(z)) {
            std::cerr << "byte widths (-x -y -z) must be 1, 2, 4 or 8 (the reference leaves fields of other widths uninitialised)" << std::endl;
            exit(2);
        }
        if (aos.open(input + ".gesa")) {
            gesa = true;
            n = aos.n / (size_t)(x + y + z + 1);
            return;
        }
        if (lcp.open(input + ".out.lcp") && bwt.open(input + ".out") && pairsa.open(input + ".out.pairSA")) {
            n = lcp.n / (size_t)x;
            if (bwt.n < n || pairsa.n < n * (size_t)(y + z)) {
                std::cerr << "index files are inconsistent: " << n << " LCP entries of " << x << " bytes, but " << bwt.n << " BWT bytes and "
                          << pairsa.n << " pairSA bytes (" << (y + z) << " per entry): wrong -x/-y/-z or a truncated file" << std::endl;
                exit(2);
            }
            return;
        }
        std::cout << "Error: missing index files." << std::endl;
        exit(1);
    }
};

// the mapped files as a chunk source of the library's chunked jobs (e2s_egsa_source): pointers into the mappings, nothing copied
struct FileSource {
    const EgsaFiles* F; int x, y, z;
    static int get(void* user, int, uint64_t lo, uint64_t, int, const void** bwt, const void** lcp, const void** pairsa, const void** aos) {
        const FileSource* s = (const FileSource*)user;
        if (s->F->gesa) { *aos = s->F->aos.p + lo * (uint64_t)(s->x + s->y + s->z + 1); return 0; }
        *bwt = s->F->bwt.p + lo; *lcp = s->F->lcp.p + lo * (uint64_t)s->x; *pairsa = s->F->pairsa.p + lo * (uint64_t)(s->y + s->z);
        return 0;
    }
    e2s_egsa_source source() const { e2s_egsa_source r{}; r.get = &FileSource::get; r.release = nullptr; r.user = (void*)this; r.gesa = F->gesa ? 1 : 0; r.flags = 1; return r; }
};
// E2S_CHUNK_ENTRIES: SA positions per chunk (default: the library's, 1.5x10^9); inputs that fit one chunk go through in one piece
inline uint64_t chunk_entries_env() { const char* e = getenv("E2S_CHUNK_ENTRIES"); return e ? strtoull(e, nullptr, 10) : 0ull; }

// E2S_CLI_TIMING=1: wall time of the executable's phases on stderr
struct PhaseClock {
    bool on = getenv("E2S_CLI_TIMING") != nullptr;
    double t0 = now();
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    void mark(const char* what) {
        if (!on) return;
        const double t = now();
        std::cerr << "[timing] " << what << " " << (t - t0) << " ms" << std::endl;
        t0 = t;
    }
};

// E2S_CLI_JSON=<path>: a perf report of the run as one JSON object in a side file (stdout stays the reference's, byte for byte):
// entries, wall seconds since process start-up code ran, the job's own seconds and the part spent uploading, rates, counts
struct CliReport {
    double t_start = PhaseClock::now();
    void write(const char* tool, const std::string& input, const e2s_job_report& rep, uint64_t entries, uint64_t bytes_uploaded, uint64_t bytes_out) const {
        const char* path = getenv("E2S_CLI_JSON");
        if (!path || !*path) return;
        FILE* f = fopen(path, "w");
        if (!f) return;
        const double wall = (PhaseClock::now() - t_start) * 1e-3;
        fprintf(f, "{\"tool\": \"%s\", \"library\": \"%s\", \"entries\": %llu, \"seconds_wall\": %.6f, \"seconds_job\": %.6f, \"seconds_upload\": %.6f, "
                   "\"entries_per_s\": %.6g, \"bytes_uploaded\": %llu, \"upload_GBps\": %.4g, \"bytes_out\": %llu, \"chunks\": %d, "
                   "\"clusters_printed\": %llu, \"clusters_file_records\": %llu, \"max_clust_length\": %d, \"candidates\": %llu, \"variants\": %llu, \"events\": %llu, \"input\": \"",
                tool, e2s_version(), (unsigned long long)entries, wall, rep.seconds_total, rep.seconds_upload, wall > 0 ? (double)entries / wall : 0.0,
                (unsigned long long)bytes_uploaded, rep.seconds_upload > 0 ? (double)bytes_uploaded / rep.seconds_upload / 1e9 : 0.0, (unsigned long long)bytes_out, (int)rep.n_chunks,
                (unsigned long long)rep.n_closed, (unsigned long long)rep.n_records, (int)rep.max_clust_length, (unsigned long long)rep.n_candidates,
                (unsigned long long)rep.n_variants, (unsigned long long)rep.n_events);
        for (char c : input) { if (c == '"' || c == '\\') fputc('\\', f); if ((unsigned char)c >= 0x20) fputc(c, f); }
        fprintf(f, "\"}\n");
        fclose(f);
    }
};

inline void die_on(int rc, e2s_handle h, const char* what) {
    if (rc == E2S_OK) return;
    std::cerr << what << " failed (" << rc << "): " << e2s_last_error(h) << std::endl;
    exit(2);
}
