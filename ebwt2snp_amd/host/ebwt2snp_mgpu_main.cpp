// ebwt2snp_mgpu -- both tools over the GPUs of one node, host code in C++ over the C ABI, the boundary stitch over RCCL.
//
//   ebwt2snp_mgpu -i <fasta> -n <reads in sample 1> [-G <GPUs>] [ebwt2clust: -k -m] [clust2snp: -L -R -g -c -e -M(cov) -p] [-x -y -z]
//
// What it replaces: running `ebwt2clust -i F` and then `clust2snp -i F -n N` (ebwt2clust.cpp:141-198, clust2snp.cpp:969-1085;
// pipeline.sh:115,122); it writes the same two files, <F>.clusters and <F minus .fast*>.snp, byte for byte.
// One thread per GPU.  Rank r owns the r-th contiguous SA range (left halo 152, right halo 1) and keeps it -- and all reads --
// resident in HBM.  After the scan every rank publishes its 88-byte boundary descriptor and the 1.2-KB statistics the scan left
// with ONE ncclAllGather (RCCL; xGMI between the GPUs); stitch plans and the statistics step are pure host functions of the
// gathered array (e2s_stitch, e2s_auto_max_clust_length_ranks), so every rank derives the same max_clust_length.  The `.snp`
// ids are sequential over the ranks: a second, 8-byte all-gather tells a rank how many events the ranks before it hold.
// The files are written rank by rank.  -G 1 runs the same code with a one-rank communicator.
// E2S_MGPU_SHARED_DEVICE=1 puts every rank on GPU 0 and exchanges through host memory (RCCL refuses two ranks on one device):
// a rehearsal of the N-rank logic on a one-GPU box, not a way to run jobs.
#include <fcntl.h>
#include <rccl/rccl.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "egsa_files.hpp"

namespace {

struct Opts {
    std::string input; uint64_t nr_reads1 = 0; int gpus = 1;
    int k = 16, min_len = 2, lcp = 1, da = 4, pos = 1;
    int k_left = 31, k_right = 30, max_gap = 10, consensus_reads = 20, max_err = 2, mcov_out = 5; double pval = 0.99;
};

// a barrier for the rehearsal mode and the turn-taking of the writers
class Barrier {
  public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        const int gen = gen_;
        if (++cnt_ == n_) { cnt_ = 0; ++gen_; cv_.notify_all(); } else cv_.wait(lk, [&]() { return gen_ != gen; });
    }
  private:
    int n_, cnt_ = 0, gen_ = 0; std::mutex m_; std::condition_variable cv_;
};
class Turn {
  public:
    void wait_for(int r) { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&]() { return turn_ == r || fail_; }); }
    void next() { { std::lock_guard<std::mutex> lk(m_); ++turn_; } cv_.notify_all(); }
    void abort() { { std::lock_guard<std::mutex> lk(m_); fail_ = true; } cv_.notify_all(); }
  private:
    int turn_ = 0; bool fail_ = false; std::mutex m_; std::condition_variable cv_;
};

struct Shared {
    Opts o; EgsaFiles* F; int nranks; bool shared_device;
    std::vector<ncclComm_t> comms;
    std::vector<uint8_t> host_blob;           // rehearsal: the exchange through host memory
    std::vector<uint64_t> host_events;
    Barrier* bar; Turn cl_turn, snp_turn;
    int cl_fd = -1; FILE* snp_f = nullptr; std::string snp_path;
    std::atomic<int> rc{0};
    std::vector<uint64_t> closed, cands, events; int mcl = 0;
    // where only the host tail reproduces the reference (a read rank past the last read, -L outside 1..64, an irregular FASTA): every
    // rank appends its packed candidates in rank order and ONE e2s_snp_from_packed runs at the end
    bool host_tail = false; std::vector<uint8_t> pk_stream; std::vector<uint32_t> pk_off8; uint32_t pk_pw = 4;
    std::mutex log_m;
};

constexpr size_t BLOB = sizeof(e2s_boundary) + sizeof(e2s_scan_stats);

int fd_sink(void* user, const void* data, size_t bytes) {
    const int fd = *(int*)user;
    const char* p = (const char*)data;
    while (bytes) { const ssize_t w = ::write(fd, p, bytes); if (w <= 0) return 1; p += w; bytes -= (size_t)w; }
    return 0;
}
int file_sink(void* user, const void* data, size_t bytes) { return fwrite(data, 1, bytes, (FILE*)user) == bytes ? 0 : 1; }

#define STEP(call, what)                                                                                                        \
    do { const int rc_ = (call); if (rc_) { std::lock_guard<std::mutex> lk(S.log_m);                                               \
         std::cerr << "rank " << r << ": " << what << " failed (" << rc_ << "): " << (h ? e2s_last_error(h) : "") << std::endl;     \
         S.rc.store(rc_); } } while (0)

// ONE all-gather of `bytes` per rank: RCCL on the rank's stream, or host memory + a barrier in the rehearsal mode
int all_gather(Shared& S, int r, const void* mine, size_t bytes, std::vector<uint8_t>& all, void* d_send, void* d_recv, hipStream_t st) {
    all.resize(bytes * (size_t)S.nranks);
    if (S.shared_device) {
        { std::lock_guard<std::mutex> lk(S.log_m); if (S.host_blob.size() < all.size()) S.host_blob.resize(all.size()); }
        S.bar->wait();
        memcpy(S.host_blob.data() + bytes * (size_t)r, mine, bytes);
        S.bar->wait();
        memcpy(all.data(), S.host_blob.data(), all.size());
        S.bar->wait();
        return 0;
    }
    if (hipMemcpyAsync(d_send, mine, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    if (ncclAllGather(d_send, d_recv, bytes, ncclChar, S.comms[(size_t)r], st) != ncclSuccess) return 2;
    if (hipMemcpyAsync(all.data(), d_recv, all.size(), hipMemcpyDeviceToHost, st) != hipSuccess) return 3;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : 4;
}

void rank_main(Shared& S, int r) {
    const Opts& o = S.o;
    const int dev = S.shared_device ? 0 : r;
    e2s_handle h = nullptr;
    hipStream_t xs = nullptr; void *d_send = nullptr, *d_recv = nullptr;
    const uint64_t n = S.F->n;
    // contiguous SA ranges of (almost) equal size; the cuts may fall anywhere, the stitch takes care of clusters that cross them
    const uint64_t lo = n * (uint64_t)r / (uint64_t)S.nranks, hi = n * (uint64_t)(r + 1) / (uint64_t)S.nranks;
    const uint32_t hl = (uint32_t)std::min<uint64_t>(lo, E2S_MAX_CLUST_LEN + 2), hr = (uint32_t)std::min<uint64_t>(n - hi, 1);
    STEP(e2s_create(dev, &h), "e2s_create");
    if (!S.rc.load() && !S.shared_device) {
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&d_send, BLOB) != hipSuccess || hipMalloc(&d_recv, BLOB * (size_t)S.nranks) != hipSuccess) S.rc.store(E2S_ERR_HIP);
    }
    if (!S.rc.load()) {
        const int rc = e2s_upload_reads_fasta(h, o.input.c_str(), nullptr);     // every rank holds all reads (candidates name any of them)
        if (rc != E2S_OK && rc != E2S_ERR_INPUT && rc != E2S_ERR_IO) STEP(rc, "upload reads");
    }
    if (!S.rc.load()) {
        e2s_shard_desc d{};
        d.global_offset = lo; d.n_entries = hi - lo; d.n_total = n; d.halo_left = hl; d.halo_right = hr;
        d.lcp_bytes = o.lcp; d.da_bytes = o.da; d.pos_bytes = o.pos;
        const uint64_t p0 = lo - hl;
        if (S.F->gesa) STEP(e2s_upload_gesa(h, &d, S.F->aos.p + p0 * (uint64_t)(o.lcp + o.da + o.pos + 1)), "upload (.gesa)");
        else STEP(e2s_upload_bcr(h, &d, S.F->bwt.p + p0, S.F->lcp.p + p0 * (uint64_t)o.lcp, S.F->pairsa.p + p0 * (uint64_t)(o.da + o.pos)), "upload (BCR)");
    }
    // ---- ebwt2clust: scan, ONE all-gather, stitch ----
    std::vector<uint8_t> mine(BLOB, 0), all;
    if (!S.rc.load()) {
        STEP(e2s_prefilter_arm(h, o.mcov_out), "arm");
        STEP(e2s_cluster_scan(h, o.k, o.min_len, 0), "cluster scan");
        if (!S.rc.load()) {
            STEP(e2s_boundary_desc(h, (e2s_boundary*)mine.data()), "boundary");
            STEP(e2s_scan_stats_desc(h, (e2s_scan_stats*)(mine.data() + sizeof(e2s_boundary))), "scan statistics");
        }
    }
    if (all_gather(S, r, mine.data(), BLOB, all, d_send, d_recv, xs)) S.rc.store(E2S_ERR_HIP);      // (every rank takes part, failed or not)
    std::vector<e2s_boundary> allb((size_t)S.nranks); std::vector<e2s_scan_stats> alls((size_t)S.nranks);
    for (int q = 0; q < S.nranks; ++q) {
        memcpy(&allb[(size_t)q], all.data() + BLOB * (size_t)q, sizeof(e2s_boundary));
        memcpy(&alls[(size_t)q], all.data() + BLOB * (size_t)q + sizeof(e2s_boundary), sizeof(e2s_scan_stats));
    }
    int mcl = 0; uint64_t n_cand = 0, nv = 0, ne = 0, nb = 0, need_host = 0;
    if (!S.rc.load()) {
        STEP(e2s_stitch(h, allb.data(), S.nranks, r), "stitch");
        uint64_t nc = 0, nrec = 0;
        STEP(e2s_cluster_counts(h, &nc, &nrec), "counts");
        S.closed[(size_t)r] = nc;
        STEP(e2s_pack_clusters(h, 1), "pack .clusters");           // on the second stream, beside the per-cluster pass
        // ---- clust2snp: statistics (pure function of the gathered array), per-cluster pass, tail ----
        uint64_t n_clust = 0, n_bases = 0;
        const int rc = e2s_auto_max_clust_length_ranks(allb.data(), alls.data(), S.nranks, n, o.k, o.lcp, o.mcov_out, o.pval, &mcl, &n_clust, &n_bases);
        if (rc) S.rc.store(rc == E2S_ERR_REF_CRASH ? 136 : rc);
        if (r == 0) S.mcl = mcl;
    }
    if (!S.rc.load()) {
        e2s_c2s_params prm{};
        prm.nr_reads1 = o.nr_reads1; prm.k_left = o.k_left; prm.k_right = o.k_right; prm.mcov_out = o.mcov_out;
        prm.consensus_reads = o.consensus_reads; prm.max_clust_length = mcl; prm.min_len = o.min_len;
        STEP(e2s_find_candidates(h, &prm, (uint64_t)r << 40, &n_cand), "find candidates");
        S.cands[(size_t)r] = n_cand;
        if (!S.rc.load() && n_cand) {
            e2s_tail_params tp{o.k_left, o.k_right, o.max_gap, o.max_err};
            const int trc = e2s_call_variants(h, &tp, 0, &nv, &ne, &nb);   // (first_id 0: the text is written once, after the exchange below)
            if (trc == E2S_ERR_STATE) { need_host = 1; ne = 0; } else STEP(trc, "tail (consensus, distance, text)");
        }
    }
    // the second exchange: events per rank, for sequential ids
    std::vector<uint8_t> alle;
    const uint64_t mine2[2] = {ne, need_host};
    if (all_gather(S, r, mine2, 16, alle, d_send, d_recv, xs)) S.rc.store(E2S_ERR_HIP);
    uint64_t before = 0; bool host_tail = false;
    for (int q = 0; q < S.nranks; ++q) {
        uint64_t v[2]; memcpy(v, alle.data() + 16 * (size_t)q, 16);
        if (q < r) before += v[0];
        host_tail |= v[1] != 0;
    }
    S.events[(size_t)r] = host_tail ? 0 : ne;
    if (!S.rc.load() && !host_tail && n_cand) STEP(e2s_renumber_snp(h, 1 + before, &nb), "ids and text");
    // ---- the files, rank by rank ----
    S.cl_turn.wait_for(r);
    if (!S.rc.load()) { uint64_t w = 0; STEP(e2s_stream_clusters(h, fd_sink, &S.cl_fd, &w), "write .clusters"); }
    S.cl_turn.next();
    S.snp_turn.wait_for(r);
    if (!S.rc.load() && host_tail) {
        S.host_tail = true;
        const void* st = nullptr; const uint32_t* of = nullptr; uint64_t nc2 = 0, bytes = 0; uint32_t pw = 4;
        if (n_cand) STEP(e2s_candidates_packed(h, &st, &of, &nc2, &bytes, &pw), "packed candidates");
        if (!S.rc.load() && n_cand) {
            const uint64_t base8 = S.pk_stream.size() / 8;
            if (base8 + bytes / 8 >= (1ull << 32)) S.rc.store(E2S_ERR_CAPACITY);
            else {
                S.pk_stream.insert(S.pk_stream.end(), (const uint8_t*)st, (const uint8_t*)st + bytes);
                for (uint64_t i = 0; i < nc2; ++i) S.pk_off8.push_back((uint32_t)(base8 + of[i]));
                S.pk_pw = pw;
            }
        }
    } else if (!S.rc.load() && ne) {
        if (!S.snp_f) S.snp_f = fopen(S.snp_path.c_str(), "w");   // (comes into being with the first event: nothing where the reference crashes)
        if (!S.snp_f) S.rc.store(E2S_ERR_IO); else STEP(e2s_stream_snp(h, file_sink, S.snp_f), "write .snp");
    }
    S.snp_turn.next();
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (xs) (void)hipStreamDestroy(xs);
    if (h) e2s_destroy(h);
}

void help() {
    std::cout << "ebwt2snp_mgpu [options]\nebwt2clust + clust2snp over the GPUs of one node (SA-range shards, RCCL boundary stitch). Options:\n"
                 "-i <arg>   Input fasta file (REQUIRED); its EGSA (<input>.gesa or the BCR triple) must exist next to it\n"
                 "-n <arg>   Number of reads in the first sample (REQUIRED)\n"
                 "-G <arg>   Number of GPUs (default: 1)\n"
                 "-k -m      ebwt2clust: minimum LCP in clusters (16), discard clusters smaller than this (2)\n"
                 "-L -R -g -c -e -p   clust2snp: as there (31, 30, 10, 20, 2, 0.99);  -M <arg>: clust2snp's -m, minimum cluster length per individual (5)\n"
                 "-x -y -z   byte sizes of LCP / DA / pos integers in the EGSA (1, 4, 1)\n"
                 "Output: <input>.clusters and <input without .fast*>.snp, as the two tools write them.\n";
    exit(0);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) help();
    Opts o;
    int opt;
    while ((opt = getopt(argc, argv, "hi:n:G:k:m:L:R:g:c:e:p:M:x:y:z:")) != -1) {
        switch (opt) {
            case 'h': help(); break;
            case 'i': o.input = optarg; break;
            case 'n': o.nr_reads1 = (uint64_t)atoi(optarg); break;
            case 'G': o.gpus = atoi(optarg); break;
            case 'k': if (atoi(optarg)) o.k = atoi(optarg); break;
            case 'm': if (atoi(optarg)) o.min_len = atoi(optarg); break;
            case 'L': if (atoi(optarg)) o.k_left = atoi(optarg); break;
            case 'R': if (atoi(optarg)) o.k_right = atoi(optarg); break;
            case 'g': if (atoi(optarg)) o.max_gap = atoi(optarg); break;
            case 'c': if (atoi(optarg)) o.consensus_reads = atoi(optarg); break;
            case 'e': if ((int)atof(optarg)) o.max_err = (int)atof(optarg); break;
            case 'p': if (atof(optarg) != 0) o.pval = atof(optarg); break;
            case 'M': if (atoi(optarg)) o.mcov_out = atoi(optarg); break;
            case 'x': if (atoi(optarg)) o.lcp = atoi(optarg); break;
            case 'y': if (atoi(optarg)) o.da = atoi(optarg); break;
            case 'z': if (atoi(optarg)) o.pos = atoi(optarg); break;
            default: help(); return -1;
        }
    }
    if (o.input.empty() || o.nr_reads1 == 0 || o.gpus < 1) help();
    EgsaFiles F;
    F.open_or_die(o.input, o.lcp, o.da, o.pos);
    Shared S;
    S.o = o; S.F = &F; S.nranks = o.gpus;
    S.shared_device = getenv("E2S_MGPU_SHARED_DEVICE") != nullptr && atoi(getenv("E2S_MGPU_SHARED_DEVICE")) != 0;
    if ((uint64_t)S.nranks * 4 > F.n) { std::cerr << "too few entries for " << S.nranks << " ranks" << std::endl; return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::cerr << "GPU unavailable (this build has no CPU path)" << std::endl; return 2; }
    if (!S.shared_device && S.nranks > ndev) { std::cerr << "-G " << S.nranks << " but only " << ndev << " GPU(s) visible" << std::endl; return 2; }
    if (!S.shared_device) {
        // one communicator per GPU, all in this process (ncclCommInitAll), each driven by its own thread
        std::vector<int> devs((size_t)S.nranks);
        for (int r = 0; r < S.nranks; ++r) devs[(size_t)r] = r;
        S.comms.resize((size_t)S.nranks);
        const ncclResult_t nr = ncclCommInitAll(S.comms.data(), S.nranks, devs.data());
        if (nr != ncclSuccess) { std::cerr << "ncclCommInitAll failed: " << ncclGetErrorString(nr) << std::endl; return 2; }
    }
    Barrier bar(S.nranks);
    S.bar = &bar;
    S.closed.assign((size_t)S.nranks, 0); S.cands.assign((size_t)S.nranks, 0); S.events.assign((size_t)S.nranks, 0);
    const std::string cl_path = o.input + ".clusters";
    S.cl_fd = ::open(cl_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    S.snp_path = o.input.substr(0, o.input.rfind(".fast")) + ".snp";
    std::cout << "This is ebwt2snp_mgpu. Input file: " << o.input << ", " << F.n << " EGSA entries over " << S.nranks << " GPU(s)"
              << (S.shared_device ? " [rehearsal: all ranks on GPU 0, exchange through host memory]" : " [RCCL all-gather]") << std::endl;
    std::vector<std::thread> th;
    for (int r = 1; r < S.nranks; ++r) th.emplace_back(rank_main, std::ref(S), r);
    rank_main(S, 0);
    for (auto& t : th) t.join();
    if (S.cl_fd >= 0) ::close(S.cl_fd);
    if (S.snp_f) fclose(S.snp_f);
    for (auto& c : S.comms) ncclCommDestroy(c);
    uint64_t closed = 0, cands = 0, events = 0;
    for (int r = 0; r < S.nranks; ++r) { closed += S.closed[(size_t)r]; cands += S.cands[(size_t)r]; events += S.events[(size_t)r]; }
    if (!S.rc.load() && S.host_tail && !S.pk_off8.empty()) {
        S.pk_off8.push_back((uint32_t)(S.pk_stream.size() / 8));
        e2s_tail_params tp{o.k_left, o.k_right, o.max_gap, o.max_err};
        uint64_t nv = 0;
        const int trc = e2s_snp_from_packed(S.pk_stream.data(), S.pk_off8.data(), S.pk_off8.size() - 1, S.pk_pw, &tp, o.input.c_str(), S.snp_path.c_str(), &nv, &events);
        if (trc) { std::cerr << "host tail failed (" << trc << ")" << std::endl; S.rc.store(trc == E2S_ERR_REF_CRASH ? 139 : trc); }
    }
    const int rc = S.rc.load();
    if (rc == 139) { std::cerr << "the reference crashes on this input in its tail; no .snp written" << std::endl; return 139; }
    if (rc == 136) { std::cerr << "no usable cluster record: the reference divides by zero here (SIGFPE)" << std::endl; return 136; }
    if (rc) return 2;
    std::cout << "Done. " << (unsigned int)closed << " clusters saved to output file." << std::endl;
    std::cout << "Cluster sizes allowed: [" << o.mcov_out * 2 << "," << S.mcl << "]" << std::endl;
    std::cout << "Done. " << cands << " potential variants detected (some might be detected twice: on fw and rev strands)" << std::endl;
    if (cands == 0) { std::cerr << "no candidate variant: the reference crashes here (SIGSEGV, clust2snp.cpp:531); no .snp written" << std::endl; return 139; }
    std::cout << events << " events saved to " << S.snp_path << std::endl;
    return 0;
}
