// ebwt2snp_mgpu -- both tools over the GPUs of one node, host code in C++ over the C ABI, the boundary stitch over RCCL.
//
//   ebwt2snp_mgpu -i <fasta> -n <reads in sample 1> [-G <GPUs>] [ebwt2clust: -k -m] [clust2snp: -L -R -g -c -e -M(cov) -p] [-x -y -z]
//
// What it replaces: running `ebwt2clust -i F` and then `clust2snp -i F -n N` (ebwt2clust.cpp:141-198, clust2snp.cpp:969-1085;
// pipeline.sh:115,122); it writes the same two files, <F>.clusters and <F minus .fast*>.snp, byte for byte.
// One thread per GPU.  Rank r owns the r-th contiguous SA range (left halo 152, right halo 1).  BCR files of the default widths: a rank
// uploads LCP + BWT only (2 of the 7 bytes per entry: what the scan with the pre-filter inside reads) and gathers the document array and the
// positions of the few clusters that filter lets through from the mapped files (e2s_find_candidates_gathered: what the one-GPU chunked job does
// per chunk); the stepwise calls with the same two exchanges.  Other inputs (.gesa, other widths, E2S_MGPU_FULL_COLUMNS=1): all four columns
// -- and all reads, uploaded from ONE index of the FASTA built by the main thread -- resident in HBM, and the step is e2s_resident_job: scan, exchange,
// stitch + statistics, per-cluster pass, tail, all enqueued at once, the two exchanges being ncclAllGather calls on the rank's stream
// (RCCL; xGMI between the GPUs): 1.3 KB of descriptor + statistics per rank, then 16 bytes (events, ok) for the `.snp` ids, which are
// sequential over the ranks.  Where that form does not apply (or a buffer was too small: every rank learns it in the second
// exchange) the ranks run the stepwise calls with the same two exchanges.  Both files are written by all ranks AT ONCE, each at
// its own offset (pwrite): the sizes are exchanged through the process's memory.
// No rank ever waits for one that failed: before every exchange the ranks agree -- a barrier over their status in this process --
// whether all of them are still in; if not, all leave; and a rank whose exchange breaks all the same -- before the collective, in it or behind
// it -- aborts every communicator and releases every waiter (abort_exchange).  (E2S_MGPU_FAIL_RANK=r, E2S_MGPU_FAIL_AT=setup|job|xchg: inject a failure: tests;
// E2S_JOB_FAIL_RANK=r, E2S_JOB_FAIL_AT=1|2: inside e2s_resident_job, before the first exchange / between the two.)
// -G 1 runs the same code without a communicator (one rank has nothing to exchange; E2S_MGPU_FORCE_RCCL=1 under E2S_DEV=1 creates a
// one-rank communicator all the same: how the RCCL calls are exercised on a one-GPU box).  ncclCommInitAll (1.7 s) runs in a thread of
// its own beside the uploads, the FASTA is indexed beside them too, and a job that went through ends with _exit: E2S_CLI_TIMING=1 shows
// where the time of a run goes.
// E2S_MGPU_SHARED_DEVICE=1 puts every rank on GPU 0 and exchanges through host memory (RCCL refuses two ranks on one device):
// a rehearsal of the N-rank logic on a one-GPU box, not a way to run jobs.
#include <fcntl.h>
#include <cerrno>
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
#include "../csrc/e2s_env.h"

namespace {

struct Opts {
    std::string input; uint64_t nr_reads1 = 0; int gpus = 1;
    int k = 16, min_len = 2, lcp = 1, da = 4, pos = 1;
    int k_left = 31, k_right = 30, max_gap = 10, consensus_reads = 20, max_err = 2, mcov_out = 5; double pval = 0.99;
};

// a barrier over the ranks' threads: the rehearsal's exchanges, the agreement before every collective, the sizes of the outputs
// (abort(): a rank that cannot enter an exchange releases everybody who waits or will wait -- wait() then answers false, at once and for good)
class Barrier {
  public:
    explicit Barrier(int n) : n_(n) {}
    bool wait() {
        std::unique_lock<std::mutex> lk(m_);
        if (aborted_) return false;
        const int gen = gen_;
        if (++cnt_ == n_) { cnt_ = 0; ++gen_; cv_.notify_all(); } else cv_.wait(lk, [&]() { return gen_ != gen || aborted_; });
        return !aborted_;
    }
    void abort() { { std::lock_guard<std::mutex> lk(m_); aborted_ = true; } cv_.notify_all(); }
  private:
    int n_, cnt_ = 0, gen_ = 0; bool aborted_ = false; std::mutex m_; std::condition_variable cv_;
};

struct Shared;
void latch_set(Shared& S, int& what, int v);
int latch_wait(Shared& S, const int& what);

struct Shared {
    Opts o; EgsaFiles* F; int nranks; bool shared_device;
    std::vector<ncclComm_t> comms;
    bool teardown = false;                    // free every handle and communicator before the process ends (development: leak checks)
    bool use_rccl = false;                    // more than one rank on separate GPUs (or E2S_MGPU_FORCE_RCCL=1: the RCCL path with ONE rank, tests)
    // two things the ranks' threads wait for while they already create handles and upload their shards: the communicators
    // (ncclCommInitAll: 1.7 s on an MI355X box, started in a thread of its own) and the one index of the FASTA (main thread)
    std::mutex latch_m; std::condition_variable latch_cv; int comms_state = 0, reads_state = 0;     // 0 pending, 1 ready, -1 failed
    std::vector<uint8_t> host_blob;           // rehearsal: the exchange through host memory
    Barrier* bar;
    // the reads, indexed once
    const void* rd_text = nullptr; uint64_t rd_bytes = 0, rd_n = 0; const uint64_t* rd_start = nullptr; const uint32_t* rd_len = nullptr; uint32_t rd_max = 0;
    bool reads_ok = false;
    bool staged = true;                         // uploads out of the mappings through the handles' pinned buffers (e2s_set_upload_staging)
    bool mapped = true, stream_reads = false;   // the shard from the files' mappings, or with pread (E2S_MGPU_PREAD=1); a lone rank streaming the FASTA itself (E2S_MGPU_STREAM_READS=1)
    int cl_fd = -1, snp_fd = -1; std::string snp_path;
    std::atomic<int> rc{0};
    std::atomic<int> crash{0};                // 136 / 139: the reference's clust2snp dies on this input (SIGFPE / SIGSEGV) -- after ebwt2clust wrote its file
    std::vector<int> status;                  // per rank: 0 = in, else its error (agreement before every collective)
    std::vector<int> form;                    // per rank: e2s_resident_job applies
    std::vector<int> verdict;                 // per rank, after the job: 1 done, 2 done (the reference crashes here), 0 redo through the stepwise calls
    std::vector<uint64_t> closed, cands, events, cl_records, snp_bytes, uploaded, gathered; int mcl = 0;
    bool slim = false;                        // ranks upload LCP + BWT only, the flagged clusters' other columns are gathered from the files
    // ... and of the READS only those its candidates name (e2s_upload_reads_of_candidates: what get_reads touches, clust2snp.cpp:519-528),
    // once the per-cluster pass has found them.  With several ranks by default (every rank held ALL reads until round 5: 8 x 30 GB through
    // one host for c5); one rank uploads everything while the scan's columns go up (E2S_MGPU_GATHER_READS=1 / 0 decide otherwise)
    bool gather_reads = false; std::vector<uint64_t> reads_uploaded, reads_gathered;
    // where only the host tail reproduces the reference (a read rank past the last read, -L outside 1..64, an irregular FASTA): every
    // rank hands over its packed candidates and ONE e2s_snp_from_packed runs at the end
    bool host_tail = false; std::vector<std::vector<uint8_t>> pk_stream; std::vector<std::vector<uint32_t>> pk_off8; uint32_t pk_pw = 4;
    int fail_rank = -1; std::string fail_at;
    std::mutex log_m;
};

void latch_set(Shared& S, int& what, int v) { { std::lock_guard<std::mutex> lk(S.latch_m); what = v; } S.latch_cv.notify_all(); }
int latch_wait(Shared& S, const int& what) { std::unique_lock<std::mutex> lk(S.latch_m); S.latch_cv.wait(lk, [&]() { return what != 0; }); return what; }

constexpr size_t BLOB = sizeof(e2s_boundary) + sizeof(e2s_scan_stats);

struct PSink { int fd; uint64_t at; };
int pwrite_sink(void* user, const void* data, size_t bytes) {
    PSink* s = (PSink*)user;
    const char* p = (const char*)data;
    while (bytes) { const ssize_t w = ::pwrite(s->fd, p, bytes, (off_t)s->at); if (w <= 0) return 1; p += w; bytes -= (size_t)w; s->at += (uint64_t)w; }
    return 0;
}

// A rank's piece of the `.clusters` file written from its image in HBM IN THE BACKGROUND, beside the per-cluster pass and the tail: a few
// threads, each with a stream and two pinned buffers of its own and a contiguous slice of the image -- a chunk comes down (DMA) while the
// one before it goes into the page cache (pwrite).  One thread's pwrite (~10 GB/s) was the 0.11 s the one-GPU run spent "writing both files"
// after everything else was done (profiles/round4_c3_end_to_end.txt).  The image must stay as it is until join().
class ImageWriter {
  public:
    void start(int dev, const void* d_img, uint64_t bytes, int fd, uint64_t file_off, int threads) {
        started_ = true;
        const uint64_t CH = 4000000;                              // (a whole number of 10-byte records)
        const uint64_t n_ch = (bytes + CH - 1) / CH;
        threads = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, n_ch));
        rc_.assign((size_t)threads, 0);
        for (int t = 0; t < threads; ++t) th_.emplace_back([=]() {
            const uint64_t c0 = n_ch * (uint64_t)t / (uint64_t)threads, c1 = n_ch * (uint64_t)(t + 1) / (uint64_t)threads;
            hipStream_t st = nullptr; void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr};
            bool ok = hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; i < 2 && ok; ++i) ok = hipHostMalloc(&buf[i], CH, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
            auto issue = [&](uint64_t c) {
                const uint64_t off = c * CH, len = std::min<uint64_t>(CH, bytes - off);
                return hipMemcpyAsync(buf[c & 1], (const uint8_t*)d_img + off, len, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(ev[c & 1], st) == hipSuccess;
            };
            if (ok && c0 < c1) ok = issue(c0);
            for (uint64_t c = c0; c < c1 && ok; ++c) {
                ok = hipEventSynchronize(ev[c & 1]) == hipSuccess;
                if (ok && c + 1 < c1) ok = issue(c + 1);
                const uint64_t off = c * CH; uint64_t len = std::min<uint64_t>(CH, bytes - off), at = file_off + off;
                const char* p = (const char*)buf[c & 1];
                while (ok && len) { const ssize_t w = ::pwrite(fd, p, len, (off_t)at); if (w <= 0) ok = false; else { p += w; len -= (uint64_t)w; at += (uint64_t)w; } }
            }
            if (st) (void)hipStreamSynchronize(st);
            for (int i = 0; i < 2; ++i) { if (buf[i]) (void)hipHostFree(buf[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
            if (st) (void)hipStreamDestroy(st);
            rc_[(size_t)t] = ok ? 0 : 1;
        });
    }
    bool started() const { return started_; }
    int join() { int rc = 0; for (auto& t : th_) t.join(); th_.clear(); for (int v : rc_) rc |= v; return rc; }
  private:
    bool started_ = false; std::vector<std::thread> th_; std::vector<int> rc_;
};

#define STEP(call, what)                                                                                                        \
    do { const int rc_ = (call); if (rc_ && !mine_rc) { std::lock_guard<std::mutex> lk(S.log_m);                                   \
         std::cerr << "rank " << r << ": " << what << " failed (" << rc_ << "): " << (h ? e2s_last_error(h) : "") << std::endl;     \
         mine_rc = rc_; } } while (0)

// An exchange broke on this rank -- before the collective, in it or behind it: its peers may sit in a collective (or, in the rehearsal, at the
// barrier of one) that will never complete.  Every communicator is aborted (pending collectives end with an error), everybody at the
// barrier is released, and the job ends with an error on every rank.  Idempotent; any rank may call it.
void abort_exchange(Shared& S) {
    { std::lock_guard<std::mutex> lk(S.log_m); for (auto& c : S.comms) if (c) { (void)ncclCommAbort(c); c = nullptr; } }
    int z = 0; S.rc.compare_exchange_strong(z, E2S_ERR_IO);
    S.bar->abort();
}
// E2S_MGPU_FAIL_AT=xchg (E2S_DEV=1): the rank fails where an exchange begins, before it has entered the collective
bool inject_xchg(Shared& S, int r) {
    if (S.fail_rank != r || S.fail_at != "xchg") return false;
    std::lock_guard<std::mutex> lk(S.log_m); std::cerr << "rank " << r << ": injected failure (exchange)" << std::endl;
    return true;
}

// Do all ranks still stand?  Every rank publishes its status, all wait, all read the same answer: nobody enters a collective that
// somebody else will not enter.  The first failure becomes the job's exit code.
bool agree(Shared& S, int r, int my_rc) {
    S.status[(size_t)r] = my_rc;
    if (!S.bar->wait()) return false;                           // (an aborted exchange: S.rc says so)
    bool all = true;
    for (int q = 0; q < S.nranks; ++q) if (S.status[(size_t)q]) { all = false; int z = 0; S.rc.compare_exchange_strong(z, S.status[(size_t)q]); }
    if (!S.bar->wait()) return false;
    return all;
}

// ONE all-gather of `bytes` per rank between host buffers (the stepwise path): RCCL on the rank's exchange stream, or host memory + a
// barrier in the rehearsal mode
// A failure at ANY point -- the copy in front of the collective included -- ends in abort_exchange: no peer is left waiting.
int all_gather(Shared& S, int r, const void* mine, size_t bytes, std::vector<uint8_t>& all, void* d_send, void* d_recv, hipStream_t st) {
    all.assign(bytes * (size_t)S.nranks, 0);
    auto broken = [&](int code) { abort_exchange(S); return code; };
    if (inject_xchg(S, r)) return broken(9);
    if (S.shared_device) {
        { std::lock_guard<std::mutex> lk(S.log_m); if (S.host_blob.size() < all.size()) S.host_blob.resize(all.size()); }
        if (!S.bar->wait()) return 5;
        memcpy(S.host_blob.data() + bytes * (size_t)r, mine, bytes);
        if (!S.bar->wait()) return 5;
        memcpy(all.data(), S.host_blob.data(), all.size());
        if (!S.bar->wait()) return 5;
        return 0;
    }
    if (!S.use_rccl) { memcpy(all.data(), mine, bytes); return 0; }          // (one rank: nothing to exchange)
    ncclComm_t comm;
    { std::lock_guard<std::mutex> lk(S.log_m); comm = S.comms[(size_t)r]; }
    if (!comm) return 6;                                                     // (aborted by a peer)
    if (hipMemcpyAsync(d_send, mine, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return broken(1);
    if (ncclAllGather(d_send, d_recv, bytes, ncclChar, comm, st) != ncclSuccess) return broken(2);
    if (hipMemcpyAsync(all.data(), d_recv, all.size(), hipMemcpyDeviceToHost, st) != hipSuccess) return broken(3);
    return hipStreamSynchronize(st) == hipSuccess ? 0 : broken(4);
}

// the exchange of e2s_resident_job: device buffers, enqueued on the handle's stream
struct AgCtx { Shared* S; int r; };
int job_allgather(void* user, const void* d_send, void* d_recv, size_t bytes, void* stream) {
    AgCtx* c = (AgCtx*)user;
    Shared& S = *c->S;
    if (S.use_rccl) {
        ncclComm_t comm;
        { std::lock_guard<std::mutex> lk(S.log_m); comm = S.comms[(size_t)c->r]; }
        return comm && ncclAllGather(d_send, d_recv, bytes, ncclChar, comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
    }
    if (!S.shared_device) return hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : 1;   // (never called: one rank)
    // rehearsal (all ranks on one device): through host memory, which means waiting for the stream here
    std::vector<uint8_t> mine(bytes), all(bytes * (size_t)S.nranks);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(mine.data(), d_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) { abort_exchange(S); return 1; }
    { std::lock_guard<std::mutex> lk(S.log_m); if (S.host_blob.size() < all.size()) S.host_blob.resize(all.size()); }
    if (!S.bar->wait()) return 1;
    memcpy(S.host_blob.data() + bytes * (size_t)c->r, mine.data(), bytes);
    if (!S.bar->wait()) return 1;
    memcpy(all.data(), S.host_blob.data(), all.size());
    if (!S.bar->wait()) return 1;
    if (hipMemcpy(d_recv, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) { abort_exchange(S); return 1; }
    return 0;
}

void rank_main(Shared& S, int r) {
    const Opts& o = S.o;
    const int dev = S.shared_device ? 0 : r;
    e2s_handle h = nullptr, hc = nullptr, hf = nullptr;        // hc (slim): the reads, the gathered per-cluster pass and the tail; hf: a shard that needs all four columns after all
    hipStream_t xs = nullptr; void *d_send = nullptr, *d_recv = nullptr;
    int mine_rc = 0;
    PhaseClock pc;                                               // E2S_CLI_TIMING=1: rank 0's phases on stderr
    pc.on = pc.on && r == 0;
    const uint64_t n = S.F->n;
    // contiguous SA ranges of (almost) equal size; the cuts may fall anywhere, the stitch takes care of clusters that cross them
    const uint64_t lo = n * (uint64_t)r / (uint64_t)S.nranks, hi = n * (uint64_t)(r + 1) / (uint64_t)S.nranks;
    const uint32_t hl = (uint32_t)std::min<uint64_t>(lo, E2S_MAX_CLUST_LEN + 2), hr = (uint32_t)std::min<uint64_t>(n - hi, 1);
    // ---- setup: everything that can fail before the first exchange (handle, exchange buffers, reads, shard) ----
    STEP(e2s_create(dev, &h), "e2s_create");
    if (!mine_rc && S.use_rccl) {
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&d_send, BLOB) != hipSuccess || hipMalloc(&d_recv, BLOB * (size_t)S.nranks) != hipSuccess) mine_rc = E2S_ERR_HIP;
    }
    pc.mark("rank 0: first handle (HIP start-up), exchange buffers");
    // the second handle (streams, events, a dozen small allocations, a pinned mailbox: 20-45 ms) is made by a thread of its own while the shard goes up
    // on the first -- unless the FASTA is streamed on it beside the shard, which needs it now
    std::thread hc_thread; int hc_rc = 0;
    auto make_hc = [&hc, &hc_rc, dev, &S]() { hc_rc = e2s_create(dev, &hc); if (!hc_rc && S.staged) (void)e2s_set_upload_staging(hc, 1); };
    if (!mine_rc && S.slim) { if (S.stream_reads) make_hc(); else hc_thread = std::thread(make_hc); }
    auto hc_ready = [&]() {
        if (hc_thread.joinable()) hc_thread.join();
        if (hc_rc && !mine_rc) {
            std::lock_guard<std::mutex> lk(S.log_m);
            std::cerr << "rank " << r << ": e2s_create (per-cluster pass) failed (" << hc_rc << "): " << e2s_last_error(nullptr) << std::endl;
            mine_rc = hc_rc;
        }
        hc_rc = 0;
    };
    if (S.stream_reads) hc_ready();
    pc.mark("rank 0: second handle");
    // (the pointers this driver uploads from are file mappings: their copies go through the handles' pinned buffers -- E2S_MGPU_STAGING=0: the runtime's staging)
    if (S.staged && h) (void)e2s_set_upload_staging(h, 1);
#define hr_ (S.slim ? hc : h)                                    /* where the reads live (hc: once hc_ready() has run) */
    // A lone rank streams the FASTA itself (e2s_upload_reads_fasta: read, sent and indexed chunk by chunk, in one pass).  Where the reads live on a
    // handle of their own (hc) that goes on in a thread of its own WHILE the shard's columns go up on h: the two share the link and the cores,
    // and neither waits for the other's allocations and kernels; joined right before the per-cluster pass, which wants the reads resident.
    std::thread reads_thread; int reads_rc = 0; uint64_t reads_n = 0;
    auto reads_done = [&]() {
        if (reads_thread.joinable()) reads_thread.join();
        if (reads_rc == E2S_OK && reads_n) { S.reads_ok = true; S.rd_n = reads_n; S.reads_uploaded[(size_t)r] = S.rd_bytes; S.reads_gathered[(size_t)r] = reads_n; }
        else if (!mine_rc && reads_rc != E2S_OK && reads_rc != E2S_ERR_INPUT && reads_rc != E2S_ERR_IO) {      // (an irregular / unreadable FASTA: the host tail decides)
            std::lock_guard<std::mutex> lk(S.log_m);
            std::cerr << "rank " << r << ": upload reads (streamed) failed (" << reads_rc << "): " << (hr_ ? e2s_last_error(hr_) : "") << std::endl;
            mine_rc = reads_rc;
        }
    };
    if (S.stream_reads && !mine_rc && hc) reads_thread = std::thread([&reads_rc, &reads_n, hc, &o]() { reads_rc = e2s_upload_reads_fasta(hc, o.input.c_str(), &reads_n); });
    e2s_shard_desc d{};
    d.global_offset = lo; d.n_entries = hi - lo; d.n_total = n; d.halo_left = hl; d.halo_right = hr;
    d.lcp_bytes = o.lcp; d.da_bytes = o.da; d.pos_bytes = o.pos;
    const uint64_t p0 = lo - hl, n_prov = (hi - lo) + hl + hr;
    if (!mine_rc) {
        // the shard's columns: pointers into the mappings, or (E2S_MGPU_PREAD=1) read from the files with pread into pinned buffers (the _fd forms)
        if (S.slim) {
            if (S.mapped) STEP(e2s_upload_lcp_bwt(h, &d, S.F->lcp.p + p0, S.F->bwt.p + p0), "upload (LCP + BWT)");
            else STEP(e2s_upload_lcp_bwt_fd(h, &d, S.F->lcp.fd, p0, S.F->bwt.fd, p0), "upload (LCP + BWT)");
            S.uploaded[(size_t)r] = 2 * n_prov;
        } else {
            const uint64_t rec = (uint64_t)(o.lcp + o.da + o.pos + 1);
            if (S.F->gesa) {
                if (S.mapped) STEP(e2s_upload_gesa(h, &d, S.F->aos.p + p0 * rec), "upload (.gesa)");
                else STEP(e2s_upload_gesa_fd(h, &d, S.F->aos.fd, p0 * rec), "upload (.gesa)");
            } else if (S.mapped) STEP(e2s_upload_bcr(h, &d, S.F->bwt.p + p0, S.F->lcp.p + p0 * (uint64_t)o.lcp, S.F->pairsa.p + p0 * (uint64_t)(o.da + o.pos)), "upload (BCR)");
            else STEP(e2s_upload_bcr_fd(h, &d, S.F->bwt.fd, p0, S.F->lcp.fd, p0 * (uint64_t)o.lcp, S.F->pairsa.fd, p0 * (uint64_t)(o.da + o.pos)), "upload (BCR)");
            S.uploaded[(size_t)r] = rec * n_prov;
        }
    }
    hc_ready();
    pc.mark("rank 0: shard uploaded");
    // the reads: all of them now -- a lone rank streams the FASTA itself (one pass: read, sent and indexed chunk by chunk); several ranks take
    // them from the ONE index the main thread has been building meanwhile -- or, later, those this rank's candidates name
    if (S.stream_reads && !reads_thread.joinable()) {           // (all four columns: the reads live on the shard's own handle, one upload after the other)
        reads_rc = mine_rc ? mine_rc : e2s_upload_reads_fasta(hr_, o.input.c_str(), &reads_n);
        reads_done();
    } else if (S.stream_reads) {
    } else if (!S.gather_reads) {
        (void)latch_wait(S, S.reads_state);
        if (!mine_rc && S.reads_ok) { STEP(e2s_upload_reads(hr_, S.rd_text, S.rd_bytes, S.rd_start, S.rd_len, S.rd_n, S.rd_max), "upload reads"); S.reads_uploaded[(size_t)r] = S.rd_bytes; S.reads_gathered[(size_t)r] = S.rd_n; }
    }
    if (S.use_rccl && latch_wait(S, S.comms_state) < 0 && !mine_rc) mine_rc = E2S_ERR_HIP;       // (ncclCommInitAll failed: said so on stderr)
    pc.mark("rank 0: reads uploaded (waited for the FASTA index and the communicators)");
    if (S.fail_rank == r && S.fail_at == "setup") { std::lock_guard<std::mutex> lk(S.log_m); std::cerr << "rank " << r << ": injected failure (setup)" << std::endl; mine_rc = E2S_ERR_HIP; }
    e2s_resident_params rp{};
    rp.k = o.k; rp.min_len = o.min_len; rp.rank = r; rp.world = S.nranks; rp.want_image = 1;
    rp.o.nr_reads1 = o.nr_reads1; rp.o.k_left = o.k_left; rp.o.k_right = o.k_right; rp.o.mcov_out = o.mcov_out; rp.o.consensus_reads = o.consensus_reads;
    rp.o.max_gap = o.max_gap; rp.o.max_err = o.max_err; rp.o.pval = o.pval;
    if (!mine_rc && !S.slim) { rp.probe_only = 1; S.form[(size_t)r] = e2s_resident_job(h, &rp, job_allgather, nullptr, nullptr) == E2S_OK ? 1 : 0; rp.probe_only = 0; }
    ImageWriter writer;                                         // (the stepwise path starts it early; it is joined before the handles go)
    auto leave = [&]() {
        if (hc_thread.joinable()) hc_thread.join();
        if (reads_thread.joinable()) reads_thread.join();
        if (writer.started()) (void)writer.join();
        if (d_send) (void)hipFree(d_send);
        if (d_recv) (void)hipFree(d_recv);
        if (xs) (void)hipStreamDestroy(xs);
        if (hf) e2s_destroy(hf);
        if (hc) e2s_destroy(hc);
        if (h) e2s_destroy(h);
    };
    if (!agree(S, r, mine_rc)) { leave(); return; }
    bool native = !S.slim;
    for (int q = 0; q < S.nranks; ++q) native = native && S.form[(size_t)q] != 0;
    e2s_handle ht = h;                                          // where the candidates and the tail end up

    int mcl = 0; uint64_t n_cand = 0, nv = 0, ne = 0, nb = 0, need_host = 0;
    bool done = false, host_tail = false;
    // ---- the step as ONE enqueue: e2s_resident_job, its two exchanges on the handle's stream ----
    if (native) {
        AgCtx ctx{&S, r};
        e2s_job_report rep{};
        // (the job's buffers were all put in place by the probe call above, before the ranks agreed to enter: between its enqueue and the
        //  two exchanges a rank can only fail on a HIP error, and then it still enters both exchanges with a zeroed share -- e2s_resident_job's
        //  `bail`; E2S_JOB_FAIL_AT / E2S_JOB_FAIL_RANK inject such a failure INSIDE the job, E2S_MGPU_FAIL_AT=job one behind it)
        int jrc = e2s_resident_job(h, &rp, job_allgather, &ctx, &rep);
        if (S.fail_rank == r && S.fail_at == "job") { std::lock_guard<std::mutex> lk(S.log_m); std::cerr << "rank " << r << ": injected failure (job)" << std::endl; jrc = E2S_ERR_HIP; }
        int crash_code = 0;
        if (jrc == E2S_OK) S.verdict[(size_t)r] = 1;
        else if (jrc == E2S_ERR_REF_CRASH) {                       // where clust2snp dies in the reference: ebwt2clust's file is written all the same
            const char* why = e2s_last_error(h);
            crash_code = (strstr(why, "divides by zero") || strstr(why, "histogram")) ? 136 : 139;
            S.verdict[(size_t)r] = 2;
        } else if (jrc == E2S_ERR_CAPACITY) S.verdict[(size_t)r] = 0;    // (on every rank alike -- the stepwise calls below)
        else {
            STEP(jrc, "resident job");
            // the exchange itself failed on this rank: its peers may sit in a collective that will never complete
            if (jrc == E2S_ERR_IO) abort_exchange(S);
        }
        if (!agree(S, r, mine_rc)) { leave(); return; }
        // the verdict travels in the second exchange, so the ranks agree on it; should they ever not (a bug), everybody takes the stepwise
        // path rather than some rank waiting in an exchange the others never enter
        bool all_done = true;
        for (int q = 0; q < S.nranks; ++q) all_done = all_done && S.verdict[(size_t)q] != 0;
        if (all_done && S.verdict[(size_t)r] == 1) {
            done = true; mcl = rep.max_clust_length; n_cand = rep.n_candidates; nv = rep.n_variants; ne = rep.n_events; nb = rep.snp_bytes;
            S.closed[(size_t)r] = rep.n_closed;
            if (r == 0) S.mcl = mcl;
        } else if (all_done) {
            int z = 0; S.crash.compare_exchange_strong(z, crash_code);
            uint64_t nc = 0;
            if (e2s_cluster_counts(h, &nc, nullptr) == E2S_OK) S.closed[(size_t)r] = nc;
            done = true; n_cand = 0; ne = 0; nb = 0;
        }
        S.bar->wait();                                              // (every rank has read the verdicts)
        pc.mark("rank 0: e2s_resident_job (one enqueue, one wait)");
    }
    // ---- the stepwise calls with the same two exchanges ----
    if (!done) {
        std::vector<uint8_t> mine(BLOB, 0), all;
        STEP(e2s_prefilter_arm(h, o.mcov_out), "arm");
        STEP(e2s_cluster_scan(h, o.k, o.min_len, 0), "cluster scan");
        if (!mine_rc) {
            STEP(e2s_boundary_desc(h, (e2s_boundary*)mine.data()), "boundary");
            STEP(e2s_scan_stats_desc(h, (e2s_scan_stats*)(mine.data() + sizeof(e2s_boundary))), "scan statistics");
        }
        pc.mark("rank 0: scan (+ pre-filter), descriptor");
        if (!agree(S, r, mine_rc)) { leave(); return; }
        if (all_gather(S, r, mine.data(), BLOB, all, d_send, d_recv, xs)) mine_rc = E2S_ERR_HIP;
        std::vector<e2s_boundary> allb((size_t)S.nranks); std::vector<e2s_scan_stats> alls((size_t)S.nranks);
        for (int q = 0; q < S.nranks; ++q) {
            memcpy(&allb[(size_t)q], all.data() + BLOB * (size_t)q, sizeof(e2s_boundary));
            memcpy(&alls[(size_t)q], all.data() + BLOB * (size_t)q + sizeof(e2s_boundary), sizeof(e2s_scan_stats));
        }
        if (!mine_rc) {
            STEP(e2s_stitch(h, allb.data(), S.nranks, r), "stitch");
            uint64_t nc = 0, nrec = 0;
            STEP(e2s_cluster_counts(h, &nc, &nrec), "counts");
            S.closed[(size_t)r] = nc;
            STEP(e2s_pack_clusters(h, 1), "pack .clusters");           // on the second stream, beside the per-cluster pass
            // ---- clust2snp: statistics (pure function of the gathered array), per-cluster pass, tail ----
            uint64_t n_clust = 0, n_bases = 0;
            const int rc = e2s_auto_max_clust_length_ranks(allb.data(), alls.data(), S.nranks, n, o.k, o.lcp, o.mcov_out, o.pval, &mcl, &n_clust, &n_bases);
            if (rc == E2S_ERR_REF_CRASH) { int z = 0; S.crash.compare_exchange_strong(z, 136); }      // (the same on every rank: the statistics are the merged ones)
            else if (rc && !mine_rc) mine_rc = rc;
            if (r == 0) S.mcl = mcl;
        }
        // the `.clusters` piece of this rank leaves NOW, in the background: the image is being laid out (e2s_clusters_image waits for it: a
        // millisecond), every rank's record count gives every rank its place in the file
        uint64_t early_records = 0; const void* early_img = nullptr;
        if (!mine_rc) STEP(e2s_clusters_image(h, &early_img, &early_records), ".clusters image");
        S.cl_records[(size_t)r] = mine_rc ? 0 : early_records;
        if (S.bar->wait() && !mine_rc && early_records) {
            uint64_t at = 0;
            for (int q = 0; q < r; ++q) at += S.cl_records[(size_t)q] * 10;
            writer.start(dev, early_img, early_records * 10, S.cl_fd, at, 4);
        }
        pc.mark("rank 0: exchange 1, stitch, statistics, image laid out, its file being written in the background");
        if (reads_thread.joinable()) { reads_done(); pc.mark("rank 0: waited for the reads (streamed beside the shard)"); }
        if (!mine_rc && !S.crash.load()) {
            e2s_c2s_params prm{};
            prm.nr_reads1 = o.nr_reads1; prm.k_left = o.k_left; prm.k_right = o.k_right; prm.mcov_out = o.mcov_out;
            prm.consensus_reads = o.consensus_reads; prm.max_clust_length = mcl; prm.min_len = o.min_len;
            if (S.slim) {
                // the clusters the pre-filter let through, gathered from all four columns of the mapped files, through find_variants on hc
                e2s_stitch_plan_t plan{};
                e2s_chunk_params cp{}; cp.device = dev; cp.k = o.k; cp.min_len = o.min_len; cp.lcp_bytes = o.lcp; cp.da_bytes = o.da; cp.pos_bytes = o.pos; cp.n_total = n;
                e2s_c2s_options co{}; co.nr_reads1 = o.nr_reads1; co.k_left = o.k_left; co.k_right = o.k_right; co.mcov_out = o.mcov_out;
                co.consensus_reads = o.consensus_reads; co.max_gap = o.max_gap; co.max_err = o.max_err; co.pval = o.pval;
                FileSource fsrc{S.F, o.lcp, o.da, o.pos};
                const e2s_egsa_source src = fsrc.source();
                int grc = e2s_stitch_plan(allb.data(), S.nranks, r, n, o.k, o.lcp, &plan);
                uint64_t ng = 0;
                if (grc == E2S_OK) grc = e2s_find_candidates_gathered(h, hc, &cp, &co, &src, &plan, mcl, &n_cand, &ng);
                if (grc == E2S_OK) {
                    ht = hc; S.gathered[(size_t)r] = ng; S.uploaded[(size_t)r] += 7 * ng;
                    if (S.gather_reads && n_cand) {                 // the reads these candidates name, and no others
                        (void)latch_wait(S, S.reads_state);
                        uint64_t rg = 0, rb = 0;
                        if (S.reads_ok) STEP(e2s_upload_reads_of_candidates(hc, S.rd_text, S.rd_bytes, S.rd_start, S.rd_len, S.rd_n, S.rd_max, &rg, &rb), "upload the candidates' reads");
                        S.reads_uploaded[(size_t)r] = rb; S.reads_gathered[(size_t)r] = rg;
                    }
                }
                else if (grc == E2S_ERR_STATE) {
                    // (a window too dense for the pre-filter's queue, a cluster of 65536+ entries: this shard needs all four columns after all)
                    STEP(e2s_create(dev, &hf), "e2s_create (full columns)");
                    if (S.gather_reads) (void)latch_wait(S, S.reads_state);
                    if (!mine_rc && S.reads_ok && !S.gather_reads) STEP(e2s_share_reads(hf, hc), "share reads");
                    if (!mine_rc && S.reads_ok && S.gather_reads) { STEP(e2s_upload_reads(hf, S.rd_text, S.rd_bytes, S.rd_start, S.rd_len, S.rd_n, S.rd_max), "upload reads (all columns)"); S.reads_uploaded[(size_t)r] = S.rd_bytes; S.reads_gathered[(size_t)r] = S.rd_n; }
                    if (!mine_rc) STEP(e2s_upload_bcr(hf, &d, S.F->bwt.p + p0, S.F->lcp.p + p0, S.F->pairsa.p + p0 * 5), "upload (BCR, all columns)");
                    if (!mine_rc) STEP(e2s_prefilter_arm(hf, o.mcov_out), "arm");
                    if (!mine_rc) STEP(e2s_cluster_scan(hf, o.k, o.min_len, 0), "cluster scan (all columns)");
                    if (!mine_rc) STEP(e2s_stitch(hf, allb.data(), S.nranks, r), "stitch (all columns)");
                    if (!mine_rc) STEP(e2s_find_candidates(hf, &prm, (uint64_t)r << 40, &n_cand), "find candidates");
                    ht = hf; S.uploaded[(size_t)r] += 7 * n_prov;
                } else STEP(grc, "find candidates (gathered)");
            } else STEP(e2s_find_candidates(h, &prm, (uint64_t)r << 40, &n_cand), "find candidates");
            pc.mark(S.slim ? "rank 0: flagged clusters gathered from the files + per-cluster pass" : "rank 0: per-cluster pass");
            if (!mine_rc && n_cand) {
                e2s_tail_params tp{o.k_left, o.k_right, o.max_gap, o.max_err};
                const int trc = e2s_call_variants(ht, &tp, 0, &nv, &ne, &nb);   // (first_id 0: the text is written once, after the exchange below)
                if (trc == E2S_ERR_STATE) { need_host = 1; ne = 0; }
                else if (trc == E2S_ERR_REF_CRASH) { int z = 0; S.crash.compare_exchange_strong(z, 139); ne = 0; }    // the reference's SIGSEGV / throw
                else STEP(trc, "tail (consensus, distance, text)");
            }
        }
        pc.mark("rank 0: tail (consensus, distance)");
        // the second exchange: events per rank, for sequential ids
        if (!agree(S, r, mine_rc)) { leave(); return; }
        std::vector<uint8_t> alle;
        const uint64_t mine2[2] = {ne, need_host};
        if (all_gather(S, r, mine2, 16, alle, d_send, d_recv, xs)) mine_rc = E2S_ERR_HIP;
        uint64_t before = 0;
        for (int q = 0; q < S.nranks && !mine_rc; ++q) {
            uint64_t v[2]; memcpy(v, alle.data() + 16 * (size_t)q, 16);
            if (q < r) before += v[0];
            host_tail |= v[1] != 0;
        }
        if (!mine_rc && !host_tail && n_cand) STEP(e2s_renumber_snp(ht, 1 + before, &nb), "ids and text");
    }
    pc.mark("rank 0: step done (exchange 2, ids and text)");
    S.cands[(size_t)r] = n_cand;
    S.events[(size_t)r] = host_tail ? 0 : ne;
    // ---- the files: every rank learns every rank's sizes through this process's memory, then all write at once ----
    uint64_t my_records = 0;
    if (!mine_rc) { const void* img = nullptr; STEP(e2s_clusters_image(h, &img, &my_records), ".clusters image"); }
    S.cl_records[(size_t)r] = my_records;
    S.snp_bytes[(size_t)r] = (host_tail || !ne) ? 0 : nb;
    if (host_tail && !mine_rc && n_cand) {                          // the packed candidates, for ONE host tail over all ranks (main thread)
        const void* st = nullptr; const uint32_t* of = nullptr; uint64_t nc2 = 0, bytes = 0; uint32_t pw = 4;
        STEP(e2s_candidates_packed(ht, &st, &of, &nc2, &bytes, &pw), "packed candidates");
        if (!mine_rc) {
            S.pk_stream[(size_t)r].assign((const uint8_t*)st, (const uint8_t*)st + bytes);
            S.pk_off8[(size_t)r].assign(of, of + nc2);
            S.pk_pw = pw;
        }
    }
    if (host_tail) S.host_tail = true;
    if (!agree(S, r, mine_rc)) { leave(); return; }
    if (S.crash.load()) S.snp_bytes[(size_t)r] = 0;                 // (no `.snp` where the reference's clust2snp dies)
    S.bar->wait();
    uint64_t cl_at = 0, snp_at = 0, snp_total = 0;
    for (int q = 0; q < S.nranks; ++q) { if (q < r) { cl_at += S.cl_records[(size_t)q] * 10; snp_at += S.snp_bytes[(size_t)q]; } snp_total += S.snp_bytes[(size_t)q]; }
    if (r == 0 && snp_total) {                                      // (the `.snp` comes into being with the first event: nothing where the reference crashes)
        S.snp_fd = ::open(S.snp_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (S.snp_fd < 0) mine_rc = E2S_ERR_IO;
    }
    if (!agree(S, r, mine_rc)) { leave(); return; }
    if (writer.started()) { if (writer.join() && !mine_rc) { std::lock_guard<std::mutex> lk(S.log_m); std::cerr << "rank " << r << ": writing .clusters failed" << std::endl; mine_rc = E2S_ERR_IO; } }
    else if (my_records) { PSink ps{S.cl_fd, cl_at}; uint64_t w = 0; STEP(e2s_stream_clusters(h, pwrite_sink, &ps, &w), "write .clusters"); }
    if (S.snp_bytes[(size_t)r]) { PSink ps{S.snp_fd, snp_at}; STEP(e2s_stream_snp(ht, pwrite_sink, &ps), "write .snp"); }
    const bool all_fine = agree(S, r, mine_rc);
    pc.mark("rank 0: both files written");
    // (a job that went through leaves its handles to the end of the process: freeing tens of GB of HBM buffer by buffer took 0.7-1.2 s of a
    //  0.8-s job on the full c3 files; E2S_MGPU_TEARDOWN=1 under E2S_DEV=1 frees everything the slow way)
    if (!all_fine || S.teardown) leave();
}
#undef hr_

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
    const double t_main = PhaseClock::now();
    struct timespec ts_main; clock_gettime(CLOCK_REALTIME, &ts_main);
    const double epoch_main = ts_main.tv_sec + ts_main.tv_nsec * 1e-9;
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
    PhaseClock pcs;
    EgsaFiles F;
    F.open_or_die(o.input, o.lcp, o.da, o.pos);
    pcs.mark("index files mapped");
    Shared S;
    S.o = o; S.F = &F; S.nranks = o.gpus;
    S.shared_device = getenv("E2S_MGPU_SHARED_DEVICE") != nullptr && atoi(getenv("E2S_MGPU_SHARED_DEVICE")) != 0;
    if ((uint64_t)S.nranks * 4 > F.n) { std::cerr << "too few entries for " << S.nranks << " ranks" << std::endl; return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::cerr << "GPU unavailable (this build has no CPU path)" << std::endl; return 2; }
    if (!S.shared_device && S.nranks > ndev) { std::cerr << "-G " << S.nranks << " but only " << ndev << " GPU(s) visible" << std::endl; return 2; }
    // RCCL: one communicator per GPU, all in this process (ncclCommInitAll), each driven by its rank's thread -- only where there is
    // something to exchange (one rank has nothing: ncclCommInitAll alone costs 1.7 s of a 0.8-s job), and created in a thread of its own
    // while the ranks upload their shards and the FASTA is indexed; a rank waits for it right before its first exchange
    S.use_rccl = !S.shared_device && (S.nranks > 1 || (e2s_env_dev("E2S_MGPU_FORCE_RCCL") && atoi(e2s_env_dev("E2S_MGPU_FORCE_RCCL")) != 0));
    S.teardown = e2s_env_dev("E2S_MGPU_TEARDOWN") && atoi(e2s_env_dev("E2S_MGPU_TEARDOWN")) != 0;
    std::thread comm_thread;
    if (S.use_rccl) {
        S.comms.assign((size_t)S.nranks, nullptr);
        comm_thread = std::thread([&S]() {
            PhaseClock pcc;
            std::vector<int> devs((size_t)S.nranks);
            for (int r = 0; r < S.nranks; ++r) devs[(size_t)r] = r;
            const ncclResult_t nr = ncclCommInitAll(S.comms.data(), S.nranks, devs.data());
            if (nr != ncclSuccess) { std::lock_guard<std::mutex> lk(S.log_m); std::cerr << "ncclCommInitAll failed: " << ncclGetErrorString(nr) << std::endl; }
            pcc.mark("RCCL communicators (ncclCommInitAll, beside the uploads)");
            latch_set(S, S.comms_state, nr == ncclSuccess ? 1 : -1);
        });
    }
    Barrier bar(S.nranks);
    S.bar = &bar;
    S.closed.assign((size_t)S.nranks, 0); S.cands.assign((size_t)S.nranks, 0); S.events.assign((size_t)S.nranks, 0);
    S.cl_records.assign((size_t)S.nranks, 0); S.snp_bytes.assign((size_t)S.nranks, 0); S.uploaded.assign((size_t)S.nranks, 0); S.gathered.assign((size_t)S.nranks, 0);
    S.slim = !F.gesa && o.lcp == 1 && o.da == 4 && o.pos == 1 && !(getenv("E2S_MGPU_FULL_COLUMNS") && atoi(getenv("E2S_MGPU_FULL_COLUMNS")) != 0); S.status.assign((size_t)S.nranks, 0); S.form.assign((size_t)S.nranks, 0); S.verdict.assign((size_t)S.nranks, 0);
    S.pk_stream.resize((size_t)S.nranks); S.pk_off8.resize((size_t)S.nranks);
    S.reads_uploaded.assign((size_t)S.nranks, 0); S.reads_gathered.assign((size_t)S.nranks, 0);
    S.gather_reads = S.slim && S.nranks > 1;
    if (const char* e = getenv("E2S_MGPU_GATHER_READS")) S.gather_reads = S.slim && atoi(e) != 0;
    // Where the uploads read from.  The shard's columns: pointers into the files' mappings; E2S_MGPU_PREAD=1: the files themselves, with pread into
    // pinned buffers (the _fd uploads: 6 GB of LCP + BWT in 0.16-0.21 s against 0.21-0.26).  The reads: ONE index of the FASTA built over its mapping
    // by the main thread beside the shards' uploads, then uploaded; E2S_MGPU_STREAM_READS=1 (a lone rank): streamed in one pass by a thread of the
    // rank's beside its shard instead.  Measured on the whole c3 files, box by box (profiles/round5_c3_uploads_from_files.txt): the three forms end
    // within 5 % of each other -- what the shard's upload gains the FASTA index, now the longer arm, gives back; the mappings stay the default.
    S.mapped = !(getenv("E2S_MGPU_PREAD") != nullptr && atoi(getenv("E2S_MGPU_PREAD")) != 0);
    S.staged = !(getenv("E2S_MGPU_STAGING") != nullptr && atoi(getenv("E2S_MGPU_STAGING")) == 0);
    S.stream_reads = getenv("E2S_MGPU_STREAM_READS") != nullptr && atoi(getenv("E2S_MGPU_STREAM_READS")) != 0 && !S.gather_reads && S.nranks == 1;
    if (const char* e = e2s_env_dev("E2S_MGPU_FAIL_RANK")) S.fail_rank = atoi(e);
    S.fail_at = e2s_env_dev("E2S_MGPU_FAIL_AT") ? e2s_env_dev("E2S_MGPU_FAIL_AT") : "setup";
    e2s_fasta_index* fix = nullptr;
    const std::string cl_path = o.input + ".clusters";
    S.cl_fd = ::open(cl_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (S.cl_fd < 0) { std::cerr << "cannot create " << cl_path << ": " << strerror(errno) << std::endl; if (comm_thread.joinable()) comm_thread.join(); return 2; }
    S.snp_path = o.input.substr(0, o.input.rfind(".fast")) + ".snp";
    std::cout << "This is ebwt2snp_mgpu. Input file: " << o.input << ", " << F.n << " EGSA entries over " << S.nranks << " GPU(s)"
              << (S.shared_device ? " [rehearsal: all ranks on GPU 0, exchange through host memory]" : (S.use_rccl ? " [RCCL all-gather]" : " [one rank: nothing to exchange]")) << std::endl;
    std::vector<std::thread> th;
    for (int r = 0; r < S.nranks; ++r) th.emplace_back(rank_main, std::ref(S), r);
    if (S.stream_reads) {
        struct stat st{};
        if (::stat(o.input.c_str(), &st) == 0) S.rd_bytes = (uint64_t)st.st_size;
    } else {
        // the reads: ONE index of the FASTA for all ranks, built here while the ranks upload their shards (an irregular or unreadable file:
        // no GPU tail, the host tail decides)
        PhaseClock pcm;
        S.reads_ok = e2s_fasta_index_open(o.input.c_str(), &fix, &S.rd_text, &S.rd_bytes, &S.rd_start, &S.rd_len, &S.rd_n, &S.rd_max) == E2S_OK && S.rd_n > 0;
        pcm.mark("FASTA indexed (one index for all ranks, beside the shards' uploads)");
        latch_set(S, S.reads_state, 1);
    }
    for (auto& t : th) t.join();
    if (comm_thread.joinable()) comm_thread.join();
    pcs.mark("ranks done");
    if (fix && (S.teardown || S.rc.load())) e2s_fasta_index_close(fix);
    if (S.cl_fd >= 0) ::close(S.cl_fd);
    if (S.snp_fd >= 0) ::close(S.snp_fd);
    if (S.teardown || S.rc.load()) for (auto& c : S.comms) if (c) ncclCommDestroy(c);
    if (S.rc.load()) ::unlink(cl_path.c_str());                  // (a job that failed leaves no half-written `.clusters` behind)
    if (!S.rc.load() && S.crash.load()) S.rc.store(S.crash.load());   // (ebwt2clust's file stays, as after the reference's two runs)
    uint64_t closed = 0, cands = 0, events = 0;
    for (int r = 0; r < S.nranks; ++r) { closed += S.closed[(size_t)r]; cands += S.cands[(size_t)r]; events += S.events[(size_t)r]; }
    if (!S.rc.load() && S.host_tail && !S.crash.load()) {
        // every rank's packed candidates, in rank order = SA order, through ONE host tail
        std::vector<uint8_t> st; std::vector<uint32_t> of;
        for (int r = 0; r < S.nranks && !S.rc.load(); ++r) {
            const uint64_t base8 = st.size() / 8;
            if (base8 + S.pk_stream[(size_t)r].size() / 8 >= (1ull << 32)) { S.rc.store(E2S_ERR_CAPACITY); break; }
            st.insert(st.end(), S.pk_stream[(size_t)r].begin(), S.pk_stream[(size_t)r].end());
            for (uint32_t v : S.pk_off8[(size_t)r]) of.push_back((uint32_t)(base8 + v));
        }
        if (!S.rc.load() && !of.empty()) {
            of.push_back((uint32_t)(st.size() / 8));
            e2s_tail_params tp{o.k_left, o.k_right, o.max_gap, o.max_err};
            uint64_t nv = 0;
            const int trc = e2s_snp_from_packed(st.data(), of.data(), of.size() - 1, S.pk_pw, &tp, o.input.c_str(), S.snp_path.c_str(), &nv, &events);
            if (trc) { std::cerr << "host tail failed (" << trc << ")" << std::endl; S.rc.store(trc == E2S_ERR_REF_CRASH ? 139 : trc); }
        }
    }
    pcs.mark("output files closed");
    const int rc = S.rc.load();
    if (rc == 139) { std::cerr << "the reference crashes on this input in its tail; no .snp written" << std::endl; return 139; }
    if (rc == 136) { std::cerr << "no usable cluster record: the reference divides by zero here (SIGFPE)" << std::endl; return 136; }
    if (rc) return 2;
    std::cout << "Done. " << (unsigned int)closed << " clusters saved to output file." << std::endl;
    std::cout << "Cluster sizes allowed: [" << o.mcov_out * 2 << "," << S.mcl << "]" << std::endl;
    std::cout << "Done. " << cands << " potential variants detected (some might be detected twice: on fw and rev strands)" << std::endl;
    if (cands == 0) { std::cerr << "no candidate variant: the reference crashes here (SIGSEGV, clust2snp.cpp:531); no .snp written" << std::endl; return 139; }
    if (events == 0) {       // candidates, but not one event: the reference has opened its output by then and leaves it EMPTY (the file above comes with the first event)
        const int fd = ::open(S.snp_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) { std::cerr << "could not create " << S.snp_path << std::endl; return 2; }
        ::close(fd);
    }
    std::cout << events << " events saved to " << S.snp_path << std::endl;
    if (const char* jp = getenv("E2S_CLI_JSON")) {                  // a perf report in a side file (stdout stays as it is)
        if (FILE* f = *jp ? fopen(jp, "w") : nullptr) {
            // (epoch_*: wall-clock stamps of main()'s first line and of this line, so that whoever started the process can tell the loader's and
            //  the kernel's share -- before main(), behind _exit: the mappings and the GPU memory go then -- from the job's)
            struct timespec tsn; clock_gettime(CLOCK_REALTIME, &tsn);
            fprintf(f, "{\"tool\": \"ebwt2snp_mgpu\", \"library\": \"%s\", \"entries\": %llu, \"ranks\": %d, \"form\": \"%s\", \"seconds_wall\": %.6f, \"epoch_main\": %.6f, \"epoch_report\": %.6f, \"bytes_uploaded_per_rank\": [",
                    e2s_version(), (unsigned long long)F.n, S.nranks, S.slim ? "LCP + BWT uploaded, flagged clusters gathered from the files" : "all four columns resident (e2s_resident_job)",
                    (PhaseClock::now() - t_main) * 1e-3, epoch_main, tsn.tv_sec + tsn.tv_nsec * 1e-9);
            for (int r = 0; r < S.nranks; ++r) fprintf(f, "%s%llu", r ? ", " : "", (unsigned long long)S.uploaded[(size_t)r]);
            fprintf(f, "], \"entries_gathered_per_rank\": [");
            for (int r = 0; r < S.nranks; ++r) fprintf(f, "%s%llu", r ? ", " : "", (unsigned long long)S.gathered[(size_t)r]);
            fprintf(f, "], \"reads_text_bytes\": %llu, \"reads_bytes_uploaded_per_rank\": [", (unsigned long long)S.rd_bytes);
            for (int r = 0; r < S.nranks; ++r) fprintf(f, "%s%llu", r ? ", " : "", (unsigned long long)S.reads_uploaded[(size_t)r]);
            fprintf(f, "], \"reads_resident_per_rank\": [");
            for (int r = 0; r < S.nranks; ++r) fprintf(f, "%s%llu", r ? ", " : "", (unsigned long long)S.reads_gathered[(size_t)r]);
            fprintf(f, "], \"clusters_printed\": %llu, \"candidates\": %llu, \"events\": %llu, \"max_clust_length\": %d}\n", (unsigned long long)closed, (unsigned long long)cands,
                    (unsigned long long)events, S.mcl);
            fclose(f);
        }
    }
    if (!S.teardown) {                                              // both files are closed; nothing else of this process is worth an orderly teardown
        std::cout.flush(); std::cerr.flush(); fflush(nullptr);
        _exit(0);
    }
    return 0;
}
