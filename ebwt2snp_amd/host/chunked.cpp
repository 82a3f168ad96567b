// chunked.cpp -- whole jobs over an EGSA that comes from the host in chunks (include/e2s_hip.h, "whole jobs ...").
//
// The reference streams its index files with O(cluster) memory (egsa_stream::read_el, include.hpp:120-199; the loops of
// ebwt2clust.cpp:90-122 and clust2snp.cpp:806-857).  Here the array passes through ONE GPU in chunks of SA positions: two handles on
// the device, an uploader thread filling one while the other is processed, every chunk a rank of the boundary stitch the multi-GPU
// path uses (e2s_stitch_plan, halo 152 / 1) -- no new semantics, the outputs are those of one pass over the whole array.
// Host code over the public C ABI (plus e2s_share_reads: the second handle uses the first one's resident reads).
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/e2s_hip.h"
#include "../csrc/e2s_env.h"

extern "C" void e2s_internal_set_error(e2s_handle h, const char* msg);   // e2s_capi.hip: h == nullptr: this thread's handle-less message
extern "C" int e2s_share_reads(e2s_handle dst, e2s_handle src);    // e2s_capi.hip (not part of the public ABI)
extern "C" int e2s_internal_flagged_list(e2s_handle h, const uint64_t** starts, const uint16_t** lens, uint64_t* n, int* needs_full);
extern "C" int e2s_internal_record_prefilter(e2s_handle h, int mcov_out, int max_clust_length, const uint64_t** starts, const uint16_t** lens, uint64_t* n);
extern "C" void e2s_internal_phantom(int lcp_bytes, int da_bytes, int pos_bytes, int gesa_layout, uint32_t text, uint32_t suff, uint32_t lcp, uint32_t bwt, uint32_t out[4]);
extern "C" int e2s_internal_find_candidates_compact(e2s_handle h, const e2s_c2s_params* p, uint64_t n_cl, const uint64_t* gstart, const uint64_t* off,
                                                    const uint16_t* len, const uint8_t* bwt, const uint8_t* lcp, const uint8_t* pos, const uint32_t* da,
                                                    uint64_t n_entries, uint64_t key_lo, uint64_t key_span, uint64_t* n_cand);

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Chunk { uint64_t lo, hi; uint32_t hl, hr; };

// chunks of `ce` owned entries; right halo 1 for the scan (the stencil looks one entry ahead), max cluster length + 1 where the
// records come from a file (a cluster is handled by the chunk that holds its START there)
std::vector<Chunk> plan_chunks(uint64_t n, uint64_t ce, uint32_t right_halo) {
    std::vector<Chunk> v;
    if (ce == 0) ce = 1500000000ull;                            // 10.5 GB of EGSA per chunk at default widths, two in flight
    ce = std::max<uint64_t>(ce, 4096);
    for (uint64_t lo = 0; lo < n; lo += ce) {
        Chunk c;
        c.lo = lo; c.hi = std::min(n, lo + ce);
        if (n - c.hi < 2 && c.hi < n) c.hi = n;                 // (the last rank must own at least two entries)
        c.hl = (uint32_t)std::min<uint64_t>(lo, E2S_MAX_CLUST_LEN + 2);
        c.hr = (uint32_t)std::min<uint64_t>(n - c.hi, right_halo);
        v.push_back(c);
        if (c.hi == n) break;
    }
    return v;
}

// two handles on one device and an uploader thread: chunk r goes into handle r & 1 while chunk r - 1 is processed on the other
class ChunkPipe {
  public:
    // mode 0: the LCP column (ebwt2clust), 1: all four columns, 2: LCP + BWT (the scan with clust2snp's pre-filter inside)
    ChunkPipe(const e2s_chunk_params& p, const e2s_egsa_source& src, const std::vector<Chunk>& chunks, int mode)
        : p_(p), src_(src), chunks_(chunks), mode_(mode) {}
    ~ChunkPipe() {
        stop();
        for (auto& h : h_) if (h) e2s_destroy(h);
        if (rc_) e2s_internal_set_error(nullptr, err_.c_str());      // the uploader's failure is the job's message (e2s_last_error(NULL))
    }
    // the two handles first -- the caller makes the reads resident on them from ITS thread -- and only then the uploader thread, which
    // works on the handles from then on (a handle is one thread's at a time)
    int create() {
        for (auto& h : h_) if (!h) {
            int rc = e2s_create(p_.device, &h);
            if (rc) { err_ = e2s_last_error(nullptr); return rc; }
            if (src_.flags & 1) (void)e2s_set_upload_staging(h, 1);   // (the source hands out pointers into file mappings)
        }
        return E2S_OK;
    }
    int start() {
        int rc = create();
        if (rc) return rc;
        th_ = std::thread([this]() { run(); });
        return E2S_OK;
    }
    e2s_handle handle(int slot) { return h_[slot]; }
    // waits until chunk r is resident; returns its handle (nullptr + rc on failure)
    int acquire(size_t r, e2s_handle* h) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return rc_ != 0 || ready_[r & 1] == (long)r; });
        if (rc_) return rc_;
        *h = h_[r & 1];
        return E2S_OK;
    }
    void done(size_t r) {
        { std::lock_guard<std::mutex> lk(m_); ready_[r & 1] = -1; }
        cv_.notify_all();
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(m_); quit_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    const std::string& error() const { return err_; }
    double upload_seconds() const { return up_s_; }

  private:
    void fail(int rc, const char* what, e2s_handle h) {
        { std::lock_guard<std::mutex> lk(m_); rc_ = rc ? rc : E2S_ERR_IO; err_ = std::string(what) + ": " + (h ? e2s_last_error(h) : "source"); }
        cv_.notify_all();
    }
    void run() {
        for (size_t r = 0; r < chunks_.size(); ++r) {
            const int slot = (int)(r & 1);
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return quit_ || ready_[slot] == -1; });
                if (quit_) return;
            }
            const Chunk& c = chunks_[r];
            const uint64_t lo = c.lo - c.hl, n = (c.hi - c.lo) + c.hl + c.hr;
            const void *bwt = nullptr, *lcp = nullptr, *psa = nullptr, *aos = nullptr;
            const double t0 = now_s();
            if (src_.get(src_.user, slot, lo, n, mode_ == 1 ? 1 : 0, &bwt, &lcp, &psa, &aos) != 0) { fail(E2S_ERR_IO, "EGSA source", nullptr); return; }
            e2s_shard_desc d{};
            d.global_offset = c.lo; d.n_entries = c.hi - c.lo; d.n_total = p_.n_total; d.halo_left = c.hl; d.halo_right = c.hr;
            d.lcp_bytes = p_.lcp_bytes; d.da_bytes = p_.da_bytes; d.pos_bytes = p_.pos_bytes;
            int rc;
            if (src_.gesa) rc = e2s_upload_gesa(h_[slot], &d, aos);
            else if (mode_ == 1) rc = e2s_upload_bcr(h_[slot], &d, bwt, lcp, psa);
            else if (mode_ == 2) rc = e2s_upload_lcp_bwt(h_[slot], &d, lcp, bwt);
            else rc = e2s_upload_lcp(h_[slot], &d, lcp, ((const uint8_t*)bwt)[c.hl + (c.hi - c.lo) - 1]);   // the BWT byte of the last owned entry (EOF rule)
            if (src_.release) src_.release(src_.user, slot);
            up_s_ += now_s() - t0;
            if (rc) { fail(rc, "upload", h_[slot]); return; }
            { std::lock_guard<std::mutex> lk(m_); ready_[slot] = (long)r; }
            cv_.notify_all();
        }
    }
    e2s_chunk_params p_; e2s_egsa_source src_; const std::vector<Chunk>& chunks_; int mode_;
    e2s_handle h_[2] = {nullptr, nullptr};
    std::thread th_; std::mutex m_; std::condition_variable cv_;
    long ready_[2] = {-1, -1}; bool quit_ = false; int rc_ = 0; std::string err_; double up_s_ = 0;
};

#define RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// pass A: the scan of every chunk, stitched in order; descriptors and statistics of all chunks are kept for a second pass
// (flagged: with the pre-filter inside the scan -- LCP + BWT uploaded -- the clusters it lets through, per chunk, for the per-cluster pass
// that follows once max_clust_length is known; full: a chunk only the four-column path handles)
struct Flagged { std::vector<uint64_t> st; std::vector<uint16_t> ln; bool full = false; };
struct ScanPass {
    std::vector<e2s_boundary> all; std::vector<e2s_scan_stats> stats; std::vector<e2s_hist> hists;
    uint64_t n_closed = 0, n_records = 0; double up_s = 0;
};
int scan_pass(const e2s_chunk_params& p, const e2s_egsa_source& src, const std::vector<Chunk>& chunks, e2s_byte_sink sink, void* user, ScanPass& out,
              int arm_mcov = -1, std::vector<Flagged>* flagged = nullptr) {
    const int nr = (int)chunks.size();
    out.all.assign(nr, e2s_boundary{}); out.stats.assign(nr, e2s_scan_stats{}); out.hists.assign(nr, e2s_hist{});
    if (flagged) flagged->assign((size_t)nr, Flagged{});
    ChunkPipe pipe(p, src, chunks, flagged ? 2 : 0);
    RC(pipe.start());
    for (int r = 0; r < nr; ++r) {
        e2s_handle h = nullptr;
        RC(pipe.acquire((size_t)r, &h));
        if (flagged) RC(e2s_prefilter_arm(h, arm_mcov));
        RC(e2s_cluster_scan(h, p.k, p.min_len, 0));
        RC(e2s_boundary_desc(h, &out.all[r]));
        RC(e2s_scan_stats_desc(h, &out.stats[r]));
        RC(e2s_stitch(h, out.all.data(), nr, r));                // (reads the descriptors of ranks <= r only)
        uint64_t nc = 0, nrec = 0;
        RC(e2s_cluster_counts(h, &nc, &nrec));
        RC(e2s_cluster_hist(h, &out.hists[r]));
        out.n_closed += nc;
        if (flagged) {
            const uint64_t* fs = nullptr; const uint16_t* fl = nullptr; uint64_t fn = 0; int full = 0;
            RC(e2s_internal_flagged_list(h, &fs, &fl, &fn, &full));
            Flagged& F = (*flagged)[(size_t)r];
            F.full = full != 0;
            if (!F.full) { F.st.assign(fs, fs + fn); F.ln.assign(fl, fl + fn); }
        }
        if (sink) { uint64_t w = 0; RC(e2s_stream_clusters(h, sink, user, &w)); out.n_records += w; } else out.n_records += nrec;
        pipe.done((size_t)r);
    }
    pipe.stop();
    out.up_s = pipe.upload_seconds();
    return E2S_OK;
}

// the tail of one chunk: GPU tail streamed to the sink, or -- where only the host tail reproduces the reference -- the packed
// candidates appended to host buffers for one e2s_snp_from_packed at the end
constexpr int E2S_SWITCH_TAILS = -1000;                         // internal: the GPU tail stopped applying after it had produced events
struct TailState {
    e2s_tail_params tp; uint64_t events = 0, variants = 0, cands = 0;
    bool host_mode = false; std::vector<uint8_t> stream; std::vector<uint32_t> off8; uint32_t pw = 4;
};
int chunk_tail(e2s_handle h, uint64_t n_cand, e2s_byte_sink snp_sink, void* snp_user, TailState& ts) {
    ts.cands += n_cand;
    if (!n_cand) return E2S_OK;
    if (!ts.host_mode) {
        uint64_t nv = 0, ne = 0, nb = 0;
        int rc = e2s_call_variants(h, &ts.tp, 1 + ts.events, &nv, &ne, &nb);
        if (rc == E2S_OK) {
            ts.events += ne; ts.variants += nv;
            return nb ? e2s_stream_snp(h, snp_sink, snp_user) : E2S_OK;
        }
        if (rc != E2S_ERR_STATE) return rc;
        if (ts.events) return E2S_SWITCH_TAILS;                  // (earlier chunks already wrote their events: the caller starts over, if it can)
        ts.host_mode = true;
    }
    const void* st = nullptr; const uint32_t* of = nullptr; uint64_t n = 0, bytes = 0; uint32_t pw = 4;
    RC(e2s_candidates_packed(h, &st, &of, &n, &bytes, &pw));
    const uint64_t base8 = ts.stream.size() / 8;
    if (base8 + bytes / 8 >= (1ull << 32)) return E2S_ERR_CAPACITY;
    ts.stream.insert(ts.stream.end(), (const uint8_t*)st, (const uint8_t*)st + bytes);
    for (uint64_t i = 0; i < n; ++i) ts.off8.push_back((uint32_t)(base8 + of[i]));
    ts.pw = pw;
    return E2S_OK;
}
int finish_tail(TailState& ts, const char* fasta_path, e2s_byte_sink snp_sink, void* snp_user, const char* out_path) {
    if (ts.cands == 0) return E2S_ERR_REF_CRASH;                 // zero candidates: the reference crashes before creating the file
    if (!ts.host_mode) {
        // candidates, but not one event (every candidate lost its supports or lies beyond -v): the reference has opened its output by then
        // and leaves it EMPTY -- the lazily created file of the chunks' sink never came into being
        if (out_path && ts.events == 0) { FILE* f = fopen(out_path, "w"); if (!f) return E2S_ERR_IO; fclose(f); }
        return E2S_OK;
    }
    if (!fasta_path) return E2S_ERR_STATE;
    ts.off8.push_back((uint32_t)(ts.stream.size() / 8));
    std::string tmp = out_path ? std::string(out_path) : std::string("/tmp/e2s_tail_XXXXXX");
    if (!out_path) { const int fd = mkstemp(&tmp[0]); if (fd < 0) return E2S_ERR_IO; close(fd); }
    uint64_t nv = 0, ne = 0;
    int rc = e2s_snp_from_packed(ts.stream.data(), ts.off8.data(), ts.off8.size() - 1, ts.pw, &ts.tp, fasta_path, tmp.c_str(), &nv, &ne);
    ts.variants += nv; ts.events += ne;                          // (chunks before the switch to the host tail had no events, but may have kept variants)
    if (rc == E2S_OK && !out_path) {                             // the text goes to the sink
        FILE* f = fopen(tmp.c_str(), "rb");
        if (!f) rc = E2S_ERR_IO;
        else {
            std::vector<char> buf(1 << 20);
            size_t got;
            while (rc == E2S_OK && (got = fread(buf.data(), 1, buf.size(), f)) > 0) if (snp_sink(snp_user, buf.data(), got)) rc = E2S_ERR_IO;
            fclose(f);
        }
    }
    if (!out_path) unlink(tmp.c_str());
    return rc;
}

int file_sink(void* user, const void* data, size_t bytes) { return fwrite(data, 1, bytes, (FILE*)user) == bytes ? 0 : 1; }
// a file that comes into being with its first byte (where the reference crashes before creating it, nothing must be left behind)
struct LazyFile {
    FILE* f; const char* path;
    ~LazyFile() { if (f) fclose(f); }
    void close() { if (f) { fclose(f); f = nullptr; } }
};
int lazy_file_sink(void* user, const void* data, size_t bytes) {
    LazyFile* c = (LazyFile*)user;
    if (!c->f) { c->f = fopen(c->path, "w"); if (!c->f) return 1; }
    return file_sink(c->f, data, bytes);
}

void fill_report(e2s_job_report* rep, const ScanPass* sp, const TailState* ts, int mcl, uint64_t n_clust, uint64_t n_bases, size_t n_chunks, double up_s, double t0) {
    if (!rep) return;
    if (sp) {
        rep->n_closed = sp->n_closed; rep->n_records = sp->n_records;
        memset(&rep->hist, 0, sizeof(rep->hist));
        for (const e2s_hist& hs : sp->hists) {
            for (int i = 0; i <= E2S_MAX_CLUST_LEN; ++i) rep->hist.freq[i] += hs.freq[i];
            rep->hist.n_records += hs.n_records; rep->hist.n_bases += hs.n_bases;
            if (hs.n_records) rep->hist.last_len = hs.last_len;
        }
    }
    if (ts) { rep->n_candidates = ts->cands; rep->n_variants = ts->variants; rep->n_events = ts->events; }
    rep->max_clust_length = mcl; rep->n_clust = n_clust; rep->n_bases = n_bases; rep->n_chunks = (int32_t)n_chunks;
    rep->seconds_upload = up_s; rep->seconds_total = now_s() - t0;
}

}  // namespace

extern "C" int e2s_ebwt2clust_chunked(const e2s_chunk_params* p, const e2s_egsa_source* src, e2s_byte_sink clusters_sink, void* sink_user, e2s_job_report* rep) {
    if (!p || !src || !src->get || !p->n_total) return E2S_ERR_ARG;
    const double t0 = now_s();
    if (rep) memset(rep, 0, sizeof(*rep));
    const std::vector<Chunk> chunks = plan_chunks(p->n_total, p->chunk_entries, 1);
    ScanPass sp;
    RC(scan_pass(*p, *src, chunks, clusters_sink, sink_user, sp));
    fill_report(rep, &sp, nullptr, 0, 0, 0, chunks.size(), sp.up_s, t0);
    return E2S_OK;
}

// The per-cluster pass of one chunk over clusters GATHERED on the host (the 2-bytes-per-entry job): the chunk's flagged clusters with a
// length the statistics allow, plus the records the EOF rule adds on the last chunk; their entries copied from all four columns of the
// source into compact columns, the EOF phantom entry (index n_total) worked out as read_el leaves it.
struct CompactJob {                                               // per job (two jobs may run on two threads): the gather buffers, reused from chunk to chunk, and
    std::vector<uint8_t> cb, cl, cp; std::vector<uint32_t> cd; std::vector<uint64_t> off;    // where the second phase's time goes (E2S_CLI_TIMING)
    double gather_s = 0, compact_s = 0;
};
int compact_chunk_pass(CompactJob& J, e2s_handle hc, const e2s_chunk_params& p, const e2s_c2s_options& o, const e2s_egsa_source& src, const Flagged& F,
                       const e2s_stitch_plan_t& plan, int mcl, uint64_t* n_cand) {
    *n_cand = 0;
    const double t_in = now_s();
    const int len_lo = std::max(std::max(2 * o.mcov_out, p.min_len), -1), len_hi = std::min(mcl, (int)E2S_MAX_CLUST_LEN);
    std::vector<uint64_t> st; std::vector<uint16_t> ln;
    for (size_t i = 0; i < F.st.size(); ++i) if ((int)F.ln[i] >= len_lo && (int)F.ln[i] <= len_hi) { st.push_back(F.st[i]); ln.push_back(F.ln[i]); }
    for (uint32_t t = 0; t < plan.n_tail; ++t)
        if ((int)plan.tail_len[t] >= len_lo && (int)plan.tail_len[t] <= len_hi) { st.push_back(plan.tail_start[t]); ln.push_back(plan.tail_len[t]); }
    e2s_c2s_params prm{};
    prm.nr_reads1 = o.nr_reads1; prm.k_left = o.k_left; prm.k_right = o.k_right; prm.mcov_out = o.mcov_out;
    prm.consensus_reads = o.consensus_reads; prm.max_clust_length = mcl; prm.min_len = p.min_len;
    if (st.empty()) return e2s_internal_find_candidates_compact(hc, &prm, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, n_cand);
    uint64_t gmin = ~0ull, gmax = 0, total = 0;
    for (size_t i = 0; i < st.size(); ++i) { gmin = std::min(gmin, st[i]); gmax = std::max(gmax, st[i] + ln[i]); total += ln[i]; }
    if (gmax > p.n_total + 1) return E2S_ERR_INPUT;               // a record past EOF (only index n_total, the phantom, exists beyond the array)
    const uint64_t real_hi = std::min<uint64_t>(gmax, p.n_total);                // (index n_total is the EOF phantom)
    const void *bwt = nullptr, *lcp = nullptr, *psa = nullptr, *aos = nullptr;
    if (real_hi > gmin && src.get(src.user, 2, gmin, real_hi - gmin, 1, &bwt, &lcp, &psa, &aos) != 0) return E2S_ERR_IO;      // (slot 2: see e2s_egsa_source)
    std::vector<uint8_t>&cb = J.cb, &cl = J.cl, &cp = J.cp; std::vector<uint32_t>& cd = J.cd; std::vector<uint64_t>& off = J.off;
    cb.resize(total); cl.resize(total); cp.resize(total); cd.resize(total); off.resize(st.size());
    const uint8_t* B = (const uint8_t*)bwt; const uint8_t* L = (const uint8_t*)lcp; const uint8_t* P = (const uint8_t*)psa;
    uint32_t ph[4] = {0, 0, 0, 0};
    if (gmax > p.n_total) {                                       // the EOF phantom, from the last real entry (which the range holds: the cluster reaches it)
        const uint64_t j = p.n_total - 1 - gmin;
        uint32_t d; memcpy(&d, P + 5 * j + 1, 4);
        e2s_internal_phantom(p.lcp_bytes, p.da_bytes, p.pos_bytes, 0, d, P[5 * j], L[j], B[j], ph);
    }
    uint64_t at = 0;
    for (size_t i = 0; i < st.size(); ++i) { off[i] = at; at += ln[i]; }
    // the gather itself: worker threads over the clusters (~10^6 entries per 3x10^8-entry chunk)
    uint8_t* const pb = cb.data(); uint8_t* const pl = cl.data(); uint8_t* const pp = cp.data(); uint32_t* const pd = cd.data(); const uint64_t* const po = off.data();
    auto work = [&, pb, pl, pp, pd, po](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const uint64_t a2 = po[i];
            const uint64_t real = st[i] + ln[i] <= p.n_total ? ln[i] : (st[i] < p.n_total ? p.n_total - st[i] : 0);
            const uint64_t j0 = st[i] - gmin;
            memcpy(pb + a2, B + j0, real); memcpy(pl + a2, L + j0, real);
            for (uint64_t q = 0; q < real; ++q) { pp[a2 + q] = P[5 * (j0 + q)]; uint32_t d; memcpy(&d, P + 5 * (j0 + q) + 1, 4); pd[a2 + q] = d; }
            for (uint64_t q = real; q < ln[i]; ++q) { pd[a2 + q] = ph[0]; pp[a2 + q] = (uint8_t)ph[1]; pl[a2 + q] = (uint8_t)ph[2]; pb[a2 + q] = (uint8_t)ph[3]; }
        }
    };
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(8u, std::max(1u, std::thread::hardware_concurrency())), st.size() / 4096));
    if (T == 1) work(0, st.size());
    else {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back(work, st.size() * t / T, st.size() * (t + 1) / T);
        work(0, st.size() / T);
        for (auto& x : th) x.join();
    }
    if (src.release && real_hi > gmin) src.release(src.user, 2);
    const double t_g = now_s();
    const int rc = e2s_internal_find_candidates_compact(hc, &prm, st.size(), st.data(), off.data(), ln.data(), cb.data(), cl.data(), cp.data(), cd.data(), total, gmin,
                                                        gmax - gmin, n_cand);
    J.gather_s += t_g - t_in; J.compact_s += now_s() - t_g;
    return rc;
}

// One shard's per-cluster pass over GATHERED clusters (include/e2s_hip.h): what the chunked job does per chunk, for callers that drive
// their shards themselves (bin/ebwt2snp_mgpu: a rank uploads LCP + BWT only).
extern "C" int e2s_find_candidates_gathered(e2s_handle h_scan, e2s_handle h_pass, const e2s_chunk_params* p, const e2s_c2s_options* o, const e2s_egsa_source* src,
                                            const e2s_stitch_plan_t* plan, int max_clust_length, uint64_t* n_cand, uint64_t* n_gathered) {
    if (!h_scan || !h_pass || !p || !o || !src || !src->get || !plan || !n_cand) return E2S_ERR_ARG;
    if (src->gesa || p->lcp_bytes != 1 || p->da_bytes != 4 || p->pos_bytes != 1) return E2S_ERR_STATE;      // (the compact pass reads BCR files of the default widths)
    const uint64_t* fs = nullptr; const uint16_t* fl = nullptr; uint64_t fn = 0; int full = 0;
    int rc = e2s_internal_flagged_list(h_scan, &fs, &fl, &fn, &full);
    if (rc) return rc;
    if (full) { e2s_internal_set_error(h_scan, "the pre-filter's list is not usable for this shard (a window too dense, a wrapped length): upload all four columns"); return E2S_ERR_STATE; }
    Flagged F;
    F.st.assign(fs, fs + fn); F.ln.assign(fl, fl + fn);
    CompactJob cj;
    rc = compact_chunk_pass(cj, h_pass, *p, *o, *src, F, *plan, max_clust_length, n_cand);
    if (n_gathered) { uint64_t t = 0; for (uint16_t l : F.ln) t += l; *n_gathered = t; }
    return rc;
}

extern "C" int e2s_ebwt2snp_chunked(const e2s_chunk_params* p, const e2s_c2s_options* o, const e2s_egsa_source* src, const char* fasta_path,
                                    const void* d_text, const uint64_t* d_start, const uint32_t* d_len, uint64_t n_reads, uint32_t max_len,
                                    e2s_byte_sink clusters_sink, void* clusters_user, e2s_byte_sink snp_sink, void* snp_user, e2s_job_report* rep) {
    if (!p || !o || !src || !src->get || !p->n_total || !snp_sink || (!fasta_path && !d_text)) return E2S_ERR_ARG;
    const double t0 = now_s();
    if (rep) memset(rep, 0, sizeof(*rep));
    const std::vector<Chunk> chunks = plan_chunks(p->n_total, p->chunk_entries, 1);
    const int nr = (int)chunks.size();
    // BCR files of the default widths: ONE streaming pass over 2 of the 7 bytes per entry -- LCP + BWT, the scan with clust2snp's
    // pre-filter inside; DA and positions are read only for the clusters it lets through (~0.5 % of the entries), gathered on the host
    // once statistics() has said how long a cluster may be.  (E2S_CHUNK_TWO_PASS=1, other widths, `.gesa`: pass A = the LCP column,
    // pass B = all four columns again.)
    static const int two_pass_env = e2s_env_dev("E2S_CHUNK_TWO_PASS") ? atoi(e2s_env_dev("E2S_CHUNK_TWO_PASS")) : 0;
    const bool slim = !two_pass_env && !src->gesa && p->lcp_bytes == 1 && p->da_bytes == 4 && p->pos_bytes == 1;
    // the handle of the second phase (slim), created up front: a FASTA is indexed and made resident on it by a thread of its own WHILE the
    // streaming pass runs (the reads are not needed before the tail; the handle is that thread's alone until it is joined)
    e2s_handle hc = nullptr, hf = nullptr;
    struct Guard { e2s_handle& a; e2s_handle& b; ~Guard() { if (a) e2s_destroy(a); if (b) e2s_destroy(b); } } guard{hc, hf};
    std::thread reads_thread; int reads_rc = E2S_OK;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{reads_thread};      // (joined before the handles go)
    if (slim && fasta_path) {
        RC(e2s_create(p->device, &hc));
        reads_thread = std::thread([&reads_rc, hc, fasta_path]() { reads_rc = e2s_upload_reads_fasta(hc, fasta_path, nullptr); });
    }
    // pass A: ebwt2clust: records out, descriptors and statistics of every chunk kept (slim: and the clusters the pre-filter lets through)
    ScanPass sp;
    std::vector<Flagged> flagged;
    RC(scan_pass(*p, *src, chunks, clusters_sink, clusters_user, sp, slim ? o->mcov_out : -1, slim ? &flagged : nullptr));
    int mcl = 0; uint64_t n_clust = 0, n_bases = 0;
    int rc = e2s_auto_max_clust_length(sp.hists.data(), nr, o->mcov_out, o->pval, &mcl, &n_clust, &n_bases);
    if (rc) { fill_report(rep, &sp, nullptr, 0, 0, 0, chunks.size(), sp.up_s, t0); return rc; }
    TailState ts;
    ts.tp = e2s_tail_params{o->k_left, o->k_right, o->max_gap, o->max_err};
    double up_b = 0;
    const double t_phase2 = now_s();
    CompactJob cj;
    auto reads_on = [&](e2s_handle h0, e2s_handle h1) -> int {
        if (fasta_path) {
            int r2 = e2s_upload_reads_fasta(h0, fasta_path, nullptr);
            if (r2 != E2S_OK && r2 != E2S_ERR_INPUT && r2 != E2S_ERR_IO) return r2;      // (an irregular / unreadable FASTA: the host tail decides)
            if (r2 == E2S_OK && h1) RC(e2s_share_reads(h1, h0));
        } else {
            RC(e2s_attach_reads(h0, d_text, d_start, d_len, n_reads, max_len));
            if (h1) RC(e2s_attach_reads(h1, d_text, d_start, d_len, n_reads, max_len));
        }
        return E2S_OK;
    };
    if (slim) {
        // the per-cluster pass and the tail, chunk by chunk, over gathered clusters (a chunk the pre-filter could not handle: all four columns)
        if (reads_thread.joinable()) {
            reads_thread.join();
            if (reads_rc != E2S_OK && reads_rc != E2S_ERR_INPUT && reads_rc != E2S_ERR_IO) return reads_rc;      // (an irregular / unreadable FASTA: the host tail decides)
        } else {
            RC(e2s_create(p->device, &hc));
            RC(reads_on(hc, nullptr));
        }
        for (int r = 0; r < nr; ++r) {
            uint64_t n_cand = 0;
            e2s_handle ht = hc;
            if (flagged[(size_t)r].full) {
                if (!hf) { RC(e2s_create(p->device, &hf)); RC(e2s_share_reads(hf, hc)); if (src->flags & 1) (void)e2s_set_upload_staging(hf, 1); }
                const Chunk& c = chunks[(size_t)r];
                const uint64_t lo = c.lo - c.hl, n = (c.hi - c.lo) + c.hl + c.hr;
                const void *bwt = nullptr, *lcp = nullptr, *psa = nullptr, *aos = nullptr;
                const double u0 = now_s();
                if (src->get(src->user, 2, lo, n, 1, &bwt, &lcp, &psa, &aos) != 0) return E2S_ERR_IO;
                e2s_shard_desc d{};
                d.global_offset = c.lo; d.n_entries = c.hi - c.lo; d.n_total = p->n_total; d.halo_left = c.hl; d.halo_right = c.hr;
                d.lcp_bytes = p->lcp_bytes; d.da_bytes = p->da_bytes; d.pos_bytes = p->pos_bytes;
                rc = e2s_upload_bcr(hf, &d, bwt, lcp, psa);
                if (src->release) src->release(src->user, 2);
                up_b += now_s() - u0;
                if (rc) return rc;
                RC(e2s_prefilter_arm(hf, o->mcov_out));
                RC(e2s_cluster_scan(hf, p->k, p->min_len, 0));
                RC(e2s_stitch(hf, sp.all.data(), nr, r));
                e2s_c2s_params prm{};
                prm.nr_reads1 = o->nr_reads1; prm.k_left = o->k_left; prm.k_right = o->k_right; prm.mcov_out = o->mcov_out;
                prm.consensus_reads = o->consensus_reads; prm.max_clust_length = mcl; prm.min_len = p->min_len;
                RC(e2s_find_candidates(hf, &prm, (uint64_t)r << 40, &n_cand));
                ht = hf;
            } else {
                e2s_stitch_plan_t plan;
                if (e2s_stitch_plan(sp.all.data(), nr, r, p->n_total, p->k, p->lcp_bytes, &plan) != E2S_OK) return E2S_ERR_INPUT;
                RC(compact_chunk_pass(cj, hc, *p, *o, *src, flagged[(size_t)r], plan, mcl, &n_cand));
            }
            rc = chunk_tail(ht, n_cand, snp_sink, snp_user, ts);
            if (rc == E2S_SWITCH_TAILS) return E2S_ERR_STATE;         // (the sink has the earlier chunks' events already: see include/e2s_hip.h)
            if (rc) return rc;
        }
    } else {
        // pass B: clust2snp: all four columns, the scan again with the pre-filter inside, the per-cluster pass and the tail
        ChunkPipe pipe(*p, *src, chunks, 1);
        RC(pipe.create());
        RC(reads_on(pipe.handle(0), pipe.handle(1)));
        RC(pipe.start());
        for (int r = 0; r < nr; ++r) {
            e2s_handle h = nullptr;
            RC(pipe.acquire((size_t)r, &h));
            RC(e2s_prefilter_arm(h, o->mcov_out));
            RC(e2s_cluster_scan(h, p->k, p->min_len, 0));
            RC(e2s_stitch(h, sp.all.data(), nr, r));                 // the descriptors of pass A: the same array, the same cuts
            e2s_c2s_params prm{};
            prm.nr_reads1 = o->nr_reads1; prm.k_left = o->k_left; prm.k_right = o->k_right; prm.mcov_out = o->mcov_out;
            prm.consensus_reads = o->consensus_reads; prm.max_clust_length = mcl; prm.min_len = p->min_len;
            uint64_t n_cand = 0;
            RC(e2s_find_candidates(h, &prm, (uint64_t)r << 40, &n_cand));
            rc = chunk_tail(h, n_cand, snp_sink, snp_user, ts);
            if (rc == E2S_SWITCH_TAILS) return E2S_ERR_STATE;         // (the sink has the earlier chunks' events already: see include/e2s_hip.h)
            if (rc) return rc;
            pipe.done((size_t)r);
        }
        pipe.stop();
        up_b = pipe.upload_seconds();
    }
    rc = finish_tail(ts, fasta_path, snp_sink, snp_user, nullptr);
    fill_report(rep, &sp, &ts, mcl, n_clust, n_bases, chunks.size(), sp.up_s + up_b, t0);
    if (getenv("E2S_CLI_TIMING"))
        fprintf(stderr, "[timing]   chunked job: %.1f ms in all; streaming pass %.1f ms (uploads %.1f ms); second phase: gather %.1f ms, per-cluster pass %.1f ms, the rest (tail, sinks) %.1f ms\n",
                (now_s() - t0) * 1e3, (t_phase2 - t0) * 1e3, sp.up_s * 1e3, cj.gather_s * 1e3, cj.compact_s * 1e3, (now_s() - t_phase2 - cj.gather_s - cj.compact_s) * 1e3);
    return rc;
}

extern "C" int e2s_clust2snp_chunked(const e2s_chunk_params* p, const e2s_c2s_options* o, int max_clust_length, const e2s_egsa_source* src,
                                     const void* recs, uint64_t n_records, const char* fasta_path, const char* out_path, e2s_job_report* rep) {
    if (!p || !o || !src || !src->get || !p->n_total || !fasta_path || !out_path || (!recs && n_records)) return E2S_ERR_ARG;
    const double t0 = now_s();
    if (rep) memset(rep, 0, sizeof(*rep));
    // records are handled where they START; the cluster must be resident: right halo = the largest length that matters + the EOF phantom
    const std::vector<Chunk> chunks = plan_chunks(p->n_total, p->chunk_entries, E2S_MAX_CLUST_LEN + 1);
    const int nr = (int)chunks.size();
    auto start_of = [&](uint64_t i) { uint64_t s; memcpy(&s, (const uint8_t*)recs + i * 10, 8); return s; };
    auto first_at_or_after = [&](uint64_t pos) {                  // records are ascending (H16: checked on the GPU per chunk)
        uint64_t a = 0, b = n_records;
        while (a < b) { const uint64_t m = (a + b) / 2; if (start_of(m) < pos) a = m + 1; else b = m; }
        return a;
    };
    TailState ts;
    double up_s = 0;
    // BCR files of the default widths: 2 of the 7 bytes per entry go up (the record pre-filter reads the BWT only), the clusters it lets
    // through are gathered on the host from all four columns (as in e2s_ebwt2snp_chunked)
    static const int two_pass_env = e2s_env_dev("E2S_CHUNK_TWO_PASS") ? atoi(e2s_env_dev("E2S_CHUNK_TWO_PASS")) : 0;
    const bool slim = !two_pass_env && !src->gesa && p->lcp_bytes == 1 && p->da_bytes == 4 && p->pos_bytes == 1;
    // one pass over the chunks; host_only: the host tail from the first chunk on (the second attempt, below)
    auto pass = [&](bool host_only) -> int {
        ts = TailState{};
        ts.tp = e2s_tail_params{o->k_left, o->k_right, o->max_gap, o->max_err};
        ts.host_mode = host_only;
        ChunkPipe pipe(*p, *src, chunks, slim ? 2 : 1);
        e2s_handle hc = nullptr;
        struct Guard { e2s_handle& a; ~Guard() { if (a) e2s_destroy(a); } } guard{hc};
        RC(pipe.create());
        if (slim) {                                               // the compact per-cluster pass and the tail run on a handle of their own
            RC(e2s_create(p->device, &hc));
            RC(pipe.start());                                     // (the first chunk uploads into the pipe's handles while this thread indexes the
            if (!host_only) {                                     //  FASTA and makes the reads resident on hc: 0.35 s of the full c3 files' 1.0 s)
                int rc = e2s_upload_reads_fasta(hc, fasta_path, nullptr);
                if (rc != E2S_OK && rc != E2S_ERR_INPUT && rc != E2S_ERR_IO) { pipe.stop(); return rc; }
            }
            LazyFile out{nullptr, out_path};
            CompactJob cj;
            for (int r = 0; r < nr; ++r) {
                e2s_handle h = nullptr;
                RC(pipe.acquire((size_t)r, &h));
                const uint64_t i0 = first_at_or_after(chunks[r].lo), i1 = r == nr - 1 ? n_records : first_at_or_after(chunks[r].hi);
                RC(e2s_load_clusters(h, (const uint8_t*)recs + i0 * 10, i1 - i0));
                Flagged F;
                const uint64_t* fs = nullptr; const uint16_t* fl = nullptr; uint64_t fn = 0;
                RC(e2s_internal_record_prefilter(h, o->mcov_out, max_clust_length, &fs, &fl, &fn));
                F.st.assign(fs, fs + fn); F.ln.assign(fl, fl + fn);
                pipe.done((size_t)r);                             // (the next chunk uploads while this one's clusters are gathered and passed)
                uint64_t n_cand = 0;
                e2s_chunk_params p0 = *p; p0.min_len = 0;
                RC(compact_chunk_pass(cj, hc, p0, *o, *src, F, e2s_stitch_plan_t{}, max_clust_length, &n_cand));
                RC(chunk_tail(hc, n_cand, lazy_file_sink, &out, ts));
            }
            pipe.stop();
            up_s += pipe.upload_seconds();
            out.close();
            return finish_tail(ts, fasta_path, nullptr, nullptr, out_path);
        }
        if (!host_only) {
            int rc = e2s_upload_reads_fasta(pipe.handle(0), fasta_path, nullptr);
            if (rc != E2S_OK && rc != E2S_ERR_INPUT && rc != E2S_ERR_IO) return rc;
            if (rc == E2S_OK) RC(e2s_share_reads(pipe.handle(1), pipe.handle(0)));
        }
        RC(pipe.start());
        LazyFile out{nullptr, out_path};
        for (int r = 0; r < nr; ++r) {
            e2s_handle h = nullptr;
            RC(pipe.acquire((size_t)r, &h));
            const uint64_t i0 = first_at_or_after(chunks[r].lo), i1 = r == nr - 1 ? n_records : first_at_or_after(chunks[r].hi);
            RC(e2s_load_clusters(h, (const uint8_t*)recs + i0 * 10, i1 - i0));
            e2s_c2s_params prm{};
            prm.nr_reads1 = o->nr_reads1; prm.k_left = o->k_left; prm.k_right = o->k_right; prm.mcov_out = o->mcov_out;
            prm.consensus_reads = o->consensus_reads; prm.max_clust_length = max_clust_length; prm.min_len = 0;
            uint64_t n_cand = 0;
            RC(e2s_find_candidates(h, &prm, i0, &n_cand));
            RC(chunk_tail(h, n_cand, lazy_file_sink, &out, ts));
            pipe.done((size_t)r);
        }
        pipe.stop();
        up_s += pipe.upload_seconds();
        out.close();
        return finish_tail(ts, fasta_path, nullptr, nullptr, out_path);
    };
    int rc = pass(false);
    if (rc == E2S_SWITCH_TAILS) {
        // a later chunk needs the host tail (a read rank past the last read) after earlier ones wrote their events with the GPU tail: the
        // host tail works on ALL candidates at once, so the partial file goes and the pass runs again, collecting the packed candidates
        unlink(out_path);
        if (getenv("E2S_CLI_TIMING")) fprintf(stderr, "[timing]   the GPU tail stopped applying after %llu events: the pass runs again with the host tail\n", (unsigned long long)ts.events);
        rc = pass(true);
    }
    fill_report(rep, nullptr, &ts, max_clust_length, 0, 0, chunks.size(), up_s, t0);
    return rc;
}

extern "C" int e2s_hist_of_records(const void* recs, uint64_t n_records, e2s_hist* out) {
    if (!out || (!recs && n_records)) return E2S_ERR_ARG;
    memset(out, 0, sizeof(*out));
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), n_records / 1000000));
    std::vector<e2s_hist> part(T);
    std::vector<int> bad(T, 0);
    std::vector<std::thread> th;
    auto work = [&](unsigned t) {
        e2s_hist& hs = part[t];
        memset(&hs, 0, sizeof(hs));
        const uint64_t a = n_records * t / T, b = n_records * (t + 1) / T;
        const uint8_t* p = (const uint8_t*)recs;
        uint64_t prev_end = 0;
        if (a) { uint64_t s; uint16_t l; memcpy(&s, p + (a - 1) * 10, 8); memcpy(&l, p + (a - 1) * 10 + 8, 2); prev_end = s + l; }
        for (uint64_t i = a; i < b; ++i) {
            uint64_t s; uint16_t l;
            memcpy(&s, p + i * 10, 8); memcpy(&l, p + i * 10 + 8, 2);
            if (i && s < prev_end) bad[t] = 1;                     // records must be ascending and disjoint (H16: the lock-step cursor of clust2snp.cpp:818-833)
            prev_end = s + l;
            hs.n_records++; hs.n_bases += l;
            if (l <= E2S_MAX_CLUST_LEN) hs.freq[l]++;
        }
    };
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (unsigned t = 0; t < T; ++t) {
        for (int i = 0; i <= E2S_MAX_CLUST_LEN; ++i) out->freq[i] += part[t].freq[i];
        out->n_records += part[t].n_records; out->n_bases += part[t].n_bases;
    }
    if (n_records) { uint16_t l; memcpy(&l, (const uint8_t*)recs + (n_records - 1) * 10 + 8, 2); out->last_len = l; }
    for (unsigned t = 0; t < T; ++t) if (bad[t]) return E2S_ERR_INPUT;
    return E2S_OK;
}
