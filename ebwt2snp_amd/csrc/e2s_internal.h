// e2s_internal.h -- shared between the kernels (e2s_kernels.hip) and the C-ABI layer (e2s_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/e2s_hip.h"

#define E2S_SCAN_BLOCK 256
#define E2S_SCAN_ROWS 4
#ifndef E2S_SPAN_BYTES
#define E2S_SPAN_BYTES 64        // bytes of LCP per thread and tile in the segmented scan
#endif
#define E2S_MAX_CLUST_LEN_K 150
#define E2S_STATS_STRIDE 160     // the scan's statistics: [151 freq | n_records | n_bases | ...] in 64-bit words,
#define E2S_STATS_REPL 8         // ... E2S_STATS_REPL copies of them that the scan's waves add to (range b: copy b mod 8), summed by k_seg_fix
#define E2S_SEGFIX_MAX 32768     // segments (= ranges of the scan) k_seg_fix ties together
#ifndef E2S_SCAN_RANGE_TILES
#define E2S_SCAN_RANGE_TILES 16  // a range of the scan holds at least this many tiles where the shard is large enough for several rounds (seg_scan_launch)
#endif

// device-side error bits
#define E2S_DEVERR_REMOTE 2u   // a cluster's entries are not resident in this shard
#define E2S_DEVERR_SEGFULL 8u  // an output segment of the segmented scan overflowed (rerun with larger segments)
#define E2S_DEVERR_DENSE  4u   // a 4096-entry window holds more clusters than the LDS slots (fall back)

// a.flags[] slots written by k_shard_flags
#define E2S_FLAG_BEGINS_INSIDE 0
#define E2S_FLAG_OPEN_AFTER 1
#define E2S_FLAG_END_BEFORE_LAST 2

struct ScanArgs {
    const void* lcp;          // first provided entry
    uint64_t n_prov;          // provided entries
    uint64_t prov_lo;         // global SA index of the first provided entry
    uint32_t prov_is_global0; // prov_lo == 0
    uint64_t own_lo;          // local index of the first owned entry
    uint64_t end_hi;          // ends are emitted for local indices [own_lo, end_hi)   (global j <= n_total-2)
    uint64_t start_hi;        // starts for [own_lo, start_hi)                          (= local end of owned range)
    uint32_t k_u32;           // (uint32_t)k
    uint32_t kb;              // k broadcast to every X-byte lane of a word
    uint32_t hge;             // Swar<X>::H, or 0 if k exceeds the lane range (never >= k)
    uint64_t n_tiles;
    // outputs
    uint64_t* starts;         // [cap + 1]
    uint16_t* lens;           // [cap + 1]
    uint64_t cap;
    uint64_t* total_ends;
    uint64_t* tile_fix;       // [n_tiles], preset to ~0: rank of the tile's record whose start lies in an earlier tile
    uint64_t* tile_fix_end;   // [n_tiles]: that record's end position
    uint64_t* first_end_pos;  // SA position of the shard's first end (preset ~0)
    // MODE 1 / 2
    uint32_t* tile_count;
    const uint64_t* tile_prefix;
    uint32_t* err;
    uint32_t* flags;
    uint32_t* tailv;
};

// ---- segmented scan (scan_mode 0): each persistent block owns a contiguous range of tiles and writes the
// records that END in it into its own output segment; nothing is exchanged between blocks during the scan.
struct SegDesc {                 // per block, written by k_scan_seg
    uint32_t count;              // records in the segment
    uint32_t first_unknown;      // record 0's start lies before the block's range (cluster continues from earlier)
    uint32_t open_after;         // a cluster is open after the block's last entry
    uint32_t overflow;           // the segment was too small
    uint64_t open_start;         // its start if it lies in this block's range, else ~0
    uint64_t first_end_pos;      // SA position of the block's first end (valid if count > 0)
    uint32_t kept;               // records append_entry keeps (record 0 not counted while its start is unknown: k_seg_fix adds it)
    // read back by the range's own wave when it ends (its segment's tail is in the caches; k_seg_fix's one thread paid a chain of
    // dependent round trips for the same values): 1 + index of the LAST record append_entry keeps among those whose length the scan
    // knew (0: none) and its length; the stored start of the range's last record; the length of its record 0
    uint32_t last_kept1;
    uint32_t last_kept_len;
    uint32_t last_start32;
    uint64_t first_start;        // record 0's start where it lies before the block's range (k_seg_fix, e2s_stitch)
    uint32_t first_len16;
    uint32_t reserved;
};

struct SegArgs {
    const void* lcp; uint64_t n_prov, prov_lo; uint32_t prov_is_global0;
    uint64_t own_lo, end_hi, start_hi;
    uint32_t k_u32, kb, hge;
    uint64_t n_tiles; uint32_t tiles_per_block; uint32_t n_blocks;   // range (= segment) b holds tiles_per_block tiles, +1 for b < tiles_rem; n_blocks ranges
    uint32_t tiles_rem;
    uint32_t* starts32; uint16_t* lens; uint32_t seg_cap;      // segment b = [b*seg_cap, (b+1)*seg_cap): start - first entry of the tile before range b, length
    SegDesc* desc;
    uint2* hot;                                                 // per range: (count | overflow << 31, kept | first_unknown << 31 | open_after << 30)
    int min_len; unsigned long long* stats;                     // [151 freq | n_records | n_bases] of the records append_entry keeps: zero before the launch
    unsigned long long* wrap_list; uint32_t* wrap_count;        // addresses (b*seg_cap + i) of records with true length >= 65536
};

// first tile of range b (b == n_blocks: the end of the last range = n_tiles)
static inline __host__ __device__ uint64_t e2s_range_tile0(const SegArgs& a, uint32_t b) {
    return (uint64_t)b * a.tiles_per_block + (b < a.tiles_rem ? b : a.tiles_rem);
}

// everything the host needs after a scan, gathered by k_scan_mail so that ONE small download suffices
struct ScanMail {
    uint64_t total_ends, first_end_pos, open_start, last_end_start;
    uint64_t first_rec_addr, last_rec_addr;                  // where record 0 / the last record live (segmented storage)
    uint32_t begins_inside, open_after, end_before_last, err;
    uint32_t wrap_count, first_len_prov, tail_lcp, tail_prev_lcp;
    uint32_t last_text, last_suff, last_lcp, last_bwt;      // the last provided entry (EOF phantom model)
    uint64_t img_records;                                    // records the `.clusters` image will hold (what append_entry keeps), the provisional record 0 included
};

struct CandHeader {           // == e2s_cand_header
    uint64_t record_index;
    uint32_t pair_index;
    uint32_t n0, n1;
    uint32_t right_idx, right_pos;
    uint32_t reserved;
};

struct CandArgs {
    const uint8_t* bwt; const void* lcp; const void* pos; const void* da;   // first provided entry
    int xb, yb, zb;
    uint64_t prov_lo, prov_hi, n_total;
    uint32_t ph_text, ph_suff, ph_lcp, ph_bwt;      // the EOF phantom entry (global index n_total)
    const uint64_t* starts; const uint16_t* lens; uint64_t n_rec;
    uint32_t key_is_start;                          // order key of a candidate = its cluster's start position
    uint32_t rec_per_wave;
    uint64_t nr_reads1, k_left_u, k_right_u;
    uint32_t k_left_lo, mcov_u;
    int len_lo, len_hi;
    uint32_t list_cap;
    uint8_t* cand_buf; uint64_t cand_cap; uint32_t stride;
    unsigned long long* cand_keys;                  // [cand_cap] SA-order key of every candidate slot
    unsigned long long* cand_count;
    // k_flagged_lists, packed mode (host_stream != nullptr): candidates go straight to pinned host memory, packed, at
    // reserved offsets (8-byte units); slot_off8[slot] remembers where, for the SA-order offsets written afterwards
    uint8_t* host_stream; unsigned long long* stream8_count; uint64_t stream8_cap; uint32_t* slot_off8;
    uint32_t pos_pack;                              // bytes per position in the packed record (1, 2 or 4)
    uint32_t blocks_hint;                           // k_flagged_lists: blocks to launch (0: 1024)
    uint64_t record_index_base;
    uint32_t* err;
    const uint64_t* keys;                           // k_flagged_lists: SA-order key of list entry r (nullptr: its start) -- clusters gathered into a compact buffer
    const int* len_hi_dev;                          // k_flagged_lists: the upper length bound lives on the device (the resident job: max_clust_length is worked out there)
    // k_flagged_lists run over the list in pieces (the resident job: the consensus of a piece's candidates runs beside the next piece): the
    // LAST block to finish leaves the number of candidate slots taken so far in *snap_out (done_ctr: cleared by the plan kernel)
    unsigned long long* snap_out; uint32_t* done_ctr;
};


struct FilterArgs {
    const uint8_t* bwt; const uint8_t* lcp; const uint32_t* da;     // first provided entry, 16-byte aligned
    uint64_t n_prov, prov_lo;
    uint32_t prov_is_global0;
    uint64_t own_lo, end_hi;                                        // a cluster is handled where its END is owned
    uint32_t k_u32, kb, hge;                                        // scan threshold (as in ScanArgs)
    uint32_t krb, kr_hge;                                           // k_right broadcast / 0 if it exceeds a byte
    uint64_t nr_reads1;
    uint32_t mcov_u;
    int len_lo, len_hi;
    uint64_t n_tiles;
    uint32_t* err;
    uint64_t* fl_starts; uint16_t* fl_lens; uint64_t fl_cap;        // clusters that pass, for k_flagged_lists
    unsigned long long* fl_count;
    // k_cluster_filter2: wave workers over 4096-entry tiles (worker b owns w_per tiles, +1 for b < w_rem)
    uint64_t w_tiles; uint32_t w_per, w_rem, w_blocks;
    uint64_t w_tile0;                                               // first tile of this launch (the pass may run in chunks)
};

// ---- clust2snp's tail on the GPU (e2s_tail.hip) ----
#define E2S_TAILERR_RANK  1u   // a candidate names a read past the last one (the reference repeats stale text: host tail instead)
#define E2S_TAILERR_CRASH 2u   // right_pos beyond the read's end: substr throws in the reference
struct TailRec {               // one per candidate, written by k_consensus
    uint32_t emit;             // bit 0: both supports > 0 and distance <= 3: an event of the `.snp`; bit 1: both supports > 0 (a candidate "kept")
    int32_t  L;                // indel length of distance() (0: SNP)
    uint32_t s0, s1;           // supports
    uint32_t right_len;        // length of the right context
    uint32_t base_len;         // bytes of the event's four lines without the digits of its id
    uint64_t right_at;         // where the right context starts in the reads' text
};
struct TailArgs {
    const uint8_t* stream; const uint32_t* off8; uint64_t n_cand; uint32_t pos_pack;      // packed candidates in SA order (device memory)
    const uint8_t* text; const uint64_t* rstart; const uint32_t* rlen; uint64_t n_reads;  // the reads: read r = text[rstart[r] .. +rlen[r])
    uint32_t stride;           // rstart == nullptr: every read is `stride` long and read r starts at r * stride (no index to look up)
    uint64_t text_bytes;       // extent of the text (>= 4): no load reaches past it
    // the same text as two bit planes per 32 characters (e2s_launch_reads_pack2; nullptr: not built): text2[g] = (plane of bit 0, plane of bit 1)
    // of the codes (ch >> 1) & 3 -- A 0, C 1, T 2, G 3 -- of characters 32 g .. 32 g + 31; a character outside "ACGT" has code 0 and its
    // READ's bit set in `impure`: k_consensus_pair gathers a left context as ONE 16-byte load where none of a wave's reads is flagged
    const uint2* text2; const uint32_t* impure;                  // (impure == nullptr with text2 set: no read is flagged)
    int kl, kr, max_gap, max_err;
    uint8_t* cons;             // [n_cand][2][64] consensus strings
    TailRec* rec;              // [n_cand]
    unsigned long long* part;  // [groups of 64 candidates][3]: events, text bytes without id digits, candidates kept
    unsigned long long* n_keep;
    uint32_t* err;
    const unsigned long long* n_cand_dev;                        // != nullptr: the number of candidates lives on the device (n_cand = the launch's capacity)
    // the consensus BESIDE the ordering pass (e2s_resident_job): k_consensus* take candidate SLOTS as the per-cluster pass filled them
    // (by_slot: off8 = the slot table, rec / cons indexed by slot, no group sums), and what follows the order goes through
    // slot_of (SA-ordered index -> slot): k_tail_group_sums, k_tail_write
    const uint32_t* slot_of; uint32_t by_slot;
    const unsigned long long* slot_lo_dev;                       // by_slot: this launch handles the slots [*slot_lo_dev, *n_cand_dev) (nullptr: from 0)
    // 8-byte units of `stream` that exist: where the per-cluster pass ran out of room (the resident job finds out at its one wait, after the
    // tail has run) a slot's offset points past what was written -- such a candidate is skipped, nothing outside the buffer is read
    uint64_t stream8_cap;
};

// ---- the resident job without host round trips (e2s_resident_job): what one rank publishes, what the plan kernel derives ----
#define E2S_JOB_BLOB 1344u                  // e2s_boundary (72 B) + e2s_scan_stats (1248 B), padded: one rank's share of the first all-gather
#define E2S_MAX_DEVICES 64                    /* per-device caches of launch parameters (device id modulo this) */
#define E2S_JOB_MAX_RANKS 64
#define E2S_JOB_MAX_PIECES 8
#define E2S_JOBERR_PLAN 16u                 // stat_rc: the gathered descriptors do not fit together
#define E2S_JOBERR_SEEDS 32u                //          more wrapped clusters than seed slots
#define E2S_JOBERR_IMAGE 64u                //          the `.clusters` image is too small for this rank's records
#define E2S_JOBERR_CRASH_MASK 3u            // stat_rc's low bits: where the reference's statistics() crashes (the same on every rank: not a reason to redo)
struct JobDev {                             // device copy at d_mail + 1536; reaches the host with the mail
    e2s_stitch_plan_t plan;                 // this rank's
    int32_t mcl; uint32_t stat_rc;          // max_clust_length; e2s_auto_len_core's code | E2S_JOBERR_*
    uint64_t n_clust, n_bases;              // statistics() incl. the H4 double count
    uint64_t img_records; uint32_t img_skip, n_tail_kept;
    int32_t len_hi; uint32_t first_wrapped;
    uint64_t first_id, events, snp_bytes, n_keep;     // the tail: ids from first_id, events of this rank, bytes of its text, candidates kept
    uint32_t tail_err, peers_not_ok;        // E2S_TAILERR_* of this rank; a rank of the job reported a problem in the second all-gather
    uint64_t events_before;
    uint32_t my_ok, reserved;               // this rank's verdict on its own step (what it sends in the second all-gather)
};
void e2s_launch_consensus(const TailArgs& a, uint32_t list_cap, hipStream_t st);
bool e2s_consensus_fits(int kl, uint32_t list_cap);           // the lists of a candidate fit the current device's LDS
void e2s_launch_tail_group_sums(const TailArgs& a, uint32_t ng, hipStream_t st);
void e2s_launch_tail_partials(const unsigned long long* part, unsigned long long* part_excl, uint32_t ng, unsigned long long first_id,
                              unsigned long long* totals, const unsigned long long* n_keep, const uint32_t* err, JobDev* jd, unsigned long long* blob2,
                              const uint32_t* bad_a, const uint32_t* bad_b, const uint32_t* bad_c, const unsigned long long* n_cand_dev, hipStream_t st);
void e2s_launch_reads_extent(const uint64_t* rstart, const uint32_t* rlen, uint64_t n_reads, unsigned long long* out, hipStream_t st);   // *out = max(start + len), *out zeroed by the caller
void e2s_launch_reads_compact(const uint8_t* text, const uint64_t* rstart, uint32_t len, uint64_t n_reads, uint8_t* out, hipStream_t st);
// the reads as bit planes (TailArgs::text2 / impure): out holds text_bytes / 32 + 3 pairs, impure (n_reads + 31) / 32 words + one more that is
// set when any read is flagged, both zeroed by the caller
void e2s_launch_reads_pack2(const uint8_t* text, uint64_t text_bytes, const uint64_t* rstart, const uint32_t* rlen, uint32_t stride, uint64_t n_reads,
                            uint2* out, uint32_t* impure, hipStream_t st);
void e2s_launch_tail_write(const TailArgs& a, const unsigned long long* part_excl, unsigned long long first_id, const unsigned long long* first_id_dev,
                           uint8_t* out, unsigned long long out_cap, hipStream_t st, const unsigned long long* ids2 = nullptr, int world = 1, int rank = 0,
                           JobDev* jd = nullptr, const unsigned long long* mail_src = nullptr, unsigned long long* mail_dst = nullptr, uint32_t mail_words = 0);

void e2s_launch_cluster_filter2(FilterArgs a, unsigned long long tile_lo, unsigned long long tile_hi, int max_workers, hipStream_t st);
void e2s_launch_phase_init(unsigned long long* count, unsigned long long count_value, uint16_t* lens, uint32_t n_lens, uint32_t* w0, uint32_t n0,
                           uint32_t* w1, uint32_t n1, unsigned long long* z, uint32_t nz, hipStream_t st);   // n_lens, n0, n1, nz <= 256
void e2s_launch_cand_init(unsigned long long* cand_count, unsigned long long* stream8_count, uint32_t* err, const unsigned long long* seed_starts,
                          const uint16_t* seed_lens, uint32_t n_seed_slots, unsigned long long* fl_starts, uint16_t* fl_lens,
                          unsigned long long* fl_count, bool set_fl_count, uint32_t* ko_cnt, uint32_t n_ko, hipStream_t st);   // n_seed_slots <= 256
void e2s_launch_lcp_max(const void* lcp, int xb, unsigned long long n, uint32_t* out, hipStream_t st);        // xb = 2 or 4
void e2s_launch_lcp_narrow(const void* lcp, int xb, unsigned long long n, uint8_t* out, hipStream_t st);
void e2s_launch_record_prefilter(const unsigned long long* starts, const uint16_t* lens, unsigned long long n_rec, const uint8_t* bwt,
                                 unsigned long long prov_lo, unsigned long long prov_hi, uint32_t mcov, int len_lo, int len_hi,
                                 unsigned long long* fl_starts, uint16_t* fl_lens, unsigned long long fl_cap, unsigned long long* fl_count, hipStream_t st);
void e2s_launch_snapshot(const unsigned long long* src, unsigned long long* dst, hipStream_t st);
void e2s_launch_flagged_lists(const CandArgs& a, const unsigned long long* lo_dev, const unsigned long long* n_dev, bool def_widths, hipStream_t st);
void e2s_launch_tile_scan(const uint32_t* cnt, unsigned long long* prefix, unsigned long long n, hipStream_t st);
unsigned long long e2s_pack_clusters_blocks(unsigned long long n);
void e2s_launch_pack_clusters(const unsigned long long* starts, const uint16_t* lens, unsigned long long n, int min_len, uint32_t* cnt,
                              unsigned long long* prefix, uint8_t* out, unsigned long long* total, hipStream_t st);
void e2s_launch_pack_segments(const SegArgs& a, uint32_t row_entries, uint32_t slot0_b, const unsigned long long* kept_off, uint8_t* out, unsigned long long out_cap,
                              const ScanMail* mail, uint32_t n_tail, unsigned long long t0, uint16_t l0, unsigned long long t1, uint16_t l1, hipStream_t st,
                              uint32_t max_blocks);
void e2s_launch_seg_patch(SegDesc* d, uint16_t* len0, unsigned long long first_start, uint16_t first_len, hipStream_t st);
#define E2S_WRAP_CAP 1024
int  e2s_scan_seg_resident_blocks(int xb);
void e2s_launch_scan_seg(const SegArgs& a, int xb, hipStream_t st);
int  e2s_scan_seg_fused_resident_blocks();
void e2s_launch_scan_seg_fused(const SegArgs& a, const FilterArgs& f, hipStream_t st);
void e2s_launch_seg_fix(const SegArgs& a, uint32_t* flags, uint32_t* tailv, unsigned long long* seg_off, unsigned long long* kept_off, uint32_t* err,
                        const uint8_t* bwt, const void* pos, const void* da, int xb, int yb, int zb, ScanMail* out, unsigned long long* stats_out,
                        uint8_t* blob, int is_last_rank, int tail_bwt_arg, hipStream_t st);
struct JobPlanArgs {
    const uint8_t* blobs; int world, rank; uint64_t n_total; int k, lcp_bytes, mcov; double pval; int min_len, len_lo;
    const ScanMail* mail; SegArgs seg; uint32_t row_entries;
    JobDev* out;
    unsigned long long* fl_starts; uint16_t* fl_lens;            // seed slots: the head of the flagged list
    unsigned long long* cand_count; unsigned long long* stream8_count; uint32_t* cand_err; uint32_t* ko_cnt; uint32_t n_ko;
    unsigned long long* zero8;                                   // 8 words cleared here (nullptr: none)
    // the per-cluster pass in pieces: bounds[c] = fl_count * c / n_pieces (list entries), snap[0] = 0 (candidate slots), done[c] = 0
    const unsigned long long* fl_count; uint32_t n_pieces; unsigned long long* piece_bounds; unsigned long long* piece_snap; uint32_t* piece_done;
    unsigned long long img_cap;                                  // records the `.clusters` image has room for (~0: no image wanted)
};
void e2s_launch_job_plan(const JobPlanArgs& a, hipStream_t st);
void e2s_launch_img_patch_dev(uint8_t* img, unsigned long long cap, const JobDev* jd, const ScanMail* mail, int min_len, hipStream_t st);
void e2s_launch_img_patch(uint8_t* img, uint32_t patch_first, unsigned long long first_start, uint16_t first_len, unsigned long long n_at, uint32_t n_tail,
                          unsigned long long t0, uint16_t l0, unsigned long long t1, uint16_t l1, hipStream_t st);
void e2s_launch_seg_compact(const SegArgs& a, const unsigned long long* seg_off, uint32_t row_entries, unsigned long long* out_starts, uint16_t* out_lens,
                            hipStream_t st);
void e2s_launch_unpack_clusters(const uint8_t* packed, unsigned long long n, unsigned long long* starts, uint16_t* lens, uint32_t* err, hipStream_t st);
void e2s_launch_apply_plan(unsigned long long* starts, uint16_t* lens, uint32_t patch_first, unsigned long long first_start,
                           uint16_t first_len, unsigned long long n_ends, uint32_t n_tail, unsigned long long t0, uint16_t l0,
                           unsigned long long t1, uint16_t l1, hipStream_t st);
void e2s_launch_hist_tail(const uint16_t* lens, unsigned long long n, int min_len, unsigned long long* out_last_len, hipStream_t st);
constexpr uint32_t E2S_KO_MAX_BUCKETS = 1u << 18;
size_t e2s_cand_order_scratch_bytes(unsigned long long n);
void e2s_launch_cand_offsets(const uint32_t* order, const uint32_t* slot_off8, unsigned long long n, const unsigned long long* n_dev,
                             const unsigned long long* stream8_count, uint32_t* off_out, const unsigned long long* fl_count, const uint32_t* err_a,
                             const uint32_t* err_b, unsigned long long* mail, uint32_t* bad, unsigned long long fl_cap, unsigned long long stream8_cap,
                             unsigned long long* zero, unsigned long long n_zero, uint32_t* ko_cnt, uint32_t n_ko, hipStream_t st);
uint32_t e2s_cand_order_buckets(unsigned long long n);
void e2s_launch_cand_order(const unsigned long long* keys, unsigned long long n, const unsigned long long* n_dev, unsigned long long key_lo,
                           unsigned long long key_span, void* scratch, uint32_t* order, hipStream_t st, bool counts_cleared = false);
void e2s_launch_hist(const uint16_t* lens, unsigned long long n, int min_len, unsigned long long* out, hipStream_t st);
void e2s_launch_unpack_pairsa(const uint8_t* src, unsigned long long n, int zf, int yf, void* pos, int zb, void* da, int yb, hipStream_t st);
void e2s_launch_narrow8(const uint8_t* src, unsigned long long n, uint32_t* dst, hipStream_t st);
void e2s_launch_unpack_gesa(const uint8_t* src, unsigned long long n, int yf, int zf, int xf, uint8_t* bwt, void* lcp, int xb,
                            void* pos, int zb, void* da, int yb, hipStream_t st);
