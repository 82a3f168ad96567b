// fasta_index.hpp -- the reads of a FASTA as (text, start[], len[]): what get_reads (clust2snp.cpp:147-212) hands to
// extract_variants, built once for ALL reads so that they can live in HBM.
//
// Only REGULAR files are indexed: the first line is a header ('>'), every header line is followed by at least one line that is
// no header, the file does not end with a header line.  Then record j is the j-th header line and its DNA the lines up to
// the next header, concatenated -- which is all get_reads' line juggling amounts to.  Anything else returns false and the
// caller keeps to the sequential restatement of get_reads in snp_tail.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

struct FastaIndex {
    std::vector<uint64_t> start;        // offset of read r in `text`
    std::vector<uint32_t> len;
    const uint8_t* text = nullptr;      // the mapped file itself when every record is one line long, else `joined`
    size_t text_bytes = 0;
    std::vector<uint8_t> joined;
    uint32_t max_len = 0;
};
bool e2s_index_fasta(const uint8_t* p, size_t n, unsigned threads, FastaIndex& out);

// The same index built while the file STREAMS to the GPU (e2s_upload_reads_fasta: pread into pinned buffers, no mapping): the file is cut
// into fixed chunks; a chunk's scan sees the lines that START in it.  Only the plain shape is taken on this path -- header, ONE DNA line,
// header, ONE DNA line ... -- and everything else (multi-line records, two headers in a row, a trailing header, a line longer than the
// chunk's read-ahead) sends the caller back to e2s_index_fasta over a mapping, which decides as it always did.
struct FastaChunk {
    std::vector<uint64_t> start;        // DNA lines that start in the chunk: offset in the file
    std::vector<uint32_t> len;
    uint64_t headers = 0;
    int first_kind = -1, last_kind = -1;   // 0 header, 1 DNA line; -1: no line starts in the chunk
    bool plain = true;                  // false: hh / multi-line / a line that does not end inside the buffer / a line of 2^32 bytes
    uint32_t max_len = 0, min_len = 0xFFFFFFFFu;
};
// buf holds the file's bytes [buf_off, buf_off + buf_n); the chunk is [a, b) with buf_off <= (a ? a - 1 : 0) and b <= buf_off + buf_n <= file_n.
void e2s_fasta_scan_chunk(const uint8_t* buf, uint64_t buf_off, size_t buf_n, uint64_t a, uint64_t b, uint64_t file_n, FastaChunk& out);
// the chunks in file order -> the index, in arrays that are written by several threads and never zero-filled first (2 x 10^7 reads: 240 MB).
// false: not the plain shape.
struct FastaJoined { std::unique_ptr<uint64_t[]> start; std::unique_ptr<uint32_t[]> len; uint64_t n = 0; uint32_t max_len = 0; bool uniform = false; };
bool e2s_fasta_join_chunks(std::vector<FastaChunk>& chunks, FastaJoined& out);
