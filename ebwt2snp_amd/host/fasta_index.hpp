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
