// BAM front-end of `breakfast detect` = dispatch_reads_to_bowtie (src/detect.rs:290-338) and
// read_bam_record (src/common.rs:61-67), without htslib: a BGZF reader with multi-threaded block
// inflate on zlib, a BAM record walker, the reference's read filter and the anchor FASTA writer.
// Host code only (this is SURVEY.md section 8f "next" row 1, the producer side of the boundary).
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace bam {

struct Record {                      // the fields dispatch_reads_to_bowtie looks at
    uint16_t flag = 0;
    std::string qname;               // without the trailing NUL (rust-htslib Record::qname)
    std::vector<uint32_t> cigar;     // BAM encoding: len << 4 | op
    std::string seq;                 // decoded with "=ACMGRSVTWYHKDBN" (Record::seq().as_bytes())
    bool is_unmapped() const { return (flag & 4) != 0; }
    int64_t leading_softclips() const;    // rust-htslib CigarStringView semantics (H then S counts)
    int64_t trailing_softclips() const;
};

class Reader {
public:
    // path "-" = stdin.  threads = inflate workers (>= 1).  Throws std::runtime_error with a
    // user-facing message on failure.
    Reader(const std::string& path, int threads);
    ~Reader();
    Reader(const Reader&) = delete;
    Reader& operator=(const Reader&) = delete;
    bool next(Record& rec);          // false after the last record
    const std::vector<std::string>& references() const { return refs_; }

private:
    struct Impl;
    Impl* p_;
    std::vector<std::string> refs_;
    bool read_exact(void* dst, size_t n, bool allow_eof);
};

// The read filter + anchor emission of detect.rs:300-337.  Writes to `out` what the reference
// writes to bowtie's stdin; returns the number of reads sent.
uint64_t dispatch_reads(Reader& bam, FILE* out, uint32_t anchor_len);

}  // namespace bam
