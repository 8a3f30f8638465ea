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

// One BAM record as it lies in the inflated stream (zero-copy when the record does not straddle a BGZF
// block): the read filter looks at flag, CIGAR and length only, so the whole-file scan (README.md:6's
// "40 M reads/min/core" stage) never decodes the bases or copies the name of a read it drops.
struct RecordView {
    const uint8_t* body = nullptr;   // the bytes after block_size
    uint32_t size = 0;
    uint32_t l_read_name = 0, n_cigar = 0, l_seq = 0;
    uint16_t flag = 0;
    bool is_unmapped() const { return (flag & 4) != 0; }
    uint32_t cigar(uint32_t i) const;
    int64_t leading_softclips() const;
    int64_t trailing_softclips() const;
    void qname(std::string& out) const;          // without the trailing NUL
    void seq(std::string& out) const;            // decoded with "=ACMGRSVTWYHKDBN"
};

class Reader {
public:
    // path "-" = stdin.  threads = inflate workers (>= 1).  Throws std::runtime_error with a
    // user-facing message on failure.
    Reader(const std::string& path, int threads);
    ~Reader();
    Reader(const Reader&) = delete;
    Reader& operator=(const Reader&) = delete;
    bool next(Record& rec);          // false after the last record (decodes everything)
    bool next_view(RecordView& v);   // the same walk without decoding; v is valid until the next call
    const std::vector<std::string>& references() const { return refs_; }

private:
    struct Impl;
    Impl* p_;
    std::vector<std::string> refs_;
    std::vector<uint8_t> spill_;     // a record that straddles BGZF blocks is assembled here
    bool read_exact(void* dst, size_t n, bool allow_eof);
};

// The read filter + anchor emission of detect.rs:300-337.  Writes to `out` what the reference
// writes to bowtie's stdin; returns the number of reads sent.
uint64_t dispatch_reads(Reader& bam, FILE* out, uint32_t anchor_len);

}  // namespace bam
