// Multi-threaded consumer of Bowtie's stdout = the line-pairing half of the consume loop
// (src/detect.rs:76-103): pairs `5p:`/`3p:` lines by read number and extracts fragment id, read
// sequence and the two (strand, contig, offset) triples into structure-of-arrays batches for
// bf_submit.  Host code only (SURVEY.md section 8f "next" row 2, SURVEY H8: the text side of the
// boundary is what bounds the end-to-end rate).
//
// The reference's loop is sequential: its only state is the most recent `5p:` line.  Here the
// stream is cut into blocks at line boundaries and the blocks are parsed concurrently; a block's
// lines that precede its first `5p:` line are the only ones whose fate depends on an earlier block,
// and they are resolved in stream order by the consuming thread.  Batches reach the sink in input
// order, so read indices are the reference's.
#pragma once

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/breakfast_hip.h"

namespace hits {

struct Batch {                           // one bf_batch (or, with Options::pack, bf_packed_batch) worth of parsed pairs
    std::string seq, frag;               // seq: ASCII bases, or BAM 4-bit codes (two bases a byte, every read byte-aligned)
    std::vector<uint64_t> seq_off{0}, frag_off{0};
    std::vector<uint16_t> seq_len;       // bases per read (4-bit codes only)
    std::vector<bf_hit> hits;
    bool packed = false;
    size_t n() const { return hits.size(); }
    void clear() { seq.clear(); frag.clear(); seq_off.assign(1, 0); frag_off.assign(1, 0); seq_len.clear(); hits.clear(); }
};

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };
// Options::pack only: a read holds a character outside BAM's alphabet "=ACMGRSVTWYHKDBN" (lower case, '.', ...), or
// more than 65535 bases.  Nothing of the stream is lost for a caller that can read it again as ASCII.
struct Unpackable : std::runtime_error { using std::runtime_error::runtime_error; };

// `n` ASCII bases -> ceil(n / 2) bytes of BAM 4-bit codes at `out` (high nibble first, a trailing low nibble 0);
// false when a base is outside the alphabet.  (SSSE3 where the CPU has it.)
bool pack_bases(const char* bases, size_t n, uint8_t* out);

class ContigIndex {                      // contig name -> id (index in the arrays given to bf_genome_set)
public:
    void add(const std::string& name, uint32_t id) { names_.push_back(name); ids_.push_back(id); built_ = false; }
    void build();
    int find(std::string_view name) const;          // -1 when absent
private:
    std::vector<std::string> names_;
    std::vector<uint32_t> ids_;
    std::vector<uint32_t> slots_;        // open addressing: index into names_ + 1, 0 = empty
    bool built_ = false;
};

struct Options {
    int threads = 8;                     // parser workers (>= 1)
    size_t block_bytes = 32u << 20;      // nominal block size; blocks end on a line boundary
    bool pack = false;                   // reads as BAM 4-bit codes (for bf_submit_packed); throws Unpackable where they cannot be
};

struct Stats { uint64_t bytes = 0, lines = 0, pairs = 0, blocks = 0; };

// Reads `fd` to its end (a regular file is memory-mapped, anything else is read()), parses it and
// calls sink(batch) on the calling thread, batch after batch in input order.  A sink that keeps a batch swaps it
// away; whatever it leaves in `batch` (best: a batch it is done with) has its buffers reused for a later block.  Throws ParseError for
// what makes the reference panic (malformed line, unknown contig); the message names the line.
Stats parse_stream(int fd, const ContigIndex& contigs, const std::string& genome_path, const Options& opt,
                   const std::function<void(Batch&)>& sink);

}  // namespace hits
