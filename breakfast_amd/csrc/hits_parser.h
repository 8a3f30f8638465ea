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

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <new>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/breakfast_hip.h"

namespace hits {

// Where a batch's arrays live.  The default is the heap; a driver hands the parser the library's pinned host memory
// (bf_host_alloc) as soon as its GPU context exists, and bf_submit then uploads a batch from where the parser wrote it.
struct Allocator {
    void* (*alloc)(void* user, size_t bytes);       // nullptr when there is none to be had: the array goes to the heap
    void (*release)(void* user, void* p);
    void* user;
};
const Allocator* heap_allocator();

template <class T> class Buf {           // a growable array of plain values; content is not initialised by resize()
public:
    Buf() = default;
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    Buf(Buf&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), al_(o.al_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    Buf& operator=(Buf&& o) noexcept {
        if (this != &o) { drop(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; al_ = o.al_; o.p_ = nullptr; o.n_ = o.cap_ = 0; }
        return *this;
    }
    ~Buf() { drop(); }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    bool empty() const { return n_ == 0; }
    T* data() { return p_; }
    const T* data() const { return p_; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }
    void clear() { n_ = 0; }
    void reserve(size_t cap) { if (cap > cap_) regrow(cap); }
    void resize(size_t n) { if (n > cap_) regrow(std::max(n, cap_ * 2)); n_ = n; }
    void push_back(const T& v) { if (n_ == cap_) regrow(std::max<size_t>(cap_ * 2, 1024)); p_[n_++] = v; }
    void append(const T* src, size_t k) { const size_t at = n_; resize(n_ + k); if (k) memcpy(p_ + at, src, k * sizeof(T)); }
    // empty, with the same room, in `al`'s memory (nothing happens when it is there already)
    void move_to(const Allocator* al) {
        n_ = 0;
        if (al == want_) return;
        const size_t cap = cap_;
        drop(); p_ = nullptr; cap_ = 0;
        want_ = al;
        if (cap) regrow(cap);
    }
    const Allocator* home() const { return al_; }
private:
    void drop() { if (p_) al_->release(al_->user, p_); p_ = nullptr; }
    void regrow(size_t cap) {
        const Allocator* al = want_;
        T* q = (T*)al->alloc(al->user, cap * sizeof(T));
        if (!q) { al = heap_allocator(); q = (T*)al->alloc(al->user, cap * sizeof(T)); }
        if (!q) throw std::bad_alloc();
        if (n_) memcpy(q, p_, n_ * sizeof(T));
        drop();
        p_ = q; cap_ = cap; al_ = al;
    }
    T* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    const Allocator* al_ = heap_allocator();      // where p_ came from
    const Allocator* want_ = heap_allocator();    // where the next allocation goes
};

struct Batch {                           // one bf_batch (or, with Options::pack, bf_packed_batch) worth of parsed pairs
    Buf<char> seq, frag;                 // seq: ASCII bases, or BAM 4-bit codes (two bases a byte, every read byte-aligned)
    Buf<uint64_t> seq_off, frag_off;     // n() + 1 entries each, the first is 0 (none before the first clear())
    Buf<uint16_t> seq_len;               // bases per read (4-bit codes only)
    Buf<bf_hit> hits;
    bool packed = false;
    Batch() = default;                   // holds nothing, not even the offsets' leading 0, until clear() has run
    Batch(Batch&&) = default;
    Batch& operator=(Batch&&) = default;
    size_t n() const { return hits.size(); }
    bool has_room() const { return seq.capacity() || hits.capacity(); }     // worth keeping for another block
    void clear() { seq.clear(); frag.clear(); seq_off.clear(); seq_off.push_back(0); frag_off.clear(); frag_off.push_back(0); seq_len.clear(); hits.clear(); }
    // clear(), and from now on the arrays live in `al`'s memory
    void clear(const Allocator* al) {
        seq.move_to(al); frag.move_to(al); seq_off.move_to(al); frag_off.move_to(al); seq_len.move_to(al); hits.move_to(al);
        seq_off.push_back(0); frag_off.push_back(0);
    }
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
    // where the batches' arrays are allocated: looked at whenever a block's parse starts, so the caller may point it
    // at another allocator while the parse runs (nullptr, or a null content: the heap).  An allocator must outlive
    // every batch that was filled while it was set.
    const std::atomic<const Allocator*>* allocator = nullptr;
};

struct Stats { uint64_t bytes = 0, lines = 0, pairs = 0, blocks = 0; };

// Reads `fd` to its end (a regular file is memory-mapped, anything else is read()), parses it and
// calls sink(batch) on the calling thread, batch after batch in input order.  A sink that keeps a batch swaps it
// away; whatever it leaves in `batch` (best: a batch it is done with) has its buffers reused for a later block.  Throws ParseError for
// what makes the reference panic (malformed line, unknown contig); the message names the line.
Stats parse_stream(int fd, const ContigIndex& contigs, const std::string& genome_path, const Options& opt,
                   const std::function<void(Batch&)>& sink);

}  // namespace hits
