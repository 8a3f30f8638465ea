// See hits_parser.h.  Reference semantics: src/detect.rs:76-103.

#include "hits_parser.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace hits {

// Open addressing on FNV-1a of the name: contig names are a handful of bytes, and the parser looks two of them up per
// read (std::unordered_map's hash and bucket walk were a fifth of its time).
static inline uint64_t name_hash(std::string_view s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : s) h = (h ^ c) * 0x100000001b3ull;
    return h ^ (h >> 29);
}

const Allocator* heap_allocator() {
    static const Allocator heap{[](void*, size_t bytes) { return malloc(bytes ? bytes : 1); }, [](void*, void* p) { free(p); }, nullptr};
    return &heap;
}

void ContigIndex::build() {
    size_t cap = 16;
    while (cap < names_.size() * 4) cap *= 2;
    slots_.assign(cap, 0);
    for (size_t i = 0; i < names_.size(); i++) {
        size_t at = name_hash(names_[i]) & (cap - 1);
        while (slots_[at] && names_[slots_[at] - 1] != names_[i]) at = (at + 1) & (cap - 1);
        slots_[at] = (uint32_t)i + 1;                         // last record of a name wins
    }
    built_ = true;
}

int ContigIndex::find(std::string_view name) const {
    if (slots_.empty()) return -1;
    const size_t mask = slots_.size() - 1;
    for (size_t at = name_hash(name) & mask;; at = (at + 1) & mask) {
        const uint32_t s = slots_[at];
        if (!s) return -1;
        const std::string& n = names_[s - 1];
        if (n.size() == name.size() && memcmp(n.data(), name.data(), name.size()) == 0) return (int)ids_[s - 1];
    }
}

namespace {

// BAM's base codes (SAM specification section 4.2.3: "=ACMGRSVTWYHKDBN" -> 0..15)
struct CodeTable {
    uint8_t t[256];
    CodeTable() {
        memset(t, 0xff, sizeof t);
        const char* a = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < 16; i++) t[(uint8_t)a[i]] = (uint8_t)i;
    }
};
const CodeTable kCodes;

bool pack_scalar(const char* s, size_t n, uint8_t* out) {
    unsigned bad = 0;
    size_t i = 0;
    for (; i + 1 < n; i += 2) {
        const uint8_t a = kCodes.t[(uint8_t)s[i]], b = kCodes.t[(uint8_t)s[i + 1]];
        bad |= (a | b) & 0x80;
        out[i >> 1] = (uint8_t)((a << 4) | (b & 15));
    }
    if (i < n) {
        const uint8_t a = kCodes.t[(uint8_t)s[i]];
        bad |= a & 0x80;
        out[i >> 1] = (uint8_t)(a << 4);
    }
    return bad == 0;
}

// 16 bases a step: the code of a letter is looked up by its low five bits (two 16-entry tables), checked by decoding
// it again (any other character decodes to something else), and the 16 codes are folded into 8 bytes.
__attribute__((target("ssse3"))) bool pack_ssse3(const char* s, size_t n, uint8_t* out) {
    //                                 idx:  0  A  B   C  D   E  F  G  H   I  J  K   L  M  N   O
    const __m128i lo = _mm_setr_epi8(0, 1, 14, 2, 13, 0, 0, 4, 11, 0, 0, 12, 0, 3, 15, 0);
    //                                 idx:  P  Q  R  S  T  U  V  W  X  Y   Z  [  \  =  ^  _
    const __m128i hi = _mm_setr_epi8(0, 0, 5, 6, 8, 0, 7, 9, 0, 10, 0, 0, 0, 0, 0, 0);
    const __m128i dec = _mm_setr_epi8('=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N');
    const __m128i m0f = _mm_set1_epi8(0x0f), m10 = _mm_set1_epi8(0x10), fold = _mm_set1_epi16(0x0110);
    __m128i ok = _mm_set1_epi8((char)0xff);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128((const __m128i*)(s + i));
        const __m128i ix = _mm_and_si128(v, m0f);
        const __m128i upper = _mm_cmpeq_epi8(_mm_and_si128(v, m10), m10);
        const __m128i code = _mm_or_si128(_mm_andnot_si128(upper, _mm_shuffle_epi8(lo, ix)), _mm_and_si128(upper, _mm_shuffle_epi8(hi, ix)));
        ok = _mm_and_si128(ok, _mm_cmpeq_epi8(_mm_shuffle_epi8(dec, code), v));
        const __m128i w = _mm_maddubs_epi16(code, fold);            // code[2k] * 16 + code[2k+1]
        _mm_storel_epi64((__m128i*)(out + (i >> 1)), _mm_packus_epi16(w, w));
    }
    const bool head_ok = _mm_movemask_epi8(ok) == 0xffff;
    return pack_scalar(s + i, n - i, out + (i >> 1)) && head_ok;
}

const bool kHaveSsse3 = __builtin_cpu_supports("ssse3");

}  // namespace

bool pack_bases(const char* bases, size_t n, uint8_t* out) {
    return kHaveSsse3 ? pack_ssse3(bases, n, out) : pack_scalar(bases, n, out);
}

namespace {

bool parse_u64(std::string_view s, uint64_t& v) {
    if (s.empty()) return false;
    v = 0;
    if (s.size() <= 19) {                                     // cannot overflow: no check per digit
        for (char c : s) {
            const unsigned d = (unsigned)(c - '0');
            if (d > 9) return false;
            v = v * 10 + d;
        }
        return true;
    }
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        if (v > (0xffffffffffffffffull - (uint64_t)(c - '0')) / 10) return false;    // usize::parse overflows -> Err
        v = v * 10 + (uint64_t)(c - '0');
    }
    return true;
}

// Fields of `s` split on `d`, like Rust's s.split(d): out[i] = i-th field for i < want; returns the
// number of fields found (<= want).  The last requested field ends at the next delimiter or at the end.
int split_fields(std::string_view s, char d, int want, std::string_view* out) {
    const char* p = s.data();
    const char* end = p + s.size();
    int n = 0;
    while (n < want) {
        const char* q = (const char*)memchr(p, d, (size_t)(end - p));
        out[n++] = std::string_view(p, (size_t)((q ? q : end) - p));
        if (!q) break;
        p = q + 1;
    }
    return n;
}

[[noreturn]] void bad(const char* what, std::string_view line) {
    throw ParseError(std::string(what) + ": " + std::string(line.substr(0, 200)));
}

// The read number of a line that starts "5p:<digits>:" or "3p:<digits>:" (what Bowtie writes): returns where the text
// after that second colon begins, 0 for any other line (read_number below then does what the reference's split does).
inline size_t canonical_number(std::string_view line, uint64_t& num) {
    const char* p = line.data();
    const size_t n = line.size();
    if (n < 5 || p[1] != 'p' || p[2] != ':' || (p[0] != '5' && p[0] != '3')) return 0;
    size_t i = 3;
    uint64_t v = 0;
    for (; i < n && i < 3 + 19; i++) {
        const unsigned d = (unsigned)(p[i] - '0');
        if (d > 9) break;
        v = v * 10 + d;
    }
    if (i == 3 || i >= n || p[i] != ':') return 0;
    num = v;
    return i + 1;
}

// `line.split(':').nth(1).unwrap().parse().unwrap()`, detect.rs:79
uint64_t read_number(std::string_view line) {
    std::string_view f[2];
    uint64_t v;
    if (split_fields(line, ':', 2, f) < 2 || !parse_u64(f[1], v)) bad("Malformed anchor alignment line", line);
    return v;
}

struct PairParser {
    const ContigIndex& contigs;
    const std::string& genome_path;
    bool pack = false;
    int contig(std::string_view name) const {
        const int id = contigs.find(name);
        if (id < 0)
            throw ParseError("Anchor aligned to a contig that " + genome_path +
                             ".fa does not contain (the reference panics here, detect.rs:115).");
        return id;
    }

    // "\t<strand>\t<contig>\t<pos>" from `p` on, the way Bowtie writes it: one-character strand, a contig name, digits up to
    // the end of the line (or a further tab).  False for anything else: the caller then takes the general path.
    static bool tail_fields(const char* p, const char* end, bool& plus, std::string_view& name, uint64_t& pos) {
        if (end - p < 6 || p[0] != '\t' || p[2] != '\t') return false;
        plus = p[1] == '+';
        const char* c = p + 3;
        const char* q = c;
        while (q < end && *q != '\t') q++;
        if (q == end || q == c) return false;
        name = std::string_view(c, (size_t)(q - c));
        const char* d = q + 1;
        if (d == end || end - d > 19) return false;
        uint64_t v = 0;
        for (; d < end && *d != '\t'; d++) {
            const unsigned k = (unsigned)(*d - '0');
            if (k > 9) return false;
            v = v * 10 + k;
        }
        if (d == q + 1) return false;
        pos = v;
        return true;
    }

    // The same pair when both lines look the way Bowtie writes them -- "5p:<n>:" / "3p:<n>:<fragment>:<bases>:" followed by
    // the three tab-separated fields -- in one pass over each line instead of a dozen searches (a search costs more in
    // call overhead than in bytes on lines this short).  `p5` / `p3`: where the lines' text continues after "<n>:".
    // Returns false, having changed nothing, when a line is anything else; operator() below then does what the
    // reference's split() calls do, errors included.
    bool fast(std::string_view prev, size_t p5, std::string_view line, size_t p3, Batch& out) const {
        if (p5 == 0 || p3 == 0) return false;
        const char* l = line.data();
        const char* lend = l + line.size();
        const char* f = l + p3;                                        // fragment id: up to the next ':'
        const char* fe = f;
        while (fe < lend && *fe != ':' && *fe != '\t') fe++;
        if (fe == lend || *fe != ':') return false;
        const char* sq = fe + 1;                                        // bases: up to the next ':', which a tab follows
        const char* se = (const char*)memchr(sq, ':', (size_t)(lend - sq));
        if (!se || se + 1 >= lend || se[1] != '\t') return false;
        if (memchr(sq, '\t', (size_t)(se - sq))) return false;
        bool plus5, plus3;
        std::string_view n5, n3;
        uint64_t pos5, pos3;
        if (!tail_fields(se + 1, lend, plus3, n3, pos3)) return false;
        if (!tail_fields(prev.data() + p5, prev.data() + prev.size(), plus5, n5, pos5)) return false;
        if (pos5 >= 0x80000000ull || pos3 >= 0x80000000ull) return false;
        const int c5 = contigs.find(n5), c3 = contigs.find(n3);
        if (c5 < 0 || c3 < 0) return false;
        const size_t L = (size_t)(se - sq);
        if (pack) {
            if (L > 65535) return false;
            const size_t at = out.seq.size();
            out.seq.resize(at + (L + 1) / 2);
            if (!pack_bases(sq, L, (uint8_t*)out.seq.data() + at)) { out.seq.resize(at); return false; }
            out.seq_len.push_back((uint16_t)L);
            out.packed = true;
        } else {
            out.seq.append(sq, L);
        }
        bf_hit h;
        memset(&h, 0, sizeof h);
        h.pos5 = (uint32_t)pos5; h.pos3 = (uint32_t)pos3;
        h.contig5 = (uint16_t)c5; h.contig3 = (uint16_t)c3;
        h.strands = (uint8_t)((plus5 ? 1 : 0) | (plus3 ? 2 : 0));
        out.hits.push_back(h);
        out.seq_off.push_back(out.seq.size());
        out.frag.append(f, (size_t)(fe - f)); out.frag_off.push_back(out.frag.size());
        return true;
    }

    // detect.rs:89-103 for one (5p line, 3p line) pair
    void operator()(std::string_view prev, std::string_view line, Batch& out) const {
        std::string_view a[4], c[4], m[4];
        if (split_fields(line, ':', 4, a) < 4) bad("Malformed anchor alignment line", line);   // nth(2), next()
        if (split_fields(prev, '\t', 4, c) < 4 || split_fields(line, '\t', 4, m) < 4)
            bad("Malformed anchor alignment line", line);
        uint64_t pos5, pos3;
        if (!parse_u64(c[3], pos5) || !parse_u64(m[3], pos3)) bad("Malformed anchor offset", line);
        const int c5 = contig(c[2]), c3 = contig(m[2]);
        if (pos5 >= 0x80000000ull || pos3 >= 0x80000000ull) throw ParseError("Anchor offset beyond 2^31.");
        bf_hit h;
        memset(&h, 0, sizeof h);
        h.pos5 = (uint32_t)pos5; h.pos3 = (uint32_t)pos3;
        h.contig5 = (uint16_t)c5; h.contig3 = (uint16_t)c3;
        h.strands = (uint8_t)((c[1] == "+" ? 1 : 0) | (m[1] == "+" ? 2 : 0));
        out.hits.push_back(h);
        if (pack) {
            const size_t L = a[3].size(), at = out.seq.size();
            if (L > 65535) throw Unpackable("a read of more than 65535 bases");
            out.seq.resize(at + (L + 1) / 2);
            if (!pack_bases(a[3].data(), L, (uint8_t*)out.seq.data() + at))
                throw Unpackable("a read holds a character outside BAM's base alphabet: " + std::string(a[3].substr(0, 200)));
            out.seq_len.push_back((uint16_t)L);
            out.packed = true;
        } else {
            out.seq.append(a[3].data(), a[3].size());
        }
        out.seq_off.push_back(out.seq.size());
        out.frag.append(a[2].data(), a[2].size()); out.frag_off.push_back(out.frag.size());
    }
};

struct Block {
    uint64_t seq = 0;
    const char* p = nullptr;
    size_t n = 0;
    std::unique_ptr<char[]> own;          // empty for views into a mapping
};

struct Lead { std::string line; uint64_t read_num; bool is3p; };

struct Parsed {
    Batch batch;
    std::vector<Lead> lead;               // lines before the block's first 5p line
    bool has5 = false;
    std::string last5;
    uint64_t last5_num = 0;
    std::string error;
    bool unpackable = false;
    uint64_t lines = 0, bytes = 0;
};

void parse_block(const Block& b, const PairParser& pair, Parsed& out) {
    const char* s = b.p;
    const char* end = b.p + b.n;
    std::string_view prev;
    uint64_t prev_num = 0;
    size_t prev_at = 0;                          // canonical_number() of `prev`
    out.bytes = b.n;
    out.batch.seq.reserve(pair.pack ? b.n * 3 / 8 : b.n * 3 / 4);
    out.batch.packed = pair.pack;
    out.batch.frag.reserve(b.n / 8);
    const size_t pairs = b.n / 160 + 16;        // a guess that saves the first doublings; the arrays grow as needed
    out.batch.hits.reserve(pairs); out.batch.seq_off.reserve(pairs + 1); out.batch.frag_off.reserve(pairs + 1);
    if (pair.pack) out.batch.seq_len.reserve(pairs);
    try {
        while (s < end) {
            const char* nl = (const char*)memchr(s, '\n', (size_t)(end - s));
            const char* e = nl ? nl : end;
            size_t len = (size_t)(e - s);
            if (nl && len && s[len - 1] == '\r') len--;       // BufRead::lines strips "\n" or "\r\n"
            const std::string_view line(s, len);
            s = nl ? nl + 1 : end;
            out.lines++;
            uint64_t num = 0;
            const size_t at = canonical_number(line, num);
            if (!at) num = read_number(line);
            const bool is5 = line.substr(0, 3) == "5p:", is3 = line.substr(0, 3) == "3p:";
            if (is5) { prev = line; prev_num = num; prev_at = at; out.has5 = true; continue; }      // detect.rs:81-84
            if (!out.has5) { out.lead.push_back(Lead{std::string(line), num, is3}); continue; }
            if (is3 && num != prev_num) continue;                                      // detect.rs:85-87
            if (!(is3 && pair.fast(prev, prev_at, line, at, out.batch))) pair(prev, line, out.batch);
        }
    } catch (const Unpackable& ex) {
        out.error = ex.what(); out.unpackable = true;
    } catch (const std::exception& ex) {
        out.error = ex.what();
    }
    if (out.has5) { out.last5.assign(prev); out.last5_num = prev_num; }
}

struct Pipeline {
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_room;
    std::deque<Block> work;
    std::map<uint64_t, std::unique_ptr<Parsed>> done;
    uint64_t produced = 0, consumed = 0;
    bool eof = false, abort = false;
    std::string reader_error;
    size_t max_in_flight = 4;
    std::vector<Batch> spare;             // batches whose buffers are kept for the next block (no allocator churn, no fresh pages)
};

void push_block(Pipeline& pl, Block&& b) {
    std::unique_lock<std::mutex> lk(pl.mu);
    pl.cv_room.wait(lk, [&] { return pl.abort || pl.produced - pl.consumed < pl.max_in_flight; });
    if (pl.abort) return;
    b.seq = pl.produced++;
    pl.work.push_back(std::move(b));
    pl.cv_work.notify_one();
}

void read_mapped(Pipeline& pl, const char* base, size_t size, size_t block_bytes, size_t n_workers) {
    size_t pos = 0;
    while (pos < size) {
        // towards the end the blocks get smaller, so that the workers finish together
        const size_t left = size - pos;
        const size_t want = left < block_bytes * n_workers ? std::max<size_t>(block_bytes / 8, std::min<size_t>(block_bytes, 1u << 20)) : block_bytes;
        size_t end = std::min(size, pos + want);
        if (end < size) {
            const void* back = memrchr(base + pos, '\n', end - pos);
            if (back) end = (size_t)((const char*)back - base) + 1;
            else {
                const void* fwd = memchr(base + end, '\n', size - end);
                end = fwd ? (size_t)((const char*)fwd - base) + 1 : size;
            }
        }
        Block b;
        b.p = base + pos; b.n = end - pos;
        push_block(pl, std::move(b));
        pos = end;
        { std::lock_guard<std::mutex> g(pl.mu); if (pl.abort) return; }
    }
}

void read_stream(Pipeline& pl, int fd, size_t block_bytes) {
    std::unique_ptr<char[]> carry;
    size_t carry_n = 0;
    bool at_eof = false;
    while (!at_eof) {
        size_t cap = std::max(block_bytes, carry_n * 2);
        std::unique_ptr<char[]> buf(new char[cap]);
        if (carry_n) memcpy(buf.get(), carry.get(), carry_n);
        size_t n = carry_n;
        carry_n = 0;
        const void* last_nl = nullptr;
        for (;;) {
            while (n < cap) {
                const ssize_t r = read(fd, buf.get() + n, cap - n);
                if (r < 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("read failed: ") + strerror(errno)); }
                if (r == 0) { at_eof = true; break; }
                n += (size_t)r;
            }
            last_nl = at_eof ? nullptr : memrchr(buf.get(), '\n', n);
            if (at_eof || last_nl) break;
            std::unique_ptr<char[]> bigger(new char[cap * 2]);       // one line longer than the block
            memcpy(bigger.get(), buf.get(), n);
            buf = std::move(bigger);
            cap *= 2;
        }
        size_t take = n;
        if (!at_eof) {
            take = (size_t)((const char*)last_nl - buf.get()) + 1;
            carry_n = n - take;
            if (carry_n) { carry.reset(new char[carry_n]); memcpy(carry.get(), buf.get() + take, carry_n); }
        }
        if (take) {
            Block b;
            b.p = buf.get(); b.n = take; b.own = std::move(buf);
            push_block(pl, std::move(b));
        }
        { std::lock_guard<std::mutex> g(pl.mu); if (pl.abort) return; }
    }
}

}  // namespace

Stats parse_stream(int fd, const ContigIndex& contigs, const std::string& genome_path, const Options& opt,
                   const std::function<void(Batch&)>& sink) {
    const int n_workers = std::max(1, opt.threads);
    const size_t block_bytes = std::max<size_t>(opt.block_bytes, 1);
    Pipeline pl;
    pl.max_in_flight = (size_t)n_workers * 2 + 2;

    // a regular file is mapped: blocks are views, nothing is copied before the parse
    const char* map_base = nullptr;
    size_t map_size = 0;
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && lseek(fd, 0, SEEK_CUR) == 0) {
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            map_base = (const char*)m; map_size = (size_t)st.st_size;
            (void)madvise(m, map_size, MADV_SEQUENTIAL);
        }
    }

    std::thread reader([&] {
        try {
            if (map_base) read_mapped(pl, map_base, map_size, block_bytes, (size_t)n_workers);
            else read_stream(pl, fd, block_bytes);
        } catch (const std::exception& ex) {
            std::lock_guard<std::mutex> g(pl.mu);
            pl.reader_error = ex.what();
        }
        std::lock_guard<std::mutex> g(pl.mu);
        pl.eof = true;
        pl.cv_work.notify_all();
        pl.cv_done.notify_all();
    });

    const PairParser proto{contigs, genome_path, opt.pack};
    std::atomic<uint64_t> worker_ns{0};
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int w = 0; w < n_workers; w++) {
        workers.emplace_back([&] {
            PairParser pair{contigs, genome_path, opt.pack};
            for (;;) {
                Block b;
                {
                    std::unique_lock<std::mutex> lk(pl.mu);
                    pl.cv_work.wait(lk, [&] { return pl.abort || !pl.work.empty() || pl.eof; });
                    if (pl.abort || pl.work.empty()) return;
                    b = std::move(pl.work.front());
                    pl.work.pop_front();
                }
                auto out = std::make_unique<Parsed>();
                {
                    std::lock_guard<std::mutex> g(pl.mu);
                    if (!pl.spare.empty()) { out->batch = std::move(pl.spare.back()); pl.spare.pop_back(); }
                }
                const Allocator* al = opt.allocator ? opt.allocator->load(std::memory_order_acquire) : nullptr;
                out->batch.clear(al ? al : heap_allocator());
                const auto t0 = std::chrono::steady_clock::now();
                parse_block(b, pair, *out);
                worker_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                std::lock_guard<std::mutex> g(pl.mu);
                pl.done[b.seq] = std::move(out);
                pl.cv_done.notify_all();
            }
        });
    }

    auto shut_down = [&](bool abort) {
        {
            std::lock_guard<std::mutex> g(pl.mu);
            if (abort) pl.abort = true;
            pl.cv_work.notify_all(); pl.cv_room.notify_all(); pl.cv_done.notify_all();
        }
        if (reader.joinable()) reader.join();
        for (auto& t : workers) if (t.joinable()) t.join();
        if (map_base) munmap((void*)map_base, map_size);
    };

    Stats stats;
    std::string prev;                      // the most recent 5p line of the stream so far (detect.rs:73)
    uint64_t prev_num = 0;                 // detect.rs:74
    Batch lead_batch;
    try {
        for (;;) {
            std::unique_ptr<Parsed> p;
            {
                std::unique_lock<std::mutex> lk(pl.mu);
                pl.cv_done.wait(lk, [&] { return pl.done.count(pl.consumed) || (pl.eof && pl.consumed == pl.produced); });
                auto it = pl.done.find(pl.consumed);
                if (it == pl.done.end()) {
                    if (!pl.reader_error.empty()) throw ParseError(pl.reader_error);
                    break;
                }
                p = std::move(it->second);
                pl.done.erase(it);
            }
            // the block's lines before its first 5p line, against the stream's state (detect.rs:85-103)
            lead_batch.clear();
            lead_batch.packed = opt.pack;
            for (const Lead& l : p->lead) {
                if (l.is3p && l.read_num != prev_num) continue;
                proto(prev, l.line, lead_batch);
            }
            if (p->unpackable) throw Unpackable(p->error);
            if (!p->error.empty()) throw ParseError(p->error);
            if (lead_batch.n()) { stats.pairs += lead_batch.n(); sink(lead_batch); }
            if (p->batch.n()) { stats.pairs += p->batch.n(); sink(p->batch); }
            if (p->has5) { prev = std::move(p->last5); prev_num = p->last5_num; }
            stats.bytes += p->bytes; stats.lines += p->lines; stats.blocks++;
            {
                // whatever the sink left in the batch (its own spent buffers, if it swapped) serves the next block
                std::lock_guard<std::mutex> g(pl.mu);
                if (p->batch.has_room() && pl.spare.size() < pl.max_in_flight + 8) pl.spare.push_back(std::move(p->batch));
            }
            p.reset();
            {
                std::lock_guard<std::mutex> g(pl.mu);
                pl.consumed++;
                pl.cv_room.notify_all();
            }
        }
    } catch (...) {
        shut_down(true);
        throw;
    }
    shut_down(false);
    if (getenv("BREAKFAST_HIP_PARSER_TIMING"))
        fprintf(stderr, "parser: wall %.3f s, workers busy %.3f s in total (%d threads), %llu blocks\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), worker_ns.load() * 1e-9, n_workers,
                (unsigned long long)stats.blocks);
    return stats;
}

}  // namespace hits
