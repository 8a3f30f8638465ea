// See bam_reader.h.  Follows the BGZF/BAM layout of the SAM specification (SAMv1 section 4); the
// reference reads the same bytes through rust-htslib 0.38 (Cargo.toml:16; detect.rs:292-321).

#include "bam_reader.h"

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>

#include <dlfcn.h>
#include <zlib.h>

namespace bam {

namespace {
// libdeflate (what htslib itself prefers for BGZF) inflates 2-3x faster than zlib.  The image ships the
// runtime library without its header, so the four entry points are bound by hand; without the library
// the workers fall back to zlib.
struct Deflate {
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*inflate)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    Deflate() {
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        inflate = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        if (!alloc || !release || !inflate || !crc) alloc = nullptr;
    }
    bool ok() const { return alloc != nullptr; }
};
const Deflate& deflate_lib() { static const Deflate d; return d; }

struct Block {
    uint64_t seq = 0;
    std::vector<uint8_t> comp;
    uint32_t isize = 0, crc = 0;
    std::vector<uint8_t> data;
};
uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
}  // namespace

struct Reader::Impl {
    FILE* f = nullptr;
    bool own = false;
    int threads = 1;
    std::thread io;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_todo, cv_done, cv_space;
    std::deque<Block> todo;
    std::map<uint64_t, Block> done;
    uint64_t n_read = 0;           // blocks handed out by the io thread
    uint64_t next_out = 0;         // next block the consumer wants
    bool io_finished = false, stop = false;
    std::string error;
    // consumer state
    std::vector<uint8_t> cur;
    size_t cur_pos = 0;

    void fail(const std::string& m) {
        std::lock_guard<std::mutex> g(mu);
        if (error.empty()) error = m;
        stop = true;
        cv_todo.notify_all(); cv_done.notify_all(); cv_space.notify_all();
    }

    void io_loop() {
        const size_t max_pending = (size_t)threads * 8 + 8;
        for (;;) {
            uint8_t h[18];
            size_t got = fread(h, 1, 12, f);
            if (got == 0) break;                                   // clean end of file
            if (got != 12 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { fail("not a BGZF-compressed BAM file"); return; }
            const uint32_t xlen = le16(h + 10);
            std::vector<uint8_t> extra(xlen);
            if (fread(extra.data(), 1, xlen, f) != xlen) { fail("truncated BGZF block header"); return; }
            int64_t bsize = -1;
            for (size_t o = 0; o + 4 <= xlen;) {
                const uint32_t slen = le16(extra.data() + o + 2);
                if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = le16(extra.data() + o + 4);
                o += 4 + slen;
            }
            if (bsize < 0) { fail("BGZF block without a BC field"); return; }
            const int64_t clen = bsize - (int64_t)xlen - 19;
            if (clen < 0) { fail("corrupt BGZF block size"); return; }
            Block b;
            b.comp.resize((size_t)clen);
            uint8_t tail[8];
            if (fread(b.comp.data(), 1, (size_t)clen, f) != (size_t)clen || fread(tail, 1, 8, f) != 8) { fail("truncated BGZF block"); return; }
            b.crc = le32(tail); b.isize = le32(tail + 4);
            std::unique_lock<std::mutex> lk(mu);
            cv_space.wait(lk, [&] { return stop || todo.size() + done.size() < max_pending; });
            if (stop) return;
            b.seq = n_read++;
            todo.push_back(std::move(b));
            cv_todo.notify_one();
        }
        std::lock_guard<std::mutex> g(mu);
        io_finished = true;
        cv_todo.notify_all(); cv_done.notify_all();
    }

    void work_loop() {
        const Deflate& ld = deflate_lib();
        void* dec = ld.ok() ? ld.alloc() : nullptr;
        struct Free { const Deflate& l; void* d; ~Free() { if (d) l.release(d); } } free_dec{ld, dec};
        for (;;) {
            Block b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_todo.wait(lk, [&] { return stop || !todo.empty() || io_finished; });
                if (stop) return;
                if (todo.empty()) { if (io_finished) return; continue; }
                b = std::move(todo.front());
                todo.pop_front();
            }
            b.data.resize(b.isize);
            if (b.isize && dec) {
                size_t got = 0;
                if (ld.inflate(dec, b.comp.data(), b.comp.size(), b.data.data(), b.data.size(), &got) != 0 || got != b.data.size()) {
                    fail("corrupt BGZF block (inflate failed)"); return;
                }
                if (ld.crc(0, b.data.data(), b.data.size()) != b.crc) { fail("corrupt BGZF block (CRC mismatch)"); return; }
            } else if (b.isize) {
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) { fail("zlib initialisation failed"); return; }
                zs.next_in = b.comp.data(); zs.avail_in = (uInt)b.comp.size();
                zs.next_out = b.data.data(); zs.avail_out = (uInt)b.data.size();
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.avail_out != 0) { fail("corrupt BGZF block (inflate failed)"); return; }
                if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), b.data.data(), (uInt)b.data.size()) != b.crc) { fail("corrupt BGZF block (CRC mismatch)"); return; }
            }
            std::vector<uint8_t>().swap(b.comp);
            std::lock_guard<std::mutex> g(mu);
            const uint64_t s = b.seq;
            done.emplace(s, std::move(b));
            cv_done.notify_all();
        }
    }

    // next decompressed block in file order; false at end of file
    bool next_block() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return stop || done.count(next_out) || (io_finished && next_out >= n_read); });
        if (!error.empty()) throw std::runtime_error(error);
        auto it = done.find(next_out);
        if (it == done.end()) return false;
        cur = std::move(it->second.data);
        cur_pos = 0;
        done.erase(it);
        next_out++;
        cv_space.notify_all();
        return true;
    }
};

bool Reader::read_exact(void* dst, size_t n, bool allow_eof) {
    uint8_t* d = (uint8_t*)dst;
    size_t got = 0;
    while (got < n) {
        if (p_->cur_pos == p_->cur.size()) {
            if (!p_->next_block()) {
                if (got == 0 && allow_eof) return false;
                throw std::runtime_error("truncated file");
            }
            continue;
        }
        const size_t k = std::min(n - got, p_->cur.size() - p_->cur_pos);
        memcpy(d + got, p_->cur.data() + p_->cur_pos, k);
        p_->cur_pos += k; got += k;
    }
    return true;
}

Reader::Reader(const std::string& path, int threads) : p_(new Impl()) {
    p_->threads = threads < 1 ? 1 : threads;
    if (path == "-") { p_->f = stdin; }
    else { p_->f = fopen(path.c_str(), "rb"); p_->own = true; }
    if (!p_->f) { delete p_; p_ = nullptr; throw std::runtime_error("cannot open"); }
    p_->io = std::thread([this] { p_->io_loop(); });
    for (int i = 0; i < p_->threads; i++) p_->workers.emplace_back([this] { p_->work_loop(); });
    try {
        uint8_t magic[4];
        read_exact(magic, 4, false);
        if (memcmp(magic, "BAM\1", 4) != 0) throw std::runtime_error("not a BAM file");
        uint8_t b4[4];
        read_exact(b4, 4, false);
        std::string text(le32(b4), '\0');
        if (!text.empty()) read_exact(&text[0], text.size(), false);
        read_exact(b4, 4, false);
        const uint32_t n_ref = le32(b4);
        for (uint32_t i = 0; i < n_ref; i++) {
            read_exact(b4, 4, false);
            std::string name(le32(b4), '\0');
            if (!name.empty()) read_exact(&name[0], name.size(), false);
            while (!name.empty() && name.back() == '\0') name.pop_back();
            read_exact(b4, 4, false);
            refs_.push_back(name);
        }
    } catch (...) {
        this->~Reader();
        p_ = nullptr;
        throw;
    }
}

Reader::~Reader() {
    if (!p_) return;
    {
        std::lock_guard<std::mutex> g(p_->mu);
        p_->stop = true;
        p_->cv_todo.notify_all(); p_->cv_done.notify_all(); p_->cv_space.notify_all();
    }
    if (p_->io.joinable()) p_->io.join();
    for (auto& w : p_->workers) if (w.joinable()) w.join();
    if (p_->own && p_->f) fclose(p_->f);
    delete p_;
    p_ = nullptr;
}

bool Reader::next_view(RecordView& v) {
    uint32_t block_size;
    if (p_->cur.size() - p_->cur_pos >= 4) {                     // the usual case: the length lies in the current block
        block_size = le32(p_->cur.data() + p_->cur_pos);
        p_->cur_pos += 4;
    } else {
        uint8_t b4[4];
        if (!read_exact(b4, 4, true)) return false;
        block_size = le32(b4);
    }
    if (block_size < 32) throw std::runtime_error("corrupt BAM record");
    const uint8_t* p;
    if (p_->cur.size() - p_->cur_pos >= block_size) {            // whole record inside the current block
        p = p_->cur.data() + p_->cur_pos;
        p_->cur_pos += block_size;
        // the walk touches a record's first line or two and hops on: ask for the next record's while this one is
        // looked at (the block was inflated by another thread, none of it is in this core's cache)
        if (p_->cur.size() - p_->cur_pos >= 64) __builtin_prefetch(p + block_size + 4, 0, 1);
    } else {
        spill_.resize(block_size);
        read_exact(spill_.data(), block_size, false);
        p = spill_.data();
    }
    v.body = p; v.size = block_size;
    v.l_read_name = p[8];
    v.n_cigar = le16(p + 12);
    v.flag = le16(p + 14);
    v.l_seq = le32(p + 16);
    const uint64_t need = 32ull + v.l_read_name + 4ull * v.n_cigar + ((uint64_t)v.l_seq + 1) / 2;
    if (need > block_size) throw std::runtime_error("corrupt BAM record");
    return true;
}

uint32_t RecordView::cigar(uint32_t i) const { return le32(body + 32 + l_read_name + 4 * i); }
void RecordView::qname(std::string& out) const { out.assign((const char*)body + 32, l_read_name ? l_read_name - 1 : 0); }
void RecordView::seq(std::string& out) const {
    static const char kNib[] = "=ACMGRSVTWYHKDBN";
    const uint8_t* q = body + 32 + l_read_name + 4 * n_cigar;
    out.resize(l_seq);
    for (uint32_t i = 0; i < l_seq; i++) out[i] = kNib[(q[i >> 1] >> ((~i & 1) << 2)) & 15];
}
// rust-htslib 0.38 CigarStringView::leading_softclips / trailing_softclips: the soft clip at the
// end of the alignment, looking through one hard clip.
int64_t RecordView::leading_softclips() const {
    if (n_cigar == 0) return 0;
    const uint32_t c0 = cigar(0);
    if ((c0 & 15) == 4) return c0 >> 4;
    if ((c0 & 15) == 5 && n_cigar > 1 && (cigar(1) & 15) == 4) return cigar(1) >> 4;
    return 0;
}
int64_t RecordView::trailing_softclips() const {
    if (n_cigar == 0) return 0;
    const uint32_t c0 = cigar(n_cigar - 1);
    if ((c0 & 15) == 4) return c0 >> 4;
    if ((c0 & 15) == 5 && n_cigar > 1 && (cigar(n_cigar - 2) & 15) == 4) return cigar(n_cigar - 2) >> 4;
    return 0;
}

bool Reader::next(Record& rec) {
    RecordView v;
    if (!next_view(v)) return false;
    rec.flag = v.flag;
    v.qname(rec.qname);
    rec.cigar.resize(v.n_cigar);
    for (uint32_t i = 0; i < v.n_cigar; i++) rec.cigar[i] = v.cigar(i);
    v.seq(rec.seq);
    return true;
}

// rust-htslib 0.38 CigarStringView::leading_softclips / trailing_softclips: the soft clip at the
// end of the alignment, looking through one hard clip.
int64_t Record::leading_softclips() const {
    if (cigar.empty()) return 0;
    const uint32_t op0 = cigar[0] & 15;
    if (op0 == 4) return cigar[0] >> 4;
    if (op0 == 5 && cigar.size() > 1 && (cigar[1] & 15) == 4) return cigar[1] >> 4;
    return 0;
}
int64_t Record::trailing_softclips() const {
    if (cigar.empty()) return 0;
    const size_t n = cigar.size();
    const uint32_t op0 = cigar[n - 1] & 15;
    if (op0 == 4) return cigar[n - 1] >> 4;
    if (op0 == 5 && n > 1 && (cigar[n - 2] & 15) == 4) return cigar[n - 2] >> 4;
    return 0;
}

uint64_t dispatch_reads(Reader& bam, FILE* out, uint32_t anchor_len) {
    uint64_t num_reads_sent = 0;
    RecordView read;
    std::string frag_id, seq, line;
    char num[24];
    while (bam.next_view(read)) {                                       // detect.rs:302
        if (!read.is_unmapped()) {                                      // detect.rs:305-310
            if (read.leading_softclips() < (int64_t)anchor_len && read.trailing_softclips() < (int64_t)anchor_len) continue;
        }
        if (read.l_seq < (uint64_t)anchor_len * 2) continue;            // detect.rs:311
        read.seq(seq);
        read.qname(frag_id);                                            // detect.rs:315-318
        for (char& c : frag_id) if (c == ':') c = '_';
        num_reads_sent++;
        const int nn = snprintf(num, sizeof num, "%llu", (unsigned long long)num_reads_sent);
        line.clear();
        line += ">5p:"; line.append(num, (size_t)nn); line += ":\n";    // detect.rs:326-327
        line.append(seq, 0, anchor_len);
        line += "\n>3p:"; line.append(num, (size_t)nn); line += ':';    // detect.rs:330-336
        line += frag_id; line += ':'; line += seq; line += ":\n";
        line.append(seq, seq.size() - anchor_len, anchor_len);
        line += '\n';
        fwrite(line.data(), 1, line.size(), out);
    }
    return num_reads_sent;
}

}  // namespace bam
