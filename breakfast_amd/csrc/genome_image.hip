// The packed genome image: what the GPU keeps of a genome (2 bits a base + the runs of lower-case letters + the runs of
// every other byte), made once on the host and kept in a file beside the FASTA, the way Bowtie keeps its index beside it.
// The reference reads the FASTA on every run (detect.rs:48-56: 3.1 GB for hg38, every byte kept); a caller that has an
// image hands 0.8 GB to the GPU instead of 3.1 GB, and nothing of the FASTA has to be read, joined or packed again.
// Every byte of the FASTA can be rebuilt from the image (bf_genome_read), so results are the same bytes.
//
//   bf_genome_image_write   host only: contigs -> file
//   bf_genome_image_open    maps and checks a file (size and sections, the source's size and mtime)
//   bf_genome_set_image     uploads it (pinned slots, a few host threads) and builds the resident planes on the GPU
//                           (= bf_genome_begin_image + bf_genome_write_image + bf_genome_end)
//
// File (little-endian, sections 64-byte aligned): header, contig lengths u64[n], name offsets u32[n+1], names,
// blocks {lo u64, hi u64}[n_blocks] (bit k of lo/hi = bit 0/1 of the code (byte >> 1) & 3 of base 64*block + k; contig i
// starts at block sum_{j<i} ceil(len_j / 64)), lower-case runs {start u64[], len u32[]}, other runs {start u64[],
// len u32[], byte u8[]}; starts are global base offsets (64 * block + k), ascending, runs do not overlap.

#include "bf_internal.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace bf;

namespace {

constexpr char kMagic[8] = {'B', 'F', 'G', 'E', 'N', 'O', 'M', '1'};
constexpr uint32_t kImageVersion = 1;

struct ImageHeader {                 // 192 bytes
    char magic[8];
    uint32_t version, n_contigs;
    uint64_t src_size;
    int64_t src_mtime_ns;
    uint64_t n_blocks, n_lower, n_other, names_bytes;
    uint64_t off_lens, off_name_off, off_names, off_blocks, off_lower_start, off_lower_len, off_other_start, off_other_len,
        off_other_byte, file_size;
    uint64_t reserved[6];
};
static_assert(sizeof(ImageHeader) == 192, "header layout");

struct Run { uint64_t start, len; uint8_t byte; };

inline int byte_class(uint8_t c) {   // 0: one of ACGT, 1: one of acgt, 2: anything else
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': return 0;
        case 'a': case 'c': case 'g': case 't': return 1;
        default: return 2;
    }
}

// bases [a, b) of a contig (a a multiple of 64) whose first base has global offset goff
void pack_chunk(const uint8_t* seq, uint64_t a, uint64_t b, uint64_t goff, GenomeBlock* blocks, std::vector<Run>& lower,
                std::vector<Run>& other) {
    for (uint64_t blk = a >> 6; (blk << 6) < b; blk++) {
        uint64_t lo = 0, hi = 0;
        const uint64_t e = std::min(b, (blk << 6) + 64);
        for (uint64_t i = blk << 6; i < e; i++) {
            const uint8_t c = seq[i];
            const int k = (int)(i & 63);
            const uint64_t code = (c >> 1) & 3;
            lo |= (code & 1) << k;
            hi |= (code >> 1) << k;
            const int cls = byte_class(c);
            if (cls == 0) continue;
            const uint64_t g = goff + i;
            if (cls == 1) {
                if (!lower.empty() && lower.back().start + lower.back().len == g) lower.back().len++;
                else lower.push_back(Run{g, 1, 0});
            } else {
                if (!other.empty() && other.back().start + other.back().len == g && other.back().byte == c) other.back().len++;
                else other.push_back(Run{g, 1, c});
            }
        }
        blocks[blk] = GenomeBlock{lo, hi};
    }
}

void append_runs(std::vector<Run>& all, const std::vector<Run>& part, bool by_byte) {
    for (const Run& r : part) {
        if (!all.empty() && all.back().start + all.back().len == r.start && (!by_byte || all.back().byte == r.byte) &&
            all.back().len + r.len <= 0xffffffffull)
            all.back().len += r.len;
        else if (r.len <= 0xffffffffull) all.push_back(r);
        else for (uint64_t a = 0; a < r.len; a += 0x80000000ull) all.push_back(Run{r.start + a, std::min<uint64_t>(0x80000000ull, r.len - a), r.byte});
    }
}

uint64_t align64(uint64_t x) { return (x + 63) & ~63ull; }

bool write_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
        const ssize_t w = write(fd, c, std::min<size_t>(n, 1u << 30));
        if (w <= 0) return false;
        c += w; n -= (size_t)w;
    }
    return true;
}

}  // namespace

struct bf_genome_image {
    void* map = MAP_FAILED;
    size_t size = 0;
    ImageHeader h{};
    std::vector<std::string> name_store;
    std::vector<const char*> names;
    const uint64_t* lens = nullptr;
    const uint8_t* at(uint64_t off) const { return (const uint8_t*)map + off; }
};

extern "C" {

int bf_genome_image_write(const char* path, uint32_t n_contigs, const char* const* names, const uint8_t* const* seqs,
                          const uint64_t* lens, uint64_t src_size, int64_t src_mtime_ns, uint32_t threads) {
    if (!path || !names || !seqs || !lens || n_contigs == 0) return fail(nullptr, BF_ERR_INVALID, "bf_genome_image_write: null/empty argument");
    if (n_contigs > (1u << kContigBits)) return fail(nullptr, BF_ERR_RANGE, "%u contigs; at most %u are supported", n_contigs, 1u << kContigBits);
    std::vector<uint64_t> first_block(n_contigs + 1, 0);
    for (uint32_t i = 0; i < n_contigs; i++) {
        if (lens[i] >= 0x80000000ull - (1ull << 17)) return fail(nullptr, BF_ERR_RANGE, "contig '%s' has %llu bases; the supported maximum is 2^31 - 2^17", names[i], (unsigned long long)lens[i]);
        if (lens[i] && !seqs[i]) return fail(nullptr, BF_ERR_INVALID, "bf_genome_image_write: contig %u has no bytes", i);
        first_block[i + 1] = first_block[i] + ((lens[i] + 63) >> 6);
    }
    const uint64_t n_blocks = first_block[n_contigs];
    std::vector<GenomeBlock> blocks;
    try { blocks.resize(n_blocks); } catch (...) { return fail(nullptr, BF_ERR_NOMEM, "bf_genome_image_write: no memory for %llu blocks", (unsigned long long)n_blocks); }
    struct Chunk { uint32_t contig; uint64_t a, b; std::vector<Run> lower, other; };
    std::vector<Chunk> chunks;
    constexpr uint64_t kChunk = 4u << 20;
    for (uint32_t i = 0; i < n_contigs; i++)
        for (uint64_t a = 0; a < lens[i]; a += kChunk) chunks.push_back(Chunk{i, a, std::min(lens[i], a + kChunk), {}, {}});
    uint32_t nt = threads ? threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    nt = (uint32_t)std::max<size_t>(1, std::min<size_t>(nt, chunks.size()));
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t k; (k = next.fetch_add(1)) < chunks.size();) {
            Chunk& c = chunks[k];
            pack_chunk(seqs[c.contig], c.a, c.b, first_block[c.contig] << 6, blocks.data() + first_block[c.contig], c.lower, c.other);
        }
    };
    if (nt == 1) work();
    else {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    std::vector<Run> lower, other;
    for (Chunk& c : chunks) {
        append_runs(lower, c.lower, false); append_runs(other, c.other, true);
        std::vector<Run>().swap(c.lower); std::vector<Run>().swap(c.other);
    }
    if (other.size() > 0xfffffff0ull) return fail(nullptr, BF_ERR_RANGE, "the genome holds %zu runs of bytes outside ACGTacgt; the supported maximum is 2^32 - 16", other.size());

    std::vector<uint32_t> name_off(n_contigs + 1, 0);
    std::string name_bytes;
    for (uint32_t i = 0; i < n_contigs; i++) { name_bytes += names[i]; name_off[i + 1] = (uint32_t)name_bytes.size(); }
    ImageHeader h{};
    memcpy(h.magic, kMagic, 8);
    h.version = kImageVersion; h.n_contigs = n_contigs; h.src_size = src_size; h.src_mtime_ns = src_mtime_ns;
    h.n_blocks = n_blocks; h.n_lower = lower.size(); h.n_other = other.size(); h.names_bytes = name_bytes.size();
    uint64_t off = align64(sizeof h);
    h.off_lens = off; off = align64(off + 8ull * n_contigs);
    h.off_name_off = off; off = align64(off + 4ull * (n_contigs + 1));
    h.off_names = off; off = align64(off + name_bytes.size());
    h.off_blocks = off; off = align64(off + 16ull * n_blocks);
    h.off_lower_start = off; off = align64(off + 8ull * lower.size());
    h.off_lower_len = off; off = align64(off + 4ull * lower.size());
    h.off_other_start = off; off = align64(off + 8ull * other.size());
    h.off_other_len = off; off = align64(off + 4ull * other.size());
    h.off_other_byte = off; off = align64(off + other.size());
    h.file_size = off;

    // into a temporary beside the target, renamed when complete: a reader never sees half an image
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(nullptr, BF_ERR_IO, "bf_genome_image_write: %s cannot be created", tmp.c_str());
    uint64_t at = 0;
    static const char zeros[64] = {0};
    auto section = [&](uint64_t where, const void* p, size_t n) {
        if (where < at) return false;
        while (at < where) { const size_t m = (size_t)std::min<uint64_t>(64, where - at); if (!write_all(fd, zeros, m)) return false; at += m; }
        if (n && !write_all(fd, p, n)) return false;
        at += n;
        return true;
    };
    bool ok = section(0, &h, sizeof h) && section(h.off_lens, lens, 8ull * n_contigs) && section(h.off_name_off, name_off.data(), 4ull * (n_contigs + 1)) &&
              section(h.off_names, name_bytes.data(), name_bytes.size()) && section(h.off_blocks, blocks.data(), 16ull * n_blocks);
    auto column = [&](uint64_t where, const std::vector<Run>& runs, int what) {      // 0 start, 1 len, 2 byte
        constexpr size_t kPiece = 1u << 16;
        std::vector<uint64_t> buf(kPiece);
        if (!section(where, nullptr, 0)) return false;
        for (size_t a = 0; a < runs.size(); a += kPiece) {
            const size_t m = std::min(kPiece, runs.size() - a);
            size_t bytes;
            if (what == 0) { for (size_t i = 0; i < m; i++) buf[i] = runs[a + i].start; bytes = 8 * m; }
            else if (what == 1) { uint32_t* q = (uint32_t*)buf.data(); for (size_t i = 0; i < m; i++) q[i] = (uint32_t)runs[a + i].len; bytes = 4 * m; }
            else { uint8_t* q = (uint8_t*)buf.data(); for (size_t i = 0; i < m; i++) q[i] = runs[a + i].byte; bytes = m; }
            if (!write_all(fd, buf.data(), bytes)) return false;
            at += bytes;
        }
        return true;
    };
    ok = ok && column(h.off_lower_start, lower, 0) && column(h.off_lower_len, lower, 1) && column(h.off_other_start, other, 0) &&
         column(h.off_other_len, other, 1) && column(h.off_other_byte, other, 2) && section(h.file_size, nullptr, 0);
    ok = (close(fd) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        (void)unlink(tmp.c_str());
        return fail(nullptr, BF_ERR_IO, "bf_genome_image_write: writing %s failed", path);
    }
    return BF_OK;
}

int bf_genome_image_open(const char* path, uint64_t src_size, int64_t src_mtime_ns, bf_genome_image** out) {
    if (!path || !out) return fail(nullptr, BF_ERR_INVALID, "bf_genome_image_open: null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(nullptr, BF_ERR_IO, "genome image %s cannot be opened", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < sizeof(ImageHeader)) {
        close(fd);
        return fail(nullptr, BF_ERR_IO, "%s is not a genome image (too short, or not a file)", path);
    }
    auto img = new bf_genome_image;
    img->size = (size_t)st.st_size;
    img->map = mmap(nullptr, img->size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    auto bad = [&](const char* what) {
        if (img->map != MAP_FAILED) munmap(img->map, img->size);
        delete img;
        return fail(nullptr, BF_ERR_IO, "%s: %s", path, what);
    };
    if (img->map == MAP_FAILED) return bad("cannot be mapped");
    memcpy(&img->h, img->map, sizeof(ImageHeader));
    const ImageHeader& h = img->h;
    if (memcmp(h.magic, kMagic, 8) != 0) return bad("not a genome image (magic)");
    if (h.version != kImageVersion) return bad("a genome image of another version; build it again");
    if (h.file_size != img->size) return bad("genome image is incomplete (size differs from its header's)");
    if ((src_size || src_mtime_ns) && (h.src_size != src_size || h.src_mtime_ns != src_mtime_ns))
        return bad("genome image is stale (made from a file of another size or modification time)");
    if (h.n_contigs == 0 || h.n_contigs > (1u << kContigBits)) return bad("genome image is damaged (contig count)");
    if (h.n_other > 0xfffffff0ull || h.n_blocks > (1ull << 31)) return bad("genome image is damaged (counts)");
    auto inside = [&](uint64_t off, uint64_t bytes) { return (off & 7) == 0 && off <= img->size && bytes <= img->size - off; };
    if (!inside(h.off_lens, 8ull * h.n_contigs) || !inside(h.off_name_off, 4ull * (h.n_contigs + 1)) || !inside(h.off_names, h.names_bytes) ||
        !inside(h.off_blocks, 16ull * h.n_blocks) || !inside(h.off_lower_start, 8 * h.n_lower) || !inside(h.off_lower_len, 4 * h.n_lower) ||
        !inside(h.off_other_start, 8 * h.n_other) || !inside(h.off_other_len, 4 * h.n_other) || !inside(h.off_other_byte, h.n_other))
        return bad("genome image is damaged (a section lies outside the file)");
    img->lens = (const uint64_t*)img->at(h.off_lens);
    const uint32_t* name_off = (const uint32_t*)img->at(h.off_name_off);
    uint64_t blocks = 0;
    for (uint32_t i = 0; i < h.n_contigs; i++) {
        if (name_off[i] > name_off[i + 1] || name_off[i + 1] > h.names_bytes) return bad("genome image is damaged (names)");
        if (img->lens[i] >= 0x80000000ull - (1ull << 17)) return bad("genome image is damaged (contig length)");
        img->name_store.emplace_back((const char*)img->at(h.off_names) + name_off[i], name_off[i + 1] - name_off[i]);
        blocks += (img->lens[i] + 63) >> 6;
    }
    if (blocks != h.n_blocks) return bad("genome image is damaged (blocks do not add up to the contigs)");
    for (auto& s : img->name_store) img->names.push_back(s.c_str());
    (void)madvise(img->map, img->size, MADV_WILLNEED);
    *out = img;
    return BF_OK;
}

void bf_genome_image_close(bf_genome_image* img) {
    if (!img) return;
    if (img->map != MAP_FAILED) munmap(img->map, img->size);
    delete img;
}

int bf_genome_image_info(const bf_genome_image* img, uint32_t* n_contigs, const char* const** names, const uint64_t** lens,
                         uint64_t* upload_bytes) {
    if (!img) return BF_ERR_INVALID;
    if (n_contigs) *n_contigs = img->h.n_contigs;
    if (names) *names = img->names.data();
    if (lens) *lens = img->lens;
    if (upload_bytes) *upload_bytes = 16 * img->h.n_blocks + 12 * img->h.n_lower + 13 * img->h.n_other;
    return BF_OK;
}

int bf_genome_write_image(bf_ctx* ctx, const bf_genome_image* img) {
    if (!ctx) return BF_ERR_INVALID;
    if (!img) return fail(ctx, BF_ERR_INVALID, "bf_genome_write_image: null image");
    const ImageHeader& h = img->h;
    bool begun = ctx->n_contigs == h.n_contigs && !ctx->genome_ready && ctx->genome_image;
    for (uint32_t i = 0; begun && i < h.n_contigs; i++)
        begun = ctx->contig_names[i] == img->name_store[i] && ctx->contig_host[i].len == img->lens[i];
    if (!begun) return fail(ctx, BF_ERR_STATE, "bf_genome_write_image: not after a bf_genome_begin_image with this image's contigs");
    // (no kernel reads these before bf_genome_end; bf_submit touches none of them)
    BF_TRY(ensure(ctx, ctx->g_lower_start, 8 * h.n_lower + 16)); BF_TRY(ensure(ctx, ctx->g_lower_len, 4 * h.n_lower + 16));
    BF_TRY(ensure(ctx, ctx->g_run_start, 8 * h.n_other + 16)); BF_TRY(ensure(ctx, ctx->g_run_len, 4 * h.n_other + 16));
    BF_TRY(ensure(ctx, ctx->g_run_byte, h.n_other + 16));
    struct Piece { uint8_t* dst; const uint8_t* src; uint64_t n; };
    std::vector<Piece> pieces;
    constexpr uint64_t kPiece = 8u << 20;
    auto add = [&](void* dst, uint64_t off, uint64_t bytes) {
        for (uint64_t a = 0; a < bytes; a += kPiece) pieces.push_back(Piece{(uint8_t*)dst + a, img->at(off) + a, std::min(kPiece, bytes - a)});
    };
    add(ctx->g_raw.p, h.off_blocks, 16 * h.n_blocks);
    add(ctx->g_lower_start.p, h.off_lower_start, 8 * h.n_lower); add(ctx->g_lower_len.p, h.off_lower_len, 4 * h.n_lower);
    add(ctx->g_run_start.p, h.off_other_start, 8 * h.n_other); add(ctx->g_run_len.p, h.off_other_len, 4 * h.n_other);
    add(ctx->g_run_byte.p, h.off_other_byte, h.n_other);
    const size_t nt = std::max<size_t>(1, std::min<size_t>(ctx->host_threads, pieces.size()));
    std::atomic<size_t> next{0};
    std::atomic<int> rc{BF_OK};
    auto work = [&] {
        for (size_t k; (k = next.fetch_add(1)) < pieces.size();) {
            const int r = genome_stage(ctx, pieces[k].dst, pieces[k].src, pieces[k].n, nullptr, nullptr);
            if (r != BF_OK) { rc = r; return; }
        }
    };
    if (nt <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    ctx->n_lower_runs = h.n_lower;
    ctx->n_runs = (uint32_t)h.n_other;
    return rc.load();               // (the message is kept for bf_genome_end, like bf_genome_write's)
}

int bf_genome_set_image(bf_ctx* ctx, const bf_genome_image* img) {
    if (!ctx) return BF_ERR_INVALID;
    if (!img) return fail(ctx, BF_ERR_INVALID, "bf_genome_set_image: null image");
    BF_TRY(genome_begin(ctx, img->h.n_contigs, img->names.data(), img->lens, true));
    const int rc = bf_genome_write_image(ctx, img);
    const int rc_end = bf_genome_end(ctx);
    return rc != BF_OK ? rc : rc_end;
}

}  // extern "C"
