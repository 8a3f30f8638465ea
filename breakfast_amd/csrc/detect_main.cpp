// breakfast-hip: the `breakfast detect` command line (src/detect.rs:28-46, src/main.rs:20-31) on top
// of libbreakfast_hip.so.  This host driver plays the role detect.rs's main thread plays in the
// reference: load <genome>.fa, consume bowtie's stdout (pairing 5p/3p lines exactly as
// detect.rs:76-103), hand parsed batches to the C ABI, print the .sv rows.  It contains no per-read
// arithmetic: every line of detect.rs:105-231 and 247-287 runs on the GPU.
//
//   breakfast-hip detect [options] <bam_file> <genome>
//     --anchor-len=N --anchor-mm=N --max-frag-len=N --min-evidence=N     (as the reference)
//     --anchor-hits=FILE   read bowtie's output (the 4-column `--suppress 5,6,7,8` lines) from FILE
//                          ("-" = stdin) instead of spawning bowtie on <bam_file>.
//     --gpus=N             shard the reads over GPUs 0..N-1 (contiguous input ranges, one thread + one ctx per GPU,
//                          evidence exchanged over RCCL inside the library: bf_shard_*)
//     --devices=a,b,...    the same with an explicit device list; a device named twice makes the ranks meet through
//                          the library's host-staged transport instead (RCCL wants one rank per GPU): test boxes
//     --threads=N          host worker threads: BGZF inflate, FASTA load, anchor-hit parsing [default: 8]
//     --timing             print the wall time of every host/GPU stage on stderr (TIMING lines)
//   breakfast-hip hits [--threads=N] [--block-bytes=N] <hits_file> <genome>
//     parses bowtie's output like detect does and prints one line per paired read; needs no GPU.
//   breakfast-hip anchors [--anchor-len=N] <bam_file>
//     prints the anchor FASTA that detect feeds to bowtie (detect.rs:290-338); needs no GPU.
//
// Without --anchor-hits the driver does what detect.rs:58-70 does: spawn
// `bowtie -f -p1 -v0 -m1 -B1 --suppress 5,6,7,8 <genome> -`, feed it from a producer thread that
// scans the BAM, and consume its stdout.  The stream is parsed by a pool of threads (hits_parser.cpp)
// while the main thread hands finished batches, in input order, to the GPU library.
//
// Errors follow src/common.rs:8-13: "ERROR: <msg>" on stderr, exit status -1.

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <spawn.h>
#include <sys/mman.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/breakfast_hip.h"
#include "bam_reader.h"
#include "hits_parser.h"

extern char** environ;

int filter_main(int argc, char** argv);      // text_tools.cpp: src/filter.rs
int annotate_main(int argc, char** argv);    // text_tools.cpp: src/annotate.rs

[[noreturn]] static void error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fputs("ERROR: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    // other threads may be in the middle of GPU calls or writing into pinned memory: no exit handlers (the HIP
    // runtime's would unmap that memory under them), the process just ends (common.rs:8-13: exit(-1))
    fflush(stdout); fflush(stderr);
    _exit(255);
}

static const char* USAGE =
    "\nUsage:\n  breakfast detect [options] <bam_file> <genome>\n\nOptions:\n"
    "  --anchor-len=N       Anchor length for split read analysis [default: 30]\n"
    "  --anchor-mm=N        Mismatches allowed in anchor alignments [default: 0]\n"
    "  --max-frag-len=N     Maximum fragment length [default: 5000]\n"
    "  --min-evidence=N     Minimum number of supporting DNA fragments [default: 2]\n"
    "  --anchor-hits=FILE   Read Bowtie's anchor alignments from FILE instead of running Bowtie\n"
    "  --gpus=N             Shard the reads over N GPUs [default: 1]\n"
    "  --devices=LIST       Comma-separated GPU ordinals, one rank each (instead of --gpus)\n"
    "  --threads=N          Host worker threads [default: 8]\n"
    "  --timing             Print stage wall times on stderr\n";

struct Timing {
    bool on = false;
    std::vector<std::pair<std::string, double>> rows;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void lap(const char* name) {
        const auto now = std::chrono::steady_clock::now();
        rows.emplace_back(name, std::chrono::duration<double>(now - last).count());
        last = now;
    }
    void add(const char* name, double s) { rows.emplace_back(name, s); }
    void print() const {
        if (!on) return;
        for (auto& r : rows) fprintf(stderr, "TIMING\t%s\t%.6f\n", r.first.c_str(), r.second);
        fprintf(stderr, "TIMING\ttotal\t%.6f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};

static void parallel_for(size_t n, int threads, const std::function<void(size_t)>& fn) {
    const int t = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n));
    if (t == 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int k = 0; k < t; k++)
        pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) fn(i); });
    for (auto& th : pool) th.join();
}

struct Fasta {
    std::vector<std::string> names;
    std::vector<std::unique_ptr<char[]>> seqs;      // uninitialised storage: no 3 GB zero-fill
    std::vector<uint64_t> lens;
};

static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// bio::io::fasta::Reader semantics (detect.rs:48-56): id = header up to the first whitespace, sequence
// lines right-trimmed and concatenated, case preserved; a repeated id replaces the earlier record
// (HashMap::insert).  The file is mapped and cut into pieces at line boundaries; the pieces are
// measured by `threads` workers (FastaScan), then either copied into one buffer per contig (load) or handed
// out piece by piece (stream: the GPU upload runs while the file is still being read).
struct FastaScan {
    struct Rec { size_t head, body, end; };
    struct Piece { size_t rec, a, b, out; uint64_t len; };
    const char* data = nullptr; size_t n = 0;
    void* map = MAP_FAILED; size_t map_size = 0;
    std::string fallback;
    std::vector<Rec> recs;
    std::vector<Piece> pieces;
    std::vector<std::string> names;            // by slot (first appearance of an id)
    std::vector<uint64_t> lens;
    std::vector<long> rec_slot;                // record -> slot, -1 when a later record of the same id supersedes it
    ~FastaScan() { if (map != MAP_FAILED) munmap(map, map_size); }
    template <class F> void each_line(const Piece& pc, F&& fn) const {
        const char* s = data + pc.a;
        const char* end = data + pc.b;
        while (s < end) {
            const char* nl = (const char*)memchr(s, '\n', (size_t)(end - s));
            const char* e = nl ? nl : end;
            while (e > s && is_space(e[-1])) e--;
            fn(s, (size_t)(e - s));
            s = nl ? nl + 1 : end;
        }
    }
};

static std::unique_ptr<FastaScan> scan_fasta(const std::string& path, int threads) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) error("Genome FASTA file %s could not be read.", path.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0) error("Genome FASTA file %s could not be read.", path.c_str());
    auto fs = std::make_unique<FastaScan>();
    const size_t size = (size_t)st.st_size;
    if (size == 0) { close(fd); return fs; }
    fs->map = S_ISREG(st.st_mode) ? mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
    fs->n = size;
    if (fs->map != MAP_FAILED) {
        fs->map_size = size;
        fs->data = (const char*)fs->map;
        (void)madvise(fs->map, size, MADV_WILLNEED);
    } else {                                            // not mappable (a pipe): read it whole
        char buf[1 << 16];
        ssize_t r;
        while ((r = read(fd, buf, sizeof buf)) > 0) fs->fallback.append(buf, (size_t)r);
        fs->data = fs->fallback.data(); fs->n = fs->fallback.size();
    }
    close(fd);
    const char* data = fs->data;
    const size_t n = fs->n;
    // record starts: '>' at the beginning of a line; the file is searched in slices by several threads
    {
        const size_t slices = std::max<size_t>(1, std::min<size_t>((size_t)threads * 4, n >> 24));
        std::vector<std::vector<size_t>> found(slices);
        parallel_for(slices, threads, [&](size_t k) {
            const char* p = data + n * k / slices;
            const char* end = data + n * (k + 1) / slices;
            while (p < end) {
                const char* q = (const char*)memchr(p, '>', (size_t)(end - p));
                if (!q) break;
                if (q == data || q[-1] == '\n') found[k].push_back((size_t)(q - data));
                p = q + 1;
            }
        });
        for (auto& f : found)
            for (size_t h : f) {
                if (!fs->recs.empty()) fs->recs.back().end = h;
                const char* eol = (const char*)memchr(data + h, '\n', n - h);
                fs->recs.push_back(FastaScan::Rec{h, eol ? (size_t)(eol - data) + 1 : n, n});
            }
    }
    auto& recs = fs->recs;
    // pieces of record bodies, cut at line boundaries
    const size_t kPiece = 8u << 20;
    for (size_t r = 0; r < recs.size(); r++) {
        size_t a = recs[r].body;
        while (a < recs[r].end) {
            size_t b = std::min(recs[r].end, a + kPiece);
            if (b < recs[r].end) {
                const char* nl = (const char*)memchr(data + b, '\n', recs[r].end - b);
                b = nl ? (size_t)(nl - data) + 1 : recs[r].end;
            }
            fs->pieces.push_back(FastaScan::Piece{r, a, b, 0, 0});
            a = b;
        }
    }
    parallel_for(fs->pieces.size(), threads, [&](size_t i) {
        uint64_t len = 0;
        fs->each_line(fs->pieces[i], [&](const char*, size_t l) { len += l; });
        fs->pieces[i].len = len;
    });
    std::vector<uint64_t> rec_len(recs.size(), 0);
    for (auto& pc : fs->pieces) { pc.out = rec_len[pc.rec]; rec_len[pc.rec] += pc.len; }
    // ids; a repeated id keeps its first slot and takes the later record's sequence
    std::unordered_map<std::string, size_t> index;
    std::vector<size_t> slot_rec;                        // slot -> record that supplies the sequence
    for (size_t r = 0; r < recs.size(); r++) {
        const char* h = data + recs[r].head + 1;
        const char* hend = data + (recs[r].body > recs[r].head ? recs[r].body : n);
        const char* e = h;
        while (e < hend && !is_space(*e)) e++;
        std::string id(h, (size_t)(e - h));
        auto it = index.find(id);
        if (it == index.end()) { index[id] = fs->names.size(); fs->names.push_back(id); slot_rec.push_back(r); }
        else slot_rec[it->second] = r;
    }
    fs->rec_slot.assign(recs.size(), -1);
    for (size_t sl = 0; sl < slot_rec.size(); sl++) fs->rec_slot[slot_rec[sl]] = (long)sl;
    fs->lens.resize(fs->names.size());
    for (size_t sl = 0; sl < slot_rec.size(); sl++) fs->lens[sl] = rec_len[slot_rec[sl]];
    return fs;
}

static Fasta read_fasta(const std::string& path, int threads) {
    auto fs = scan_fasta(path, threads);
    Fasta fa;
    fa.names = fs->names; fa.lens = fs->lens;
    fa.seqs.resize(fa.names.size());
    for (size_t sl = 0; sl < fa.names.size(); sl++) fa.seqs[sl].reset(new char[std::max<uint64_t>(fa.lens[sl], 1)]);
    parallel_for(fs->pieces.size(), threads, [&](size_t i) {
        const long sl = fs->rec_slot[fs->pieces[i].rec];
        if (sl < 0) return;                              // superseded by a later record of the same id
        char* out = fa.seqs[(size_t)sl].get() + fs->pieces[i].out;
        fs->each_line(fs->pieces[i], [&](const char* s, size_t l) { memcpy(out, s, l); out += l; });
    });
    return fa;
}

// Every piece of every contig to the library, from `threads` workers.  A piece's lines are joined straight into the
// pinned memory it is uploaded from (bf_genome_fill); a piece too long for that (a FASTA without line breaks) goes
// through a buffer the worker keeps, 2 MB at a time (bf_genome_write).
static void stream_fasta(const FastaScan& fs, int threads, bf_ctx* ctx) {
    constexpr size_t kChunk = 2u << 20;
    struct Fill { const FastaScan* fs; const FastaScan::Piece* piece; };
    parallel_for(fs.pieces.size(), threads, [&](size_t i) {
        const long sl = fs.rec_slot[fs.pieces[i].rec];
        if (sl < 0 || fs.pieces[i].len == 0) return;
        if (fs.pieces[i].len <= BF_GENOME_FILL_MAX) {
            Fill f{&fs, &fs.pieces[i]};
            (void)bf_genome_fill(ctx, (uint32_t)sl, fs.pieces[i].out, fs.pieces[i].len, [](void* u, uint8_t* dst, uint64_t n) {
                const Fill* f = (const Fill*)u;
                uint8_t* out = dst;
                f->fs->each_line(*f->piece, [&](const char* s, size_t l) {
                    if ((uint64_t)(out - dst) + l <= n) memcpy(out, s, l);
                    out += l;
                });
            }, &f);
            return;
        }
        thread_local std::unique_ptr<char[]> buf;
        thread_local size_t cap = 0;
        size_t fill = 0;
        uint64_t at = fs.pieces[i].out;
        auto flush = [&] { if (fill) (void)bf_genome_write(ctx, (uint32_t)sl, at, (const uint8_t*)buf.get(), fill); at += fill; fill = 0; };
        fs.each_line(fs.pieces[i], [&](const char* s, size_t l) {
            if (fill + l > cap) {
                flush();
                if (l > cap) { cap = std::max(l, kChunk + (64u << 10)); buf.reset(new char[cap]); }
            }
            memcpy(buf.get() + fill, s, l);
            fill += l;
            if (fill >= kChunk) flush();
        });
        flush();
    });
}

struct Child { pid_t pid = -1; int to_child = -1, from_child = -1; };

// Command::new("bowtie").args(...).stdin(piped).stdout(piped).spawn(), detect.rs:60-63
static Child spawn_bowtie(const std::string& genome_path) {
    int in_pipe[2], out_pipe[2];
    if (pipe2(in_pipe, O_CLOEXEC) != 0 || pipe2(out_pipe, O_CLOEXEC) != 0) error("Could not start Bowtie process.");
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, in_pipe[0], 0);
    posix_spawn_file_actions_adddup2(&fa, out_pipe[1], 1);
    std::string g = genome_path;
    const char* argv[] = {"bowtie", "-f", "-p1", "-v0", "-m1", "-B1", "--suppress", "5,6,7,8", g.c_str(), "-", nullptr};
    Child c;
    const int rc = posix_spawnp(&c.pid, "bowtie", &fa, nullptr, (char* const*)argv, environ);
    posix_spawn_file_actions_destroy(&fa);
    close(in_pipe[0]); close(out_pipe[1]);
    if (rc != 0) { close(in_pipe[1]); close(out_pipe[0]); error("Could not start Bowtie process."); }
    c.to_child = in_pipe[1]; c.from_child = out_pipe[0];
    return c;
}

static std::unique_ptr<bam::Reader> open_bam(const std::string& path, int threads) {
    try {
        return std::make_unique<bam::Reader>(path, threads);
    } catch (const std::exception& e) {
        if (path == "-") error("Failed to read BAM file from standard input.");          // detect.rs:293-294
        error("Could not open BAM file '%s'", path.c_str());                              // detect.rs:296-297
    }
}

static bool parse_u64(std::string_view s, uint64_t& v) {
    if (s.empty()) return false;
    v = 0;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        v = v * 10 + (uint64_t)(c - '0');
    }
    return true;
}

static void write_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t w = write(fd, p, std::min<size_t>(n, 1u << 30));
        if (w < 0) { if (errno == EINTR) continue; error("Could not write the output: %s", strerror(errno)); }
        p += w; n -= (size_t)w;
    }
}

// bf_host_alloc memory for the parser's batches, handed out in power-of-two blocks from a few large slabs (one call
// into the driver per slab, not per array).  A released block is handed out again at once: the parser only lets go of
// arrays it is filling or of batches whose uploads have been waited for.
struct PinnedHome {
    static constexpr size_t kHead = 64, kMinBlock = 64u << 10, kSlab = 256u << 20;
    bf_ctx* ctx = nullptr;
    std::mutex mu;
    std::vector<std::vector<void*>> spare;       // by class c: blocks of kMinBlock << c bytes (head included)
    uint8_t* slab = nullptr;
    size_t slab_left = 0;
    void* alloc(size_t bytes) {
        size_t c = 0;
        while ((kMinBlock << c) - kHead < bytes) c++;
        const size_t block = kMinBlock << c;
        std::lock_guard<std::mutex> g(mu);
        if (spare.size() <= c) spare.resize(c + 1);
        uint8_t* p = nullptr;
        if (!spare[c].empty()) { p = (uint8_t*)spare[c].back(); spare[c].pop_back(); }
        else {
            if (slab_left < block) {                  // (what is left of the old slab stays unused)
                void* q = nullptr;
                const size_t want = std::max(block, kSlab);
                if (bf_host_alloc(ctx, want, &q) != BF_OK) return nullptr;
                slab = (uint8_t*)q; slab_left = want;
            }
            p = slab; slab += block; slab_left -= block;
            *(uint64_t*)p = c;
        }
        return p + kHead;
    }
    void release(void* q) {
        uint8_t* p = (uint8_t*)q - kHead;
        std::lock_guard<std::mutex> g(mu);
        spare[(size_t)*(uint64_t*)p].push_back(p);
    }
};

// One parsed batch (or reads [a, b) of it) to the library, in the form the parser produced it
static int submit_parsed(bf_ctx* ctx, const hits::Batch& src, uint64_t a, uint64_t b) {
    if (src.packed) {
        bf_packed_batch pb;
        pb.n_reads = b - a;
        pb.seq_nibbles = (const uint8_t*)src.seq.data(); pb.seq_off = src.seq_off.data() + a; pb.seq_len = src.seq_len.data() + a;
        pb.frag_bytes = (const uint8_t*)src.frag.data(); pb.frag_off = src.frag_off.data() + a;
        pb.hits = src.hits.data() + a;
        return bf_submit_packed(ctx, &pb);
    }
    bf_batch bb;
    bb.n_reads = b - a;
    bb.seq_bytes = (const uint8_t*)src.seq.data(); bb.seq_off = src.seq_off.data() + a;
    bb.frag_bytes = (const uint8_t*)src.frag.data(); bb.frag_off = src.frag_off.data() + a;
    bb.hits = src.hits.data() + a;
    return bf_submit(ctx, &bb);
}

// Reads go to the GPU as BAM 4-bit codes when they are known to be in BAM's alphabet (they come out of this program's
// own BAM reader) or when the hit list is a file that can be read again as ASCII should a read turn out otherwise.
static bool can_reread(int fd) {
    struct stat st;
    return fd > 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
}

// The packed image of <genome>.fa, if `breakfast-hip index` has made one and the FASTA has not changed since (size and
// modification time are kept in the image): 0.27 of the bytes go to the GPU and the FASTA is not read at all.
static std::string genome_image_path(const std::string& genome_path) { return genome_path + ".bf2bit"; }
static int64_t mtime_ns(const struct stat& st) { return (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec; }
static bf_genome_image* open_genome_image(const std::string& genome_path) {
    if (getenv("BREAKFAST_HIP_NO_IMAGE")) return nullptr;                       // (A/B switch)
    struct stat st;
    if (stat((genome_path + ".fa").c_str(), &st) != 0) return nullptr;          // the FASTA path reports it
    bf_genome_image* img = nullptr;
    if (bf_genome_image_open(genome_image_path(genome_path).c_str(), (uint64_t)st.st_size, mtime_ns(st), &img) != BF_OK) {
        if (access(genome_image_path(genome_path).c_str(), F_OK) == 0)
            fprintf(stderr, "WARNING: %s (reading the FASTA instead; `breakfast-hip index` makes a new image)\n", bf_last_error(nullptr));
        return nullptr;
    }
    return img;
}

// `breakfast-hip detect --gpus=N`: the reads are parsed into host memory, cut into N contiguous input ranges (the
// reference's tie-breaks are input order, detect.rs:193-198, so rank r must hold the reads before rank r+1's), and
// every range is driven by its own thread and ctx through the library's sharded path.  Rows are merged for printing
// by bf_shard_format_sv.
static int detect_sharded(const bf_config& cfg0, const std::vector<int>& devices, const std::string& genome_path, int in_fd,
                          const hits::Options& popt, int threads, Timing& timing) {
    const uint32_t W = (uint32_t)devices.size();
    bool shared_gpu = false;
    for (uint32_t a = 0; a < W; a++) for (uint32_t b = a + 1; b < W; b++) shared_gpu = shared_gpu || devices[a] == devices[b];
    uint8_t uid[128] = {0};
    if (!shared_gpu && bf_shard_unique_id(uid) != BF_OK) error("%s", bf_last_error(nullptr));
    char shm_name[64];
    snprintf(shm_name, sizeof shm_name, "/breakfast_hip_%d", (int)getpid());

    bf_genome_image* image = open_genome_image(genome_path);
    Fasta fa;
    if (!image) fa = read_fasta(genome_path + ".fa", threads);
    fprintf(stderr, "Reading reference genome into memory...\n");
    timing.lap(image ? "image_open" : "fasta_load");
    hits::ContigIndex contigs;
    std::vector<const char*> names; std::vector<const uint8_t*> seqs;
    if (image) {
        uint32_t n = 0; const char* const* nm = nullptr;
        bf_genome_image_info(image, &n, &nm, nullptr, nullptr);
        for (uint32_t i = 0; i < n; i++) { names.push_back(nm[i]); contigs.add(nm[i], i); }
    }
    for (size_t i = 0; i < fa.names.size(); i++) {
        names.push_back(fa.names[i].c_str());
        seqs.push_back((const uint8_t*)fa.seqs[i].get());
        contigs.add(fa.names[i], (uint32_t)i);
    }
    if (names.empty()) error("Genome FASTA file %s.fa holds no sequences.", genome_path.c_str());
    contigs.build();

    fprintf(stderr, "Splitting unaligned reads into %u bp anchors and aligning against the genome...\n", cfg0.anchor_len);
    std::vector<hits::Batch> all;
    uint64_t n_total = 0;
    hits::Options opt = popt;
    for (;;) {
        try {
            hits::parse_stream(in_fd, contigs, genome_path, opt, [&](hits::Batch& b) {
                n_total += b.n();
                all.emplace_back();
                std::swap(all.back(), b);
            });
            break;
        } catch (const hits::Unpackable& e) {                      // once more, as ASCII
            if (!can_reread(in_fd) || lseek(in_fd, 0, SEEK_SET) != 0) error("%s", e.what());
            opt.pack = false; all.clear(); n_total = 0;
        } catch (const std::exception& e) { error("%s", e.what()); }
    }
    if (in_fd > 0) close(in_fd);
    timing.lap("parse");

    std::vector<uint64_t> first(all.size() + 1, 0);               // global index of every batch's first read
    for (size_t k = 0; k < all.size(); k++) first[k + 1] = first[k] + all[k].n();
    std::vector<bf_ctx*> ctxs(W, nullptr);
    std::vector<bf_shard_result> results(W);
    std::vector<std::string> errors(W);
    // The ranks are threads of this process.  A rank that fails on its own (no such device, no memory for the genome, a
    // rejected batch) must not leave the others waiting inside a collective: RCCL has no time-out.  So every rank first
    // does everything that can fail locally, the ranks then agree -- on the host -- that all of them got there, and only
    // then does anyone enter bf_shard_init_* / bf_shard_finish; the same again between the two.
    struct Agree {
        std::mutex mu; std::condition_variable cv;
        uint32_t n, arrived = 0; uint64_t gen = 0; bool ok = true, last = true;
        bool all_ok(bool mine) {
            std::unique_lock<std::mutex> lk(mu);
            ok = ok && mine;
            if (++arrived == n) { last = ok; ok = true; arrived = 0; gen++; cv.notify_all(); return last; }
            const uint64_t g = gen;
            cv.wait(lk, [&] { return gen != g; });
            return last;
        }
    } agree;
    agree.n = W;
    std::vector<std::thread> pool;
    for (uint32_t r = 0; r < W; r++) {
        pool.emplace_back([&, r] {
            bf_config cfg = cfg0;
            cfg.device = devices[r];
            bf_ctx* ctx = nullptr;
            auto check = [&](int rc) { if (rc != BF_OK && errors[r].empty()) errors[r] = bf_last_error(ctx); return rc == BF_OK; };
            const uint64_t lo = n_total * r / W, hi = n_total * (r + 1) / W;
            // 1. everything local: ctx, genome, this rank's reads (bf_last_error(NULL) is per thread)
            bool ok = check(bf_ctx_create(&cfg, &ctxs[r]));
            ctx = ctxs[r];
            ok = ok && check(image ? bf_genome_set_image(ctx, image) : bf_genome_set(ctx, (uint32_t)names.size(), names.data(), seqs.data(), fa.lens.data()));
            for (size_t k = 0; ok && k < all.size(); k++) {          // the parts of the parsed batches inside [lo, hi)
                const uint64_t a = std::max(lo, first[k]), b = std::min(hi, first[k + 1]);
                if (a < b) ok = check(submit_parsed(ctx, all[k], a - first[k], b - first[k]));
            }
            if (!agree.all_ok(ok)) return;
            // 2. join the other ranks
            ok = check(shared_gpu ? bf_shard_init_host(ctx, r, W, lo, shm_name, 64u << 20) : bf_shard_init_rccl(ctx, r, W, lo, uid));
            if (!agree.all_ok(ok)) return;
            // 3. the sharded path
            check(bf_shard_finish(ctx, &results[r]));
        });
    }
    for (auto& th : pool) th.join();
    for (uint32_t r = 0; r < W; r++) if (!errors[r].empty()) error("rank %u: %s", r, errors[r].c_str());
    timing.lap("gpu_sharded_detect");
    fprintf(stderr, "Found %llu rearrangement supporting reads.\n", (unsigned long long)n_total);
    fprintf(stderr, "Sorting rearrangement supporting reads by position...\n");
    fprintf(stderr, "Identifying rearrangements based on clusters of discordant reads...\n");
    const int64_t size = bf_shard_format_sv(ctxs[0], W, results.data(), nullptr, 0);
    if (size < 0) error("could not render the result");
    std::unique_ptr<char[]> out(new char[(size_t)size + 1]);
    if (bf_shard_format_sv(ctxs[0], W, results.data(), out.get(), (uint64_t)size) != size) error("could not render the result");
    timing.lap("format_sv");
    fflush(stdout);
    write_all(1, out.get(), (size_t)size);
    timing.lap("write_stdout");
    for (bf_ctx* c : ctxs) bf_ctx_destroy(c);
    bf_genome_image_close(image);
    return 0;
}

// write end of the pipe to the waiting parent (detect only; -1: there is none)
static int g_done_fd = -1;
// everything the caller waits for is done: tell the parent, let go of the standard streams (a caller that reads our
// stdout or stderr through a pipe waits for their last holder), and leave the cleaning up to what follows
static void work_done(int code) {
    fflush(stdout); fflush(stderr);
    if (g_done_fd < 0) return;
    const unsigned char c = (unsigned char)code;
    if (write(g_done_fd, &c, 1) != 1) {}
    close(g_done_fd); g_done_fd = -1;
    close(0); close(1); close(2);
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "filter") == 0) return filter_main(argc, argv);        // main.rs:26
    if (argc >= 2 && strcmp(argv[1], "annotate") == 0) return annotate_main(argc, argv);    // main.rs:27
    const bool anchors_only = argc >= 2 && strcmp(argv[1], "anchors") == 0;
    const bool hits_only = argc >= 2 && strcmp(argv[1], "hits") == 0;
    const bool index_only = argc >= 2 && strcmp(argv[1], "index") == 0;
    if (argc < 2 || (strcmp(argv[1], "detect") != 0 && !anchors_only && !hits_only && !index_only)) {
        fputs("\nBreakfast is a software for detecting chromosomal rearrangements in DNA/RNA sequencing data.\n\n"
              "Usage:\n  breakfast-hip detect [options] <bam_file> <genome>\n  breakfast-hip filter [options] <sample_sheet>\n"
              "  breakfast-hip annotate <sv_path> <bed_path>\n  breakfast-hip anchors [--anchor-len=N] <bam_file>\n"
              "  breakfast-hip hits [--threads=N] [--block-bytes=N] <hits_file> <genome>\n"
              "  breakfast-hip index [--threads=N] <genome>      (writes <genome>.bf2bit beside <genome>.fa; detect uses it)\n", stderr);
        return -1;
    }
    // `detect` does its work in a child process, started before anything touches the GPU.  When the output is written
    // the child says so and the parent returns; releasing 10 GB of device memory, the pinned buffers and the runtime
    // (0.18 s for configs[1], most of it inside the driver after exit) then happens behind the caller's back instead of
    // in front of it.  A child that ends without saying so hands its exit status on.  BREAKFAST_HIP_NO_DETACH=1: one process.
    if (!anchors_only && !hits_only && !index_only && !getenv("BREAKFAST_HIP_NO_DETACH")) {
        int pfd[2];
        if (pipe2(pfd, O_CLOEXEC) == 0) {                   // (not for Bowtie to inherit)
            fflush(stdout); fflush(stderr);
            const pid_t parent = getpid();
            const pid_t pid = fork();
            if (pid > 0) {
                close(pfd[1]);
                unsigned char code = 0;
                ssize_t r;
                while ((r = read(pfd[0], &code, 1)) < 0 && errno == EINTR) {}
                if (r == 1) _exit(code);
                int st = 0;
                while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
                _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 255);
            }
            if (pid == 0) {
                close(pfd[0]); g_done_fd = pfd[1];
                // the worker never outlives the process the caller started by more than its own cleaning up: a caller
                // that kills that process (a time limit, ^C) takes the worker with it
                prctl(PR_SET_PDEATHSIG, SIGKILL);
                if (getppid() != parent) _exit(255);
            }
            else { close(pfd[0]); close(pfd[1]); }          // fork failed: everything in this process
        }
    }
    bf_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.anchor_len = 30; cfg.max_frag_len = 5000; cfg.min_evidence = 2; cfg.device = -1;
    std::string hits_path;
    int threads = 8;
    std::vector<int> devices;                     // empty: one GPU, the current device
    hits::Options popt;
    int reads_form = -1;                          // -1: chosen from where the reads come from; 0 ASCII; 1 BAM 4-bit codes
    Timing timing;
    std::vector<std::string> pos;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char* opt, uint64_t& out) {
            const size_t n = strlen(opt);
            if (a.compare(0, n, opt) != 0) return false;
            if (!parse_u64(std::string_view(a).substr(n), out)) error("Invalid arguments.\n%s", USAGE);
            return true;
        };
        uint64_t v;
        if (val("--anchor-len=", v)) cfg.anchor_len = (uint32_t)std::min<uint64_t>(v, 0xffffffffu);
        else if (val("--anchor-mm=", v)) {}                       // parsed and ignored, detect.rs:44
        else if (val("--max-frag-len=", v)) cfg.max_frag_len = (uint32_t)std::min<uint64_t>(v, 0xffffffffu);
        else if (val("--min-evidence=", v)) cfg.min_evidence = (uint32_t)std::min<uint64_t>(v, 0xffffffffu);
        else if (val("--threads=", v)) threads = (int)std::min<uint64_t>(std::max<uint64_t>(v, 1), 64);
        else if (val("--block-bytes=", v)) popt.block_bytes = (size_t)std::max<uint64_t>(v, 1);
        else if (val("--gpus=", v)) { devices.clear(); for (uint64_t d = 0; d < std::min<uint64_t>(std::max<uint64_t>(v, 1), 64); d++) devices.push_back((int)d); }
        else if (a.compare(0, 10, "--devices=") == 0) {
            devices.clear();
            std::string_view list = std::string_view(a).substr(10);
            while (!list.empty()) {
                const size_t c = list.find(',');
                uint64_t d;
                if (!parse_u64(list.substr(0, c), d) || d > 1023) error("Invalid arguments.\n%s", USAGE);
                devices.push_back((int)d);
                list = c == std::string_view::npos ? std::string_view() : list.substr(c + 1);
            }
            if (devices.empty() || devices.size() > 64) error("Invalid arguments.\n%s", USAGE);
        }
        else if (a == "--timing") timing.on = true;
        else if (a == "--ascii-reads") reads_form = 0;
        else if (a == "--packed-reads") reads_form = 1;
        else if (a.compare(0, 14, "--anchor-hits=") == 0) hits_path = a.substr(14);
        else if (a.size() > 1 && a[0] == '-' && a != "-") error("Invalid arguments.\n%s", USAGE);
        else pos.push_back(a);
    }
    popt.threads = threads;
    if (anchors_only) {
        if (pos.size() != 1) error("Invalid arguments.\nUsage:\n  breakfast-hip anchors [--anchor-len=N] <bam_file>");
        auto bam = open_bam(pos[0], threads);
        try {
            bam::dispatch_reads(*bam, stdout, cfg.anchor_len);
        } catch (const std::exception& e) { error("Failed reading BAM record: %s", e.what()); }   // common.rs:64
        fflush(stdout);
        return 0;
    }
    if (index_only) {
        // <genome>.fa -> <genome>.bf2bit, the packed image `detect` hands to the GPU instead of the FASTA (host work only)
        if (pos.size() != 1) error("Invalid arguments.\nUsage:\n  breakfast-hip index [--threads=N] <genome>");
        const std::string fa_path = pos[0] + ".fa";
        struct stat st;
        if (stat(fa_path.c_str(), &st) != 0) error("Genome FASTA file %s could not be read.", fa_path.c_str());
        Fasta fa = read_fasta(fa_path, threads);
        if (fa.names.empty()) error("Genome FASTA file %s holds no sequences.", fa_path.c_str());
        timing.lap("fasta_load");
        std::vector<const char*> names; std::vector<const uint8_t*> seqs;
        for (size_t i = 0; i < fa.names.size(); i++) { names.push_back(fa.names[i].c_str()); seqs.push_back((const uint8_t*)fa.seqs[i].get()); }
        if (bf_genome_image_write(genome_image_path(pos[0]).c_str(), (uint32_t)names.size(), names.data(), seqs.data(), fa.lens.data(),
                                  (uint64_t)st.st_size, mtime_ns(st), (uint32_t)threads) != BF_OK)
            error("%s", bf_last_error(nullptr));
        timing.lap("pack_and_write");
        timing.print();
        return 0;
    }
    if (hits_only) {
        // the consume loop's parsing half alone (detect.rs:76-103): one line per paired read
        if (pos.size() != 2) error("Invalid arguments.\nUsage:\n  breakfast-hip hits [--threads=N] [--block-bytes=N] <hits_file> <genome>");
        Fasta fa = read_fasta(pos[1] + ".fa", threads);
        hits::ContigIndex contigs;
        for (size_t i = 0; i < fa.names.size(); i++) contigs.add(fa.names[i], (uint32_t)i);
        contigs.build();
        const int fd = pos[0] == "-" ? 0 : open(pos[0].c_str(), O_RDONLY);
        if (fd < 0) error("Cannot open file %s for reading.", pos[0].c_str());
        popt.pack = reads_form == 1;
        // tests: BREAKFAST_HIP_TEST_ALLOCATOR=k moves the parser's batches to another allocator after k batches, the way
        // `detect` moves them into pinned memory once the GPU context exists (counts come out on stderr)
        struct Counting { std::atomic<uint64_t> allocs{0}, releases{0}, live_bytes{0}; } counting;
        const hits::Allocator test_mem{
            [](void* u, size_t bytes) -> void* {
                Counting* c = (Counting*)u;
                uint64_t* q = (uint64_t*)malloc(bytes + 16);
                if (!q) return nullptr;
                q[0] = bytes; q[1] = 0x5eedf00d;
                c->allocs++; c->live_bytes += bytes;
                memset(q + 2, 0xa5, bytes);                         // nothing may rely on fresh memory being zero
                return q + 2;
            },
            [](void* u, void* ptr) {
                Counting* c = (Counting*)u;
                uint64_t* q = (uint64_t*)ptr - 2;
                if (q[1] != 0x5eedf00d) { fprintf(stderr, "test allocator: released a pointer it never handed out\n"); abort(); }
                q[1] = 0;
                c->releases++; c->live_bytes -= q[0];
                free(q);
            }, &counting};
        std::atomic<const hits::Allocator*> batch_home{nullptr};
        const char* switch_after = getenv("BREAKFAST_HIP_TEST_ALLOCATOR");
        uint64_t batches = 0;
        if (switch_after) popt.allocator = &batch_home;
        try {
            uint64_t counted = 0;
            const bool count_only = getenv("BREAKFAST_HIP_COUNT_ONLY") != nullptr;     // parser timing: no text
            hits::parse_stream(fd, contigs, pos[1], popt, [&](hits::Batch& b) {
                counted += b.n();
                if (switch_after && batches++ == (uint64_t)atoll(switch_after)) batch_home.store(&test_mem, std::memory_order_release);
                for (size_t i = 0; i < (count_only ? 0 : b.n()); i++) {
                    const bf_hit& h = b.hits[i];
                    fwrite(b.frag.data() + b.frag_off[i], 1, b.frag_off[i + 1] - b.frag_off[i], stdout);
                    fputc('\t', stdout);
                    if (b.packed) {                                 // --packed-reads: the 4-bit codes, spelled out again
                        const uint8_t* q = (const uint8_t*)b.seq.data() + b.seq_off[i];
                        for (uint32_t k = 0; k < b.seq_len[i]; k++) fputc("=ACMGRSVTWYHKDBN"[(q[k >> 1] >> ((~k & 1) * 4)) & 15], stdout);
                    } else
                        fwrite(b.seq.data() + b.seq_off[i], 1, b.seq_off[i + 1] - b.seq_off[i], stdout);
                    fprintf(stdout, "\t%u\t%u\t%u\t%u\t%u\n", h.strands, h.contig5, h.pos5, h.contig3, h.pos3);
                }
            });
            if (count_only) fprintf(stderr, "%llu pairs\n", (unsigned long long)counted);
        } catch (const std::exception& e) { error("%s", e.what()); }
        if (switch_after)
            fprintf(stderr, "test allocator: %llu allocations, %llu releases, %llu bytes live\n", (unsigned long long)counting.allocs.load(),
                    (unsigned long long)counting.releases.load(), (unsigned long long)counting.live_bytes.load());
        fflush(stdout);
        return 0;
    }
    if (pos.size() != 2) error("Invalid arguments.\n%s", USAGE);
    const std::string sam_path = pos[0], genome_path = pos[1];

    // Bowtie and the BAM producer start BEFORE the GPU is touched (the child must be spawned from a
    // process that has not initialised the device); the pipes simply fill until we start consuming.
    int in_fd = -1;
    Child child;
    std::thread producer;
    std::unique_ptr<bam::Reader> bam_reader;
    if (hits_path.empty()) {
        bam_reader = open_bam(sam_path, threads);
        child = spawn_bowtie(genome_path);
        const uint32_t A = cfg.anchor_len;
        const int wfd = child.to_child;
        bam::Reader* rd = bam_reader.get();
        producer = std::thread([rd, wfd, A] {                      // thread::spawn(...), detect.rs:68-70
            FILE* w = fdopen(wfd, "w");
            try {
                bam::dispatch_reads(*rd, w, A);
            } catch (const std::exception& e) { error("Failed reading BAM record: %s", e.what()); }
            fclose(w);
        });
        in_fd = child.from_child;
    } else {
        in_fd = hits_path == "-" ? 0 : open(hits_path.c_str(), O_RDONLY);
        if (in_fd < 0) error("Cannot open file %s for reading.", hits_path.c_str());
    }
    timing.lap("start_up");
    popt.pack = reads_form >= 0 ? reads_form == 1 : (hits_path.empty() || can_reread(in_fd));

    if (devices.size() > 1) {
        const int rc = detect_sharded(cfg, devices, genome_path, in_fd, popt, threads, timing);
        if (producer.joinable()) producer.join();
        if (child.pid > 0) { int st = 0; waitpid(child.pid, &st, 0); }
        timing.print();
        work_done(rc);
        return rc;
    }
    if (devices.size() == 1) cfg.device = devices[0];
    cfg.reserved[1] = (uint32_t)std::max(2, threads / 4);              // staging threads of bf_submit (they check the batches)
    if (const char* e = getenv("BREAKFAST_HIP_STAGE_THREADS")) cfg.reserved[1] = (uint32_t)std::max(1, atoi(e));
    bf_ctx* ctx = nullptr;
    double genome_s = 0, submit_s = 0;
    size_t longest_name = 0;                      // of the contigs (bounds a row's length when the table is rendered)
    // one attempt at "genome + reads into the GPU"; false when the reads have to be taken as ASCII after all
    bool announce = true;
    auto load_inputs = [&]() -> bool {
        // The GPU context comes up on a thread of its own (most of that is the driver) while the FASTA is measured
        // (names, lengths) and the main thread ALREADY parses Bowtie's output: the parser only needs the contig names.
        // The same thread then sends the FASTA's pieces to the GPU from a few workers; bf_submit accepts reads from
        // bf_genome_begin on (their kernels wait for bf_genome_end): the submit thread below waits for that moment.
        std::string ctx_error;
        double ctx_at = 0, ready_at = 0, first_submit_at = 0;      // seconds since the process started (--timing)
        auto since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - timing.t0).count(); };
        std::thread ctx_thread([&] { if (bf_ctx_create(&cfg, &ctx) != BF_OK) ctx_error = bf_last_error(nullptr); ctx_at = since_start(); });
        bf_genome_image* image = open_genome_image(genome_path);
        std::unique_ptr<FastaScan> fs;
        if (!image) fs = scan_fasta(genome_path + ".fa", threads);
        if (announce) fprintf(stderr, "Reading reference genome into memory...\n");
        timing.lap(image ? "image_open" : "fasta_scan");
        hits::ContigIndex contigs;
        std::vector<const char*> names;
        const uint64_t* contig_lens = nullptr;
        if (image) {
            uint32_t n = 0; const char* const* nm = nullptr;
            bf_genome_image_info(image, &n, &nm, &contig_lens, nullptr);
            for (uint32_t i = 0; i < n; i++) { names.push_back(nm[i]); contigs.add(nm[i], i); }
        } else {
            for (size_t i = 0; i < fs->names.size(); i++) { names.push_back(fs->names[i].c_str()); contigs.add(fs->names[i], (uint32_t)i); }
            contig_lens = fs->lens.data();
        }
        if (names.empty()) error("Genome FASTA file %s.fa holds no sequences.", genome_path.c_str());
        for (const char* nm : names) longest_name = std::max(longest_name, strlen(nm));
        contigs.build();
        // half of the threads parse, a quarter joins and uploads the FASTA, the rest is for bf_submit's checks, the submit
        // thread and the reader (measured with 16 threads: 8 + 4 + 4 gives 0.285-0.294 s, 11 + 3 + 3 gives 0.298-0.306 s)
        int g_threads = std::max(1, threads / 4);
        if (const char* e = getenv("BREAKFAST_HIP_GENOME_THREADS")) g_threads = std::max(1, atoi(e));
        if (const char* e = getenv("BREAKFAST_HIP_PARSER_THREADS")) popt.threads = std::max(1, atoi(e));
        genome_s = 0;
        // the parser's batches move into the library's pinned memory once the ctx exists: bf_submit then uploads them
        // from where the parser wrote them instead of copying them into its staging slots first
        PinnedHome home;
        const hits::Allocator pinned_mem{[](void* u, size_t bytes) { return ((PinnedHome*)u)->alloc(bytes); },
                                         [](void* u, void* q) { ((PinnedHome*)u)->release(q); }, &home};
        std::atomic<const hits::Allocator*> batch_home{nullptr};
        popt.allocator = &batch_home;
        std::mutex gpu_mu;
        std::condition_variable gpu_cv;
        bool gpu_ready = false;
        std::thread genome_thread([&] {
            ctx_thread.join();
            if (!ctx) error("%s", ctx_error.c_str());
            if ((image ? bf_genome_begin_image(ctx, (uint32_t)names.size(), names.data(), contig_lens)
                       : bf_genome_begin(ctx, (uint32_t)names.size(), names.data(), contig_lens)) != BF_OK) error("%s", bf_last_error(ctx));
            // a hit list in a file: the resident arrays are sized once from a sample of its first lines (growing them
            // later waits for all queued work and copies what is there)
            struct stat hst;
            std::vector<char> sample(1 << 16);
            ssize_t got = 0;
            if (can_reread(in_fd) && fstat(in_fd, &hst) == 0 && hst.st_size > (off_t)sample.size() && (got = pread(in_fd, sample.data(), sample.size(), 0)) > 0) {
                uint64_t lines = 0, seq = 0, frag = 0;
                for (const char* q = sample.data(); q < sample.data() + got;) {
                    const char* nl = (const char*)memchr(q, '\n', (size_t)(sample.data() + got - q));
                    if (!nl) break;
                    if (nl - q > 3 && q[0] == '3' && q[1] == 'p') {        // 3p:<number>:<fragment id>:<sequence>:
                        const char* c1 = (const char*)memchr(q + 3, ':', (size_t)(nl - q - 3));
                        const char* c2 = c1 ? (const char*)memchr(c1 + 1, ':', (size_t)(nl - c1 - 1)) : nullptr;
                        const char* c3 = c2 ? (const char*)memchr(c2 + 1, ':', (size_t)(nl - c2 - 1)) : nullptr;
                        if (c3) { frag += (uint64_t)(c2 - c1 - 1); seq += (uint64_t)(c3 - c2 - 1); }
                    }
                    lines++; q = nl + 1;
                }
                if (lines >= 2) {
                    const double scale = (double)hst.st_size / (double)got * 1.03;
                    const uint64_t n_est = (uint64_t)((double)(lines / 2 + 1) * scale) + 4096;
                    const uint64_t seq_est = (uint64_t)((double)(popt.pack ? seq / 2 + lines : seq) * scale) + (1u << 20);
                    (void)bf_reserve(ctx, n_est, seq_est, (uint64_t)((double)frag * scale) + (1u << 20));   // an estimate: failing is not fatal
                }
            }
            ready_at = since_start();
            home.ctx = ctx;
            batch_home.store(&pinned_mem, std::memory_order_release);
            { std::lock_guard<std::mutex> g(gpu_mu); gpu_ready = true; }
            gpu_cv.notify_all();
            const auto t0 = std::chrono::steady_clock::now();
            if (image) (void)bf_genome_write_image(ctx, image);          // (bf_genome_end repeats its error)
            else stream_fasta(*fs, g_threads, ctx);
            genome_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });

        if (announce) fprintf(stderr, "Splitting unaligned reads into %u bp anchors and aligning against the genome...\n", cfg.anchor_len);
        announce = false;

        // parsed batches go to a submit thread through a short queue (in input order: one producer, one consumer), so that
        // the thread that pairs lines across block boundaries is not held up while a batch is staged for the GPU
        submit_s = 0;
        // (with a genome image nothing joins FASTA lines: the uploader's copies take 35 ms of a quarter of the threads, the
        // parser gets the rest but one -- 16 threads: 11 parse, parse_and_submit 0.141-0.150 -> 0.124-0.127 s)
        if (!getenv("BREAKFAST_HIP_PARSER_THREADS"))
            popt.threads = image ? std::max(1, threads - std::max(2, threads / 4) - 1) : std::max(1, threads >= 8 ? threads / 2 : threads - g_threads);
        {
            std::mutex mu;
            std::condition_variable cv;
            std::deque<hits::Batch> queue;
            std::vector<hits::Batch> spent;           // submitted batches: their buffers go back to the parser
            std::vector<hits::Batch> in_flight;       // ... those in pinned memory after their uploads have left it
            bool done = false;
            std::thread submitter([&] {
                { std::unique_lock<std::mutex> lk(gpu_mu); gpu_cv.wait(lk, [&] { return gpu_ready; }); }
                for (;;) {
                    hits::Batch b;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return done || !queue.empty(); });
                        if (queue.empty()) return;
                        b = std::move(queue.front());
                        queue.pop_front();
                        cv.notify_all();
                    }
                    const auto t0 = std::chrono::steady_clock::now();
                    if (first_submit_at == 0) first_submit_at = since_start();
                    if (submit_parsed(ctx, b, 0, b.n()) != BF_OK) error("%s", bf_last_error(ctx));
                    submit_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    const hits::Allocator* heap = hits::heap_allocator();
                    const bool pageable = b.seq.home() == heap && b.frag.home() == heap && b.seq_off.home() == heap &&
                                          b.frag_off.home() == heap && b.seq_len.home() == heap && b.hits.home() == heap;
                    if (pageable) {                         // staged when bf_submit returns
                        std::lock_guard<std::mutex> g(mu);
                        spent.push_back(std::move(b));
                        continue;
                    }
                    in_flight.push_back(std::move(b));
                    if (in_flight.size() < 6) continue;
                    if (bf_submit_wait(ctx) != BF_OK) error("%s", bf_last_error(ctx));
                    std::lock_guard<std::mutex> g(mu);
                    for (auto& f : in_flight) spent.push_back(std::move(f));
                    in_flight.clear();
                }
            });
            try {
                hits::parse_stream(in_fd, contigs, genome_path, popt, [&](hits::Batch& b) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return queue.size() < 8; });
                    queue.emplace_back();
                    if (!spent.empty()) { queue.back() = std::move(spent.back()); spent.pop_back(); }
                    std::swap(queue.back(), b);
                    cv.notify_all();
                });
            } catch (const hits::Unpackable& e) {
                // a read outside BAM's alphabet: everything once more with ASCII reads (only offered for a hit list that
                // can be read again; such reads do not come out of a BAM file)
                if (!can_reread(in_fd) || lseek(in_fd, 0, SEEK_SET) != 0) error("%s", e.what());
                { std::lock_guard<std::mutex> g(mu); queue.clear(); done = true; cv.notify_all(); }
                submitter.join();
                genome_thread.join();
                bf_genome_image_close(image);
                spent.clear(); in_flight.clear();          // they live in the ctx's pinned memory
                popt.allocator = nullptr;
                bf_ctx_destroy(ctx);
                ctx = nullptr;
                return false;
            } catch (const std::exception& e) { error("%s", e.what()); }
            { std::lock_guard<std::mutex> g(mu); done = true; cv.notify_all(); }
            submitter.join();
        }
        if (in_fd > 0) close(in_fd);
        if (producer.joinable()) producer.join();
        if (child.pid > 0) { int st = 0; waitpid(child.pid, &st, 0); }     // the reference never checks it either
        popt.allocator = nullptr;
        timing.lap("parse_and_submit");
        timing.add("  of_which_bf_submit", submit_s);
        timing.add("  at_ctx_created", ctx_at);
        timing.add("  at_gpu_accepts_reads", ready_at);
        timing.add("  at_first_bf_submit", first_submit_at);
        genome_thread.join();
        if (bf_genome_end(ctx) != BF_OK) error("%s", bf_last_error(ctx));
        fs.reset();                                                 // the genome now lives in HBM
        bf_genome_image_close(image);
        timing.lap("genome_to_hbm_tail");
        timing.add("  genome_stream_beside_the_parser", genome_s);

        return true;
    };
    if (!load_inputs()) {
        popt.pack = false;
        popt.threads = threads;
        if (!load_inputs()) error("could not read the anchor hits");
    }

    bf_result res;
    uint64_t n_reads = 0;
    bf_stats(ctx, &n_reads, nullptr);
    fprintf(stderr, "Found %llu rearrangement supporting reads.\n", (unsigned long long)n_reads);
    fprintf(stderr, "Sorting rearrangement supporting reads by position...\n");
    fprintf(stderr, "Identifying rearrangements based on clusters of discordant reads...\n");
    if (bf_finish(ctx, &res) != BF_OK) error("%s", bf_last_error(ctx));
    timing.lap("gpu_localise_clusterise");
    // the rows are rendered a slice of clusters at a time while the previous slice is being written
    fflush(stdout);
    {
        const uint64_t nc = res.n_clusters;
        int slices = nc >= 4096 ? 8 : 1;
        if (const char* e = getenv("BREAKFAST_HIP_SV_SLICES")) slices = std::max(1, std::min(64, atoi(e)));    // (A/B switch)
        const size_t hdr = strlen("CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tSUPPORTING READS\tSIGNATURE\tNOTES\n");
        std::vector<std::unique_ptr<char[]>> bufs((size_t)slices);
        auto slice = [&](int k) {
            bf_result part = res;
            const uint64_t c0 = nc * (uint64_t)k / slices, c1 = nc * (uint64_t)(k + 1) / slices;
            part.n_clusters = c1 - c0;
            part.clusters = res.clusters ? res.clusters + c0 : nullptr;
            part.cluster_junction_off = res.cluster_junction_off ? res.cluster_junction_off + c0 : nullptr;
            return part;
        };
        // A slice's text is at most 4 bytes per record byte (two bytes of a record are eight bases) plus a '|' and a ';' per
        // read, plus a row's fixed part per cluster: a buffer of that size is filled by ONE bf_format_sv call (a size query
        // first would decode every record once more: 17 of 59 ms on configs[1]); untouched memory costs nothing.
        std::thread writer;
        double fill_s = 0, wait_s = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        for (int k = 0; k < slices; k++) {
            const bf_result part = slice(k);
            const auto t1 = now();
            uint64_t rec_bytes = res.junction_size, n_rd = res.n_reads;
            if (res.cluster_junction_off && part.n_clusters) {
                rec_bytes = part.cluster_junction_off[part.n_clusters] - part.cluster_junction_off[0];
                n_rd = part.clusters[part.n_clusters - 1].read_end - part.clusters[0].read_begin;
            } else if (!part.n_clusters) { rec_bytes = 0; n_rd = 0; }
            const uint64_t bound = hdr + 4 * rec_bytes + 2 * n_rd + part.n_clusters * (2 * longest_name + 64) + 64;
            bufs[k].reset(new char[(size_t)bound]);
            const int64_t size = bf_format_sv(ctx, &part, bufs[k].get(), bound);
            if (size < 0) error("could not render the result");
            if ((uint64_t)size > bound) {                          // (the bound is generous; if it ever is not: the exact size)
                bufs[k].reset(new char[(size_t)size + 1]);
                if (bf_format_sv(ctx, &part, bufs[k].get(), (uint64_t)size) != size) error("could not render the result");
            }
            const auto t2 = now();
            const char* p = bufs[k].get() + (k ? hdr : 0);       // every slice starts with the header line: once is enough
            const size_t len = (size_t)size - (k ? hdr : 0);
            if (writer.joinable()) writer.join();
            const auto t3 = now();
            fill_s += secs(t1, t2); wait_s += secs(t2, t3);
            writer = std::thread([p, len] { write_all(1, p, len); });
        }
        const auto t4 = now();
        if (writer.joinable()) writer.join();
        wait_s += secs(t4, now());
        timing.lap("format_and_write");
        timing.add("  of_which_formatting", fill_s);
        timing.add("  of_which_waiting_for_the_writer", wait_s);
    }
    if (getenv("BREAKFAST_HIP_TEARDOWN")) { bf_ctx_destroy(ctx); timing.lap("teardown"); }
    timing.print();
    // everything is written: the process ends here and the driver reclaims the GPU memory
    work_done(0);
    _exit(0);
}
