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
//     --threads=N          BGZF inflate workers of the BAM front-end [default: 4]
//   breakfast-hip anchors [--anchor-len=N] <bam_file>
//     prints the anchor FASTA that detect feeds to bowtie (detect.rs:290-338); needs no GPU.
//
// Without --anchor-hits the driver does what detect.rs:58-70 does: spawn
// `bowtie -f -p1 -v0 -m1 -B1 --suppress 5,6,7,8 <genome> -`, feed it from a producer thread that
// scans the BAM, and consume its stdout.  A parser thread turns lines into batches while the main
// thread hands finished batches to the GPU library.
//
// Errors follow src/common.rs:8-13: "ERROR: <msg>" on stderr, exit status -1.

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
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
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/breakfast_hip.h"
#include "bam_reader.h"

extern char** environ;

[[noreturn]] static void error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fputs("ERROR: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    exit(-1);
}

static const char* USAGE =
    "\nUsage:\n  breakfast detect [options] <bam_file> <genome>\n\nOptions:\n"
    "  --anchor-len=N       Anchor length for split read analysis [default: 30]\n"
    "  --anchor-mm=N        Mismatches allowed in anchor alignments [default: 0]\n"
    "  --max-frag-len=N     Maximum fragment length [default: 5000]\n"
    "  --min-evidence=N     Minimum number of supporting DNA fragments [default: 2]\n"
    "  --anchor-hits=FILE   Read Bowtie's anchor alignments from FILE instead of running Bowtie\n";

struct Fasta {
    std::vector<std::string> names;
    std::vector<std::string> seqs;
};

static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// bio::io::fasta::Reader semantics (detect.rs:48-56): id = header up to the first whitespace, sequence
// lines right-trimmed and concatenated, case preserved; a repeated id replaces the earlier record.
static Fasta read_fasta(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) error("Genome FASTA file %s could not be read.", path.c_str());
    Fasta fa;
    std::unordered_map<std::string, size_t> index;
    char* line = nullptr;
    size_t cap = 0;
    ssize_t n;
    long cur = -1;
    while ((n = getline(&line, &cap, f)) >= 0) {
        size_t len = (size_t)n;
        const bool header = len > 0 && line[0] == '>';
        while (len > 0 && is_space(line[len - 1])) len--;
        if (header) {
            size_t e = 1;
            while (e < len && !is_space(line[e])) e++;
            std::string id(line + 1, e - 1);
            auto it = index.find(id);
            if (it == index.end()) {
                index[id] = fa.names.size();
                fa.names.push_back(id);
                fa.seqs.emplace_back();
                cur = (long)fa.names.size() - 1;
            } else {
                cur = (long)it->second;
                fa.seqs[cur].clear();
            }
        } else if (cur >= 0) {
            fa.seqs[cur].append(line, len);
        }
    }
    free(line);
    fclose(f);
    return fa;
}

struct Batch {                          // one bf_batch worth of parsed anchor-hit pairs
    std::string seq, frag;
    std::vector<uint64_t> seq_off{0}, frag_off{0};
    std::vector<bf_hit> hits;
};

// parser thread -> main thread hand-over (at most two batches in flight)
struct BatchQueue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Batch>> q;
    bool closed = false;
    std::string error;
    void push(std::unique_ptr<Batch> b) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.size() < 2; });
        q.push_back(std::move(b));
        cv.notify_all();
    }
    void close(const std::string& err = "") {
        std::lock_guard<std::mutex> g(mu);
        closed = true;
        if (error.empty()) error = err;
        cv.notify_all();
    }
    std::unique_ptr<Batch> pop() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty() || closed; });
        if (q.empty()) return nullptr;
        auto b = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return b;
    }
};

static bool parse_u64(std::string_view s, uint64_t& v) {
    if (s.empty()) return false;
    v = 0;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        v = v * 10 + (uint64_t)(c - '0');
    }
    return true;
}

// nth field (0-based) of `s` split on `d`, like Rust's split(d).nth(n)
static bool field(std::string_view s, char d, int n, std::string_view& out) {
    size_t a = 0;
    for (int i = 0;; i++) {
        size_t b = s.find(d, a);
        if (i == n) { out = s.substr(a, b == std::string_view::npos ? std::string_view::npos : b - a); return true; }
        if (b == std::string_view::npos) return false;
        a = b + 1;
    }
}

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

// the consume loop, detect.rs:76-103: pair 5p/3p lines by read number, extract the fields
static void parse_hits(FILE* in, const std::unordered_map<std::string, uint32_t>& contig_id, const std::string& genome_path,
                       BatchQueue& out) {
    const size_t kBatch = 1u << 20;
    auto cur = std::make_unique<Batch>();
    std::string prev;
    uint64_t prev_read_num = 0;
    char* lineb = nullptr;
    size_t cap = 0;
    ssize_t n;
    auto bad = [&](const char* what, std::string_view line) { throw ParseError(std::string(what) + ": " + std::string(line)); };
    while ((n = getline(&lineb, &cap, in)) >= 0) {
        size_t len = (size_t)n;
        if (len && lineb[len - 1] == '\n') len--;
        if (len && lineb[len - 1] == '\r') len--;
        std::string_view line(lineb, len);
        std::string_view f1;
        uint64_t read_num;
        if (!field(line, ':', 1, f1) || !parse_u64(f1, read_num)) bad("Malformed anchor alignment line", line);
        if (line.substr(0, 3) == "5p:") { prev.assign(line); prev_read_num = read_num; continue; }
        else if (line.substr(0, 3) == "3p:" && read_num != prev_read_num) continue;
        std::string_view frag, seq, st, mst, chr, mchr, p, mp;
        if (!field(line, ':', 2, frag) || !field(line, ':', 3, seq)) bad("Malformed anchor alignment line", line);
        std::string_view pv(prev);
        if (!field(pv, '\t', 1, st) || !field(pv, '\t', 2, chr) || !field(pv, '\t', 3, p) ||
            !field(line, '\t', 1, mst) || !field(line, '\t', 2, mchr) || !field(line, '\t', 3, mp))
            bad("Malformed anchor alignment line", line);
        uint64_t pos5, pos3;
        if (!parse_u64(p, pos5) || !parse_u64(mp, pos3)) bad("Malformed anchor offset", line);
        auto c5 = contig_id.find(std::string(chr)), c3 = contig_id.find(std::string(mchr));
        if (c5 == contig_id.end() || c3 == contig_id.end())
            throw ParseError("Anchor aligned to a contig that " + genome_path + ".fa does not contain (the reference panics here, detect.rs:115).");
        if (pos5 >= 0x80000000ull || pos3 >= 0x80000000ull) throw ParseError("Anchor offset beyond 2^31.");
        bf_hit h;
        memset(&h, 0, sizeof h);
        h.pos5 = (uint32_t)pos5; h.pos3 = (uint32_t)pos3;
        h.contig5 = (uint16_t)c5->second; h.contig3 = (uint16_t)c3->second;
        h.strands = (uint8_t)((st == "+" ? 1 : 0) | (mst == "+" ? 2 : 0));
        cur->hits.push_back(h);
        cur->seq.append(seq); cur->seq_off.push_back(cur->seq.size());
        cur->frag.append(frag); cur->frag_off.push_back(cur->frag.size());
        if (cur->hits.size() >= kBatch) { out.push(std::move(cur)); cur = std::make_unique<Batch>(); }
    }
    free(lineb);
    if (!cur->hits.empty()) out.push(std::move(cur));
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

int main(int argc, char** argv) {
    const bool anchors_only = argc >= 2 && strcmp(argv[1], "anchors") == 0;
    if (argc < 2 || (strcmp(argv[1], "detect") != 0 && !anchors_only)) {
        fputs("\nBreakfast is a software for detecting chromosomal rearrangements in DNA/RNA sequencing data.\n\n"
              "Usage:\n  breakfast-hip detect [options] <bam_file> <genome>\n  breakfast-hip anchors [--anchor-len=N] <bam_file>\n", stderr);
        return -1;
    }
    bf_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.anchor_len = 30; cfg.max_frag_len = 5000; cfg.min_evidence = 2; cfg.device = -1;
    std::string hits_path;
    int threads = 4;
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
        else if (a.compare(0, 14, "--anchor-hits=") == 0) hits_path = a.substr(14);
        else if (a.size() > 1 && a[0] == '-' && a != "-") error("Invalid arguments.\n%s", USAGE);
        else pos.push_back(a);
    }
    if (anchors_only) {
        if (pos.size() != 1) error("Invalid arguments.\nUsage:\n  breakfast-hip anchors [--anchor-len=N] <bam_file>");
        auto bam = open_bam(pos[0], threads);
        try {
            bam::dispatch_reads(*bam, stdout, cfg.anchor_len);
        } catch (const std::exception& e) { error("Failed reading BAM record: %s", e.what()); }   // common.rs:64
        fflush(stdout);
        return 0;
    }
    if (pos.size() != 2) error("Invalid arguments.\n%s", USAGE);
    const std::string sam_path = pos[0], genome_path = pos[1];

    // Bowtie and the BAM producer start BEFORE the GPU is touched (the child must be spawned from a
    // process that has not initialised the device); the pipes simply fill until we start consuming.
    FILE* in = nullptr;
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
        in = fdopen(child.from_child, "r");
    } else {
        in = hits_path == "-" ? stdin : fopen(hits_path.c_str(), "rb");
        if (!in) error("Cannot open file %s for reading.", hits_path.c_str());
    }

    bf_ctx* ctx = nullptr;
    if (bf_ctx_create(&cfg, &ctx) != BF_OK) error("%s", bf_last_error(nullptr));

    Fasta fa = read_fasta(genome_path + ".fa");
    fprintf(stderr, "Reading reference genome into memory...\n");
    {
        std::vector<const char*> names; std::vector<const uint8_t*> seqs; std::vector<uint64_t> lens;
        for (size_t i = 0; i < fa.names.size(); i++) {
            names.push_back(fa.names[i].c_str());
            seqs.push_back((const uint8_t*)fa.seqs[i].data());
            lens.push_back(fa.seqs[i].size());
        }
        if (names.empty()) error("Genome FASTA file %s.fa holds no sequences.", genome_path.c_str());
        if (bf_genome_set(ctx, (uint32_t)names.size(), names.data(), seqs.data(), lens.data()) != BF_OK)
            error("%s", bf_last_error(ctx));
    }
    std::unordered_map<std::string, uint32_t> contig_id;
    for (size_t i = 0; i < fa.names.size(); i++) contig_id[fa.names[i]] = (uint32_t)i;
    for (auto& s : fa.seqs) std::string().swap(s);             // the genome now lives in HBM

    fprintf(stderr, "Splitting unaligned reads into %u bp anchors and aligning against the genome...\n", cfg.anchor_len);

    BatchQueue queue;
    std::thread parser([&] {
        try { parse_hits(in, contig_id, genome_path, queue); queue.close(); }
        catch (const std::exception& e) { queue.close(e.what()); }
    });
    while (auto b = queue.pop()) {
        bf_batch bb;
        bb.n_reads = b->hits.size();
        bb.seq_bytes = (const uint8_t*)b->seq.data(); bb.seq_off = b->seq_off.data();
        bb.frag_bytes = (const uint8_t*)b->frag.data(); bb.frag_off = b->frag_off.data();
        bb.hits = b->hits.data();
        if (bf_submit(ctx, &bb) != BF_OK) error("%s", bf_last_error(ctx));
    }
    parser.join();
    if (!queue.error.empty()) error("%s", queue.error.c_str());
    if (in != stdin) fclose(in);
    if (producer.joinable()) producer.join();
    if (child.pid > 0) { int st = 0; waitpid(child.pid, &st, 0); }     // the reference never checks it either

    bf_result res;
    if (bf_localise(ctx) != BF_OK) error("%s", bf_last_error(ctx));
    uint64_t n_reads = 0;
    bf_stats(ctx, &n_reads, nullptr);
    fprintf(stderr, "Found %llu rearrangement supporting reads.\n", (unsigned long long)n_reads);
    fprintf(stderr, "Sorting rearrangement supporting reads by position...\n");
    fprintf(stderr, "Identifying rearrangements based on clusters of discordant reads...\n");
    if (bf_clusterise(ctx, &res) != BF_OK) error("%s", bf_last_error(ctx));
    const int64_t size = bf_format_sv(ctx, &res, nullptr, 0);
    if (size < 0) error("could not render the result");
    std::string out((size_t)size, '\0');
    bf_format_sv(ctx, &res, out.data(), (uint64_t)size);
    fwrite(out.data(), 1, out.size(), stdout);
    fflush(stdout);
    bf_ctx_destroy(ctx);
    return 0;
}
