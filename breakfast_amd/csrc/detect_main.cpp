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
//
// Errors follow src/common.rs:8-13: "ERROR: <msg>" on stderr, exit status -1.

#include <cerrno>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/breakfast_hip.h"

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

struct Batcher {
    bf_ctx* ctx;
    std::string seq, frag;
    std::vector<uint64_t> seq_off{0}, frag_off{0};
    std::vector<bf_hit> hits;
    void flush() {
        if (hits.empty()) return;
        bf_batch b;
        b.n_reads = hits.size();
        b.seq_bytes = (const uint8_t*)seq.data(); b.seq_off = seq_off.data();
        b.frag_bytes = (const uint8_t*)frag.data(); b.frag_off = frag_off.data();
        b.hits = hits.data();
        if (bf_submit(ctx, &b) != BF_OK) error("%s", bf_last_error(ctx));
        seq.clear(); frag.clear(); hits.clear();
        seq_off.assign(1, 0); frag_off.assign(1, 0);
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

int main(int argc, char** argv) {
    if (argc < 2 || strcmp(argv[1], "detect") != 0) {
        fputs("\nBreakfast is a software for detecting chromosomal rearrangements in DNA/RNA sequencing data.\n\n"
              "Usage:\n  breakfast-hip detect [options] <bam_file> <genome>\n", stderr);
        return -1;
    }
    bf_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.anchor_len = 30; cfg.max_frag_len = 5000; cfg.min_evidence = 2; cfg.device = -1;
    std::string hits_path;
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
        else if (a.compare(0, 14, "--anchor-hits=") == 0) hits_path = a.substr(14);
        else if (a.size() > 1 && a[0] == '-' && a != "-") error("Invalid arguments.\n%s", USAGE);
        else pos.push_back(a);
    }
    if (pos.size() != 2) error("Invalid arguments.\n%s", USAGE);
    const std::string genome_path = pos[1];

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
    if (hits_path.empty())
        error("The BAM front-end (detect.rs:290-338) and the Bowtie child process (detect.rs:60-70) are not part of "
              "this build yet: pass Bowtie's output with --anchor-hits=FILE.");
    FILE* in = hits_path == "-" ? stdin : fopen(hits_path.c_str(), "rb");
    if (!in) error("Cannot open file %s for reading.", hits_path.c_str());

    // the consume loop, detect.rs:76-103
    Batcher bt{ctx};
    std::string prev;
    uint64_t prev_read_num = 0;
    char* lineb = nullptr;
    size_t cap = 0;
    ssize_t n;
    const size_t kBatch = 1u << 20;
    while ((n = getline(&lineb, &cap, in)) >= 0) {
        size_t len = (size_t)n;
        if (len && lineb[len - 1] == '\n') len--;
        if (len && lineb[len - 1] == '\r') len--;
        std::string_view line(lineb, len);
        std::string_view f1;
        uint64_t read_num;
        if (!field(line, ':', 1, f1) || !parse_u64(f1, read_num)) error("Malformed anchor alignment line: %.*s", (int)len, lineb);
        if (line.substr(0, 3) == "5p:") { prev.assign(line); prev_read_num = read_num; continue; }
        else if (line.substr(0, 3) == "3p:" && read_num != prev_read_num) continue;
        std::string_view frag, seq, st, mst, chr, mchr, p, mp;
        if (!field(line, ':', 2, frag) || !field(line, ':', 3, seq)) error("Malformed anchor alignment line: %.*s", (int)len, lineb);
        std::string_view pv(prev);
        if (!field(pv, '\t', 1, st) || !field(pv, '\t', 2, chr) || !field(pv, '\t', 3, p) ||
            !field(line, '\t', 1, mst) || !field(line, '\t', 2, mchr) || !field(line, '\t', 3, mp))
            error("Malformed anchor alignment line: %.*s", (int)len, lineb);
        uint64_t pos5, pos3;
        if (!parse_u64(p, pos5) || !parse_u64(mp, pos3)) error("Malformed anchor offset: %.*s", (int)len, lineb);
        auto c5 = contig_id.find(std::string(chr)), c3 = contig_id.find(std::string(mchr));
        if (c5 == contig_id.end() || c3 == contig_id.end())
            error("Anchor aligned to a contig that %s.fa does not contain (the reference panics here, detect.rs:115).", genome_path.c_str());
        if (pos5 >= 0x80000000ull || pos3 >= 0x80000000ull) error("Anchor offset beyond 2^31.");
        bf_hit h;
        memset(&h, 0, sizeof h);
        h.pos5 = (uint32_t)pos5; h.pos3 = (uint32_t)pos3;
        h.contig5 = (uint16_t)c5->second; h.contig3 = (uint16_t)c3->second;
        h.strands = (uint8_t)((st == "+" ? 1 : 0) | (mst == "+" ? 2 : 0));
        bt.hits.push_back(h);
        bt.seq.append(seq); bt.seq_off.push_back(bt.seq.size());
        bt.frag.append(frag); bt.frag_off.push_back(bt.frag.size());
        if (bt.hits.size() >= kBatch) bt.flush();
    }
    free(lineb);
    if (in != stdin) fclose(in);
    bt.flush();

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
