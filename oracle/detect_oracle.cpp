// CPU oracle (C++17, single thread) for `breakfast detect` -- TEST INFRASTRUCTURE ONLY.
//
// A restatement of the reference's algorithm, /root/reference/src/detect.rs:76-287,
// keeping the reference's own data structures (one heap-allocated Evidence per read,
// HashMap genome, stable merge sort with string compares, O(n*w) greedy clustering,
// O(c^2) frag-id dedup) so that it doubles as the timed CPU baseline
// (bench.py `cpu_baseline`, kind "port").  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this; the product never does.
//
// PARITY UNPINNED by the reference: it ships no tests/fixtures/golden vectors and cannot
// be built in this image (no cargo/rustc; bowtie/htslib absent).  Pinned instead by the
// hand-checked known-answer vector of SURVEY.md section 8c and by agreement with the
// independent pure-Python restatement (oracle/detect_oracle.py) on every fixture.
//
// Third-party arithmetic restated (crate `bio` 0.39, Cargo.toml:14, not vendored):
//   alphabets::dna::complement/revcomp: identity except AGCTYRWSKMDVHBN->TCGARYWSMKHBDVN
//   (and lowercase); io::fasta::Reader: id = first word of header, lines right-trimmed.
// All usize arithmetic wraps at 64 bits (release build, README.md:19).

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace {

using Bytes = std::vector<uint8_t>;

struct Complement {
    uint8_t t[256];
    Complement() {
        for (int i = 0; i < 256; i++) t[i] = (uint8_t)i;
        const char* a = "AGCTYRWSKMDVHBN";
        const char* b = "TCGARYWSMKHBDVN";
        for (int i = 0; a[i]; i++) {
            t[(uint8_t)a[i]] = (uint8_t)b[i];
            t[(uint8_t)a[i] + 32] = (uint8_t)(b[i] + 32);
        }
    }
};
const Complement COMP;

inline uint8_t complement(uint8_t b) { return COMP.t[b]; }          // detect.rs:121,130
Bytes revcomp(const Bytes& s) {                                      // detect.rs:112
    Bytes r(s.size());
    for (size_t i = 0; i < s.size(); i++) r[i] = complement(s[s.size() - 1 - i]);
    return r;
}
inline uint8_t to_lower(uint8_t b) { return (b >= 'A' && b <= 'Z') ? (uint8_t)(b + 32) : b; }

// struct Evidence, detect.rs:13-26 (+ a few extras for kernel-level parity checks)
struct Evidence {
    std::string chr;
    uint64_t pos;
    bool strand;
    std::string mchr;
    uint64_t mpos;
    bool mstrand;
    Bytes sequence;
    Bytes signature;
    Bytes frag_id;
    // extras
    uint32_t bp;
    bool swapped;
    uint64_t input_index;   // index of the read in the consume order / batch
    uint32_t full_len;
};

using Genome = std::unordered_map<std::string, Bytes>;              // detect.rs:52

inline uint8_t get_or_n(const Bytes& s, uint64_t idx) {              // .get(idx).unwrap_or(&b'N')
    return idx < s.size() ? s[idx] : (uint8_t)'N';
}

// One iteration of the consume loop body, detect.rs:92-187.
void localise(const Genome& genome, uint64_t anchor_len, std::string_view frag_id, Bytes seq,
              bool strand, std::string_view chr, uint64_t pos, bool mstrand,
              std::string_view mchr, uint64_t mpos, std::vector<Evidence>& evidence,
              uint64_t input_index) {
    const uint64_t full_len = seq.size();
    bool swapped = false;
    if (chr > mchr || (chr == mchr && pos > mpos)) {                 // detect.rs:108-113
        std::swap(chr, mchr);
        std::swap(pos, mpos);
        bool tmp = strand; strand = !mstrand; mstrand = !tmp;
        seq = revcomp(seq);
        swapped = true;
    }
    auto it = genome.find(std::string(chr));
    auto mit = genome.find(std::string(mchr));
    if (it == genome.end() || mit == genome.end()) {                 // genome[chr] panics, :115
        fprintf(stderr, "oracle: contig missing from genome (reference would panic)\n");
        abort();
    }
    const Bytes& chr_seq = it->second;
    Bytes left(full_len), right(full_len);
    if (strand) {                                                    // detect.rs:116-122
        for (uint64_t k = 0; k < full_len; k++) left[k] = get_or_n(chr_seq, pos - 1 + k);
    } else {
        for (uint64_t k = 0; k < full_len; k++)
            left[k] = complement(get_or_n(chr_seq, pos + anchor_len - 2 - k));
    }
    const Bytes& mchr_seq = mit->second;
    if (mstrand) {                                                   // detect.rs:125-131
        for (uint64_t k = 0; k < full_len; k++)
            right[k] = get_or_n(mchr_seq, mpos + anchor_len - full_len - 1 + k);
    } else {
        for (uint64_t k = 0; k < full_len; k++)
            right[k] = complement(get_or_n(mchr_seq, mpos + full_len - 2 - k));
    }

    std::vector<uint64_t> mismatches(full_len, 0);                   // detect.rs:135-153
    for (uint64_t k = 0; k < anchor_len; k++) mismatches[anchor_len] += (seq[k] != left[k]);
    for (uint64_t k = anchor_len; k < full_len; k++) mismatches[anchor_len] += (seq[k] != right[k]);
    for (uint64_t k = anchor_len + 1; k + anchor_len < full_len; k++)
        mismatches[k] = mismatches[k - 1] + (seq[k - 1] != left[k - 1]) - (seq[k - 1] != right[k - 1]);
    uint64_t bp = anchor_len;
    uint64_t least = mismatches[anchor_len];
    for (uint64_t k = anchor_len + 1; k + anchor_len < full_len; k++)
        if (mismatches[k] < least) { bp = k; least = mismatches[k]; }

    Bytes junction(full_len + 1, ' ');                               // detect.rs:155-162
    for (uint64_t k = 0; k < bp; k++) junction[k] = seq[k] == left[k] ? seq[k] : to_lower(seq[k]);
    junction[bp] = '|';
    for (uint64_t k = bp; k < full_len; k++)
        junction[k + 1] = seq[k] == right[k] ? seq[k] : to_lower(seq[k]);

    Bytes signature(17, ' ');                                        // detect.rs:165-168
    if (bp < 8 || bp + 8 > full_len) {
        fprintf(stderr, "oracle: signature index out of range (reference would panic)\n");
        abort();
    }
    for (int k = 0; k < 8; k++) signature[k] = left[bp - 8 + k];
    signature[8] = '|';
    for (int k = 0; k < 8; k++) signature[9 + k] = right[bp + k];

    uint64_t left_bp = strand ? pos + bp - 1 : pos + anchor_len - bp;              // :172-176
    uint64_t right_bp = mstrand ? mpos + anchor_len - (full_len - bp) : mpos + (full_len - bp);

    Evidence e;
    e.chr = std::string(chr); e.pos = left_bp; e.strand = strand;
    e.mchr = std::string(mchr); e.mpos = right_bp; e.mstrand = mstrand;
    e.sequence = std::move(junction); e.signature = std::move(signature);
    e.frag_id.assign(frag_id.begin(), frag_id.end());
    e.bp = (uint32_t)bp; e.swapped = swapped; e.input_index = input_index;
    e.full_len = (uint32_t)full_len;
    evidence.push_back(std::move(e));
}

// detect.rs:247-287
std::vector<const Evidence*> remove_frag_id_duplicates(const std::vector<const Evidence*>& evidence) {
    std::vector<const Evidence*> filtered;
    std::vector<int32_t> redundant_with(evidence.size(), -1);
    for (size_t a = 0; a < evidence.size(); a++) {
        if (redundant_with[a] >= 0) continue;
        redundant_with[a] = (int32_t)a;
        int num_redundant = 1;
        for (size_t b = a + 1; b < evidence.size(); b++) {
            if (redundant_with[b] >= 0) continue;
            if (evidence[a]->frag_id == evidence[b]->frag_id) {
                redundant_with[b] = (int32_t)a;
                num_redundant += 1;
            }
        }
        if (num_redundant == 1) { filtered.push_back(evidence[a]); continue; }
        size_t best = 0, best_score = 0;
        for (size_t k = 0; k < evidence.size(); k++) {
            if (redundant_with[k] != (int32_t)a) continue;
            const Bytes& seq = evidence[k]->sequence;
            size_t left_len = std::find(seq.begin(), seq.end(), (uint8_t)'|') - seq.begin();
            size_t right_len = seq.size() - left_len - 1;
            size_t score = std::min(left_len, right_len) * 10000 + std::max(left_len, right_len);
            if (score > best_score) { best = k; best_score = score; }
        }
        filtered.push_back(evidence[best]);
    }
    return filtered;
}

struct Run {
    std::vector<Evidence> evidence;       // in consume order until clustered (then sorted)
    std::string sv;
    uint64_t n_clusters = 0, n_support = 0;
    // consume-order snapshots for kernel-level parity (filled before sorting)
    std::vector<uint32_t> bp, len;
    std::vector<uint8_t> flags;           // bit0 strand, bit1 mstrand, bit2 swapped
    std::vector<uint64_t> lpos, rpos, input_index;
    std::vector<uint8_t> signature;       // 17 bytes each
    std::vector<uint8_t> junction; std::vector<uint64_t> junction_off;
};

void snapshot(Run& r) {
    size_t n = r.evidence.size();
    r.junction_off.push_back(0);
    for (size_t i = 0; i < n; i++) {
        const Evidence& e = r.evidence[i];
        r.bp.push_back(e.bp); r.len.push_back(e.full_len);
        r.flags.push_back((uint8_t)((e.strand ? 1 : 0) | (e.mstrand ? 2 : 0) | (e.swapped ? 4 : 0)));
        r.lpos.push_back(e.pos); r.rpos.push_back(e.mpos); r.input_index.push_back(e.input_index);
        r.signature.insert(r.signature.end(), e.signature.begin(), e.signature.end());
        r.junction.insert(r.junction.end(), e.sequence.begin(), e.sequence.end());
        r.junction_off.push_back(r.junction.size());
    }
}

void append_u64(std::string& s, uint64_t v) {
    char buf[24]; int n = snprintf(buf, sizeof buf, "%llu", (unsigned long long)v); s.append(buf, n);
}

// Sort, cluster, dedup, print: detect.rs:193-241.
// emit_text = false (bench.py's CPU baseline only): everything but the row text, so that the timed scope is the
// GPU step's -- evidence in, cluster table out; the printing is bf_format_sv's job there and is timed apart.
void cluster_and_format(Run& r, uint64_t max_frag_len, uint64_t min_evidence, bool emit_text = true) {
    std::vector<Evidence>& evidence = r.evidence;
    std::stable_sort(evidence.begin(), evidence.end(), [](const Evidence& a, const Evidence& b) {
        int c = a.chr.compare(b.chr);                     // byte-wise, like Rust String Ord
        if (c != 0) return c < 0;
        return a.pos < b.pos;
    });
    std::string& out = r.sv;
    out += "CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\t"
           "SUPPORTING READS\tSIGNATURE\tNOTES\n";
    std::vector<bool> reported(evidence.size(), false);
    for (size_t p = 0; p < evidence.size(); p++) {
        if (reported[p]) continue;
        const Evidence& read = evidence[p];
        std::vector<const Evidence*> cluster{&read};
        for (size_t s = p + 1; s < evidence.size(); s++) {
            const Evidence& e = evidence[s];
            if (e.chr != read.chr) break;
            if (e.pos - read.pos > max_frag_len) break;
            if (e.mchr != read.mchr) continue;
            int64_t d = (int64_t)e.mpos - (int64_t)read.mpos;
            if ((d < 0 ? -d : d) > (int64_t)max_frag_len) continue;
            if (e.strand != read.strand) continue;
            if (e.mstrand != read.mstrand) continue;
            if (e.signature != read.signature) continue;
            cluster.push_back(&e);
            reported[s] = true;
        }
        if (cluster.size() < min_evidence) continue;
        cluster = remove_frag_id_duplicates(cluster);
        if (cluster.size() < min_evidence) continue;
        r.n_clusters++;
        if (!emit_text) { r.n_support += cluster.size(); continue; }
        out += read.chr; out += '\t'; out += read.strand ? '+' : '-'; out += '\t';
        append_u64(out, read.pos); out += "\t\t";
        out += read.mchr; out += '\t'; out += read.mstrand ? '+' : '-'; out += '\t';
        append_u64(out, read.mpos); out += "\t\t";
        for (size_t k = 0; k < cluster.size(); k++) {
            if (k > 0) out += ';';
            out.append((const char*)cluster[k]->sequence.data(), cluster[k]->sequence.size());
        }
        out += "\t\t\n";
        r.n_support += cluster.size();
    }
}

std::vector<std::string_view> split(std::string_view s, char d) {
    std::vector<std::string_view> v;
    size_t a = 0;
    for (;;) {
        size_t b = s.find(d, a);
        if (b == std::string_view::npos) { v.push_back(s.substr(a)); break; }
        v.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return v;
}

uint64_t parse_usize(std::string_view s) {
    if (s.empty()) { fprintf(stderr, "oracle: cannot parse integer (reference would panic)\n"); abort(); }
    uint64_t v = 0;
    for (char c : s) {
        if (c < '0' || c > '9') { fprintf(stderr, "oracle: cannot parse integer (reference would panic)\n"); abort(); }
        v = v * 10 + (uint64_t)(c - '0');
    }
    return v;
}

// The consume loop over bowtie's stdout, detect.rs:72-188.
void consume_lines(const Genome& genome, std::string_view text, uint64_t anchor_len, Run& r) {
    std::string_view prev;
    uint64_t prev_read_num = 0;
    uint64_t n_used = 0;
    size_t a = 0;
    while (a < text.size()) {
        size_t b = text.find('\n', a);
        if (b == std::string_view::npos) b = text.size();
        std::string_view line = text.substr(a, b - a);
        a = b + 1;
        if (!line.empty() && line.back() == '\r') line.remove_suffix(1);     // BufRead::lines
        auto fields = split(line, ':');
        if (fields.size() < 2) { fprintf(stderr, "oracle: malformed line (reference would panic)\n"); abort(); }
        uint64_t read_num = parse_usize(fields[1]);                          // detect.rs:79
        if (line.substr(0, 3) == "5p:") { prev = line; prev_read_num = read_num; continue; }
        else if (line.substr(0, 3) == "3p:" && read_num != prev_read_num) continue;
        if (fields.size() < 4) { fprintf(stderr, "oracle: malformed line (reference would panic)\n"); abort(); }
        std::string_view frag_id = fields[2];                                // detect.rs:90
        Bytes seq(fields[3].begin(), fields[3].end());                       // detect.rs:91
        auto cols = split(prev, '\t');
        auto mcols = split(line, '\t');
        if (cols.size() < 4 || mcols.size() < 4) { fprintf(stderr, "oracle: malformed line (reference would panic)\n"); abort(); }
        bool strand = cols[1] == "+";
        bool mstrand = mcols[1] == "+";
        localise(genome, anchor_len, frag_id, std::move(seq), strand, cols[2], parse_usize(cols[3]),
                 mstrand, mcols[2], parse_usize(mcols[3]), r.evidence, n_used);
        n_used++;
    }
}

Genome parse_fasta(std::string_view text) {                                 // detect.rs:48-56
    Genome genome;
    std::string name; bool have = false; Bytes seq;
    size_t a = 0;
    while (a <= text.size()) {
        size_t b = text.find('\n', a);
        if (b == std::string_view::npos) b = text.size();
        std::string_view raw = text.substr(a, b - a);
        a = b + 1;
        std::string_view line = raw;
        while (!line.empty() && (line.back() == ' ' || line.back() == '\t' || line.back() == '\r' ||
                                 line.back() == '\n' || line.back() == '\v' || line.back() == '\f'))
            line.remove_suffix(1);
        if (!raw.empty() && raw[0] == '>') {
            if (have) genome[name] = std::move(seq);
            seq = Bytes();
            std::string_view h = line.substr(1);
            size_t e = 0;
            while (e < h.size() && !(h[e] == ' ' || h[e] == '\t' || h[e] == '\r' || h[e] == '\v' || h[e] == '\f')) e++;
            name = std::string(h.substr(0, e));
            have = true;
        } else if (have) {
            seq.insert(seq.end(), line.begin(), line.end());
        }
        if (b == text.size()) break;
    }
    if (have) genome[name] = std::move(seq);
    return genome;
}

}  // namespace

extern "C" {

struct oracle_hit {       // same 16-byte layout as include/breakfast_hip.h bf_hit
    uint32_t pos5, pos3;
    uint16_t contig5, contig3;
    uint8_t strands;      // bit0: 5p anchor on '+', bit1: 3p anchor on '+'
    uint8_t pad[3];
};

void* oracle_genome_from_fasta(const char* text, uint64_t n) {
    return new Genome(parse_fasta(std::string_view(text, n)));
}
void* oracle_genome_new() { return new Genome(); }
void oracle_genome_add(void* g, const char* name, const uint8_t* seq, uint64_t len) {
    (*(Genome*)g)[name] = Bytes(seq, seq + len);
}
void oracle_genome_free(void* g) { delete (Genome*)g; }

// Text path: exactly what detect::main does with bowtie's stdout.
void* oracle_detect_text(void* g, const char* lines, uint64_t n, uint64_t anchor_len,
                         uint64_t max_frag_len, uint64_t min_evidence, int keep_evidence) {
    Run* r = new Run();
    consume_lines(*(Genome*)g, std::string_view(lines, n), anchor_len, *r);
    if (keep_evidence) snapshot(*r);
    cluster_and_format(*r, max_frag_len, min_evidence);
    return r;
}

// Batch path: the same per-read math, sort, clustering and dedup on an in-memory parsed batch
// (what the consume loop holds after detect.rs:89-103).  Used for the timed CPU baseline.
// stage: 1 = localise only, 2 = localise + sort/cluster/dedup/format, 3 = as 2 without the row text (timing).
void* oracle_detect_batch(void* g, const char* const* contig_names, uint64_t n_reads,
                          const uint8_t* seq_bytes, const uint64_t* seq_off,
                          const uint8_t* frag_bytes, const uint64_t* frag_off,
                          const oracle_hit* hits, uint64_t anchor_len, uint64_t max_frag_len,
                          uint64_t min_evidence, int keep_evidence, int stage) {
    Run* r = new Run();
    const Genome& genome = *(Genome*)g;
    for (uint64_t i = 0; i < n_reads; i++) {
        const oracle_hit& h = hits[i];
        Bytes seq(seq_bytes + seq_off[i], seq_bytes + seq_off[i + 1]);
        std::string_view frag((const char*)frag_bytes + frag_off[i], frag_off[i + 1] - frag_off[i]);
        localise(genome, anchor_len, frag, std::move(seq), (h.strands & 1) != 0,
                 contig_names[h.contig5], h.pos5, (h.strands & 2) != 0, contig_names[h.contig3],
                 h.pos3, r->evidence, i);
    }
    if (keep_evidence) snapshot(*r);
    if (stage >= 2) cluster_and_format(*r, max_frag_len, min_evidence, stage != 3);
    return r;
}

const char* oracle_sv(void* run, uint64_t* n) { Run* r = (Run*)run; *n = r->sv.size(); return r->sv.data(); }
uint64_t oracle_n_evidence(void* run) { return ((Run*)run)->evidence.size(); }
uint64_t oracle_n_clusters(void* run) { return ((Run*)run)->n_clusters; }
uint64_t oracle_n_support(void* run) { return ((Run*)run)->n_support; }
const uint32_t* oracle_ev_bp(void* run) { return ((Run*)run)->bp.data(); }
const uint32_t* oracle_ev_len(void* run) { return ((Run*)run)->len.data(); }
const uint8_t* oracle_ev_flags(void* run) { return ((Run*)run)->flags.data(); }
const uint64_t* oracle_ev_lpos(void* run) { return ((Run*)run)->lpos.data(); }
const uint64_t* oracle_ev_rpos(void* run) { return ((Run*)run)->rpos.data(); }
const uint64_t* oracle_ev_index(void* run) { return ((Run*)run)->input_index.data(); }
const uint8_t* oracle_ev_signature(void* run) { return ((Run*)run)->signature.data(); }
const uint8_t* oracle_ev_junction(void* run) { return ((Run*)run)->junction.data(); }
const uint64_t* oracle_ev_junction_off(void* run) { return ((Run*)run)->junction_off.data(); }
void oracle_run_free(void* run) { delete (Run*)run; }

}  // extern "C"

#ifdef ORACLE_MAIN
// CLI: detect_oracle [--anchor-len=N] [--max-frag-len=N] [--min-evidence=N] <bowtie_lines_file> <genome>
// reads <genome>.fa like the reference (detect.rs:48) and bowtie's stdout from a file (or "-").
static std::string slurp(const char* path) {
    FILE* f = strcmp(path, "-") == 0 ? stdin : fopen(path, "rb");
    if (!f) { fprintf(stderr, "ERROR: cannot open %s\n", path); exit(255); }
    std::string s; char buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    if (f != stdin) fclose(f);
    return s;
}
int main(int argc, char** argv) {
    uint64_t A = 30, F = 5000, E = 2;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "--anchor-len=", 13)) A = strtoull(argv[i] + 13, 0, 10);
        else if (!strncmp(argv[i], "--max-frag-len=", 15)) F = strtoull(argv[i] + 15, 0, 10);
        else if (!strncmp(argv[i], "--min-evidence=", 15)) E = strtoull(argv[i] + 15, 0, 10);
        else if (!strncmp(argv[i], "--anchor-mm=", 12)) {}
        else pos.push_back(argv[i]);
    }
    if (pos.size() != 2) { fprintf(stderr, "usage: detect_oracle [options] <bowtie_lines> <genome>\n"); return 255; }
    std::string fa = slurp((std::string(pos[1]) + ".fa").c_str());
    std::string lines = slurp(pos[0]);
    void* g = oracle_genome_from_fasta(fa.data(), fa.size());
    void* r = oracle_detect_text(g, lines.data(), lines.size(), A, F, E, 0);
    fprintf(stderr, "Found %llu rearrangement supporting reads.\n", (unsigned long long)oracle_n_evidence(r));
    uint64_t n; const char* sv = oracle_sv(r, &n);
    fwrite(sv, 1, n, stdout);
    return 0;
}
#endif
