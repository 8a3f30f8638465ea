// `breakfast filter` (src/filter.rs) and `breakfast annotate` (src/annotate.rs): the two text tools
// downstream of detect's .sv format.  Host code only -- the data are
// kilobytes to megabytes of text and there is nothing here for a GPU to do; they are built so that a
// user of the reference finds the whole detect -> filter -> annotate chain, and as a regression on
// the .sv wire format (columns 0,1,2,4,5,6,8 are parsed back, filter.rs:139-146).
//
// Error convention: src/common.rs:8-13, "ERROR: <msg>" on stderr and exit status -1.  Inputs on which
// the reference panics (.unwrap() on a failed parse, slice out of range) are errors here too, with a
// message that names the line; they are marked "panics" below.

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <sys/stat.h>
#include <zlib.h>

namespace {

[[noreturn]] void error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fputs("ERROR: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    exit(-1);
}

// common.rs:26-57.  "-" is stdin; a path ending in ".gz" is inflated (the reference pipes it through
// `gunzip -c`; zlib's gz reader gives the same bytes); read_line drops one trailing '\n' and nothing else.
class FileReader {
public:
    explicit FileReader(const std::string& path) {
        const bool gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
        if (path == "-") f_ = stdin;
        else if (gz) { g_ = gzopen(path.c_str(), "rb"); if (g_) gzbuffer(g_, 1 << 18); }
        else f_ = fopen(path.c_str(), "rb");
        if (!f_ && !g_) error("Cannot open file %s for reading.", path.c_str());
    }
    ~FileReader() {
        if (g_) gzclose(g_);
        if (f_ && f_ != stdin) fclose(f_);
        free(buf_);
    }
    FileReader(const FileReader&) = delete;
    FileReader& operator=(const FileReader&) = delete;
    bool read_line(std::string& line) {
        line.clear();
        bool any = false;
        if (f_) {
            const ssize_t n = getline(&buf_, &cap_, f_);
            if (n < 0) { if (ferror(f_)) error("I/O error while reading from file."); }
            else { any = n > 0; line.assign(buf_, (size_t)n); }
        } else {
            char chunk[1 << 16];
            while (gzgets(g_, chunk, sizeof chunk)) {
                any = true;
                const size_t n = strlen(chunk);
                line.append(chunk, n);
                if (n && chunk[n - 1] == '\n') break;
            }
            if (!any && !gzeof(g_)) error("I/O error while reading from file.");
        }
        if (!line.empty() && line.back() == '\n') line.pop_back();
        return any;
    }
private:
    FILE* f_ = nullptr;
    gzFile g_ = nullptr;
    char* buf_ = nullptr;
    size_t cap_ = 0;
};

std::vector<std::string_view> split(std::string_view s, char d) {
    std::vector<std::string_view> out;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(d, a);
        if (b == std::string_view::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}

bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
std::string_view trim(std::string_view s) {               // str::trim on ASCII input
    while (!s.empty() && is_ws(s.front())) s.remove_prefix(1);
    while (!s.empty() && is_ws(s.back())) s.remove_suffix(1);
    return s;
}

// usize / u32 ::from_str: optional '+', decimal digits, no overflow
bool parse_uint(std::string_view s, uint64_t max, uint64_t& v) {
    if (!s.empty() && s.front() == '+') s.remove_prefix(1);
    if (s.empty()) return false;
    v = 0;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        const uint64_t d = (uint64_t)(c - '0');
        if (v > (max - d) / 10) return false;
        v = v * 10 + d;
    }
    return true;
}

struct Options {
    std::vector<std::string> positional;
    std::unordered_map<std::string, std::string> values;   // --name=value or --name value
    std::unordered_set<std::string> flags;
};

// the subset of docopt the three usage strings need: options with a value take "--opt=V" or "--opt V"
Options parse_options(int argc, char** argv, int first, const std::vector<std::string>& valued,
                      const std::vector<std::string>& boolean, const char* usage) {
    Options o;
    for (int i = first; i < argc; i++) {
        const std::string a = argv[i];
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            const size_t eq = a.find('=');
            const std::string name = a.substr(0, eq);
            if (std::find(valued.begin(), valued.end(), name) != valued.end()) {
                if (eq != std::string::npos) o.values[name] = a.substr(eq + 1);
                else if (i + 1 < argc) o.values[name] = argv[++i];
                else error("Invalid arguments.\n%s", usage);
            } else if (eq == std::string::npos && std::find(boolean.begin(), boolean.end(), name) != boolean.end()) {
                o.flags.insert(name);
            } else {
                error("Invalid arguments.\n%s", usage);
            }
        } else if (a.size() > 1 && a[0] == '-') {
            error("Invalid arguments.\n%s", usage);
        } else {
            o.positional.push_back(a);
        }
    }
    return o;
}

// ---------------------------------------------------------------------------------------------
// filter

const char* FILTER_USAGE =
    "\nUsage:\n  breakfast filter [options] <sample_sheet>\n\nOptions:\n"
    "  --min-reads=N        Minimum required number of supporting reads [default: 4]\n"
    "  --min-distance=N     Minimum distance between breakpoints [default: 10]\n"
    "  --keep-duplicates    Do not merge supporting reads with identical sequences\n"
    "  --output-dir=PATH    Directory for output files [default: annotated]\n";

struct Rearrangement {                    // filter.rs:23-31
    std::string chr_1; uint64_t pos_1; char strand_1;
    std::string chr_2; uint64_t pos_2; char strand_2;
    bool operator==(const Rearrangement& o) const {
        return pos_1 == o.pos_1 && pos_2 == o.pos_2 && strand_1 == o.strand_1 && strand_2 == o.strand_2 &&
               chr_1 == o.chr_1 && chr_2 == o.chr_2;
    }
};
struct RearrangementHash {
    size_t operator()(const Rearrangement& r) const {
        size_t h = std::hash<std::string>()(r.chr_1) * 1000003u ^ std::hash<std::string>()(r.chr_2);
        h = h * 1000003u ^ (size_t)r.pos_1;
        h = h * 1000003u ^ (size_t)r.pos_2;
        return h * 4u + (r.strand_1 == '+' ? 2u : 0u) + (r.strand_2 == '+' ? 1u : 0u);
    }
};
struct Evidence { std::string sample; size_t reads; };          // filter.rs:18-21
struct Sample { std::string name, patient; bool germline; };    // filter.rs:33-37

// filter.rs:139-146; the reference panics on a position that does not parse or an empty strand column
Rearrangement read_rearrangement(const std::vector<std::string_view>& cols, const std::string& line) {
    Rearrangement r;
    if (cols.size() < 7) error("Rearrangement line has too few columns (the reference panics):\n%s\n", line.c_str());
    if (!parse_uint(cols[2], UINT64_MAX, r.pos_1) || !parse_uint(cols[6], UINT64_MAX, r.pos_2))
        error("Rearrangement position is not a number (the reference panics):\n%s\n", line.c_str());
    if (cols[1].empty() || cols[5].empty())
        error("Rearrangement strand is empty (the reference panics):\n%s\n", line.c_str());
    // `chars().nth(0)`: the strand columns detect writes are '+' or '-' (one byte)
    r.chr_1 = std::string(cols[0]); r.strand_1 = cols[1][0];
    r.chr_2 = std::string(cols[4]); r.strand_2 = cols[5][0];
    return r;
}

// filter.rs:148-156.  sort_by_key / sort are stable; dedup_by_key keeps the first of a run.
void deduplicate_reads(std::vector<std::string_view>& reads, const std::string& line) {
    for (auto r : reads)
        if (r.size() < 20) error("Supporting read shorter than 20 bases (the reference panics):\n%s\n", line.c_str());
    auto last20 = [](std::string_view r) { return r.substr(r.size() - 20); };
    auto first20 = [](std::string_view r) { return r.substr(0, 20); };
    std::stable_sort(reads.begin(), reads.end(), [&](std::string_view a, std::string_view b) { return last20(a) < last20(b); });
    reads.erase(std::unique(reads.begin(), reads.end(), [&](std::string_view a, std::string_view b) { return last20(a) == last20(b); }),
                reads.end());
    std::stable_sort(reads.begin(), reads.end());
    reads.erase(std::unique(reads.begin(), reads.end(), [&](std::string_view a, std::string_view b) { return first20(a) == first20(b); }),
                reads.end());
}

std::string lower_ascii(std::string_view s) {
    std::string o(s);
    for (char& c : o) if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
    return o;
}

void mkdir_p(const std::string& path) {                  // fs::create_dir_all, result ignored (filter.rs:104)
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty()) (void)mkdir(cur.c_str(), 0777);
        }
        if (i < path.size()) cur += path[i];
    }
}

}  // namespace

int filter_main(int argc, char** argv) {
    const Options o = parse_options(argc, argv, 2, {"--min-reads", "--min-distance", "--output-dir"}, {"--keep-duplicates"}, FILTER_USAGE);
    if (o.positional.size() != 1) error("Invalid arguments.\n%s", FILTER_USAGE);
    auto opt = [&](const char* name, const char* dflt) {
        auto it = o.values.find(name);
        return it == o.values.end() ? std::string(dflt) : it->second;
    };
    uint64_t min_reads, min_distance;
    if (!parse_uint(opt("--min-reads", "4"), UINT64_MAX, min_reads)) error("--min-reads must be a positive integer.");   // filter.rs:41
    if (!parse_uint(opt("--min-distance", "10"), UINT64_MAX, min_distance))
        error("--min-distance must be a non-negative integer (the reference panics).");                                  // filter.rs:42
    const bool merge_duplicates = !o.flags.count("--keep-duplicates");
    const std::string output_dir = opt("--output-dir", "annotated");

    fprintf(stderr, "Output files will be written in directory %s\n", output_dir.c_str());
    fprintf(stderr, "Reading sample sheet into memory...\n");
    std::vector<Sample> samples;
    std::string line;
    {
        FileReader sheet(o.positional[0]);
        sheet.read_line(line);
        const auto headers = split(line, '\t');
        auto col_of = [&](const char* h) {
            for (size_t i = 0; i < headers.size(); i++) if (headers[i] == h) return i;
            return (size_t)99;
        };
        const size_t patient_col = col_of("PATIENT"), sample_col = col_of("SAMPLE"), type_col = col_of("TYPE");
        if (patient_col == 99 || sample_col == 99 || type_col == 99)
            error("Required headers PATIENT, SAMPLE and TYPE were not all found on the first row of the sample sheet.");
        const size_t min_columns = std::max(patient_col, std::max(sample_col, type_col));
        while (sheet.read_line(line)) {
            if (trim(line).empty()) continue;
            const auto cols = split(line, '\t');
            // filter.rs:66 tests `< min_columns` and then indexes column min_columns: a line with exactly
            // min_columns columns panics there; both cases are the same error here
            if (cols.size() <= min_columns) error("Sample sheet line has too few columns:\n%s\n", line.c_str());
            const std::string_view patient = trim(cols[patient_col]), sample = trim(cols[sample_col]);
            const std::string sample_type = lower_ascii(trim(cols[type_col]));
            if (patient.empty() || sample.empty() || sample_type.empty())
                error("Missing PATIENT, SAMPLE or TYPE column on line:\n%s\n", line.c_str());
            const bool germline = sample_type == "wbc" || sample_type == "gdna" || sample_type == "germline";
            samples.push_back(Sample{std::string(sample), std::string(patient), germline});
        }
    }

    fprintf(stderr, "Reading rearrangements into memory...\n");
    std::unordered_map<Rearrangement, std::vector<Evidence>, RearrangementHash> rearrangements;
    for (const Sample& sample : samples) {
        FileReader sv(sample.name + ".sv");
        sv.read_line(line);                                       // discard the header line
        while (sv.read_line(line)) {
            const auto cols = split(line, '\t');
            if (cols.size() < 9) error("File %s.sv has a line with too few columns:\n%s\n", sample.name.c_str(), line.c_str());
            const Rearrangement r = read_rearrangement(cols, line);
            auto reads = split(cols[8], ';');
            if (merge_duplicates) deduplicate_reads(reads, line);
            rearrangements[r].push_back(Evidence{sample.name, reads.size()});
        }
    }

    fprintf(stderr, "Generating output files:\n");
    mkdir_p(output_dir);
    for (const Sample& sample : samples) {
        std::unordered_set<std::string> negative_controls;
        for (const Sample& s : samples)
            if (s.patient != sample.patient || (!sample.germline && s.germline)) negative_controls.insert(s.name);
        if (negative_controls.empty()) error("No negative controls found for sample %s.", sample.name.c_str());

        const std::string out_path = output_dir + "/" + sample.name + ".sv";
        FILE* out = fopen(out_path.c_str(), "wb");
        if (!out) error("Cannot create file %s.", out_path.c_str());
        FileReader sv(sample.name + ".sv");
        sv.read_line(line);
        fwrite(line.data(), 1, line.size(), out); fputc('\n', out);
        while (sv.read_line(line)) {
            const auto cols = split(line, '\t');
            const Rearrangement r = read_rearrangement(cols, line);
            const uint64_t gap = r.pos_1 > r.pos_2 ? r.pos_1 - r.pos_2 : r.pos_2 - r.pos_1;
            if (r.chr_1 == r.chr_2 && r.strand_1 == r.strand_2 && gap < min_distance) continue;   // too adjacent
            if (cols.size() < 9) error("File %s.sv has a line with too few columns:\n%s\n", sample.name.c_str(), line.c_str());
            auto reads = split(cols[8], ';');
            if (merge_duplicates) deduplicate_reads(reads, line);
            if (reads.size() < min_reads) continue;
            size_t max_control_reads = 0;
            auto it = rearrangements.find(r);
            if (it != rearrangements.end())
                for (const Evidence& e : it->second)
                    if (negative_controls.count(e.sample)) max_control_reads = std::max(max_control_reads, e.reads);
            if (reads.size() < 50 * max_control_reads) continue;
            std::string row(cols[0]);
            for (int k = 1; k < 8; k++) { row += '\t'; row += cols[k]; }
            row += '\t';
            for (size_t k = 0; k < reads.size(); k++) { if (k) row += ';'; row += reads[k]; }
            for (size_t k = 9; k < cols.size(); k++) { row += '\t'; row += cols[k]; }
            row += '\n';
            fwrite(row.data(), 1, row.size(), out);
        }
        if (fclose(out) != 0) error("Cannot create file %s.", out_path.c_str());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// annotate

namespace {

const char* ANNOTATE_USAGE = "\nUsage:\n  breakfast annotate [options] <sv_path> <bed_path>\n";

struct Feature { uint32_t start, end; std::string name; };       // annotate.rs:19-24 (chr is the map key)

uint32_t distance(uint32_t pos, const Feature& f) {              // annotate.rs:9-17
    if (pos < f.start) return f.start - pos;
    if (pos > f.end) return pos - f.end;
    return 0;
}

}  // namespace

int annotate_main(int argc, char** argv) {
    const Options o = parse_options(argc, argv, 2, {}, {}, ANNOTATE_USAGE);
    if (o.positional.size() != 2) error("Invalid arguments.\n%s", ANNOTATE_USAGE);
    FileReader sv(o.positional[0]);
    std::string line;
    sv.read_line(line);
    std::string out = line + "\n";                               // annotate.rs:33-34

    // features per contig, in BED order (the sort below is stable on distance, annotate.rs:69)
    std::unordered_map<std::string, std::vector<Feature>> features;
    {
        FileReader bed(o.positional[1]);
        while (bed.read_line(line)) {
            const auto cols = split(line, '\t');
            if (cols.size() < 4) error("Feature BED file has too few columns on this line:\n%s\n", line.c_str());
            uint64_t s, e;
            if (!parse_uint(cols[1], 0xfffffffeull, s) || !parse_uint(cols[2], 0xffffffffull, e))
                error("Feature BED coordinates are not 32-bit numbers (the reference panics):\n%s\n", line.c_str());
            features[std::string(cols[0])].push_back(Feature{(uint32_t)s + 1, (uint32_t)e, std::string(cols[3])});
        }
    }
    auto nearby = [&](std::string_view chr, uint32_t pos, std::string& dst) {
        auto it = features.find(std::string(chr));
        if (it == features.end()) return;
        std::vector<std::pair<uint32_t, const Feature*>> near;
        for (const Feature& f : it->second) {
            const uint32_t d = distance(pos, f);
            if (d > 100000) continue;
            near.emplace_back(d, &f);
        }
        std::stable_sort(near.begin(), near.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (size_t k = 0; k < near.size(); k++) {
            dst += near[k].second->name; dst += " ("; dst += std::to_string(near[k].first); dst += ')';
            if (k + 1 < near.size()) dst += ", ";
        }
    };
    while (sv.read_line(line)) {
        if (line.compare(0, 3, "chr") != 0) continue;            // annotate.rs:53
        const auto cols = split(line, '\t');
        if (cols.size() < 11) error("Rearrangement line has fewer than 11 columns (the reference panics):\n%s\n", line.c_str());
        uint64_t p1, p2;
        if (!parse_uint(cols[2], 0xffffffffull, p1) || !parse_uint(cols[6], 0xffffffffull, p2))
            error("Rearrangement position is not a 32-bit number (the reference panics):\n%s\n", line.c_str());
        out += cols[0]; out += '\t'; out += cols[1]; out += '\t'; out += std::to_string(p1); out += '\t';
        nearby(cols[0], (uint32_t)p1, out);
        out += '\t';
        out += cols[4]; out += '\t'; out += cols[5]; out += '\t'; out += std::to_string(p2); out += '\t';
        nearby(cols[4], (uint32_t)p2, out);
        out += '\t'; out += cols[8]; out += '\t'; out += cols[9]; out += '\t'; out += cols[10]; out += '\n';
        if (out.size() > (1u << 20)) { fwrite(out.data(), 1, out.size(), stdout); out.clear(); }
    }
    fwrite(out.data(), 1, out.size(), stdout);
    fflush(stdout);
    return 0;
}
