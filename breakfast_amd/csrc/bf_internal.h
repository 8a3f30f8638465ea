// Internal declarations shared by the HIP translation units of libbreakfast_hip.so.
// Device data layout is documented in DESIGN.md ("Data layout in HBM").
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <thread>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/breakfast_hip.h"

// Wire format parameters of the evidence exchange that all ranks must agree on (the maximum over the ranks of each field)
struct bf_wire_local {
    uint32_t max_frag_len_bytes;  // longest fragment id on this rank
    uint32_t has_exact_sig;       // 1 if any evidence carries a non-ACGT signature
};

namespace bf {

// ---- genome image ---------------------------------------------------------------------------
// One 16-byte block per 64 bases of the concatenated genome: bit k of `lo`/`hi` = low/high bit of
// the 2-bit code of base 64*block + k, code = (ascii >> 1) & 3 (A=0 C=1 T=2 G=3, so complement is
// code ^ 2).  Bases that are not one of "ACGT" (case-sensitive) get an arbitrary code and set the
// block's bit in the coarse exception bitmap; such windows are routed to the byte-exact kernel,
// which reads the raw image.  Every contig starts on a block boundary.
struct GenomeBlock { uint64_t lo, hi; };

// The block image is stored as OVERLAPPING 128-byte lines: line j holds blocks 4j .. 4j+7, so every
// block lives twice (in line b/4 and, as its upper half, in line b/4 - 1) and any run of up to 5
// consecutive blocks -- a flank window of up to 256 bases at any offset -- lies inside ONE line.
// A window gather therefore costs one 128-byte memory request instead of 1.375 on average
// (a 64-byte run at 16-byte granularity straddles a line boundary 3 times in 8).  Twice the 2-bit
// image (1.5 GB for hg38) is nothing against 288 GB of HBM.
__host__ __device__ inline uint64_t block_slot(uint64_t b) { return ((b >> 2) << 3) + (b & 3); }
__host__ __device__ inline uint64_t block_image_slots(uint64_t n_blocks) { return ((n_blocks >> 2) + 2) << 3; }

// Exception plane: one 64-bit word per 64-base block, bit k = "base k is not one of ACGT" (N, soft-masked
// lowercase, IUPAC).  Stored the same way as overlapping 128-byte lines (line j = words 8j .. 8j+15), so
// the up to five consecutive words of a window are one memory request.  Only windows that the coarse
// bitmaps flag ever read it.
__host__ __device__ inline uint64_t exc_slot(uint64_t b) { return ((b >> 3) << 4) + (b & 7); }
__host__ __device__ inline uint64_t exc_image_slots(uint64_t n_blocks) { return ((n_blocks >> 3) + 2) << 4; }

// The reference keeps every byte of the FASTA (detect.rs:48-56) and compares bytes (detect.rs:137-145, :223).  The raw
// image (one byte per base: 3.1 GB of hg38's 5.4 GB resident) is gone after the pack: a byte is rebuilt from its 2-bit
// code ((byte >> 1) & 3: A C T G = 0 1 2 3 in either case), its exception bit, a plane "one of acgt", and for every
// other byte (N, n, IUPAC, anything) a table of the runs of equal such bytes, sorted by start, searched by bisection
// (hg38: a few hundred runs of N).  Only the byte-exact kernel and signatures with a byte outside ACGT ask.
struct GenomeBytes {
    const GenomeBlock* blocks;
    const uint64_t* exc;         // overlapping lines, exc_slot()
    const uint64_t* lower;       // one word per 64-base block, bit k = base k is one of "acgt"
    const uint64_t* run_start;   // global base offsets, ascending
    const uint8_t* run_byte;
    uint32_t n_runs;
#ifdef __HIPCC__
    __device__ __forceinline__ uint8_t at(uint64_t g) const {
        const uint64_t b = g >> 6;
        const int k = (int)(g & 63);
        const GenomeBlock w = blocks[block_slot(b)];
        const uint32_t code = (uint32_t)((w.lo >> k) & 1) | ((uint32_t)((w.hi >> k) & 1) << 1);
        const uint32_t plain = 0x47544341u >> (8 * code);                 // "ACTG"
        if (!((exc[exc_slot(b)] >> k) & 1)) return (uint8_t)plain;
        if ((lower[b] >> k) & 1) return (uint8_t)(plain | 0x20);
        uint32_t lo_ = 0, hi_ = n_runs;                                   // last run that starts at or before g
        while (lo_ < hi_) {
            const uint32_t mid = (lo_ + hi_) >> 1;
            if (run_start[mid] <= g) lo_ = mid + 1; else hi_ = mid;
        }
        return lo_ ? run_byte[lo_ - 1] : (uint8_t)'N';
    }
#endif
};

struct ContigDev {      // 16 B, indexed by contig id
    uint64_t goff;      // global base offset of the contig's first base (multiple of 64)
    uint32_t len;
    uint32_t rank;      // rank of the contig name in byte-wise order (Rust String Ord)
};

constexpr int kContigBits = 14;                 // <= 16384 contigs
constexpr int kKeySigBits = 32;
constexpr int kKeySigxBit = 32;                 // signature holds a non-ACGT byte (sig = hash32 of raw)
constexpr int kKeyMstrandBit = 33;
constexpr int kKeyStrandBit = 34;
constexpr int kKeyMrankShift = 35;
// The two contig ranks sit above bit 35 and take `cb` bits each, cb = ceil(log2(n_contigs)) chosen
// per genome (5 for the 25 hg38 contigs), so the radix sort only walks bits that can differ.
__host__ __device__ inline int key_rank_shift(int cb) { return kKeyMrankShift + cb; }
__host__ __device__ inline int key_bits(int cb) { return kKeyMrankShift + 2 * cb; }

// evidence flags (low bits mirror BF_EV_*)
constexpr uint32_t kEvStrand = 1, kEvMstrand = 2, kEvSwapped = 4, kEvExact = 8, kEvSigx = 16;

// ---- junction records (include/breakfast_hip.h, "Junction records") ------------------------------
// A surviving read's junction string crosses PCIe (and xGMI on the sharded path) as a self-describing
// record: 2-3 bits per base instead of 8 when the read is (almost) plain ACGT.  `max_packed` = 64 * W of the
// ctx's localise kernel; `n_exc` = read bases outside "ACGT" (saturating u16 from the render record).
// Header word (little endian u32): bits 0-1 kind (0 text, 1 dense, 2 sparse), bit 2 "the read was reverse-complemented"
// (detect.rs:112); packed kinds: L in bits 3-11, bp in 12-20, n_exc in 21-26, listed lower-case positions in 27-31;
// text: bp in bits 8-23 and L in a second word.
constexpr uint32_t kJrText = 0u, kJrDense = 1u, kJrSparse = 2u, kJrRef = 3u;
__host__ __device__ inline bool junction_packable(uint32_t L, uint32_t n_exc, uint32_t max_packed) {
    return L <= max_packed && L <= 256u && 8u * n_exc <= L;
}
// Packed records come in two forms: with a lower-case (mismatch) bit per base, or -- most reads differ from
// their flanks in a base or two -- with the lower-case positions listed (a byte each) after the code bits.
__host__ __device__ inline bool junction_sparse(uint32_t L, uint32_t n_lower) {
    return n_lower <= 31u && n_lower < ((L + 7u) >> 3);
}
__host__ __device__ inline uint32_t junction_head_packed(bool sparse, bool swapped, uint32_t L, uint32_t bp, uint32_t n_exc, uint32_t n_lower) {
    return (sparse ? kJrSparse : kJrDense) | (swapped ? 4u : 0u) | (L << 3) | (bp << 12) | (n_exc << 21) | ((sparse ? n_lower : 0u) << 27);
}
__host__ __device__ inline uint32_t junction_head_text(bool swapped, uint32_t bp) { return kJrText | (swapped ? 4u : 0u) | ((bp & 0xffffu) << 8); }
__host__ __device__ inline uint32_t junction_record_bytes(uint32_t L, uint32_t n_exc, uint32_t n_lower, uint32_t max_packed) {
    if (!junction_packable(L, n_exc, max_packed)) return (8u + L + 1u + 3u) & ~3u;
    const uint32_t u = (L + 7u) >> 3;
    return ((junction_sparse(L, n_lower) ? 4u + 2u * u + 2u * n_exc + n_lower : 4u + 3u * u + 2u * n_exc) + 3u) & ~3u;
}
// Reference records (BF_FLAG_READS_ON_HOST): the bases stay in the caller's host arrays, the record names the read.
// Header: kind 3, bit 2 swapped, bit 3 wide (u16 positions), bits 4-15 n_lower (kJrRefDense: a bit per base), bits
// 16-31 bp; then the read's number (u32), then the lower-case positions (or u32 L + ceil(L/8) mask bytes).
constexpr uint32_t kJrRefDense = 4095u;
__host__ __device__ inline uint32_t junction_ref_bytes(uint32_t L, uint32_t n_lower) {
    if (n_lower >= kJrRefDense) return (8u + 4u + ((L + 7u) >> 3) + 3u) & ~3u;
    return (8u + n_lower * (L > 256u ? 2u : 1u) + 3u) & ~3u;
}
__host__ __device__ inline uint32_t junction_head_ref(bool swapped, uint32_t L, uint32_t n_lower, uint32_t bp) {
    return kJrRef | (swapped ? 4u : 0u) | (L > 256u ? 8u : 0u) | ((n_lower >= kJrRefDense ? kJrRefDense : n_lower) << 4) | ((bp & 0xffffu) << 16);
}
// What the cluster stage gathers per evidence record by sort permutation, in one 8-byte word instead of four
// arrays (a random gather costs a whole line whatever it uses): junction record bytes (20 bits), evidence
// flags (5 bits), breakpoint offset and read length.
__host__ __device__ inline uint64_t read_info(uint32_t jrec, uint32_t flags, uint32_t bp, uint32_t len) {
    return (uint64_t)(jrec & 0xfffffu) | ((uint64_t)(flags & 0x1fu) << 20) | ((uint64_t)(bp & 0xffffu) << 32) | ((uint64_t)(len & 0xffffu) << 48);
}
__host__ __device__ inline uint32_t info_jrec(uint64_t i) { return (uint32_t)i & 0xfffffu; }
__host__ __device__ inline uint32_t info_flags(uint64_t i) { return (uint32_t)(i >> 20) & 0x1fu; }
__host__ __device__ inline uint32_t info_bp(uint64_t i) { return (uint32_t)(i >> 32) & 0xffffu; }
__host__ __device__ inline uint32_t info_len(uint64_t i) { return (uint32_t)(i >> 48); }
// ... and that word sits beside the record's 64-bit fragment-id hash, 16 aligned bytes: the wave that enumerates a
// cluster's members (fill_members_kernel) needs both of every member, and a gather through the sort permutation costs
// a 128-byte line whatever it uses -- one line per member instead of two (round 3: 2.26 -> 1.2 GB per step).
struct alignas(16) RecWord { uint64_t info, fhash; };

// ---- wire record of the multi-GPU exchange ---------------------------------------------------
struct WireHeader {     // 40 B, followed by frag bytes (fixed stride) and optionally 16 raw sig bytes
    uint64_t key;
    int32_t lpos, rpos;
    uint64_t fhash;
    uint32_t src_idx;
    uint16_t bp, len;
    uint16_t frag_len, flags;
    uint32_t jrec;       // bytes of the read's junction record: the owner lays out its result without asking
};
static_assert(sizeof(WireHeader) == 40, "wire header layout");

// ---- a few host threads that stay (bf_submit's checks and staging copies) ----------------------
// run(n, f): f(0) on the caller, f(1) .. f(n-1) on pool threads; returns when all are done.  One run at a time.
class HostPool {
public:
    ~HostPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run(int n, const std::function<void(int)>& f) {
        if (n <= 1) { f(0); return; }
        {
            std::lock_guard<std::mutex> g(mu_);
            while ((int)th_.size() < n - 1) { const int k = (int)th_.size() + 1; th_.emplace_back([this, k] { work(k); }); }
            fn_ = &f; n_ = n; pending_ = n - 1; gen_++;
        }
        cv_.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    void work(int k) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* f = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (k < n_) f = fn_;
            }
            if (!f) continue;
            (*f)(k);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// ---- device buffer ---------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return (T*)p; }
};

struct KernelTimer;

// Owner-side evidence view: what sort/sweep/dedup consume.  With one rank it aliases the
// source-side arrays produced by localise; with several it points at imported records.
struct OwnerView {
    uint64_t n = 0;
    const uint64_t* key = nullptr;
    const int32_t* lpos = nullptr;
    const int32_t* rpos = nullptr;
    const uint16_t* bp = nullptr;
    const uint16_t* len = nullptr;
    const uint16_t* flags = nullptr;
    const uint8_t* frag = nullptr;       // fragment id bytes
    const uint64_t* frag_off = nullptr;  // byte offset of record i's fragment id
    const uint16_t* frag_len = nullptr;
    const uint8_t* sigx = nullptr;       // 16 raw signature bytes per slot
    const uint32_t* sigx_slot = nullptr; // record -> slot (only read when the key's sigx bit is set)
    const uint64_t* src = nullptr;       // imported records: source rank << 32 | index on that rank, ONE word (emit gathers it
                                         // per supporting read); nullptr = all from this rank, record i is read i
    const RecWord* rw = nullptr;         // per record, ONE 16-byte load for the gathers of the dedup: read_info() word + fragment-id hash
};

struct HostBuf {                         // pinned host staging (hipHostMalloc): full-rate D2H copies
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return (T*)p; }
};

struct ClusterOut {                      // host copies of the cluster stage's results
    uint64_t n_clusters = 0, n_reads = 0, junction_size = 0;
    HostBuf clusters;                    // bf_cluster[n_clusters]
    HostBuf reads;                       // bf_read[n_reads]
    HostBuf src;                         // uint32 src_rank[n_reads], src_idx[n_reads] (multi-rank only)
    HostBuf junction;                    // junction records
    HostBuf cjo;                         // uint64 cluster_junction_off[n_clusters + 1] (single rank)
};

struct Timing {
    std::vector<std::string> names;
    std::vector<float> ms;
    std::vector<uint64_t> launches;
    std::vector<bf_kernel_time> view;
    void add(const char* name, float ms_);
    void clear();
};

}  // namespace bf

struct bf_ctx {
    bf_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;          // small kernels that need not wait in line (print-order sorts beside the dedup)
    hipStream_t copy_stream = nullptr;          // result D2H, overlapped with the render kernel
    hipEvent_t ev[14] = {};                     // untimed events chaining the two streams
    std::string err;

    // genome
    uint32_t n_contigs = 0;
    int contig_bits = 1;                        // bits per contig rank in the class key
    std::vector<std::string> contig_names;
    std::vector<uint32_t> contig_rank;          // by id
    std::vector<uint32_t> contig_by_rank;       // rank -> id
    std::vector<bf::ContigDev> contig_host;
    uint64_t genome_bases = 0;                  // padded total (multiple of 64)
    bf::DevBuf g_blocks, g_exc, g_raw, g_coarse, g_coarse2, g_contigs;     // g_raw: only between bf_genome_begin and bf_genome_end
    bf::DevBuf g_lower, g_run_start, g_run_byte;
    uint32_t n_runs = 0;
    // hand-over as a packed image (bf_genome_set_image): g_raw stages the plain blocks, the runs come as they are
    bool genome_image = false;
    bf::DevBuf g_lower_start, g_lower_len, g_run_len;         // released at bf_genome_end
    uint64_t n_lower_runs = 0;

    // host -> device hand-over (bf_submit): uploads run on in_stream from pinned memory -- the caller's own when it
    // came from bf_host_alloc, else a ring of pinned staging slots filled by a few host threads -- and the reads
    // of every uploaded piece are localised behind it on `stream`, while the next piece is still on PCIe
    hipStream_t in_stream = nullptr;
    hipEvent_t ev_in = nullptr;                 // recorded on in_stream after the uploads queued so far
    bool in_pending = false;                    // ... and `stream` has not been told to wait for it yet
    struct StageSlot { uint8_t* p = nullptr; hipEvent_t ev = nullptr; bool used = false; };
    std::vector<StageSlot> slots;               // one per host thread, kStageSlotBytes each (allocated on first use)
    // 4-bit batches: the offsets are not uploaded, the device rebuilds them from the lengths (a read takes ceil(len / 2)
    // bytes, fragment ids lie back to back): reads [r0, r1) start at these byte offsets; done before anything reads them
    // (two_bit: the piece came as 2-bit codes -- bf_submit_packed2 --, its bytes lie at d_seq2 + seq2_base and are expanded
    //  to 4-bit codes at seq_base once the offsets are there; exceptions [exc0, exc1) of the resident list are its own)
    struct OffsetPiece { uint64_t r0, r1, seq_base, frag_base; bool two_bit = false; uint64_t seq2_base = 0, exc0 = 0, exc1 = 0; };
    std::vector<OffsetPiece> off_pending;
    struct WideSlot { uint8_t* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
    WideSlot wide[2];                           // piece-wide offset staging for batches in bf_host_alloc memory, used in turn
    uint32_t wide_next = 0;
    std::vector<std::pair<const uint8_t*, size_t>> pinned;   // bf_host_alloc ranges
    // BF_FLAG_READS_ON_HOST: where the caller keeps the bases of resident reads [first, first + n)
    struct HostReads { uint64_t first, n; const uint8_t* seq; const uint64_t* seq_off; const uint16_t* seq_len; };
    std::vector<HostReads> host_reads;
    std::mutex pinned_mu;                       // bf_host_alloc / bf_host_free may come from any thread, beside other calls
    uint32_t host_threads = 0;                  // bf_config.reserved[1], or chosen from the machine
    bf::HostPool* pool = nullptr;               // bf_submit's helpers: threads that stay (a submit of 70k reads is 0.2 ms of work,
                                                // starting and joining three threads for it took as long again)
    uint64_t cap_reads = 0;                     // capacity of every per-read array below (inputs and evidence)
    uint64_t n_localised = 0;                   // reads [0, n_localised) have evidence under (plane_words, mask_words)
    std::vector<uint64_t> rank_hist;            // resident reads per left contig rank after the reorient step (detect.rs:108-113:
                                                // the smaller of the two anchors' ranks), counted by bf_submit's staging threads:
                                                // the cluster stage cuts its contig groups from it without asking the GPU
    bool consumed = false;                      // a bf_finish has clustered exactly this read set: the next one without
                                                // new reads runs the whole path again (bf_finish is idempotent)

    // genome hand-over in pieces (bf_genome_begin / bf_genome_write / bf_genome_end): any thread may write
    bool genome_ready = false;                  // bf_genome_end (or bf_genome_set) has packed the image: kernels may read it
    struct GenomeSlot { uint8_t* p = nullptr; hipEvent_t ev = nullptr; bool used = false; std::mutex mu; };
    std::vector<GenomeSlot> gslots;
    std::atomic<uint32_t> gslot_next{0};
    std::mutex genome_mu;                       // guards genome_err
    std::string genome_err; int genome_err_code = 0;

    // source-side resident reads
    bool packed_reads = false;                  // d_seq holds 4-bit base codes (bf_submit_packed), d_len came with them
    uint64_t n_reads = 0;
    uint64_t seq_bytes = 0, frag_bytes = 0;
    uint64_t seq2_bytes = 0, n_exc = 0;          // bf_submit_packed2: 2-bit bytes uploaded since bf_reset (kept until then: a
                                                // piece is expanded by the stream that localises it), exceptions resident
    bf::DevBuf d_seq2, d_exc_read, d_exc_pos, d_exc_code;
    uint32_t max_len = 0, min_len = 0xffffffffu, max_frag = 0;
    bf::DevBuf d_seq, d_seq_off, d_frag, d_frag_off, d_hits;
    // derived by localise
    bool localised = false;
    uint32_t plane_words = 0, mask_words = 0;
    uint64_t n_exact = 0;
    uint64_t n_sigx = 0;                        // evidence whose signature holds a non-ACGT byte
    uint64_t n_text_records = 0;                // resident reads whose junction record cannot be packed
    uint64_t* h_mail = nullptr;                 // pinned (behind h_counts): kMailWords 64-bit words the cluster stage's counts come
                                                // back in -- several device words queued with mail_fetch(), ONE wait for all of them
                                                // (a copy into pageable memory is a round trip of its own, ~25 us of idle GPU each)
    std::vector<uint64_t> tables_key;           // what c_tables holds (index bases + contig count): uploaded when it changes
    std::vector<std::pair<uint32_t, uint32_t>> group_ranks;   // cluster_groups: first and last left contig rank of every group
    bool order_full = false;                    // the short print-order sort met something it does not handle (long tie runs, a
                                                // position outside its mapped range): the ctx uses the full two-sort order from then on
    bool saw_overflow = false;                  // a chunk too large for the LDS sort has come up: class_order waits for the count again
    bf::HostBuf h_wide;                         // pinned, grown on demand (wide_words): longer read-backs that ride on a wait the
                                                // path makes anyway -- the imported view's contig histogram, per-rank words
    uint32_t* h_counts = nullptr;               // pinned (bf_ctx_create): words 0-2 the localise kernels' counters on their way
                                                // to the three fields above, words 8-11 the read-backs of bf_render_junctions_dev_static
    void* scratch = nullptr;                    // this device's copy of the static scratch array (localise.hip, scratch_setup)
    void* shard = nullptr;                      // bf::Shard (shard.hip): transport and buffers of the in-library sharded path
    bool counts_pending = false;                // ... not yet read back: settle_counts() after the next synchronisation
    bf::DevBuf d_len, d_frag_len, d_fhash;
    bf::DevBuf d_key, d_lpos, d_rpos, d_bp, d_flags, d_mask, d_jrec;   // d_jrec: RecWord per read (read_info() word + d_fhash's hash)
    bf::DevBuf d_exact_list, d_counters, d_sigx, d_sigx_slot;

    // multi-rank
    uint32_t rank = 0, n_ranks = 1;
    uint64_t index_base = 0;
    std::vector<uint64_t> src_index_base;
    bool imported = false;
    bool clustered = false;                     // ctx->out / c_rsrc describe the last bf_clusterise
    uint32_t wire_frag_stride = 0, wire_has_sigx = 0, wire_rec_bytes = 0;
    uint64_t n_imported = 0;
    bf::DevBuf d_send, d_part_key, d_part_idx, d_part_tmp;
    bf::DevBuf i_key, i_lpos, i_rpos, i_bp, i_len, i_flags, i_frag, i_frag_off, i_frag_len, i_sigx,
        i_sigx_slot, i_src, i_jrec;
    bool reads_on_device = false;               // the finished read table waits in c_reads for bf_result_reads
    uint64_t rsrc_stride = 0;                   // c_rsrc = 4 arrays of this many entries (source rank, source index, record, bytes)

    // cluster stage scratch + results
    bf::DevBuf c_tmp, c_a32, c_b32, c_c32, c_d32, c_k64a, c_k64b, c_slpos, c_srpos, c_reported,
        c_csize, c_flag, c_scan, c_qseed, c_qsize, c_qoff, c_mrec, c_mjrec, c_leader, c_rep, c_qnrep,
        c_ctl, c_sel, c_ord, c_clusters, c_reads, c_misc, c_pos2, c_joff, c_jbytes, c_rsrc, c_tables, c_mkey_s, c_slot, c_big, c_cjo, c_aoff, c_err, c_tmp2, c_qord, c_gidx, c_hist, c_comp, c_bstart, c_chunk;
    bf::ClusterOut out;
    std::vector<uint64_t> render_off_host;
    std::vector<uint8_t> render_bytes_host;
    uint64_t n_evidence = 0;

    // evidence fetch staging
    std::vector<uint64_t> ev_key, ev_mask;
    std::vector<int32_t> ev_lpos, ev_rpos;
    std::vector<uint16_t> ev_bp, ev_flags;

    bf::Timing timing;
    void* timing_pending = nullptr;             // std::vector<PendingEvent>* (api.hip)
    bf::DevBuf r_idx, r_off, r_bytes;           // bf_render_junctions staging
    std::atomic<uint64_t> device_bytes{0};       // (bf_genome_write_image allocates beside bf_submit)
};

namespace bf {

// error helpers ---------------------------------------------------------------------------------
int fail(bf_ctx* ctx, int code, const char* fmt, ...);
#define BF_HIP(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return bf::fail(ctx, BF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                               \
    } while (0)
#define BF_TRY(expr)                 \
    do {                             \
        int r_ = (expr);             \
        if (r_ != BF_OK) return r_;  \
    } while (0)

// BREAKFAST_TRACE=1: every C-ABI call and every host wait for the GPU writes "enter"/"leave" and "wait"/"done"
// lines (pid, rank, seconds, call or stream, source line) to stderr, so that a hang names the wait that never
// returned -- with several ranks sharing one GPU the process that is stuck is not the one whose collective times out.
bool trace_on();
void trace_line(const bf_ctx* ctx, const char* what, const char* name, const char* file, int line);
int traced_sync(bf_ctx* ctx, hipStream_t s, const char* name, const char* file, int line);
#define BF_SYNC(ctx, s) BF_TRY(bf::traced_sync(ctx, s, #s, __FILE__, __LINE__))
struct TraceCall {
    const bf_ctx* ctx; const char* fn;
    TraceCall(const bf_ctx* c, const char* f) : ctx(c), fn(f) { if (trace_on()) trace_line(ctx, "enter", fn, "", 0); }
    ~TraceCall() { if (trace_on()) trace_line(ctx, "leave", fn, "", 0); }
};
#define BF_TRACE_CALL(ctx) bf::TraceCall trace_call_(ctx, __func__)

constexpr int kMailWords = 32;
constexpr int kMailRecordBytes = 6;    // cluster_groups leaves the sum of all records' junction record bytes here
// queues the copy of n 64-bit device words into mailbox slots [slot, slot + n); valid after the next wait for ctx->stream
inline int mail_fetch(bf_ctx* ctx, int slot, const void* d_src, int n = 1) {
    BF_HIP(ctx, hipMemcpyAsync(ctx->h_mail + slot, d_src, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    return BF_OK;
}
int ensure(bf_ctx* ctx, DevBuf& b, size_t bytes, bool keep = false);
// n 64-bit words of pinned host memory for a read-back (valid until the next call with a larger n: ask before queueing copies)
int wide_words(bf_ctx* ctx, size_t n, uint64_t** out);
int ensure_host(bf_ctx* ctx, HostBuf& b, size_t bytes);
void release(bf_ctx* ctx, DevBuf& b);

// timed region helper: records HIP events on the ctx stream around a group of launches
struct Scope {
    bf_ctx* ctx; const char* name; hipEvent_t a = nullptr, b = nullptr;
    Scope(bf_ctx* c, const char* n);
    ~Scope();
};

// Host code is single-threaded per ctx: while this object lives, everything queued through ctx->stream (and the
// sort / scan workspace c_tmp) goes to the side stream with a workspace of its own instead.
struct OnSideStream {
    bf_ctx* ctx;
    explicit OnSideStream(bf_ctx* c) : ctx(c) { std::swap(ctx->stream, ctx->side_stream); std::swap(ctx->c_tmp, ctx->c_tmp2); }
    ~OnSideStream() { std::swap(ctx->stream, ctx->side_stream); std::swap(ctx->c_tmp, ctx->c_tmp2); }
};

// localise.hip ----------------------------------------------------------------------------------
int genome_unpack(bf_ctx* ctx, uint64_t g0, uint64_t n, uint8_t* dev_out);    // the bytes of global offsets [g0, g0 + n)
int genome_pack(bf_ctx* ctx);
int genome_from_image(bf_ctx* ctx);                 // plain blocks + runs -> the same resident images
int genome_begin(bf_ctx* ctx, uint32_t n_contigs, const char* const* names, const uint64_t* lens, bool image);
int genome_stage(bf_ctx* ctx, uint8_t* dst, const uint8_t* bytes, uint64_t n, bf_fill_fn fill, void* user);                       // raw image -> blocks + coarse bitmap
int run_localise(bf_ctx* ctx, bool defer_counts = false, bool all = true);   // read_meta + packed + exact kernels; all = false:
                                                    // only the reads bf_submit has not localised behind their upload yet
int localise_uploaded(bf_ctx* ctx, uint64_t n, uint32_t max_len);
int offsets_from_lengths(bf_ctx* ctx, uint64_t n_uploaded, uint32_t max_len);   // 4-bit batches: rebuilds the offset arrays of the uploaded pieces   // bf_submit: the packed kernel over reads [n_localised, n)
int grow_reads(bf_ctx* ctx, uint64_t cap);          // per-read arrays for `cap` reads, contents kept
int settle_counts(bf_ctx* ctx);                     // n_exact / n_sigx / n_text_records valid on return
int render_prepare(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, DevBuf& d_off, int n_chunks, uint64_t* chunk_off,
                   const uint32_t* d_sizes = nullptr, bool wait = true);
int render_prepare_wait(bf_ctx* ctx, uint64_t n, int n_chunks, uint64_t* chunk_off);     // after render_prepare(..., wait = false)   // d_sizes: record bytes per list entry when the caller has them
int64_t junction_decode(const uint8_t* rec, uint64_t avail, uint8_t* out, uint64_t cap, uint64_t* stored = nullptr, const bf_ctx* ctx = nullptr);   // api.hip, host
int render_range(bf_ctx* ctx, uint64_t a, uint64_t b, const uint32_t* d_local_idx, const DevBuf& d_off, DevBuf& d_bytes);
constexpr uint64_t kRenderSlack = 1u << 17;    // room of one junction record (a text record of a 65535-base read)
int render_range_at(bf_ctx* ctx, uint64_t a, uint64_t b, const uint32_t* d_local_idx, const uint64_t* d_off, uint8_t* d_bytes,
                    uint64_t cap = ~0ull);        // records that would start at or beyond d_bytes + cap are not written
int run_render(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, DevBuf& d_off, DevBuf& d_bytes,
               uint64_t* total_bytes);              // junction strings for local reads

int render_dev(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, uint64_t* d_off, uint8_t* d_bytes,
               uint64_t d_bytes_cap, uint64_t* total);
int render_sizes_checked(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, uint32_t* d_len, uint32_t* d_bad);   // record bytes per list
                                                    // entry; entries outside the resident reads are counted in *d_bad, not read
void shard_destroy(bf_ctx* ctx);                    // shard.hip
int scratch_setup(bf_ctx* ctx);                     // this device's address of the static scratch array -> ctx->scratch

// cluster.hip -----------------------------------------------------------------------------------
// One group of whole contig ranks (api.hip cuts the evidence into a few, so that a group's result crosses PCIe
// while the next group is clustered): the record ids of the group, where its rows and reads go in the result,
// and the capacity of the per-read result arrays.
struct ClusterGroup {
    const uint32_t* idx; uint64_t m;
    uint64_t cl_base, rd_base, cap;
    uint64_t ns = 0, nr = 0;           // out: clusters and supporting reads this group added
    int g = -1;                        // the group's number in the cut cluster_groups made (ctx->group_ranks[g]: its contig ranks)
};
// Queues the cut; waits for nothing.  want_record_bytes: the sum of all records' junction record bytes is on its way to
// ctx->h_mail[kMailRecordBytes] (valid after the caller's next wait for ctx->stream).  view_hist: records per left contig rank
// of an imported view.
int cluster_groups(bf_ctx* ctx, const OwnerView& v, uint32_t max_groups, std::vector<uint64_t>& bounds, bool want_record_bytes,
                   const std::vector<uint64_t>* view_hist = nullptr);
int rank_histogram(bf_ctx* ctx, const OwnerView& v, std::vector<uint64_t>& hist);
int run_cluster(bf_ctx* ctx, const OwnerView& v, ClusterGroup* g = nullptr);   // fills ctx->out (without junction text)
int run_partition(bf_ctx* ctx, const bf_wire_local& agreed, void* dev_send, uint64_t dev_send_bytes, uint64_t* counts,
                  void** dev_out, uint32_t* record_bytes);
// rank_hist: the imported records per left contig rank, counted behind the import (same wait); dev_self: where the records
// of source rank `self` are read instead of their place in dev_recv (the rank's own piece, left in its send buffer)
int run_import(bf_ctx* ctx, const void* dev_recv, const uint64_t* src_counts, std::vector<uint64_t>* rank_hist = nullptr,
               const uint8_t* dev_self = nullptr, uint32_t self = 0);
// primitives.hip -------------------------------------------------------------------------------
// stable LSD radix sort of (key, value) pairs on key bits [begin_bit, end_bit); result in kout/vout, inputs kept
int radix_sort_pairs(bf_ctx* ctx, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n,
                     unsigned begin_bit, unsigned end_bit);
int radix_sort_pairs(bf_ctx* ctx, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n,
                     unsigned begin_bit, unsigned end_bit);
struct RsV3 { uint32_t a, b, c; };
int radix_sort_kv3(bf_ctx* ctx, const uint64_t* kin, uint64_t* kout, const RsV3* vin, RsV3* vout, uint64_t n,
                   unsigned begin_bit, unsigned end_bit);
int scan_u32_to_u64(bf_ctx* ctx, const uint32_t* in, uint64_t* out, uint64_t n);  // exclusive, out has n+1
uint64_t scan_workspace_bytes(uint64_t n);
int scan_u32_to_u64_ws(bf_ctx* ctx, const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* workspace);   // allocates nothing
int scan_max_u32_inplace(bf_ctx* ctx, uint32_t* data, uint64_t n);                // running maximum
int patch_reads(bf_ctx* ctx, const uint64_t* d_joff, uint64_t n);
int patch_reads_off(bf_ctx* ctx, const uint64_t* d_joff, uint64_t n, uint64_t rd_base = 0, uint64_t byte_base = 0);
int cluster_record_offsets(bf_ctx* ctx, const uint64_t* d_joff, uint64_t* d_out, uint64_t n_clusters, uint64_t byte_base = 0);
int order_requests_range(bf_ctx* ctx, uint64_t rd_base, uint64_t nr, uint32_t* d_req, uint64_t** d_start);

}  // namespace bf
