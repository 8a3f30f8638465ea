/*
 * breakfast_hip.h -- C ABI of the MI355X (gfx950) implementation of the `breakfast detect` hot path.
 *
 * The reference (annalam/breakfast @ 2024_08_07) has no FFI seam: `detect::main` is one function
 * (src/detect.rs:39-242).  This ABI is the seam a maintainer would cut into it; each entry point
 * names the reference lines it replaces.  INTEGRATION.md shows the Rust `extern "C"` block and the
 * patched consume loop.  Everything is plain C: opaque handle, POD structs, pointers + sizes, int
 * status codes.  Nothing throws or aborts across the boundary; on failure a function returns a
 * negative BF_ERR_* code and bf_last_error() describes it (the caller maps that to the reference's
 * convention `ERROR: <msg>` on stderr + exit(-1), src/common.rs:8-13).
 *
 * Threading: like the reference's consume loop (one thread, detect.rs:76) a bf_ctx is
 * single-caller.  One ctx drives one GPU; multi-GPU runs use one ctx (one process or thread) per GPU and the
 * bf_shard_* calls below.
 *
 * The library has NO CPU fallback: without a usable gfx950 device every compute entry point fails
 * with BF_ERR_NO_DEVICE.
 */
#ifndef BREAKFAST_HIP_H
#define BREAKFAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_ABI_VERSION 14

enum {
    BF_OK = 0,
    BF_ERR_INVALID = -1,     /* bad argument / unsupported configuration            */
    BF_ERR_NO_DEVICE = -2,   /* no usable HIP device                                */
    BF_ERR_HIP = -3,         /* a HIP runtime call failed                           */
    BF_ERR_STATE = -4,       /* call sequence violated (e.g. submit before genome)  */
    BF_ERR_RANGE = -5,       /* value outside the supported range (see limits)      */
    BF_ERR_NOMEM = -6,
    BF_ERR_IO = -7           /* a file cannot be read or written, or is not what was asked for (genome images) */
};

/* Supported ranges (violations -> BF_ERR_RANGE / BF_ERR_INVALID, never silent truncation):
 *   anchor_len in [8, 32767] (the reference panics below 8: signature index, detect.rs:166)
 *   max_frag_len < 2^31, contigs <= 16384, contig length < 2^31 - 2^17 (BAM's own limit is 2^31-1),
 *   anchor positions < 2^31, read length in [2*anchor_len, 65535] (the reference's producer only
 *   emits reads with len >= 2*anchor_len, detect.rs:311), at most 2^32-1 reads per ctx.           */

typedef struct bf_ctx bf_ctx;

/* Options of `breakfast detect` (detect.rs:28-46).  --anchor-mm is accepted by the reference CLI
 * but ignored (detect.rs:44 is commented out), so it has no field here. */
typedef struct {
    uint32_t anchor_len;     /* --anchor-len   [default 30] */
    uint32_t max_frag_len;   /* --max-frag-len [default 5000] */
    uint32_t min_evidence;   /* --min-evidence [default 2] */
    int32_t  device;         /* HIP device ordinal, -1 = current device */
    uint32_t flags;          /* BF_FLAG_* */
    uint32_t reserved[3];    /* [0]: contig groups of the cluster stage (0 = chosen from the input size; A/B),
                                [1]: host threads that stage pageable bf_submit batches (0 = chosen, at most 16),
                                [2]: reads per upload piece of bf_submit (0 = 2^20; tests) */
} bf_config;

#define BF_FLAG_KEEP_EVIDENCE 1u   /* keep per-read evidence readable via bf_evidence_fetch (tests) */
#define BF_FLAG_FORCE_EXACT   2u   /* route every read through the byte-exact kernel (tests)        */
/* (4, 8, 16 and 1024 were round-1 A/B variants and the stage-call routing flag; removed with ABI version 12) */
#define BF_FLAG_SORTED_DEDUP   32u /* A/B + tests: fragment-id dedup by the sorted-run kernels even when every cluster
                                      fits the in-wave form (they are the path of clusters above 256 members) */
#define BF_FLAG_WEAK_DEDUP_HASH 64u /* tests: the in-wave dedup groups on 3 hash bits, so that its byte verification fails
                                       and the pair-by-pair redo runs */
#define BF_FLAG_ONE_GROUP    128u   /* A/B + tests: cluster all contigs in one go instead of a few contig ranks at a time with
                                      each group's records copied to the host behind the next group's kernels */
#define BF_FLAG_GROUPS_ALWAYS 256u  /* tests: the groups even for inputs below the size where they pay (4M reads) */
#define BF_FLAG_ONE_STREAM   512u   /* profiling: every kernel and copy of the ctx on ONE stream, nothing overlaps (a kernel trace then
                                      shows each kernel's own duration, not its share of a GPU it has to share with the result copy) */
#define BF_FLAG_READS_ON_HOST 2048u /* the caller keeps every submitted batch's sequence arrays (seq_bytes / seq_nibbles, seq_off,
                                       seq_len) valid and unchanged until it is done with the results: junction records then
                                       name the read instead of carrying its bases (kind 3 below; 12 bytes instead of 44 for a
                                       150-base read), and bf_format_sv / bf_junction_text_of read the bases where the caller
                                       left them.  One ctx, one process: not with bf_shard_init_*. */

/* One pair of anchor alignments = the fields the consume loop parses out of a `5p:`/`3p:` line
 * pair (detect.rs:94-103): bowtie's strand, contig and 1-based leftmost offset (-B1) of each anchor.
 * Contig ids index the arrays given to bf_genome_set. 16 bytes. */
typedef struct {
    uint32_t pos5;           /* `pos`  (detect.rs:102) */
    uint32_t pos3;           /* `mpos` (detect.rs:103) */
    uint16_t contig5;        /* `chr`  (detect.rs:100) */
    uint16_t contig3;        /* `mchr` (detect.rs:101) */
    uint8_t  strands;        /* bit0: 5p anchor on '+' (detect.rs:98), bit1: 3p on '+' (:99) */
    uint8_t  pad[3];
} bf_hit;

/* A batch of candidate reads in input (bowtie output) order, structure-of-arrays. */
typedef struct {
    uint64_t        n_reads;
    const uint8_t*  seq_bytes;   /* full read sequences as received (ASCII), concatenated (detect.rs:91) */
    const uint64_t* seq_off;     /* n_reads+1 offsets into seq_bytes */
    const uint8_t*  frag_bytes;  /* fragment ids (QNAME with ':'->'_'), concatenated (detect.rs:90)      */
    const uint64_t* frag_off;    /* n_reads+1 offsets into frag_bytes */
    const bf_hit*   hits;        /* n_reads anchor-hit pairs */
} bf_batch;

/* One reported rearrangement = one printed row (detect.rs:233-240). */
typedef struct {
    int64_t  lpos;           /* seed's left breakpoint position; negative values are the reference's */
    int64_t  rpos;           /* wrapped usize and print as (uint64_t)value (detect.rs:172-181)       */
    uint64_t seed_index;     /* global input index of the seed read */
    uint64_t read_begin;     /* supporting reads = reads[read_begin .. read_end) */
    uint64_t read_end;
    uint32_t contig;         /* `chr`  of the seed (after reorientation, detect.rs:108-113) */
    uint32_t mcontig;        /* `mchr` */
    uint8_t  strand;         /* 1 = '+' */
    uint8_t  mstrand;
    uint8_t  pad[6];
} bf_cluster;

/* Junction records.  A supporting read's case-marked sequence with '|' at the breakpoint (detect.rs:155-162)
 * leaves the GPU as a self-describing record, because the result copy over PCIe is the longest stage of
 * the path and 2-3 bits per base say what 8 bits of text say.  Records start on 4-byte boundaries and are zero-padded
 * to a multiple of 4 bytes; little endian.  First a header word (u32):
 *     bits 0-1    kind: 0 text, 1 packed dense, 2 packed sparse, 3 reference (BF_FLAG_READS_ON_HOST only)
 *     bit  2      the read was reverse-complemented by the reorient step (detect.rs:108-113)
 *   packed kinds (reads of up to 256 bases with at most one base in eight outside ACGT):
 *     bits 3-11   L, the read length            bits 12-20  bp, position of '|' in the string (0 <= bp <= L)
 *     bits 21-26  n_exc, exception entries      bits 27-31  n_lower, listed lower-case positions (sparse; else 0)
 *   followed by
 *     ceil(L/8) x { u8 lo, u8 hi }              sparse: bit j of unit u = base k = 8u+j of the printed sequence,
 *     ceil(L/8) x { u8 lo, u8 hi, u8 lower }    dense:  letter "ACTG"[lo + 2*hi], lower case when `lower` is set
 *     n_exc x { u8 k, u8 byte }                 base k prints as `byte` (N and other IUPAC letters)
 *     n_lower x u8 k                            (sparse) base k prints in lower case: the usual case, a mismatch or two
 *   text kind (longer reads, or many bases outside ACGT): bits 8-23 of the header = bp, then
 *     u32 L, L+1 bytes: the string itself
 *   reference kind (3; only under BF_FLAG_READS_ON_HOST): the bases stay where the caller submitted them from
 *     bit 3       wide: lower-case positions are u16 (reads of more than 256 bases), else u8
 *     bits 4-15   n_lower, listed lower-case positions; 4095: a bit per base instead (u32 L, ceil(L/8) bytes)
 *     bits 16-31  bp
 *     u32 read    the read's number among the ctx's resident reads (submit order)
 *     n_lower x (u8 | u16) k
 * A record's length follows from its header, so records can be walked one after the other.
 * bf_junction_text() expands one record, bf_result_junction_text() all records of a result,
 * bf_format_sv() expands them while it prints.  44 bytes instead of 151 for a 150-base read with one mismatch. */

/* One supporting read that survived frag-id dedup (detect.rs:247-287), in printed order. */
typedef struct {
    uint64_t index;          /* global input index (position in submit order, + ctx base)                         */
    uint64_t junction_off;   /* its junction record (see above) starts at junction_bytes + junction_off; breakpoint */
                             /* offset (detect.rs:147-153) and reorientation are in the record's header             */
} bf_read;                   /* 16 bytes */

/* What the reference prints (detect.rs:233-240) is the cluster rows and the strings of their reads, so that is what
 * bf_finish / bf_clusterise bring to the host: rows, junction records in printed order (a cluster's records are
 * contiguous, each says how long it is) and one offset per cluster.  The read table -- which input read each
 * string came from, and each record's own offset -- is a quarter of the bytes; it is complete on the device and
 * comes over when bf_result_reads() asks for it. */
typedef struct {
    uint64_t          n_evidence;     /* the count the reference logs at detect.rs:190 */
    uint64_t          n_clusters;
    uint64_t          n_reads;
    const bf_cluster* clusters;       /* output order = the reference's print order */
    const bf_read*    reads;          /* NULL until bf_result_reads() */
    const uint8_t*    junction_bytes;
    uint64_t          junction_size;
    const uint64_t*   cluster_junction_off;  /* n_clusters + 1: the records of cluster c's reads, in read order, are
                                                junction_bytes[off[c] .. off[c+1]); NULL when `reads` locates them */
} bf_result;

/* ---- lifetime ------------------------------------------------------------------------------ */

/* Replaces option parsing + set-up, detect.rs:40-46.  Fails loudly when no GPU is usable. */
int bf_ctx_create(const bf_config* cfg, bf_ctx** out);
void bf_ctx_destroy(bf_ctx* ctx);
/* Message for the last failing call on this ctx (or, with ctx == NULL, the last failed create). */
const char* bf_last_error(const bf_ctx* ctx);
int bf_abi_version(void);

/* Replaces the genome HashMap, detect.rs:48-56 + lookups at :115/:124.  Takes each contig's bytes
 * exactly as stored in the FASTA (case preserved).  Packs them on the GPU into the resident
 * 2-bit block image + exception map + a plane "one of acgt" + a table of the runs of every other byte; the bytes
 * themselves are released after the pack (every one of them can be rebuilt: bf_genome_read).  Contig order by byte-wise name compare
 * (Rust String Ord, detect.rs:108,194) is derived here.  Inputs may be freed on return. */
int bf_genome_set(bf_ctx* ctx, uint32_t n_contigs, const char* const* names,
                  const uint8_t* const* seqs, const uint64_t* lens);
/* The same in pieces, for a FASTA loader that uploads while it is still reading (detect.rs:48-56 reads the file
 * record by record): bf_genome_begin fixes names and lengths (from then on bf_submit accepts reads; their kernels
 * wait for the genome), bf_genome_write hands over any piece of any contig -- from ANY thread, in any order;
 * bytes may be freed on return; its errors are returned and repeated by bf_genome_end --, bf_genome_end packs. */
int bf_genome_begin(bf_ctx* ctx, uint32_t n_contigs, const char* const* names, const uint64_t* lens);
int bf_genome_write(bf_ctx* ctx, uint32_t contig, uint64_t offset, const uint8_t* bytes, uint64_t n);
/* bf_genome_write without the copy: the library hands `fill(user, dst, n)` the pinned staging memory the piece is
 * uploaded from and the caller writes the n bases there (e.g. a FASTA reader joining lines); n <= BF_GENOME_FILL_MAX.
 * `fill` runs on the calling thread before the call returns.  Thread-safe like bf_genome_write. */
#define BF_GENOME_FILL_MAX (8u << 20)
typedef void (*bf_fill_fn)(void* user, uint8_t* dst, uint64_t n);
int bf_genome_fill(bf_ctx* ctx, uint32_t contig, uint64_t offset, uint64_t n, bf_fill_fn fill, void* user);
int bf_genome_end(bf_ctx* ctx);
/* The bytes of a piece of a contig as the ctx holds them after bf_genome_end: out[0..n) = what bf_genome_write was
 * given for [offset, offset + n), case and every byte outside ACGT included (the reference's genome[chr][i],
 * detect.rs:118-130; rebuilt on the device from the packed form, copied to `out` in host memory).  For checks. */
int bf_genome_read(bf_ctx* ctx, uint32_t contig, uint64_t offset, uint64_t n, uint8_t* out);

/* ---- the genome as a packed image ------------------------------------------------------------
 * No counterpart in the reference, which reads the FASTA on every run (detect.rs:48-56); the counterpart in its
 * world is Bowtie's index beside the FASTA.  An image holds what the GPU keeps of a genome -- 2 bits a base, the runs
 * of lower-case letters, the runs of every other byte -- so that 0.8 GB instead of hg38's 3.1 GB cross PCIe and nothing
 * of the FASTA is read, joined or packed again.  Every byte of the FASTA can be rebuilt from it (bf_genome_read): the
 * results are the same bytes.  The format is described in breakfast_amd/csrc/genome_image.hip.
 *
 * bf_genome_image_write: host only (no GPU is touched), `threads` workers (0 = chosen from the machine).  src_size and
 * src_mtime_ns name the file the contigs came from (st_size, st_mtim in ns; 0, 0 = unnamed): bf_genome_image_open
 * refuses an image whose source does not match the values it is given (0, 0 = do not check).  The file is written
 * beside its target and renamed when complete.
 * bf_genome_image_open maps and checks the file; the tables bf_genome_image_info returns live until _close.
 * bf_genome_set_image = bf_genome_set from an image; in steps, for a caller that submits reads meanwhile:
 * bf_genome_begin_image (for bf_genome_begin: the contig table bf_genome_image_info gave), bf_genome_write_image (for the
 * bf_genome_write calls: uploads everything, from any one thread, beside bf_submit on others), bf_genome_end.
 * Messages of the calls without a ctx: bf_last_error(NULL). */
typedef struct bf_genome_image bf_genome_image;
int bf_genome_image_write(const char* path, uint32_t n_contigs, const char* const* names, const uint8_t* const* seqs,
                          const uint64_t* lens, uint64_t src_size, int64_t src_mtime_ns, uint32_t threads);
int bf_genome_image_open(const char* path, uint64_t src_size, int64_t src_mtime_ns, bf_genome_image** out);
int bf_genome_image_info(const bf_genome_image* img, uint32_t* n_contigs, const char* const** names, const uint64_t** lens,
                         uint64_t* upload_bytes);
void bf_genome_image_close(bf_genome_image* img);
int bf_genome_begin_image(bf_ctx* ctx, uint32_t n_contigs, const char* const* names, const uint64_t* lens);
int bf_genome_write_image(bf_ctx* ctx, const bf_genome_image* img);
int bf_genome_set_image(bf_ctx* ctx, const bf_genome_image* img);

/* ---- the hot path -------------------------------------------------------------------------- */

/* Appends a batch of parsed anchor-hit pairs (what the consume loop holds after detect.rs:89-103)
 * to the ctx's HBM-resident read set.  ASYNCHRONOUS, as the reference consumes reads while Bowtie still emits
 * them (detect.rs:76-188): the call validates the batch (offsets, lengths, contig ids, positions: errors are
 * reported here, and a rejected batch leaves the ctx as it was), queues the uploads on the ctx's copy-in stream
 * and queues the per-read kernels of detect.rs:105-187 behind every uploaded piece of about a million reads, so
 * that reads are localised while later ones are still on PCIe; bf_finish is left with the cluster stage.
 *   - arrays in ordinary (pageable) memory are staged through pinned slots by a few host threads
 *     (bf_config.reserved[1] = how many, 0 = chosen from the machine, at most 16): they may be freed on return;
 *   - arrays that lie in memory from bf_host_alloc (pinned; e.g. the buffers a parser writes its batches into) are
 *     uploaded from where they are, without a copy: they must stay untouched until bf_submit_wait, bf_finish or
 *     bf_localise returns.  Each of the five arrays is looked at on its own. */
int bf_submit(bf_ctx* ctx, const bf_batch* batch);

/* The same with the reads as BAM 4-bit base codes instead of ASCII (SAMv1 4.2: "=ACMGRSVTWYHKDBN" -> 0..15, two bases
 * per byte, high nibble first; every read starts on a byte and takes ceil(len / 2) bytes: seq_off are BYTE offsets,
 * seq_len the bases).  The reads of detect.rs:91 come from a BAM file, whose decoder (detect.rs:321, rust-htslib) emits
 * exactly these sixteen letters, so the packing loses nothing; it halves what crosses PCIe and what every kernel that
 * touches a read has to fetch.  The two offset arrays are read by the host only (validation): what goes to the GPU are
 * the lengths, 2 + 2 bytes a read instead of 8 + 8, and the device rebuilds the offsets with a scan per uploaded piece.
 * A caller whose reads hold other bytes (lower case, ...) uses bf_submit.  A ctx holds
 * its reads in one form: mixing the two calls without bf_reset is BF_ERR_STATE. */
typedef struct {
    uint64_t        n_reads;
    const uint8_t*  seq_nibbles;
    const uint64_t* seq_off;     /* n_reads+1 byte offsets into seq_nibbles */
    const uint16_t* seq_len;     /* n_reads: bases per read */
    const uint8_t*  frag_bytes;
    const uint64_t* frag_off;
    const bf_hit*   hits;
} bf_packed_batch;
int bf_submit_packed(bf_ctx* ctx, const bf_packed_batch* batch);

/* The same reads at TWO bits a base on PCIe (round 3; what crosses the link is the longest part of a host-to-host step,
 * and a read is > 99 % plain ACGT): A = 0, C = 1, G = 2, T = 3, four bases a byte, base k of a read in bits 2(k & 3) and
 * 2(k & 3) + 1 of its byte k >> 2; every read starts on a byte and takes ceil(len / 4) bytes (seq_off: BYTE offsets).
 * The bases that are none of the four travel beside the stream as exceptions {read, position, BAM 4-bit code}, ascending
 * by (read, position); the stream holds any code there.  The ctx keeps its reads as 4-bit codes, as after
 * bf_submit_packed (the two calls may be mixed): every uploaded piece is expanded on the device behind its upload, the
 * exceptions patched in, before the per-read kernels see it -- results are those of bf_submit_packed on the same reads.
 * Same hand-over rules as bf_submit (asynchronous; arrays in bf_host_alloc memory are uploaded from where they lie; the
 * three exception arrays are copied before the call returns).  Not with BF_FLAG_READS_ON_HOST (records that name a read
 * are expanded from 4-bit or ASCII arrays).  No counterpart in the reference (detect.rs:91 holds reads as ASCII). */
typedef struct {
    uint64_t        n_reads;
    const uint8_t*  seq_2bit;
    const uint64_t* seq_off;      /* n_reads+1 byte offsets into seq_2bit */
    const uint16_t* seq_len;      /* n_reads: bases per read */
    uint64_t        n_exceptions;
    const uint32_t* exc_read;     /* n_exceptions: read index inside this batch, ascending */
    const uint16_t* exc_pos;      /* ... base position inside the read, ascending within a read */
    const uint8_t*  exc_code;     /* ... BAM 4-bit code of that base: any of 0..15 except 1, 2, 4, 8 (A, C, G, T) */
    const uint8_t*  frag_bytes;
    const uint64_t* frag_off;
    const bf_hit*   hits;
} bf_packed2_batch;
int bf_submit_packed2(bf_ctx* ctx, const bf_packed2_batch* batch);

/* Pinned host memory for batches (bf_submit uploads from it without staging).  Owned by the ctx: freed by
 * bf_host_free or bf_ctx_destroy.  These two calls may come from any thread, also while another thread is inside
 * another call on the ctx (a parser's workers allocate while a submit thread submits); bf_host_free waits for the
 * queued uploads.  No counterpart in the reference (its reads never leave host memory). */
int bf_host_alloc(bf_ctx* ctx, uint64_t bytes, void** out);
int bf_host_free(bf_ctx* ctx, void* p);
/* Returns when every upload queued by bf_submit has left host memory (pinned batch buffers may be reused). */
int bf_submit_wait(bf_ctx* ctx);
/* Optional: sizes the resident arrays for this many reads / sequence bytes / fragment id bytes up front, so that no
 * bf_submit has to grow them (growing waits for all queued work and copies).  bf_reset keeps the memory. */
int bf_reserve(bf_ctx* ctx, uint64_t n_reads, uint64_t seq_bytes, uint64_t frag_bytes);

/* Replaces detect.rs:105-187 (reorient, flank gather, split scan, junction masks, signature,
 * coordinates) for every resident read: kernels encode_reads + localise (+ exact path). */
int bf_localise(bf_ctx* ctx);

/* Replaces detect.rs:190-231 + 247-287 (stable sort, greedy clustering, frag-id dedup) and builds
 * the junction strings of the surviving reads on the GPU (detect.rs:155-162).  The result is owned
 * by the ctx and valid until the next bf_clusterise/bf_finish/bf_reset/bf_ctx_destroy. */
int bf_clusterise(bf_ctx* ctx, bf_result* out);

/* bf_localise (only for the reads bf_submit has not localised behind their upload yet) + bf_clusterise.
 * Idempotent: called again with no bf_submit in between, it runs the whole path over the resident reads again. */
int bf_finish(bf_ctx* ctx, bf_result* out);

/* Completes `result` (the one the last bf_finish / bf_clusterise of this ctx returned) with the read table:
 * copies it from the device if it is not on the host yet and sets result->reads.  The table is owned by the ctx
 * like the rest of the result.  No counterpart in the reference, which prints sequences only (detect.rs:236-239). */
int bf_result_reads(bf_ctx* ctx, bf_result* result);

/* Drops the resident reads (the genome stays). */
int bf_reset(bf_ctx* ctx);

/* Blocks until all GPU work queued by this ctx has completed. */
int bf_sync(bf_ctx* ctx);

/* Replaces the print statements detect.rs:201,233-240: renders `result` as the exact stdout bytes
 * of `breakfast detect`.  Returns the byte count (or a negative error); writes at most `cap`
 * bytes.  Call with buf == NULL to query the size. */
int64_t bf_format_sv(const bf_ctx* ctx, const bf_result* result, char* buf, uint64_t cap);

/* The string detect.rs:155-162 builds for one read, from its junction record: writes len+1 bytes (no
 * terminator) when buf is given and cap suffices; returns len+1, or BF_ERR_INVALID for a malformed
 * record.  `avail` = bytes readable from `record` on (at least the record; e.g. junction_size - junction_off).
 * Pure host code, needs no ctx. */
int64_t bf_junction_text(const uint8_t* record, uint64_t avail, char* buf, uint64_t cap);
/* bf_junction_text for the records of `ctx`: the only way to expand a reference record (BF_FLAG_READS_ON_HOST), which
 * names its read and leaves the bases in the caller's arrays. */
int64_t bf_junction_text_of(const bf_ctx* ctx, const uint8_t* record, uint64_t avail, char* buf, uint64_t cap);
/* Stored size of the record in bytes (a multiple of 4), or BF_ERR_INVALID. */
int64_t bf_junction_record_bytes(const uint8_t* record, uint64_t avail);

/* All junction strings of a result, concatenated in read order: text_off receives n_reads+1 offsets,
 * text (capacity cap) the bytes when given and large enough.  Returns the total size or a negative error
 * (BF_ERR_INVALID also for results that hold reference records: those need the ctx, bf_junction_text_of). */
int64_t bf_result_junction_text(const bf_result* result, uint8_t* text, uint64_t cap, uint64_t* text_off);

/* ---- junction strings on request ---------------------------------------------------------------- */

/* Renders the junction strings (detect.rs:155-162) of the given resident reads (device work) and expands them to
 * TEXT on the host.  out_bytes/out_off (n+1 offsets) are owned by the ctx until the next call.  Needs bf_localise
 * (or bf_finish) first.  The reference builds this string for every read (detect.rs:155-162); here only the reads
 * that survive are rendered by bf_finish, and this call serves a caller that wants others (tests compare ALL reads). */
int bf_render_junctions(bf_ctx* ctx, uint64_t n, const uint32_t* local_index,
                        const uint8_t** out_bytes, const uint64_t** out_off);

/* The same for a DEVICE index list into caller-owned device memory, as junction RECORDS (format above): dev_off
 * receives n+1 byte offsets, *total their sum.  With dev_bytes == NULL only offsets and *total are produced (size
 * query); otherwise dev_bytes (capacity dev_bytes_cap >= *total) receives the records.  An index outside the resident
 * reads is an error, never dereferenced.  dev_bytes and dev_local_index must be 4-byte aligned, dev_off 8-byte
 * aligned.  Returns after its GPU work has completed.  No counterpart in the reference. */
int bf_render_junctions_dev(bf_ctx* ctx, uint64_t n, const uint32_t* dev_local_index, uint64_t* dev_off,
                            uint8_t* dev_bytes, uint64_t dev_bytes_cap, uint64_t* total);

/* bf_render_junctions_dev for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same results, but
 * the call makes no hipMalloc / hipFree / hipHostMalloc and creates no events: the per-entry record sizes, the
 * scan's tile totals and the bad-index counter -- which bf_render_junctions_dev keeps in ctx-owned device memory
 * that grows on demand -- live in a static device array of 4 MiB + 2.5 KiB that every GPU the library runs on holds
 * until the process ends.  At most BF_STATIC_RENDER_MAX_READS list entries per call (BF_ERR_RANGE beyond; split
 * the list).  All ctxs of a process on one GPU share that GPU's array: calls that overlap on one GPU race (the
 * array is not locked); one after the other they are independent (what must start at zero is cleared). */
#define BF_STATIC_RENDER_MAX_READS (1u << 20)
int bf_render_junctions_dev_static(bf_ctx* ctx, uint64_t n, const uint32_t* dev_local_index, uint64_t* dev_off,
                                   uint8_t* dev_bytes, uint64_t dev_bytes_cap, uint64_t* total);
/* Upper bound of a resident read's junction record in bytes (sizing of bf_render_junctions_dev buffers). */
uint64_t bf_junction_record_bound(const bf_ctx* ctx);

/* ---- the sharded path in one call (north_star item 4) ------------------ */
/* The path shards by contiguous input ranges; rank r's reads have global indices [index_base, index_base + n).
 * Evidence of one breakpoint class must meet on one rank.  The calls below run the whole sharded path inside
 * the library, from C, with no orchestration language in between: one ctx per GPU, each driven by its own thread
 * or process; the ranks call the same functions in the same order.  (Until ABI version 11 a set of stage calls --
 * bf_set_rank, bf_exchange_*, bf_result_attach_* -- left the all-to-all to the caller; nothing used them any more.)
 *
 *   bf_shard_unique_id      rank 0 makes the 128-byte RCCL id (ncclGetUniqueId); the launcher hands it to every rank
 *   bf_shard_init_rccl      joins the communicator (ncclCommInitRank): evidence crosses xGMI, device to device.
 *                           RCCL is loaded at run time (librccl.so.1); one rank per GPU, as RCCL requires
 *   bf_shard_init_host      a host-staged transport instead: ranks meet in a POSIX shared-memory segment `name`
 *                           with `slot_bytes` of staging per rank (threads or processes of one machine).  Ranks may
 *                           share a GPU: what a one-GPU test box can run, and the fallback where RCCL is absent
 *   bf_shard_finish         the sharded bf_finish.  Every rank: localises its reads (contiguous input range
 *                           [index_base, index_base + n), detect.rs:105-187); packs the evidence by
 *                           owner = mix64(class key) % n_ranks; ONE all-to-all brings every class together
 *                           (records arrive ordered by source rank, then input order, so the stable sort keeps the
 *                           reference's tie-breaks, detect.rs:193-198); clusters what it owns (detect.rs:190-231,
 *                           247-287).  The case-marked sequences never travel: an owner sends each source rank the
 *                           list of its surviving reads (4 bytes each), the source renders their junction records
 *                           and copies them to ITS host memory, and the owner's `src_rank` says on which rank each
 *                           supporting read's record lives.  Idempotent like bf_finish.
 *   bf_shard_format_sv      detect.rs:201,233-240 for the ranks' results together: owners' rows merged by the seed's
 *                           rank in the reference's sort (contig by name, position as the wrapped usize, input index).
 *                           Pure host code; `results[r]` = rank r's bf_shard_result (all n_ranks of them).
 * No counterpart in the reference, which is one process. */
typedef struct { uint32_t src_rank; uint32_t slot; } bf_place;    /* record = results[src_rank].record_off[slot] */
typedef struct {
    uint64_t          n_local;         /* reads this rank localised (sum over ranks = the count of detect.rs:190) */
    uint64_t          n_evidence;      /* evidence records this rank owns (clustered here) */
    /* owner side */
    uint64_t          n_clusters, n_reads;
    const bf_cluster* clusters;        /* this owner's rows, in print order among themselves */
    const uint8_t*    src_rank;        /* n_reads: the rank whose host memory holds supporting read i's record */
    uint32_t          n_groups;        /* the cluster stage ran in this many rounds (the same on every rank) */
    uint32_t          reserved;
    const uint64_t*   group_read_begin;/* n_groups + 1: supporting reads [b[g], b[g+1]) belong to round g */
    /* source side: junction records of this rank's reads that support a row of any owner.  They lie in the order they
     * were asked for: round by round, inside a round owner by owner, inside that in the owner's own result order.
     * So whoever walks owner o's rows in order finds the records o asked of source s one after the other from
     * block_off[g * n_ranks + o] on (a record says how long it is, bf_junction_record_bytes): 1 byte per supporting
     * read crosses PCIe for the book-keeping instead of a 8-byte place and a 8-byte record offset. */
    uint64_t          n_records;
    const uint64_t*   block_off;       /* n_groups * n_ranks + 1 byte offsets into record_bytes */
    const uint8_t*    record_bytes;
    uint64_t          record_size;
    /* random access, on demand (bf_shard_result_places; NULL before): */
    const bf_place*   places;          /* n_reads: supporting read i of the rows above */
    const uint64_t*   record_off;      /* n_records + 1 offsets into record_bytes */
} bf_shard_result;
int bf_shard_unique_id(uint8_t id[128]);
int bf_shard_init_rccl(bf_ctx* ctx, uint32_t rank, uint32_t n_ranks, uint64_t index_base, const uint8_t id[128]);
int bf_shard_init_host(bf_ctx* ctx, uint32_t rank, uint32_t n_ranks, uint64_t index_base, const char* name, uint64_t slot_bytes);
int bf_shard_finish(bf_ctx* ctx, bf_shard_result* out);
/* Fetches the per-read tables that bf_shard_finish leaves on the device (like bf_result_reads for bf_finish): the owner's
 * `places` and the source's `record_off`, for callers that want a supporting read's record without walking the rows.
 * `result` must be what this ctx's last bf_shard_finish filled. */
int bf_shard_result_places(bf_ctx* ctx, bf_shard_result* result);
int64_t bf_shard_format_sv(const bf_ctx* ctx, uint32_t n_ranks, const bf_shard_result* results, char* buf, uint64_t cap);
/* The same without a ctx (a merging process that holds no GPU): contig names as given to bf_genome_set. */
int64_t bf_shard_format_sv_names(uint32_t n_contigs, const char* const* names, uint32_t n_ranks, const bf_shard_result* results,
                                 char* buf, uint64_t cap);

/* ---- introspection (tests, bench) ---------------------------------------------------------- */

/* Per-read evidence in input order (needs BF_FLAG_KEEP_EVIDENCE): arrays of n entries each, any
 * may be NULL.  mask = combined mismatch mask (bit k of word k/64: read base k mismatches the
 * flank it was assigned to), mask_words u64 per read. */
typedef struct {
    uint64_t  n;
    uint32_t  mask_words;
    uint64_t* class_key;     /* packed (contig rank, mcontig rank, strands, signature) */
    int32_t*  lpos;
    int32_t*  rpos;
    uint16_t* bp;
    uint16_t* flags;         /* BF_EV_* */
    uint64_t* mask;          /* n * mask_words, read-major */
} bf_evidence_view;
#define BF_EV_STRAND   1u
#define BF_EV_MSTRAND  2u
#define BF_EV_SWAPPED  4u
#define BF_EV_EXACT    8u     /* went through the byte-exact kernel */
#define BF_EV_SIGX     16u    /* signature contains a non-ACGT reference byte */
int bf_evidence_fetch(bf_ctx* ctx, bf_evidence_view* view);

/* Kernel timing of the last bf_localise / bf_clusterise in milliseconds (HIP events on the ctx's
 * stream): names are stable identifiers, e.g. "encode_reads", "localise", "sort", "sweep". */
typedef struct { const char* name; float ms; uint64_t launches; } bf_kernel_time;
int bf_kernel_times(bf_ctx* ctx, const bf_kernel_time** out, uint32_t* n);

/* Number of resident reads / bytes of HBM held by the ctx. */
int bf_stats(const bf_ctx* ctx, uint64_t* n_reads, uint64_t* device_bytes);

#ifdef __cplusplus
}
#endif
#endif /* BREAKFAST_HIP_H */
