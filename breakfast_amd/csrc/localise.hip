// Per-read kernels of the `breakfast detect` hot path for gfx950 (MI355X):
//   genome_pack    raw FASTA bytes -> 2-bit block image + coarse exception bitmap   (detect.rs:48-56)
//   read_meta      fragment ids -> 64-bit hashes, lengths                            (detect.rs:90)
//   localise_fast  read -> bit planes in registers (detect.rs:91), reorient + two flank gathers + split scan +
//                  signature + coords                                                (detect.rs:105-187)
//   localise_exact the same on raw bytes for reads/windows the packed path cannot represent
//   render         case-marked junction strings of surviving reads                   (detect.rs:155-162)
//
// Mapping (wave64): one lane per read.  All per-read state (3 read planes, 2x2 window planes,
// 2 mismatch masks) lives in registers as W 64-bit words, bit k <-> read base k, so the split scan
// is popcount + a branch-free running sum and never touches LDS.  The only uncoalesced traffic is the
// inherent one, the two 16*(W+1)-byte window gathers per read.  (Round 1's A/B variants -- a separate
// wave-ballot encode kernel, a cooperative LDS-staged gather -- lost on measurement and were removed in
// round 3: profiles/r01*)

#include "bf_internal.h"

#include <chrono>

namespace bf {

__constant__ uint8_t kComp[256] = {
#include "bf_comp_table.inc"
};

__device__ __forceinline__ uint8_t comp_byte(uint8_t c) { return kComp[c]; }

// Reads may be resident as BAM 4-bit base codes (bf_submit_packed; SAMv1 4.2: "=ACMGRSVTWYHKDBN", two bases per
// byte, high nibble first, every read starting on a byte) instead of ASCII: half the bytes on PCIe and in every
// kernel that touches a read.  base_at() is the byte the reference would see (detect.rs:91 decodes the same table).
__constant__ uint8_t kNibbleBase[16] = {'=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};
__device__ __forceinline__ uint8_t base_at(const uint8_t* __restrict__ seq, uint64_t so, int k, bool packed) {
    if (!packed) return seq[so + (uint64_t)k];
    const uint8_t b = seq[so + (uint64_t)(k >> 1)];
    return kNibbleBase[(k & 1) ? (b & 15) : (b >> 4)];
}
__device__ __forceinline__ bool is_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// ---------------------------------------------------------------------------------------------
// genome_pack: thread t packs bases [64t, 64t+64).  A wave's 64 exception flags form one word
// of the coarse bitmap (ballot), so block t's flag is bit t%64 of coarse[t/64].
struct PackedBlock { uint64_t lo, hi, ex, lw, st; };       // st: bases that start a run of equal bytes outside ACGTacgt

__device__ __forceinline__ PackedBlock pack_block(const uint8_t* __restrict__ raw, uint64_t t) {
    const uint4* src = (const uint4*)(raw + t * 64);
    PackedBlock o{0, 0, 0, 0, 0};
    uint8_t prev = t ? raw[t * 64 - 1] : (uint8_t)'A';
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 v = src[q];
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint8_t c = (uint8_t)(wds[d] >> (8 * b));
                const int k = q * 16 + d * 4 + b;
                const uint64_t code = (c >> 1) & 3;
                const bool upper = is_acgt(c), lower = is_acgt((uint8_t)(c ^ 0x20)) && (c & 0x20);
                o.lo |= (code & 1) << k;
                o.hi |= (code >> 1) << k;
                o.ex |= (uint64_t)!upper << k;
                o.lw |= (uint64_t)lower << k;
                o.st |= (uint64_t)(!upper && !lower && c != prev) << k;
                prev = c;
            }
        }
    }
    return o;
}

__global__ void __launch_bounds__(256) genome_pack_kernel(const uint8_t* __restrict__ raw,
                                                          GenomeBlock* __restrict__ blocks,
                                                          uint64_t* __restrict__ exc_plane,
                                                          uint64_t* __restrict__ lower_plane,
                                                          uint64_t* __restrict__ coarse,
                                                          uint32_t* __restrict__ coarse2,
                                                          uint32_t* __restrict__ wave_runs,
                                                          uint64_t n_blocks /* multiple of 64 */) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_blocks) return;  // whole waves only: n_blocks and blockDim are multiples of 64
    const PackedBlock o = pack_block(raw, t);
    const uint64_t lo = o.lo, hi = o.hi, ex = o.ex;
    // overlapping lines: block t is the lower-half slot t%4 of line t/4 and the upper-half slot of line t/4 - 1
    blocks[block_slot(t)] = GenomeBlock{lo, hi};
    if (t >= 4) blocks[block_slot(t - 4) + 4] = GenomeBlock{lo, hi};
    if (ex) {                                            // the plane is zero-initialised
        exc_plane[exc_slot(t)] = ex;
        if (t >= 8) exc_plane[exc_slot(t - 8) + 8] = ex;
    }
    lower_plane[t] = o.lw;
    const uint64_t flags = __ballot(ex != 0);
    uint32_t runs = (uint32_t)__popcll(o.st);
    if (__ballot(runs != 0)) for (int d = 32; d > 0; d >>= 1) runs += (uint32_t)__shfl_xor((int)runs, d);
    if ((threadIdx.x & 63) == 0) {
        coarse[t >> 6] = flags;
        // second level: one bit per coarse word (4096 bases); zero-initialised, set bits are rare
        if (flags) atomicOr(&coarse2[t >> 11], 1u << ((t >> 6) & 31));
        wave_runs[t >> 6] = runs;
    }
}

// second pass over the waves that hold run starts: the table {start, byte}, in ascending order of start
__global__ void __launch_bounds__(256) genome_runs_kernel(const uint8_t* __restrict__ raw, const uint32_t* __restrict__ wave_runs,
                                                          const uint64_t* __restrict__ wave_off, uint64_t* __restrict__ run_start,
                                                          uint8_t* __restrict__ run_byte, uint64_t n_blocks) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_blocks || wave_runs[t >> 6] == 0) return;          // wave-uniform
    uint64_t st = pack_block(raw, t).st;
    const uint32_t mine = (uint32_t)__popcll(st);
    uint32_t incl = mine;                                         // inclusive prefix over the wave's lanes
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += up; }
    uint64_t at = wave_off[t >> 6] + (incl - mine);
    while (st) {
        const int k = __ffsll((long long)st) - 1;
        st &= st - 1;
        run_start[at] = t * 64 + (uint64_t)k;
        run_byte[at] = raw[t * 64 + (uint64_t)k];
        at++;
    }
}

// lengths + FNV-1a/64 with a final avalanche of the fragment id bytes
__global__ void __launch_bounds__(256) read_meta_kernel(uint64_t r0, uint64_t n, const uint64_t* __restrict__ seq_off,
                                                        const uint8_t* __restrict__ frag,
                                                        const uint64_t* __restrict__ frag_off,
                                                        uint16_t* __restrict__ len,
                                                        uint16_t* __restrict__ frag_len,
                                                        uint64_t* __restrict__ fhash, bool packed) {
    const uint64_t r = r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;     // reads [r0, n)
    if (r >= n) return;
    if (!packed) len[r] = (uint16_t)(seq_off[r + 1] - seq_off[r]);                // (4-bit reads bring their lengths along)
    const uint64_t o = frag_off[r];
    const uint32_t fl = (uint32_t)(frag_off[r + 1] - o);
    frag_len[r] = (uint16_t)fl;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t k = 0; k < fl; k++) { h ^= frag[o + k]; h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    fhash[r] = h;
}

// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void reverse_bits(uint64_t (&p)[W], int L) {
    // p'[k] = p[L-1-k] for k < L, 0 above.  Reverse all 64W bits, then shift right by 64W - L.
    uint64_t t[W];
#pragma unroll
    for (int w = 0; w < W; w++) t[w] = __brevll(p[W - 1 - w]);
    int s = 64 * W - L;
#pragma unroll
    for (int i = 0; i < W - 1; i++) {
        if (s >= 64) {
#pragma unroll
            for (int w = 0; w < W - 1; w++) t[w] = t[w + 1];
            t[W - 1] = 0;
            s -= 64;
        }
    }
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t up = (w + 1 < W) ? t[w + 1] : 0ull;
        p[w] = (t[w] >> s) | ((up << 1) << (63 - s));
    }
}

template <int W>
__device__ __forceinline__ void load_window(const GenomeBlock* __restrict__ blocks, uint64_t g,
                                            uint64_t (&lo)[W], uint64_t (&hi)[W]) {
    // bits of bases [g, g + 64W): W+1 consecutive 16-byte blocks, funnel-shifted by g % 64
    const GenomeBlock* __restrict__ p = blocks + block_slot(g >> 6);   // W+1 <= 5 blocks: inside one 128-byte line
    const int sh = (int)(g & 63);
    GenomeBlock x[W + 1];
#pragma unroll
    for (int i = 0; i <= W; i++) x[i] = p[i];
#pragma unroll
    for (int w = 0; w < W; w++) {
        lo[w] = (x[w].lo >> sh) | ((x[w + 1].lo << 1) << (63 - sh));
        hi[w] = (x[w].hi >> sh) | ((x[w + 1].hi << 1) << (63 - sh));
    }
}

template <int W>
__device__ __forceinline__ void load_window_exc(const uint64_t* __restrict__ exc_plane, uint64_t g, uint64_t (&ex)[W]) {
    const uint64_t* __restrict__ p = exc_plane + exc_slot(g >> 6);        // W+1 <= 5 words: inside one line
    const int sh = (int)(g & 63);
    uint64_t x[W + 1];
#pragma unroll
    for (int i = 0; i <= W; i++) x[i] = p[i];
#pragma unroll
    for (int w = 0; w < W; w++) ex[w] = (x[w] >> sh) | ((x[w + 1] << 1) << (63 - sh));
}

// bits of the lower-case plane for the 8 bases from g on (bit j = base g + j); the plane ends in a spare word
__device__ __forceinline__ uint32_t lower8(const uint64_t* __restrict__ lower, uint64_t g) {
    const uint64_t w0 = lower[g >> 6], w1 = lower[(g >> 6) + 1];
    const int sh = (int)(g & 63);
    return (uint32_t)((w0 >> sh) | ((w1 << 1) << (63 - sh))) & 0xffu;
}

// 32-bit stand-in for a signature that holds a non-ACGT byte (the sweep compares the 16 raw bytes)
__device__ __forceinline__ uint32_t sigx_hash(const uint8_t (&sb)[16]) {
    uint32_t h = 0x811c9dc5u;
#pragma unroll
    for (int k = 0; k < 16; k++) { h ^= sb[k]; h *= 0x01000193u; }
    return h;
}

template <int W>
__device__ __forceinline__ uint32_t extract8(const uint64_t (&p)[W], int off) {
    // bits [off, off+8) of the multi-word value (off + 8 <= 64W)
    uint64_t v = 0;
    const int wi = off >> 6, sh = off & 63;
#pragma unroll
    for (int w = 0; w < W; w++) {
        if (w == wi) {
            const uint64_t up = (w + 1 < W) ? p[w + 1] : 0ull;
            v = (p[w] >> sh) | ((up << 1) << (63 - sh));
        }
    }
    return (uint32_t)(v & 0xff);
}

__device__ __forceinline__ uint64_t low_mask(int nbits) {   // nbits in [0, 64]
    return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1);
}

// ---------------------------------------------------------------------------------------------
// In-register encoding of one read (fused path): the lane loads its own ASCII bases as 16-byte groups
// (4-byte aligned; every fetched line is fully used by this lane and its neighbours) and turns each
// dword of 4 bases into three 4-bit nibbles (low code bit, high code bit, "not ACGT") with SWAR
// arithmetic: code = (c >> 1) & 3 (A0 C1 T2 G3); a byte is ACGT iff it equals
// 0x41 + 2*c0 + 0x13*c1 - 0x0F*(c0 & c1) = 'A','C','T','G' for codes 0..3.
struct __attribute__((aligned(4))) U4A { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t gather4(uint32_t v) {   // bits 0,8,16,24 -> bits 0..3
    const uint32_t a = v | (v >> 7);
    return (a | (a >> 14)) & 0xfu;
}

template <int W>
__device__ __forceinline__ void encode_ascii(const uint8_t* __restrict__ seq, uint64_t o, int L,
                                             uint64_t (&lo)[W], uint64_t (&hi)[W], uint64_t (&ex)[W]) {
    constexpr int ND = 16 * W;                              // dwords of 4 bases
    const uint32_t sh8 = (uint32_t)(o & 3);
    const U4A* __restrict__ src = (const U4A*)(seq + (o - sh8));
    const int need = (int)((sh8 + (uint32_t)L + 3u) >> 2);  // aligned dwords holding the read
    uint32_t d[ND + 4];
#pragma unroll
    for (int g = 0; g <= ND / 4; g++) {
        U4A v{0, 0, 0, 0};
        if (4 * g < need) v = src[g];
        d[4 * g] = v.x; d[4 * g + 1] = v.y; d[4 * g + 2] = v.z; d[4 * g + 3] = v.w;
    }
    uint32_t plo[2 * W], phi[2 * W], pex[2 * W];
#pragma unroll
    for (int i = 0; i < 2 * W; i++) plo[i] = phi[i] = pex[i] = 0;
    // Per dword of 4 bases (VALU-bound: this loop was 0.4 ms of the kernel's 1.1 ms with shift/or bit
    // gathers): the 2-bit codes select the expected letter through v_perm_b32 used as a 4-entry table,
    // and v_dot4_u32_u8 with weights 1,2,4,8 (16..128 for the odd dword) gathers one bit per byte into a
    // nibble, accumulating two dwords = 8 bases per plane byte.
#pragma unroll
    for (int g = 0; g < ND / 4; g++) {
        if (__any(4 * g < need)) {                          // wave-uniform skip of the unused tail
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++) {
                uint32_t blo = 0, bhi = 0, bex = 0;         // 8 bases -> one byte per plane
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int j = 4 * g + 2 * q2 + h;
                    const uint32_t x = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh8);
                    const uint32_t sel = (x >> 1) & 0x03030303u;                     // A0 C1 T2 G3 per byte
                    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, sel);   // "ACTG"[code]
                    const uint32_t e = x ^ expect;
                    const uint32_t nz = ((((e & 0x7f7f7f7fu) + 0x7f7f7f7fu) | e) >> 7) & 0x01010101u;
                    const uint32_t wgt = h ? 0x80402010u : 0x08040201u;
                    blo = __builtin_amdgcn_udot4(sel & 0x01010101u, wgt, blo, false);
                    bhi = __builtin_amdgcn_udot4((sel >> 1) & 0x01010101u, wgt, bhi, false);
                    bex = __builtin_amdgcn_udot4(nz, wgt, bex, false);
                }
                const int j0 = 4 * g + 2 * q2;              // even dword index: 8 bases at bit 4*j0 of the plane
                const int slot = j0 >> 3, bit = 4 * (j0 & 7);
                plo[slot] |= blo << bit;
                phi[slot] |= bhi << bit;
                pex[slot] |= bex << bit;
            }
        }
    }
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t vm = low_mask(min(max(L - 64 * w, 0), 64));
        lo[w] = (((uint64_t)plo[2 * w + 1] << 32) | plo[2 * w]) & vm;
        hi[w] = (((uint64_t)phi[2 * w + 1] << 32) | phi[2 * w]) & vm;
        ex[w] = (((uint64_t)pex[2 * w + 1] << 32) | pex[2 * w]) & vm;
    }
}

// The same three planes from 4-bit base codes.  One dword = 8 bases; the two nibbles of every byte are swapped first
// so that nibble j of the dword is base j; then lo = "C or G" = bit1|bit2, hi = "G or T" = bit2|bit3 of the one-hot
// code (A=1 C=2 G=4 T=8 -> 2-bit codes A0 C1 T2 G3 as in the genome image), exception = the nibble's popcount is not 1.
__device__ __forceinline__ uint32_t gather_nibble_bits(uint32_t t) {     // bits 0,4,...,28 -> bits 0..7
    uint32_t y = (t | (t >> 3)) & 0x03030303u;
    y = (y | (y >> 6)) & 0x000f000fu;
    return (y | (y >> 12)) & 0xffu;
}

template <int W>
__device__ __forceinline__ void encode_nibbles(const uint8_t* __restrict__ seq, uint64_t o, int L,
                                               uint64_t (&lo)[W], uint64_t (&hi)[W], uint64_t (&ex)[W]) {
    constexpr int ND = 8 * W;                               // dwords of 8 bases
    const uint32_t sh8 = (uint32_t)(o & 3);
    const U4A* __restrict__ src = (const U4A*)(seq + (o - sh8));
    const int need = (int)((sh8 + (uint32_t)((L + 1) >> 1) + 3u) >> 2);   // aligned dwords holding the read
    uint32_t d[ND + 4];
#pragma unroll
    for (int g = 0; g <= ND / 4; g++) {
        U4A v{0, 0, 0, 0};
        if (4 * g < need) v = src[g];
        d[4 * g] = v.x; d[4 * g + 1] = v.y; d[4 * g + 2] = v.z; d[4 * g + 3] = v.w;
    }
    uint32_t plo[2 * W], phi[2 * W], pex[2 * W];
#pragma unroll
    for (int i = 0; i < 2 * W; i++) plo[i] = phi[i] = pex[i] = 0;
#pragma unroll
    for (int g = 0; g < ND / 4; g++) {
        if (__any(4 * g < need)) {                          // wave-uniform skip of the unused tail
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int j = 4 * g + q;                    // dword j = bases [8j, 8j + 8)
                const uint32_t x = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh8);
                const uint32_t n = ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu);
                const uint32_t blo = gather_nibble_bits(((n >> 1) | (n >> 2)) & 0x11111111u);
                const uint32_t bhi = gather_nibble_bits(((n >> 2) | (n >> 3)) & 0x11111111u);
                uint32_t c = (n & 0x55555555u) + ((n >> 1) & 0x55555555u);
                c = (c & 0x33333333u) + ((c >> 2) & 0x33333333u);          // popcount of every nibble
                const uint32_t e = c ^ 0x11111111u;                           // zero where exactly one bit is set
                const uint32_t bex = gather_nibble_bits((e | (e >> 1) | (e >> 2) | (e >> 3)) & 0x11111111u);
                const int slot = j >> 2, bit = 8 * (j & 3);
                plo[slot] |= blo << bit;
                phi[slot] |= bhi << bit;
                pex[slot] |= bex << bit;
            }
        }
    }
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t vm = low_mask(min(max(L - 64 * w, 0), 64));
        lo[w] = (((uint64_t)plo[2 * w + 1] << 32) | plo[2 * w]) & vm;
        hi[w] = (((uint64_t)phi[2 * w + 1] << 32) | phi[2 * w]) & vm;
        ex[w] = (((uint64_t)pex[2 * w + 1] << 32) | pex[2 * w]) & vm;
    }
}

template <int W>
__device__ __forceinline__ void encode_read(const uint8_t* __restrict__ seq, uint64_t o, int L, bool packed,
                                            uint64_t (&lo)[W], uint64_t (&hi)[W], uint64_t (&ex)[W]) {
    if (packed) encode_nibbles<W>(seq, o, L, lo, hi, ex); else encode_ascii<W>(seq, o, L, lo, hi, ex);
}

struct LocaliseArgs {
    uint64_t r0, n;          // this launch covers reads [r0, n): bf_submit localises every uploaded piece on its own
    int A;
    int cb;                  // bits per contig rank in the class key
    uint32_t force_exact;
    const bf_hit* hits;
    const uint16_t* len;          // 4-bit reads: their lengths (input)
    uint16_t* len_out;            // ASCII reads: their lengths (output)
    const uint8_t* seq;           // the reads, ASCII or 4-bit codes
    const uint64_t* seq_off;      // ... and where each starts (bytes)
    uint32_t packed;              // 1: 4-bit codes, lengths in `len`
    uint32_t by_ref;              // 1: junction records name the read (BF_FLAG_READS_ON_HOST): their size in `info`
    const ContigDev* contigs;
    const GenomeBlock* blocks;
    const uint64_t* coarse;
    const uint32_t* coarse2;
    const uint64_t* exc;          // exception plane (overlapping lines), read only by flagged windows
    GenomeBytes gb;               // raw genome bytes, rebuilt only for signatures that hold a non-ACGT byte
    uint8_t* sigx;                // 16 raw signature bytes per slot
    uint32_t* sigx_slot;          // read -> slot
    uint64_t* key;
    int32_t* lpos;
    int32_t* rpos;
    uint16_t* bp;
    uint16_t* flags;
    uint64_t* meta;          // render record per read: mask_words mask words + {bp, flags, len}
    int meta_stride;         // u64 per record = mask_words + 1
    RecWord* rw;             // read_info() (junction record bytes, flags, bp, len in one word) beside the fragment-id hash:
    const uint64_t* fhash;   // one 16-byte record per read for the cluster stage's gathers (fhash: read_meta_kernel's output)
    uint32_t* exact_list;
    uint32_t* counters;      // [0] = reads routed to the exact kernel, [1] = raw-signature slots handed out,
                             // [2] = reads whose junction record is text (not packable)
};

struct Oriented {            // the read's two anchors after detect.rs:108-113
    ContigDev c, mc;
    int64_t pos, mpos;
    bool strand, mstrand, swapped;
};

__device__ __forceinline__ Oriented orient(const bf_hit& h, const ContigDev* __restrict__ contigs) {
    Oriented o;
    o.c = contigs[h.contig5];
    o.mc = contigs[h.contig3];
    o.pos = h.pos5;
    o.mpos = h.pos3;
    o.strand = (h.strands & 1) != 0;
    o.mstrand = (h.strands & 2) != 0;
    // `chr > mchr || (chr == mchr && pos > mpos)`: names compare byte-wise == ranks compare
    o.swapped = o.c.rank > o.mc.rank || (o.c.rank == o.mc.rank && o.pos > o.mpos);
    if (o.swapped) {
        const ContigDev tc = o.c; o.c = o.mc; o.mc = tc;
        const int64_t tp = o.pos; o.pos = o.mpos; o.mpos = tp;
        const bool ts = o.strand; o.strand = !o.mstrand; o.mstrand = !ts;
    }
    return o;
}

__device__ __forceinline__ uint64_t make_key(const Oriented& o, bool sigx, uint32_t sig, int cb) {
    return ((uint64_t)o.c.rank << key_rank_shift(cb)) | ((uint64_t)o.mc.rank << kKeyMrankShift) |
           ((uint64_t)o.strand << kKeyStrandBit) | ((uint64_t)o.mstrand << kKeyMstrandBit) |
           ((uint64_t)sigx << kKeySigxBit) | sig;
}

// (No occupancy attribute here: at W=3 the kernel takes 102 VGPRs = 4 waves/SIMD; forcing 5, 6 or 8 waves
// spills and measured 1.12, 1.45 and 2.30 ms against 1.07 ms.)
template <int W, bool PACKED>
__global__ void __launch_bounds__(256) localise_fast_kernel(const LocaliseArgs a) {
    const uint64_t r = a.r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const uint64_t so = a.seq_off[r];
    int L;
    if (PACKED) L = a.len[r];
    else { L = (int)(a.seq_off[r + 1] - so); a.len_out[r] = (uint16_t)L; }
    const int A = a.A;
    const bf_hit h = a.hits[r];
    const uint64_t fh = a.fhash[r];          // (only passed on: it travels beside the info word from here)
    const Oriented o = orient(h, a.contigs);

    // forward-strand window starts, 0-based inside the contig (detect.rs:116-131)
    const int64_t ls = o.strand ? o.pos - 1 : o.pos + A - 1 - L;
    const int64_t rs = o.mstrand ? o.mpos + A - 1 - L : o.mpos - 1;
    bool ok = L <= 64 * W && !a.force_exact;
    ok = ok && ls >= 0 && ls + L <= (int64_t)o.c.len && rs >= 0 && rs + L <= (int64_t)o.mc.len;
    const uint64_t lg = o.c.goff + (uint64_t)(ok ? ls : 0);
    const uint64_t rg = o.mc.goff + (uint64_t)(ok ? rs : 0);

    // read bases are requested first and the flank windows right after, so the encode arithmetic
    // runs while the window gathers are still in flight (vmcnt retires in issue order)
    uint64_t rlo[W], rhi[W], rex[W];
    if (L <= 64 * W) {
        if (PACKED) encode_nibbles<W>(a.seq, so, L, rlo, rhi, rex); else encode_ascii<W>(a.seq, so, L, rlo, rhi, rex);
    } else {
#pragma unroll
        for (int w = 0; w < W; w++) rlo[w] = rhi[w] = rex[w] = 0;
    }
    uint64_t llo[W], lhi[W], qlo[W], qhi[W];
    load_window<W>(a.blocks, lg, llo, lhi);
    load_window<W>(a.blocks, rg, qlo, qhi);
    bool flagged = false;        // a window touches a 64-base block that holds a byte outside ACGT
    {
        // exception test, two levels.  Level 2 (one bit per 4096 bases, ~100 KB for hg38, L2-resident)
        // clears almost every window; only flagged ones read the per-64-base bitmap (6 MB for hg38:
        // it does not fit a 4 MiB XCD L2, and probing it for every read cost 1.5 of the 5.0 memory
        // requests per read that rocprofv3 counted).
        const uint64_t lb = lg >> 6, rb = rg >> 6, le = (lg + L - 1) >> 6, re = (rg + L - 1) >> 6;
        const uint32_t l0 = (uint32_t)(lb >> 6), l1 = (uint32_t)(le >> 6), r0 = (uint32_t)(rb >> 6), r1 = (uint32_t)(re >> 6);
        const uint32_t hit2 = ((a.coarse2[l0 >> 5] >> (l0 & 31)) | (a.coarse2[l1 >> 5] >> (l1 & 31)) |
                               (a.coarse2[r0 >> 5] >> (r0 & 31)) | (a.coarse2[r1 >> 5] >> (r1 & 31))) & 1u;
        if (hit2) {
            const int lnb = (int)(le - lb) + 1, rnb = (int)(re - rb) + 1;
            const uint64_t c0 = a.coarse[lb >> 6], c1 = a.coarse[(lb >> 6) + 1];
            const uint64_t d0 = a.coarse[rb >> 6], d1 = a.coarse[(rb >> 6) + 1];
            const int ls_ = (int)(lb & 63), rs_ = (int)(rb & 63);
            const uint64_t lbits = ((c0 >> ls_) | ((c1 << 1) << (63 - ls_))) & low_mask(lnb);
            const uint64_t rbits = ((d0 >> rs_) | ((d1 << 1) << (63 - rs_))) & low_mask(rnb);
            flagged = (lbits | rbits) != 0;
        }
    }
    if (!ok) {
        const uint32_t slot = atomicAdd(&a.counters[0], 1u);
        a.exact_list[slot] = (uint32_t)r;
        return;
    }
    // Reference bytes outside ACGT (N, soft-masked lowercase -- half of UCSC's hg38.fa --, IUPAC) stay on
    // this path: the comparison is on raw bytes (detect.rs:137-145), so such a base differs from every
    // read base that IS one of ACGT, which is all the packed planes need to know.  Only a read base that is
    // itself outside ACGT facing such a reference base (N against N) cannot be decided here.
    uint64_t lex[W], qex[W];
#pragma unroll
    for (int w = 0; w < W; w++) lex[w] = qex[w] = 0;
    if (flagged) {
        load_window_exc<W>(a.exc, lg, lex);
        load_window_exc<W>(a.exc, rg, qex);
    }

    uint64_t vm[W];
#pragma unroll
    for (int w = 0; w < W; w++) vm[w] = low_mask(min(max(L - 64 * w, 0), 64));

    if (o.swapped) {              // seq = revcomp(seq), detect.rs:112
        reverse_bits<W>(rlo, L); reverse_bits<W>(rhi, L); reverse_bits<W>(rex, L);
#pragma unroll
        for (int w = 0; w < W; w++) rhi[w] ^= vm[w];
    }
    if (!o.strand) {              // left flank read backwards + complemented, detect.rs:121
        reverse_bits<W>(llo, L); reverse_bits<W>(lhi, L);
#pragma unroll
        for (int w = 0; w < W; w++) lhi[w] ^= vm[w];
    }
    if (!o.mstrand) {             // detect.rs:130
        reverse_bits<W>(qlo, L); reverse_bits<W>(qhi, L);
#pragma unroll
        for (int w = 0; w < W; w++) qhi[w] ^= vm[w];
    }
    if (flagged) {                // the complement of a byte outside ACGT is outside ACGT: only the order changes
#pragma unroll
        for (int w = 0; w < W; w++) { lex[w] &= vm[w]; qex[w] &= vm[w]; }      // bits of the bases after the window
        if (!o.strand) reverse_bits<W>(lex, L);
        if (!o.mstrand) reverse_bits<W>(qex, L);
        uint64_t both = 0;
#pragma unroll
        for (int w = 0; w < W; w++) both |= rex[w] & (lex[w] | qex[w]);
        if (both) {               // non-ACGT read base against non-ACGT reference base: bytes decide
            const uint32_t slot = atomicAdd(&a.counters[0], 1u);
            a.exact_list[slot] = (uint32_t)r;
            return;
        }
    }

    // mismatch masks: a base outside "ACGT" on one side differs from every ACGT base on the other
    uint64_t ml[W], mr[W];
#pragma unroll
    for (int w = 0; w < W; w++) {
        ml[w] = ((rlo[w] ^ llo[w]) | (rhi[w] ^ lhi[w]) | rex[w] | lex[w]) & vm[w];
        mr[w] = ((rlo[w] ^ qlo[w]) | (rhi[w] ^ qhi[w]) | rex[w] | qex[w]) & vm[w];
    }

    // split scan, detect.rs:135-153: mm[A] by popcount, then the running sum; first minimum wins
    int mm = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t lowA = low_mask(min(max(A - 64 * w, 0), 64));
        mm += __popcll(ml[w] & lowA) + __popcll(mr[w] & ~lowA);
    }
    int best = mm, bp = A;
#pragma unroll
    for (int w = 0; w < W; w++) {
        // j = k-1 runs over [A, L-A-2]; inside word w that is jj in [j0, j1)
        const int j0 = max(A - 64 * w, 0), j1 = min(L - A - 1 - 64 * w, 64);
        if (j0 < j1) {
            uint64_t x = ml[w] >> j0, y = mr[w] >> j0;
            for (int jj = j0; jj < j1; jj++) {
                mm += (int)(x & 1) - (int)(y & 1);
                x >>= 1; y >>= 1;
                if (mm < best) { best = mm; bp = 64 * w + jj + 1; }
            }
        }
    }

    // signature, detect.rs:165-168: 8 reference bases left of bp and 8 right of it
    uint32_t sig = extract8<W>(llo, bp - 8) | (extract8<W>(lhi, bp - 8) << 8) |
                   (extract8<W>(qlo, bp) << 16) | (extract8<W>(qhi, bp) << 24);
    const bool sigx = flagged && (extract8<W>(lex, bp - 8) | extract8<W>(qex, bp)) != 0;
    if (sigx) {
        // a signature byte outside ACGT: the class is decided on the 16 raw bytes (case-sensitive,
        // detect.rs:223); the key carries their hash and the bytes are kept aside for the sweep
        // Codes and exception bits of the sixteen bases are in registers (oriented, complemented); what is missing is
        // which of the exceptions are one of "acgt" (the complement of a lower-case letter is lower case: one 8-bit
        // window of the lower-case plane per side, in genome order) and the bytes that are neither (bisection).
        uint8_t sb[16];
        const uint64_t lg0 = o.c.goff + (uint64_t)(o.strand ? o.pos - 1 + (bp - 8) : o.pos + A - 2 - (bp - 1));     // detect.rs:118,121
        const uint64_t rg0 = o.mc.goff + (uint64_t)(o.mstrand ? o.mpos + A - L - 1 + bp : o.mpos + L - 2 - (bp + 7));  // :127,130
        uint32_t llw = lower8(a.gb.lower, lg0), rlw = lower8(a.gb.lower, rg0);
        if (!o.strand) llw = __brev(llw) >> 24;
        if (!o.mstrand) rlw = __brev(rlw) >> 24;
        const uint32_t lx = extract8<W>(lex, bp - 8), rx = extract8<W>(qex, bp);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t lcode = ((sig >> k) & 1) | (((sig >> (8 + k)) & 1) << 1), rcode = ((sig >> (16 + k)) & 1) | (((sig >> (24 + k)) & 1) << 1);
            uint8_t cl = (uint8_t)(0x47544341u >> (8 * lcode)), cr = (uint8_t)(0x47544341u >> (8 * rcode));         // "ACTG"
            if ((lx >> k) & 1) {
                if ((llw >> k) & 1) cl |= 0x20;
                else { const uint8_t c = a.gb.at(o.strand ? lg0 + k : lg0 + 7 - k); cl = o.strand ? c : comp_byte(c); }
            }
            if ((rx >> k) & 1) {
                if ((rlw >> k) & 1) cr |= 0x20;
                else { const uint8_t c = a.gb.at(o.mstrand ? rg0 + k : rg0 + 7 - k); cr = o.mstrand ? c : comp_byte(c); }
            }
            sb[k] = cl; sb[8 + k] = cr;
        }
        sig = sigx_hash(sb);
        const uint32_t slot = atomicAdd(&a.counters[1], 1u);
#pragma unroll
        for (int k = 0; k < 16; k++) a.sigx[(uint64_t)slot * 16 + k] = sb[k];
        a.sigx_slot[r] = slot;
    }

    // coordinates, detect.rs:172-181 (signed; a negative value is the reference's wrapped usize)
    const int64_t lp = o.strand ? o.pos + bp - 1 : o.pos + A - bp;
    const int64_t rp = o.mstrand ? o.mpos + A - (L - bp) : o.mpos + (L - bp);

    a.key[r] = make_key(o, sigx, sig, a.cb);
    a.lpos[r] = (int32_t)lp;
    a.rpos[r] = (int32_t)rp;
    a.bp[r] = (uint16_t)bp;
    const uint16_t fl = (uint16_t)((o.strand ? kEvStrand : 0) | (o.mstrand ? kEvMstrand : 0) | (o.swapped ? kEvSwapped : 0) |
                                   (sigx ? kEvSigx : 0));
    uint64_t* rec = a.meta + r * (uint64_t)a.meta_stride;
    uint32_t n_lower = 0;        // bases printed in lower case (detect.rs:157,161)
#pragma unroll
    for (int w = 0; w < W; w++) {
        const uint64_t lowB = low_mask(min(max(bp - 64 * w, 0), 64));
        const uint64_t m = (ml[w] & lowB) | (mr[w] & ~lowB);
        if (w < a.meta_stride - 1) rec[w] = m;
        n_lower += (uint32_t)__popcll(m);
    }
    for (int w = W; w < a.meta_stride - 1; w++) rec[w] = 0;
    uint32_t n_exc = 0;          // read bases outside "ACGT": what the packed junction record lists one by one
#pragma unroll
    for (int w = 0; w < W; w++) n_exc += (uint32_t)__popcll(rex[w]);
    if (!junction_packable((uint32_t)L, n_exc, 64u * W)) atomicAdd(&a.counters[2], 1u);
    a.rw[r] = RecWord{read_info(a.by_ref ? junction_ref_bytes((uint32_t)L, n_lower) : junction_record_bytes((uint32_t)L, n_exc, n_lower, 64u * W), fl, (uint32_t)bp, (uint32_t)L), fh};
    rec[a.meta_stride - 1] = (uint64_t)(uint16_t)bp | ((uint64_t)fl << 16) | ((uint64_t)(uint16_t)L << 32) | ((uint64_t)n_exc << 48);
    a.flags[r] = fl;
}

// ---------------------------------------------------------------------------------------------
// Byte-exact path: one wave per routed read, on the bytes of the FASTA (rebuilt: GenomeBytes).  Handles windows that
// leave the contig (-> 'N', detect.rs:118-130), reference bytes outside ACGT (soft-masked, N,
// IUPAC), and reads longer than the packed path's 64*W.
struct ExactArgs {
    uint64_t n_total;
    int A;
    int cb;
    const uint32_t* exact_list;
    uint32_t* counters;      // [0] = length of exact_list, [1] = raw-signature slots handed out, [3] = list entries
                             // earlier launches have done (reads localised behind their upload add to the list)
    const bf_hit* hits;
    const uint8_t* seq;
    const uint64_t* seq_off;
    const uint16_t* len;     // read lengths when the reads are 4-bit codes
    uint32_t packed;
    const ContigDev* contigs;
    GenomeBytes gb;
    uint64_t* key;
    int32_t* lpos;
    int32_t* rpos;
    uint16_t* bp;
    uint16_t* flags;
    uint64_t* meta;
    int mask_words;          // record stride = mask_words + 1
    uint32_t max_packed;     // longest read a packed junction record can hold
    uint32_t by_ref;         // junction records name the read (BF_FLAG_READS_ON_HOST)
    uint8_t* sigx;           // 16 bytes per slot
    uint32_t* sigx_slot;     // read -> slot
    RecWord* rw;             // read_info() word + fragment-id hash
    const uint64_t* fhash;
};

struct ExactRead {
    const uint8_t* seq; uint64_t so; bool packed; int L; bool swapped;
    __device__ __forceinline__ uint8_t raw_base(int k) const { return base_at(seq, so, k, packed); }
    GenomeBytes gb;
    uint64_t lgoff; int64_t llen; int64_t lbase; bool lfwd;
    uint64_t rgoff; int64_t rlen; int64_t rbase; bool rfwd;
    __device__ __forceinline__ uint8_t s(int k) const {
        return swapped ? comp_byte(raw_base(L - 1 - k)) : raw_base(k);
    }
    __device__ __forceinline__ uint8_t left(int k) const {
        const int64_t i = lfwd ? lbase + k : lbase - k;
        const uint8_t c = (i >= 0 && i < llen) ? gb.at(lgoff + (uint64_t)i) : (uint8_t)'N';
        return lfwd ? c : comp_byte(c);
    }
    __device__ __forceinline__ uint8_t right(int k) const {
        const int64_t i = rfwd ? rbase + k : rbase - k;
        const uint8_t c = (i >= 0 && i < rlen) ? gb.at(rgoff + (uint64_t)i) : (uint8_t)'N';
        return rfwd ? c : comp_byte(c);
    }
};

// One WAVE per routed read, any read length.  Lane l looks at base 64c + l of chunk c; a wave ballot of
// "read byte != flank byte" IS the chunk's 64-bit mismatch word, so the arithmetic below is the packed
// kernel's (popcounts for mm[A], prefix popcounts for the running sum, lowest k among the minima), only
// with words that come from byte compares.  Three passes over the chunks (mm[A]; the scan; the combined
// mask once bp is known): the bytes are cache-resident after the first.
__global__ void __launch_bounds__(256) localise_exact_kernel(const ExactArgs a) {
    const int lane = threadIdx.x & 63;
    // fixed grid, every wave takes list entries wave, wave + n_waves, ...: the list's length stays on the device
    // (the host does not wait for the packed kernel to learn it)
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6, n_list = a.counters[0], first = a.counters[3];
    for (uint32_t slot = first + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6); slot < n_list; slot += n_waves) {     // wave-uniform
    const uint32_t r = a.exact_list[slot];
    const bf_hit h = a.hits[r];
    const Oriented o = orient(h, a.contigs);
    const int A = a.A;
    ExactRead x;
    x.seq = a.seq; x.so = a.seq_off[r]; x.packed = a.packed != 0;
    x.L = a.packed ? (int)a.len[r] : (int)(a.seq_off[r + 1] - a.seq_off[r]);
    x.swapped = o.swapped;
    const int L = x.L;
    x.gb = a.gb; x.lgoff = o.c.goff; x.llen = o.c.len; x.lfwd = o.strand;
    x.lbase = o.strand ? o.pos - 1 : o.pos + A - 2;                 // detect.rs:118,121
    x.rgoff = o.mc.goff; x.rlen = o.mc.len; x.rfwd = o.mstrand;
    x.rbase = o.mstrand ? o.mpos + A - L - 1 : o.mpos + L - 2;      // detect.rs:127,130
    const int n_chunks = (L + 63) >> 6;

    auto words = [&](int c, uint64_t& ml, uint64_t& mr) {            // mismatch words of chunk c (wave-uniform results)
        const int k = 64 * c + lane;
        bool dl = false, dr = false;
        if (k < L) { const uint8_t sb = x.s(k); dl = sb != x.left(k); dr = sb != x.right(k); }
        ml = __ballot(dl); mr = __ballot(dr);
    };
    auto low = [](int nbits) -> uint64_t { return nbits <= 0 ? 0ull : nbits >= 64 ? ~0ull : ((1ull << nbits) - 1); };

    // mm[A] = mismatches of [0, A) against the left flank + of [A, L) against the right flank (detect.rs:137-146)
    int mm = 0;
    uint32_t n_exc = 0;                                              // read bases outside "ACGT" (junction record)
    for (int c = 0; c < n_chunks; c++) {
        uint64_t ml, mr; words(c, ml, mr);
        const uint64_t lowA = low(A - 64 * c);
        mm += __popcll(ml & lowA) + __popcll(mr & ~lowA);
        const int k = 64 * c + lane;
        n_exc += (uint32_t)__popcll(__ballot(k < L && !is_acgt(x.raw_base(k))));
    }
    n_exc = min(n_exc, 0xffffu);
    // candidates k in [A+1, L-A): mm[k] = mm[k-1] + ml[k-1] - mr[k-1]; j = k-1 runs over [A, L-A-2]
    uint64_t best = ((uint64_t)(uint32_t)mm << 32) | (uint32_t)A;    // (mm, k): lexicographic min = first minimum
    for (int c = A >> 6; c < n_chunks && 64 * c <= L - A - 2; c++) {
        uint64_t ml, mr; words(c, ml, mr);
        const int j0 = max(A - 64 * c, 0), j1 = min(L - A - 1 - 64 * c, 64);     // [j0, j1) inside this chunk
        if (j0 >= j1) continue;
        const uint64_t range = low(j1) & ~low(j0);
        const uint64_t upto = low(lane + 1) & range;                 // steps j0 .. lane of this chunk
        uint64_t cand = ~0ull;
        if (lane >= j0 && lane < j1) {
            const int v = mm + __popcll(ml & upto) - __popcll(mr & upto);
            cand = ((uint64_t)(uint32_t)v << 32) | (uint32_t)(64 * c + lane + 1);
        }
        for (int d = 32; d > 0; d >>= 1) {
            const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(cand >> 32), d) << 32) |
                                   (uint32_t)__shfl_xor((int)(uint32_t)cand, d);
            cand = min(cand, other);
        }
        best = min(best, cand);
        mm += __popcll(ml & range) - __popcll(mr & range);
    }
    const int bp = (int)(uint32_t)best;

    // combined mask: left-flank mismatches below bp, right-flank mismatches from bp on (detect.rs:155-162)
    uint32_t n_lower = 0;
    for (int c = 0; c < a.mask_words; c++) {
        uint64_t ml = 0, mr = 0;
        if (c < n_chunks) words(c, ml, mr);
        const uint64_t lowB = low(bp - 64 * c);
        const uint64_t m = (ml & lowB) | (mr & ~lowB);
        n_lower += (uint32_t)__popcll(m);
        if (lane == 0) a.meta[(uint64_t)r * (a.mask_words + 1) + c] = m;
    }

    // signature (detect.rs:165-168): lanes 0..7 hold left[bp-8+k], lanes 8..15 right[bp+k-8]
    uint8_t sc = 'A';
    if (lane < 8) sc = x.left(bp - 8 + lane);
    else if (lane < 16) sc = x.right(bp + lane - 8);
    const bool plain = (__ballot(lane < 16 && !is_acgt(sc)) == 0);
    const uint32_t code = (sc >> 1) & 3;
    const uint64_t blo = __ballot(lane < 16 && (code & 1)), bhi = __ballot(lane < 16 && (code >> 1));
    uint32_t sig = (uint32_t)(blo & 0xff) | ((uint32_t)(bhi & 0xff) << 8) | ((uint32_t)((blo >> 8) & 0xff) << 16) |
                   ((uint32_t)((bhi >> 8) & 0xff) << 24);
    if (!plain) {
        uint8_t sigb[16];
#pragma unroll
        for (int k = 0; k < 16; k++) sigb[k] = (uint8_t)__shfl((int)sc, k);
        sig = sigx_hash(sigb);
        uint32_t xs = 0;
        if (lane == 0) xs = atomicAdd(&a.counters[1], 1u);
        xs = (uint32_t)__shfl((int)xs, 0);
        if (lane < 16) a.sigx[(uint64_t)xs * 16 + lane] = sc;
        if (lane == 0) a.sigx_slot[r] = xs;
    }
    if (lane != 0) continue;
    const int64_t lp = o.strand ? o.pos + bp - 1 : o.pos + A - bp;
    const int64_t rp = o.mstrand ? o.mpos + A - (L - bp) : o.mpos + (L - bp);
    a.key[r] = make_key(o, !plain, sig, a.cb);
    a.lpos[r] = (int32_t)lp;
    a.rpos[r] = (int32_t)rp;
    a.bp[r] = (uint16_t)bp;
    const uint16_t fl = (uint16_t)((o.strand ? kEvStrand : 0) | (o.mstrand ? kEvMstrand : 0) |
                                   (o.swapped ? kEvSwapped : 0) | kEvExact | (plain ? 0 : kEvSigx));
    a.flags[r] = fl;
    if (!junction_packable((uint32_t)L, n_exc, a.max_packed)) atomicAdd(&a.counters[2], 1u);
    a.rw[r] = RecWord{read_info(a.by_ref ? junction_ref_bytes((uint32_t)L, n_lower) : junction_record_bytes((uint32_t)L, n_exc, n_lower, a.max_packed), fl, (uint32_t)bp, (uint32_t)L), a.fhash[r]};
    a.meta[(uint64_t)r * (a.mask_words + 1) + a.mask_words] =
        (uint64_t)(uint16_t)bp | ((uint64_t)fl << 16) | ((uint64_t)(uint16_t)L << 32) | ((uint64_t)n_exc << 48);
    }
}

__global__ void exact_done_kernel(uint32_t* counters) { counters[3] = counters[0]; }

// ---------------------------------------------------------------------------------------------
// Junction records of surviving reads (format: include/breakfast_hip.h, "Junction records").
// Record sizes from the render records' last word {bp, flags, len, n_exc}.
__global__ void __launch_bounds__(256) render_len_kernel(uint64_t n, const uint32_t* __restrict__ idx,
                                                         const RecWord* __restrict__ rw, uint32_t* __restrict__ out_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out_len[i] = info_jrec(rw[idx[i]].info);
}

// lengths for a caller-supplied (device) index list: an index outside the resident reads is counted,
// not dereferenced
__global__ void __launch_bounds__(256) render_len_checked_kernel(uint64_t n, const uint32_t* __restrict__ idx,
                                                                 const RecWord* __restrict__ rw, uint64_t n_reads,
                                                                 uint32_t* __restrict__ out_len, uint32_t* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = idx[i];
    if (r >= n_reads) { out_len[i] = 0; atomicAdd(bad, 1u); return; }
    out_len[i] = info_jrec(rw[r].info);
}

// render: 16 lanes per surviving read, 4 reads per wave, 16 reads per block.  The junction string is
// produced in tiles of 64 ALIGNED output dwords: the group stages the source bytes the tile needs
// (aligned dword loads of the read, forwards or backwards) and the mask words in LDS, then every
// lane assembles whole output dwords from LDS bytes -- complement when the read was reoriented
// (detect.rs:112), lower case where the mask says mismatch (:157,161), '|' at the breakpoint -- and
// stores them with one dword store; only the first and last dword of a string, which it shares with
// its neighbours in the output, are stored bytewise.  Everything a read needs besides its bases sits
// in ONE render record (mask words + bp/flags/len).
constexpr int kRenderTileDw = 64;                            // output dwords per tile
constexpr int kRenderStageDw = kRenderTileDw + 4;            // source dwords a tile can touch
constexpr int kRenderMaskW = (4 * kRenderTileDw) / 64 + 2;   // mask words a tile can touch

// 8 waves/SIMD (64 VGPRs, 40 bytes of scratch) instead of the 4 the compiler picks at 100 VGPRs: the kernel
// waits on scattered loads, 1.65 -> 1.35 ms per 3.9M reads.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
render_kernel(uint64_t n, const uint32_t* __restrict__ idx,
                                                     const uint64_t* __restrict__ joff,
                                                     const uint8_t* __restrict__ seq,
                                                     const uint64_t* __restrict__ seq_off,
                                                     const uint64_t* __restrict__ meta, int meta_stride,
                                                     uint32_t max_packed, uint8_t* __restrict__ out, uint64_t cap) {
    __shared__ uint8_t comp[256];
    __shared__ uint32_t stage[16][kRenderStageDw];
    __shared__ uint64_t smask[16][kRenderMaskW];
    comp[threadIdx.x] = kComp[threadIdx.x];
    __syncthreads();
    const int g = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const uint64_t i = (uint64_t)blockIdx.x * 16 + g;
    if (i >= n) return;
    const uint32_t r = idx[i];
    const uint64_t o = seq_off[r];
    const uint64_t* rec = meta + (uint64_t)r * meta_stride;
    const uint64_t tail = rec[meta_stride - 1];
    const int64_t bp = (int64_t)(tail & 0xffff), L = (int64_t)((tail >> 32) & 0xffff);
    const bool swapped = ((tail >> 16) & kEvSwapped) != 0;
    if (junction_packable((uint32_t)L, (uint32_t)(tail >> 48), max_packed)) return;      // render_packed_kernel's
    if (joff[i] >= cap) return;                                          // (beyond the caller's buffer: the caller reports it)
    if (l16 == 0) {                                                      // text record: {head, len} + the string
        uint32_t* hd = (uint32_t*)(out + joff[i]);                       // records start on 4-byte boundaries
        hd[0] = junction_head_text(swapped, (uint32_t)bp); hd[1] = (uint32_t)L;
        for (int64_t q = 8 + L + 1; q & 3; q++) out[joff[i] + q] = 0;    // padding
    }
    const int64_t dst0 = (int64_t)joff[i] + 8, out_len = L + 1;
    const int64_t first_dw = dst0 >> 2, last_dw = (dst0 + out_len - 1) >> 2;
    const uint8_t* sbytes = (const uint8_t*)stage[g];
    for (int64_t t0 = first_dw; t0 <= last_dw; t0 += kRenderTileDw) {
        // output positions [q_lo, q_hi) of this tile need read bases [k_lo, k_hi)
        const int64_t q_lo = max((int64_t)0, 4 * t0 - dst0), q_hi = min(out_len, 4 * (t0 + kRenderTileDw) - dst0);
        const int64_t k_lo = max((int64_t)0, q_lo - 1), k_hi = min(L, q_hi);
        const int64_t src_lo = (int64_t)o + (swapped ? L - k_hi : k_lo), src_hi = (int64_t)o + (swapped ? L - k_lo : k_hi);
        const int64_t base_dw = src_lo >> 2;
        const int n_dw = (int)(((src_hi + 3) >> 2) - base_dw);
        for (int d = l16; d < n_dw; d += 16) stage[g][d] = ((const uint32_t*)seq)[base_dw + d];
        const int64_t w_lo = k_lo >> 6;
        const int n_w = k_hi > k_lo ? (int)(((k_hi - 1) >> 6) - w_lo) + 1 : 0;
        if (l16 < n_w) smask[g][l16] = rec[w_lo + l16];
        // LDS operations of one wave execute in order; only the compiler must not reorder them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int w = l16; w < kRenderTileDw; w += 16) {
            const int64_t dw = t0 + w;
            if (dw > last_dw) break;
            uint32_t val = 0, valid = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int64_t q = 4 * dw + b - dst0;
                if (q < 0 || q >= out_len) continue;
                uint32_t c = '|';
                if (q != bp) {
                    const int64_t k = q < bp ? q : q - 1;
                    const int64_t sa = (int64_t)o + (swapped ? L - 1 - k : k);
                    c = sbytes[sa - 4 * base_dw];
                    if (swapped) c = comp[c];
                    const uint64_t m = smask[g][(k >> 6) - w_lo];
                    if (((m >> (k & 63)) & 1) && c >= 'A' && c <= 'Z') c += 32;       // to_ascii_lowercase
                }
                val |= c << (8 * b);
                valid |= 1u << b;
            }
            if (valid == 0xfu) {
                ((uint32_t*)out)[dw] = val;
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (valid & (1u << b)) out[4 * dw + b] = (uint8_t)(val >> (8 * b));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Text records of reads that are resident as 4-bit codes: 16 lanes per read, a byte at a time (such records are rare:
// reads above the packed kernel's 256 bases, or with more than one base in eight outside ACGT).
__global__ void __launch_bounds__(256) render_text_nibbles_kernel(uint64_t n, const uint32_t* __restrict__ idx,
                                                                  const uint64_t* __restrict__ joff,
                                                                  const uint8_t* __restrict__ seq,
                                                                  const uint64_t* __restrict__ seq_off,
                                                                  const uint64_t* __restrict__ meta, int meta_stride,
                                                                  uint32_t max_packed, uint8_t* __restrict__ out, uint64_t cap) {
    const int g = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const uint64_t i = (uint64_t)blockIdx.x * 16 + g;
    if (i >= n) return;
    const uint32_t r = idx[i];
    const uint64_t so = seq_off[r];
    const uint64_t* rec = meta + (uint64_t)r * meta_stride;
    const uint64_t tail = rec[meta_stride - 1];
    const int bp = (int)(tail & 0xffff), L = (int)((tail >> 32) & 0xffff);
    const bool swapped = ((tail >> 16) & kEvSwapped) != 0;
    if (junction_packable((uint32_t)L, (uint32_t)(tail >> 48), max_packed)) return;      // render_packed_kernel's
    if (joff[i] >= cap) return;
    uint8_t* o = out + joff[i];
    if (l16 == 0) {
        uint32_t* hd = (uint32_t*)o;
        hd[0] = junction_head_text(swapped, (uint32_t)bp); hd[1] = (uint32_t)L;
        for (int q = 8 + L + 1; q & 3; q++) o[q] = 0;                                    // padding
    }
    for (int q = l16; q <= L; q += 16) {                                                 // L + 1 output bytes
        uint32_t c = '|';
        if (q != bp) {
            const int k = q < bp ? q : q - 1;
            c = swapped ? comp_byte(base_at(seq, so, L - 1 - k, true)) : base_at(seq, so, k, true);
            if (((rec[k >> 6] >> (k & 63)) & 1) && c >= 'A' && c <= 'Z') c += 32;          // to_ascii_lowercase
        }
        o[8 + q] = (uint8_t)c;
    }
}

// Packed junction records: one lane per surviving read whose record is packable (at most 64*W bases, few
// bases outside ACGT).  The lane re-encodes its read from the ASCII bases exactly as the localise kernel did
// (same function, so the exception count in the render record is the one found here), re-orients it when the
// read was reverse-complemented (detect.rs:112), and writes the record of include/breakfast_hip.h ("Junction
// records"): a 4-byte header, per 8 bases {lo byte, hi byte} -- or {lo, hi, lower-case} when the lower-case bases are
// many --, the bases outside ACGT as {position, printed byte}, the lower-case positions as bytes
// -- 44 bytes for a 150-base read with one mismatch against 151 bytes of text: the result copy to the host
// is the longest stage of a step, and the .sv formatter expands records at memcpy speed.
template <int W>
__global__ void __launch_bounds__(256) render_packed_kernel(uint64_t n, const uint32_t* __restrict__ idx,
                                                            const uint64_t* __restrict__ joff,
                                                            const uint8_t* __restrict__ seq,
                                                            const uint64_t* __restrict__ seq_off,
                                                            const uint64_t* __restrict__ meta, int meta_stride,
                                                            uint8_t* __restrict__ out, bool packed, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t at0 = joff[i];
    if (at0 >= cap) return;                                              // (beyond the caller's buffer: the caller reports it)
    const uint32_t r = idx[i];
    const uint64_t* __restrict__ rec = meta + (uint64_t)r * meta_stride;
    const uint64_t tail = rec[meta_stride - 1];
    const int L = (int)((tail >> 32) & 0xffff), bp = (int)(tail & 0xffff);
    const uint32_t n_exc = (uint32_t)(tail >> 48);
    if (!junction_packable((uint32_t)L, n_exc, 64u * W)) return;         // text record: render_kernel's
    const bool swapped = ((tail >> 16) & kEvSwapped) != 0;
    const uint64_t so = seq_off[r];
    uint64_t lo[W], hi[W], ex[W], cs[W];
    encode_read<W>(seq, so, L, packed, lo, hi, ex);
#pragma unroll
    for (int w = 0; w < W; w++) cs[w] = w < meta_stride - 1 ? rec[w] : 0ull;
    if (swapped) {
        reverse_bits<W>(lo, L); reverse_bits<W>(hi, L); reverse_bits<W>(ex, L);
#pragma unroll
        for (int w = 0; w < W; w++) hi[w] ^= low_mask(min(max(L - 64 * w, 0), 64));
    }
    uint32_t n_lower = 0;
#pragma unroll
    for (int w = 0; w < W; w++) n_lower += (uint32_t)__popcll(cs[w]);
    const bool sparse = junction_sparse((uint32_t)L, n_lower);
    // the record as a byte stream, written a dword at a time (zero padding at the end comes with the last one)
    uint32_t* __restrict__ o32 = (uint32_t*)(out + at0);
    uint64_t acc = 0;
    int have = 0, at = 0;
    auto put = [&](uint32_t v, int nbytes) {
        acc |= (uint64_t)v << (8 * have);
        have += nbytes;
        if (have >= 4) { o32[at++] = (uint32_t)acc; acc >>= 32; have -= 4; }
    };
    put(junction_head_packed(sparse, swapped, (uint32_t)L, (uint32_t)bp, n_exc, n_lower), 4);
#pragma unroll
    for (int u = 0; u < 8 * W; u++) {                                    // 8 bases a step: {lo, hi[, lower case]} bytes
        if (8 * u < L) {
            const int sh = 8 * (u & 7);
            const uint32_t l8 = (uint32_t)(lo[u >> 3] >> sh) & 0xffu, h8 = (uint32_t)(hi[u >> 3] >> sh) & 0xffu;
            if (sparse) put(l8 | (h8 << 8), 2);
            else put(l8 | (h8 << 8) | (((uint32_t)(cs[u >> 3] >> sh) & 0xffu) << 16), 3);
        }
    }
    if (n_exc) {                                                         // rare: N and other IUPAC letters, as printed
#pragma unroll
        for (int w = 0; w < W; w++) {
            uint64_t m = ex[w];
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int k = 64 * w + b;
                uint32_t c = swapped ? comp_byte(base_at(seq, so, L - 1 - k, packed)) : base_at(seq, so, k, packed);
                if (((cs[w] >> b) & 1) && c >= 'A' && c <= 'Z') c += 32;  // to_ascii_lowercase, detect.rs:157,161
                put((uint32_t)k | (c << 8), 2);
            }
        }
    }
    if (sparse && n_lower) {                                             // lower-case positions, a byte each
#pragma unroll
        for (int w = 0; w < W; w++) {
            uint64_t m = cs[w];
            while (m) {
                put(64u * w + (uint32_t)__builtin_ctzll(m), 1);
                m &= m - 1;
            }
        }
    }
    if (have) o32[at] = (uint32_t)acc;
}

// Reference records (BF_FLAG_READS_ON_HOST): the caller still holds the bases, so a record is the read's number, the
// breakpoint, the orientation and where the lower-case bases are -- everything from the read's render record, no
// gather of the read itself.  One lane per surviving read.
__global__ void __launch_bounds__(256) render_ref_kernel(uint64_t n, const uint32_t* __restrict__ idx,
                                                         const uint64_t* __restrict__ joff,
                                                         const uint64_t* __restrict__ meta, int meta_stride,
                                                         uint8_t* __restrict__ out, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || joff[i] >= cap) return;
    const uint32_t r = idx[i];
    const uint64_t* __restrict__ rec = meta + (uint64_t)r * meta_stride;
    // the record's first words and its tail in one round trip (the usual record has three mask words)
    uint64_t m4[4];
#pragma unroll
    for (int w = 0; w < 4; w++) m4[w] = w < meta_stride - 1 ? rec[w] : 0ull;
    const uint64_t tail = rec[meta_stride - 1];
    const uint32_t L = (uint32_t)((tail >> 32) & 0xffff), bp = (uint32_t)(tail & 0xffff);
    const bool swapped = ((tail >> 16) & kEvSwapped) != 0;
    const int words = min(meta_stride - 1, (int)((L + 63u) >> 6));
    auto word = [&](int w) -> uint64_t { return w == 0 ? m4[0] : w == 1 ? m4[1] : w == 2 ? m4[2] : w == 3 ? m4[3] : rec[w]; };
    uint32_t n_lower = 0;
    for (int w = 0; w < words; w++) n_lower += (uint32_t)__popcll(word(w));
    uint32_t* __restrict__ o32 = (uint32_t*)(out + joff[i]);
    uint64_t acc = 0;
    int have = 0, at = 0;
    auto put = [&](uint32_t v, int nbytes) {
        acc |= (uint64_t)v << (8 * have);
        have += nbytes;
        if (have >= 4) { o32[at++] = (uint32_t)acc; acc >>= 32; have -= 4; }
    };
    put(junction_head_ref(swapped, L, n_lower, bp), 4);
    put(r, 4);
    if (n_lower >= kJrRefDense) {
        put(L, 4);
        for (uint32_t b = 0; b < ((L + 7u) >> 3); b++) put((uint32_t)(rec[b >> 3] >> (8 * (b & 7))) & 0xffu, 1);
    } else {
        const int pb = L > 256u ? 2 : 1;
        for (int w = 0; w < words; w++) {
            uint64_t m = word(w);
            while (m) {
                put(64u * (uint32_t)w + (uint32_t)__builtin_ctzll(m), pb);
                m &= m - 1;
            }
        }
    }
    if (have) o32[at] = (uint32_t)acc;
}

// ---------------------------------------------------------------------------------------------
// host side

static inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

int genome_pack(bf_ctx* ctx) {
    const uint64_t n_blocks = ctx->genome_bases / 64;     // multiple of 64 by construction
    const uint64_t n_waves = n_blocks / 64;
    Scope t(ctx, "genome_pack");
    DevBuf wave_runs, wave_off;
    BF_TRY(ensure(ctx, ctx->g_lower, (n_blocks + 1) * 8));
    BF_HIP(ctx, hipMemsetAsync(ctx->g_lower.as<uint64_t>() + n_blocks, 0, 8, ctx->stream));
    BF_TRY(ensure(ctx, wave_runs, n_waves * 4));
    if (int rc = ensure(ctx, wave_off, (n_waves + 1) * 8)) { release(ctx, wave_runs); return rc; }
    auto done = [&](int rc) { release(ctx, wave_runs); release(ctx, wave_off); return rc; };
    hipLaunchKernelGGL(genome_pack_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, ctx->stream,
                       ctx->g_raw.as<uint8_t>(), ctx->g_blocks.as<GenomeBlock>(), ctx->g_exc.as<uint64_t>(), ctx->g_lower.as<uint64_t>(),
                       ctx->g_coarse.as<uint64_t>(), ctx->g_coarse2.as<uint32_t>(), wave_runs.as<uint32_t>(), n_blocks);
    if (hipGetLastError() != hipSuccess) return done(fail(ctx, BF_ERR_HIP, "genome_pack_kernel failed to launch"));
    // the bytes that are none of ACGTacgt, as runs: count per wave -> offsets -> the table, sized exactly
    if (int rc = scan_u32_to_u64(ctx, wave_runs.as<uint32_t>(), wave_off.as<uint64_t>(), n_waves)) return done(rc);
    uint64_t total = 0;
    if (hipMemcpyAsync(&total, wave_off.as<uint64_t>() + n_waves, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return done(fail(ctx, BF_ERR_HIP, "genome_pack: reading the run count back failed"));
    if (total > 0xfffffff0ull) return done(fail(ctx, BF_ERR_RANGE, "the genome holds %llu runs of bytes outside ACGTacgt; the supported maximum is 2^32 - 16", (unsigned long long)total));
    ctx->n_runs = (uint32_t)total;
    if (int rc = ensure(ctx, ctx->g_run_start, (total + 1) * 8)) return done(rc);
    if (int rc = ensure(ctx, ctx->g_run_byte, total + 16)) return done(rc);
    if (total) {
        hipLaunchKernelGGL(genome_runs_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, ctx->stream,
                           ctx->g_raw.as<uint8_t>(), wave_runs.as<uint32_t>(), wave_off.as<uint64_t>(),
                           ctx->g_run_start.as<uint64_t>(), ctx->g_run_byte.as<uint8_t>(), n_blocks);
        if (hipGetLastError() != hipSuccess) return done(fail(ctx, BF_ERR_HIP, "genome_runs_kernel failed to launch"));
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return done(fail(ctx, BF_ERR_HIP, "genome_pack failed"));
    release(ctx, ctx->g_raw);                             // every byte can be rebuilt from what is resident now
    return done(BF_OK);
}

// The same images from a packed genome image (genome_image.hip): thread t takes block t's two code words as they are and
// builds its words of the lower-case and exception planes from the runs that reach into it (bisection for the first such
// run; runs ascend and do not overlap -- damaged tables can set wrong bits, never touch memory outside the arrays).
__device__ __forceinline__ uint64_t run_bits(const uint64_t* __restrict__ start, const uint32_t* __restrict__ len, uint64_t n, uint64_t g0) {
    uint64_t lo = 0, hi = n;                                           // first run that ends after g0
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (start[mid] + len[mid] <= g0) lo = mid + 1; else hi = mid;
    }
    uint64_t bits = 0;
    for (uint64_t i = lo; i < n; i++) {
        const uint64_t s = start[i];
        if (s >= g0 + 64) break;
        const uint64_t e = min(s + (uint64_t)len[i], g0 + 64), a = max(s, g0);
        if (e > a) bits |= low_mask((int)(e - g0)) & ~low_mask((int)(a - g0));
    }
    return bits;
}

__global__ void __launch_bounds__(256) genome_image_kernel(const GenomeBlock* __restrict__ plain,
                                                           const uint64_t* __restrict__ lower_start, const uint32_t* __restrict__ lower_len, uint64_t n_lower,
                                                           const uint64_t* __restrict__ other_start, const uint32_t* __restrict__ other_len, uint64_t n_other,
                                                           GenomeBlock* __restrict__ blocks, uint64_t* __restrict__ exc_plane,
                                                           uint64_t* __restrict__ lower_plane, uint64_t* __restrict__ coarse,
                                                           uint32_t* __restrict__ coarse2, uint32_t* __restrict__ bad, uint64_t n_blocks) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_blocks) return;  // whole waves only
    const GenomeBlock b = plain[t];
    blocks[block_slot(t)] = b;
    if (t >= 4) blocks[block_slot(t - 4) + 4] = b;
    const uint64_t lw = run_bits(lower_start, lower_len, n_lower, t * 64), ot = run_bits(other_start, other_len, n_other, t * 64);
    const uint64_t ex = lw | ot;
    if (lw & ot) atomicOr(bad, 1u);                      // a base in a run of both kinds: not an image this library wrote
    if (ex) {
        exc_plane[exc_slot(t)] = ex;
        if (t >= 8) exc_plane[exc_slot(t - 8) + 8] = ex;
    }
    lower_plane[t] = lw;
    const uint64_t flags = __ballot(ex != 0);
    if ((threadIdx.x & 63) == 0) {
        coarse[t >> 6] = flags;
        if (flags) atomicOr(&coarse2[t >> 11], 1u << ((t >> 6) & 31));
    }
}

// runs ascend, do not overlap and end inside the genome
__global__ void __launch_bounds__(256) runs_check_kernel(const uint64_t* __restrict__ start, const uint32_t* __restrict__ len, uint64_t n,
                                                         uint64_t genome_bases, uint32_t* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = start[i] + len[i];
    if (len[i] == 0 || e > genome_bases || (i + 1 < n && start[i + 1] < e)) atomicOr(bad, 2u);
}

int genome_from_image(bf_ctx* ctx) {
    const uint64_t n_blocks = ctx->genome_bases / 64;
    Scope t(ctx, "genome_pack");
    DevBuf bad;
    BF_TRY(ensure(ctx, ctx->g_lower, (n_blocks + 1) * 8));
    BF_TRY(ensure(ctx, bad, 16));
    auto done = [&](int rc) {
        release(ctx, bad); release(ctx, ctx->g_raw); release(ctx, ctx->g_lower_start); release(ctx, ctx->g_lower_len); release(ctx, ctx->g_run_len);
        if (rc != BF_OK) ctx->n_contigs = 0;              // the staged image is gone: the next genome starts with its own begin
        return rc;
    };
    uint32_t h_bad = 0;
    const uint64_t nl = ctx->n_lower_runs, no = ctx->n_runs;
    bool ok = hipMemsetAsync(ctx->g_lower.as<uint64_t>() + n_blocks, 0, 8, ctx->stream) == hipSuccess &&
              hipMemsetAsync(bad.p, 0, 16, ctx->stream) == hipSuccess;
    if (ok) {
        if (nl) hipLaunchKernelGGL(runs_check_kernel, dim3(grid_for(nl, 256)), dim3(256), 0, ctx->stream, ctx->g_lower_start.as<uint64_t>(), ctx->g_lower_len.as<uint32_t>(), nl, ctx->genome_bases, bad.as<uint32_t>());
        if (no) hipLaunchKernelGGL(runs_check_kernel, dim3(grid_for(no, 256)), dim3(256), 0, ctx->stream, ctx->g_run_start.as<uint64_t>(), ctx->g_run_len.as<uint32_t>(), no, ctx->genome_bases, bad.as<uint32_t>());
        hipLaunchKernelGGL(genome_image_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, ctx->stream, ctx->g_raw.as<GenomeBlock>(),
                           ctx->g_lower_start.as<uint64_t>(), ctx->g_lower_len.as<uint32_t>(), nl,
                           ctx->g_run_start.as<uint64_t>(), ctx->g_run_len.as<uint32_t>(), no,
                           ctx->g_blocks.as<GenomeBlock>(), ctx->g_exc.as<uint64_t>(), ctx->g_lower.as<uint64_t>(),
                           ctx->g_coarse.as<uint64_t>(), ctx->g_coarse2.as<uint32_t>(), bad.as<uint32_t>(), n_blocks);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!ok) return done(fail(ctx, BF_ERR_HIP, "building the genome from its image failed"));
    if (h_bad) return done(fail(ctx, BF_ERR_INVALID, "the genome image is damaged (its runs %s)", (h_bad & 2) ? "do not ascend or leave the genome" : "overlap"));
    return done(BF_OK);
}

static GenomeBytes genome_bytes(const bf_ctx* ctx) {
    return GenomeBytes{ctx->g_blocks.as<GenomeBlock>(), ctx->g_exc.as<uint64_t>(), ctx->g_lower.as<uint64_t>(),
                       ctx->g_run_start.as<uint64_t>(), ctx->g_run_byte.as<uint8_t>(), ctx->n_runs};
}

__global__ void __launch_bounds__(256) genome_unpack_kernel(const GenomeBytes gb, uint64_t g0, uint64_t n, uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gb.at(g0 + i);
}

int genome_unpack(bf_ctx* ctx, uint64_t g0, uint64_t n, uint8_t* dev_out) {
    if (n == 0) return BF_OK;
    hipLaunchKernelGGL(genome_unpack_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, genome_bytes(ctx), g0, n, dev_out);
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}

template <int W>
static void launch_fast(bf_ctx* ctx, const LocaliseArgs& a) {
    const dim3 g(grid_for(a.n - a.r0, 256)), b(256);
    if (a.packed) hipLaunchKernelGGL((localise_fast_kernel<W, true>), g, b, 0, ctx->stream, a);
    else hipLaunchKernelGGL((localise_fast_kernel<W, false>), g, b, 0, ctx->stream, a);
}

int settle_counts(bf_ctx* ctx) {
    if (!ctx->counts_pending) return BF_OK;
    BF_SYNC(ctx, ctx->stream);
    ctx->n_exact = ctx->h_counts[0]; ctx->n_sigx = ctx->h_counts[1]; ctx->n_text_records = ctx->h_counts[2];
    ctx->counts_pending = false;
    return BF_OK;
}

static inline uint32_t mask_words_for(uint32_t max_len) { return (max_len + 63) / 64; }
static inline uint32_t plane_words_for(uint32_t wm) { return wm <= 2 ? 2 : wm == 3 ? 3 : 4; }

// Every per-read array for `cap` reads, contents kept: the inputs bf_submit appends to, and -- once reads have been
// localised behind their upload -- the evidence the packed kernel wrote for them.  Waits for all queued work.
int grow_reads(bf_ctx* ctx, uint64_t cap) {
    if (cap <= ctx->cap_reads) return BF_OK;
    if (ctx->in_stream) BF_SYNC(ctx, ctx->in_stream);
    BF_SYNC(ctx, ctx->stream);
    BF_TRY(ensure(ctx, ctx->d_seq_off, (cap + 1) * 8, true));
    BF_TRY(ensure(ctx, ctx->d_frag_off, (cap + 1) * 8, true));
    BF_TRY(ensure(ctx, ctx->d_hits, cap * sizeof(bf_hit), true));
    if (ctx->packed_reads) {                               // inputs in this mode: the lengths came with the reads
        BF_TRY(ensure(ctx, ctx->d_len, cap * 2, true));
        BF_TRY(ensure(ctx, ctx->d_frag_len, cap * 2, true));
    }
    if (ctx->n_localised) {
        const uint64_t stride = (uint64_t)ctx->mask_words + 1;
        BF_TRY(ensure(ctx, ctx->d_len, cap * 2, true)); BF_TRY(ensure(ctx, ctx->d_frag_len, cap * 2, true));
        BF_TRY(ensure(ctx, ctx->d_fhash, cap * 8, true)); BF_TRY(ensure(ctx, ctx->d_key, cap * 8, true));
        BF_TRY(ensure(ctx, ctx->d_lpos, cap * 4, true)); BF_TRY(ensure(ctx, ctx->d_rpos, cap * 4, true));
        BF_TRY(ensure(ctx, ctx->d_bp, cap * 2, true)); BF_TRY(ensure(ctx, ctx->d_flags, cap * 2, true));
        BF_TRY(ensure(ctx, ctx->d_mask, stride * cap * 8, true)); BF_TRY(ensure(ctx, ctx->d_jrec, cap * sizeof(RecWord), true));
        BF_TRY(ensure(ctx, ctx->d_exact_list, cap * 4, true)); BF_TRY(ensure(ctx, ctx->d_sigx_slot, cap * 4, true));
        BF_TRY(ensure(ctx, ctx->d_sigx, cap * 16, true));
    }
    ctx->cap_reads = cap;
    return BF_OK;
}

// 4-bit batches come without their offset arrays (16 of a read's 121 bytes on PCIe): a read takes ceil(len / 2) bytes
// and fragment ids lie back to back, so an exclusive scan of the lengths of an uploaded piece, started at the piece's
// first byte (known on the host), IS the offset array.  Per piece and array: widen, scan, add the base.
// halve: 0 = the length itself (fragment ids), 1 = ceil(len / 2) (4-bit codes), 2 = ceil(len / 4) (2-bit codes)
__global__ void widen_lengths_kernel(uint32_t* __restrict__ out, const uint16_t* __restrict__ len, uint64_t n, int halve) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = halve == 2 ? ((uint32_t)len[i] + 3u) >> 2 : halve ? ((uint32_t)len[i] + 1u) >> 1 : (uint32_t)len[i];
}
// bf_submit_packed2: a piece that crossed PCIe at two bits a base becomes the resident 4-bit form (BAM codes: A = 1, C = 2,
// G = 4, T = 8 = 1 << code; two bases a byte, high nibble first).  One lane per 16 bases of a read: four bytes in (bits
// 2(k & 3).. of byte k >> 2 hold base k), eight bytes out; lanes of one read touch consecutive bytes on both sides.  The
// last group of a read writes only the bytes the read has and leaves the padding nibble of an odd length zero.
__global__ void __launch_bounds__(256) expand_2bit_kernel(uint8_t* __restrict__ out, const uint64_t* __restrict__ out_off /* per read, absolute */,
                                                          const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off /* per read, from 0 */,
                                                          const uint16_t* __restrict__ len, uint64_t n, uint32_t groups) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = x / groups;
    const uint32_t g = (uint32_t)(x - r * groups);
    if (r >= n) return;
    const uint32_t L = len[r];
    if (16u * g >= L) return;
    const uint32_t nb = min(16u, L - 16u * g);                       // bases of this group
    const uint8_t* src = in + in_off[r] + 4u * g;
    uint32_t w;                                                      // (whole word: d_seq2 ends in 64 spare bytes; bits past nb are masked below)
    __builtin_memcpy(&w, src, 4);
    uint64_t o = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t hi = 1u << ((w >> (4u * j)) & 3u), lo = 1u << ((w >> (4u * j + 2u)) & 3u);
        const uint32_t byte = (2u * j < nb ? hi << 4 : 0u) | (2u * j + 1u < nb ? lo : 0u);
        o |= (uint64_t)byte << (8u * j);
    }
    uint8_t* dst = out + out_off[r] + 8u * g;
    if (nb == 16u) __builtin_memcpy(dst, &o, 8);
    else for (uint32_t k = 0; k < (nb + 1u) >> 1; k++) dst[k] = (uint8_t)(o >> (8u * k));
}
// ... and the bases that are none of ACGT put in place: exception j names (read, position, 4-bit code).  A nibble is
// cleared and set with two atomics on its aligned word: two exceptions may share a byte, never a nibble.
__global__ void patch_exceptions_kernel(uint8_t* __restrict__ seq, const uint64_t* __restrict__ seq_off, const uint32_t* __restrict__ exc_read,
                                        const uint16_t* __restrict__ exc_pos, const uint8_t* __restrict__ exc_code, uint64_t n) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t pos = exc_pos[j];
    const uint64_t at = seq_off[exc_read[j]] + (pos >> 1);
    const uint32_t shift = 8u * (uint32_t)((uintptr_t)(seq + at) & 3u) + ((pos & 1u) ? 0u : 4u);
    unsigned int* word = (unsigned int*)((uintptr_t)(seq + at) & ~(uintptr_t)3);
    atomicAnd(word, ~(0xfu << shift));
    atomicOr(word, ((uint32_t)exc_code[j] & 0xfu) << shift);
}
__global__ void add_base_kernel(uint64_t* __restrict__ off, uint64_t base, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) off[i] += base;
}
int offsets_from_lengths(bf_ctx* ctx, uint64_t n_uploaded, uint32_t max_len) {
    size_t done = 0;
    for (const bf_ctx::OffsetPiece& pc : ctx->off_pending) {
        if (pc.r1 > n_uploaded) break;
        const uint64_t m = pc.r1 - pc.r0;
        BF_TRY(ensure(ctx, ctx->c_a32, m * 4, false));
        for (int which = 0; which < 2 && m; which++) {
            const uint16_t* len = (which ? ctx->d_frag_len.as<uint16_t>() : ctx->d_len.as<uint16_t>()) + pc.r0;
            uint64_t* off = (which ? ctx->d_frag_off.as<uint64_t>() : ctx->d_seq_off.as<uint64_t>()) + pc.r0;
            hipLaunchKernelGGL(widen_lengths_kernel, dim3(grid_for(m, 256)), dim3(256), 0, ctx->stream, ctx->c_a32.as<uint32_t>(), len, m, which ? 0 : 1);
            BF_TRY(scan_u32_to_u64(ctx, ctx->c_a32.as<uint32_t>(), off, m));                   // m + 1 entries, from 0
            hipLaunchKernelGGL(add_base_kernel, dim3(grid_for(m + 1, 256)), dim3(256), 0, ctx->stream, off, which ? pc.frag_base : pc.seq_base, m + 1);
        }
        if (pc.two_bit && m) {                               // the piece's bytes are 2-bit codes in d_seq2: expand, patch
            BF_TRY(ensure(ctx, ctx->c_scan, (m + 1) * 8, false));
            const uint16_t* len = ctx->d_len.as<uint16_t>() + pc.r0;
            hipLaunchKernelGGL(widen_lengths_kernel, dim3(grid_for(m, 256)), dim3(256), 0, ctx->stream, ctx->c_a32.as<uint32_t>(), len, m, 2);
            BF_TRY(scan_u32_to_u64(ctx, ctx->c_a32.as<uint32_t>(), ctx->c_scan.as<uint64_t>(), m));
            const uint32_t groups = (std::max<uint32_t>(max_len, 1) + 15u) / 16u;
            hipLaunchKernelGGL(expand_2bit_kernel, dim3(grid_for(m * groups, 256)), dim3(256), 0, ctx->stream, ctx->d_seq.as<uint8_t>(),
                               ctx->d_seq_off.as<uint64_t>() + pc.r0, ctx->d_seq2.as<uint8_t>() + pc.seq2_base, ctx->c_scan.as<uint64_t>(), len, m, groups);
            if (pc.exc1 > pc.exc0)
                hipLaunchKernelGGL(patch_exceptions_kernel, dim3(grid_for(pc.exc1 - pc.exc0, 256)), dim3(256), 0, ctx->stream, ctx->d_seq.as<uint8_t>(),
                                   ctx->d_seq_off.as<uint64_t>(), ctx->d_exc_read.as<uint32_t>() + pc.exc0, ctx->d_exc_pos.as<uint16_t>() + pc.exc0,
                                   ctx->d_exc_code.as<uint8_t>() + pc.exc0, pc.exc1 - pc.exc0);
        }
        BF_HIP(ctx, hipGetLastError());
        done++;
    }
    ctx->off_pending.erase(ctx->off_pending.begin(), ctx->off_pending.begin() + (long)done);
    return BF_OK;
}

// The packed kernel (and the fragment-id hashes) for the reads that are resident but not localised yet,
// [n_localised, n): called by bf_submit behind every uploaded piece (n = the reads uploaded so far, max_len = the
// longest of them) and by run_localise for what is left.
// The evidence layout depends on the longest read seen (mask words per render record; the plane words of the
// kernel instance, which also bound a packed junction record): a piece that changes it starts the whole set over.
int localise_uploaded(bf_ctx* ctx, uint64_t n, uint32_t max_len) {
    if (n == 0) return BF_OK;
    const uint32_t A = ctx->cfg.anchor_len;
    const uint32_t Wm = mask_words_for(max_len), W = plane_words_for(Wm);
    if (ctx->n_localised && Wm != ctx->mask_words) ctx->n_localised = 0;
    const uint64_t r0 = ctx->n_localised;
    if (r0 >= n) return BF_OK;
    const bool keep = r0 > 0;
    const uint64_t cap = std::max<uint64_t>(ctx->cap_reads, n);
    ctx->plane_words = W; ctx->mask_words = Wm;
    BF_TRY(ensure(ctx, ctx->d_len, cap * 2, keep || ctx->packed_reads)); BF_TRY(ensure(ctx, ctx->d_frag_len, cap * 2, keep || ctx->packed_reads));
    BF_TRY(ensure(ctx, ctx->d_fhash, cap * 8, keep)); BF_TRY(ensure(ctx, ctx->d_key, cap * 8, keep));
    BF_TRY(ensure(ctx, ctx->d_lpos, cap * 4, keep)); BF_TRY(ensure(ctx, ctx->d_rpos, cap * 4, keep));
    BF_TRY(ensure(ctx, ctx->d_bp, cap * 2, keep)); BF_TRY(ensure(ctx, ctx->d_flags, cap * 2, keep));
    BF_TRY(ensure(ctx, ctx->d_mask, (size_t)(Wm + 1) * cap * 8, keep)); BF_TRY(ensure(ctx, ctx->d_jrec, cap * sizeof(RecWord), keep));
    BF_TRY(ensure(ctx, ctx->d_exact_list, cap * 4, keep)); BF_TRY(ensure(ctx, ctx->d_sigx_slot, cap * 4, keep));
    // raw signature bytes: one 16-byte slot per read at most (handed out with an atomic counter by both kernels)
    BF_TRY(ensure(ctx, ctx->d_sigx, (size_t)cap * 16, keep));
    BF_TRY(ensure(ctx, ctx->d_counters, 64));
    if (r0 == 0) BF_HIP(ctx, hipMemsetAsync(ctx->d_counters.p, 0, 64, ctx->stream));
    if (ctx->in_pending) {                                 // the uploads these kernels read
        BF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));
        ctx->in_pending = false;
    }
    BF_TRY(offsets_from_lengths(ctx, n, max_len));

    {
        Scope t(ctx, "read_meta");
        hipLaunchKernelGGL(read_meta_kernel, dim3(grid_for(n - r0, 256)), dim3(256), 0, ctx->stream, r0, n,
                           ctx->d_seq_off.as<uint64_t>(), ctx->d_frag.as<uint8_t>(),
                           ctx->d_frag_off.as<uint64_t>(), ctx->d_len.as<uint16_t>(),
                           ctx->d_frag_len.as<uint16_t>(), ctx->d_fhash.as<uint64_t>(), ctx->packed_reads);
    }
    BF_HIP(ctx, hipGetLastError());

    LocaliseArgs a{};
    a.r0 = r0; a.n = n; a.A = (int)A; a.cb = ctx->contig_bits; a.force_exact = (ctx->cfg.flags & BF_FLAG_FORCE_EXACT) ? 1u : 0u;
    a.hits = ctx->d_hits.as<bf_hit>(); a.len = ctx->d_len.as<uint16_t>();
    a.len_out = ctx->d_len.as<uint16_t>();
    a.seq = ctx->d_seq.as<uint8_t>(); a.seq_off = ctx->d_seq_off.as<uint64_t>(); a.contigs = ctx->g_contigs.as<ContigDev>();
    a.packed = ctx->packed_reads ? 1u : 0u;
    a.by_ref = (ctx->cfg.flags & BF_FLAG_READS_ON_HOST) ? 1u : 0u;
    a.blocks = ctx->g_blocks.as<GenomeBlock>(); a.coarse = ctx->g_coarse.as<uint64_t>();
    a.coarse2 = ctx->g_coarse2.as<uint32_t>();
    a.exc = ctx->g_exc.as<uint64_t>(); a.gb = genome_bytes(ctx);
    a.sigx = ctx->d_sigx.as<uint8_t>(); a.sigx_slot = ctx->d_sigx_slot.as<uint32_t>();
    a.key = ctx->d_key.as<uint64_t>(); a.lpos = ctx->d_lpos.as<int32_t>(); a.rpos = ctx->d_rpos.as<int32_t>();
    a.bp = ctx->d_bp.as<uint16_t>(); a.flags = ctx->d_flags.as<uint16_t>(); a.meta = ctx->d_mask.as<uint64_t>();
    a.meta_stride = (int)Wm + 1; a.rw = ctx->d_jrec.as<RecWord>(); a.fhash = ctx->d_fhash.as<uint64_t>();
    a.exact_list = ctx->d_exact_list.as<uint32_t>(); a.counters = ctx->d_counters.as<uint32_t>();
    {
        Scope t(ctx, "localise");
        if (W == 2) launch_fast<2>(ctx, a);
        else if (W == 3) launch_fast<3>(ctx, a);
        else launch_fast<4>(ctx, a);
    }
    BF_HIP(ctx, hipGetLastError());
    ctx->n_localised = n;
    return BF_OK;
}

int run_localise(bf_ctx* ctx, bool defer_counts, bool all) {
    const uint64_t n = ctx->n_reads;
    ctx->n_exact = 0; ctx->n_sigx = 0; ctx->n_text_records = 0;
    ctx->localised = false;
    if (n == 0) { ctx->localised = true; ctx->plane_words = 1; ctx->mask_words = 1; ctx->n_localised = 0; return BF_OK; }
    const uint32_t A = ctx->cfg.anchor_len;
    if (ctx->min_len < 2 * A)
        return fail(ctx, BF_ERR_RANGE, "a read of length %u is shorter than 2 * anchor_len = %u "
                    "(the reference only emits reads with len >= 2*anchor_len, detect.rs:311)",
                    ctx->min_len, 2 * A);
    if (all) ctx->n_localised = 0;
    BF_TRY(localise_uploaded(ctx, n, ctx->max_len));
    const uint32_t Wm = ctx->mask_words, W = ctx->plane_words;
    {
        ExactArgs e{};
        e.n_total = n; e.A = (int)A; e.cb = ctx->contig_bits; e.exact_list = ctx->d_exact_list.as<uint32_t>();
        e.counters = ctx->d_counters.as<uint32_t>(); e.hits = ctx->d_hits.as<bf_hit>();
        e.seq = ctx->d_seq.as<uint8_t>(); e.seq_off = ctx->d_seq_off.as<uint64_t>();
        e.len = ctx->d_len.as<uint16_t>(); e.packed = ctx->packed_reads ? 1u : 0u;
        e.contigs = ctx->g_contigs.as<ContigDev>(); e.gb = genome_bytes(ctx);
        e.key = ctx->d_key.as<uint64_t>(); e.lpos = ctx->d_lpos.as<int32_t>(); e.rpos = ctx->d_rpos.as<int32_t>();
        e.bp = ctx->d_bp.as<uint16_t>(); e.flags = ctx->d_flags.as<uint16_t>(); e.meta = ctx->d_mask.as<uint64_t>();
        e.mask_words = (int)Wm; e.max_packed = 64u * W; e.by_ref = (ctx->cfg.flags & BF_FLAG_READS_ON_HOST) ? 1u : 0u; e.rw = ctx->d_jrec.as<RecWord>(); e.fhash = ctx->d_fhash.as<uint64_t>(); e.sigx = ctx->d_sigx.as<uint8_t>(); e.sigx_slot = ctx->d_sigx_slot.as<uint32_t>();
        Scope t(ctx, "localise_exact");
        const unsigned blocks = (unsigned)std::min<uint64_t>(grid_for(n * 64, 256), 16384);   // oversubscribed: entries differ in length
        hipLaunchKernelGGL(localise_exact_kernel, dim3(blocks), dim3(256), 0, ctx->stream, e);
        hipLaunchKernelGGL(exact_done_kernel, dim3(1), dim3(1), 0, ctx->stream, e.counters);
        BF_HIP(ctx, hipGetLastError());
    }
    BF_HIP(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counters.p, 12, hipMemcpyDeviceToHost, ctx->stream));
    ctx->counts_pending = true;
    ctx->localised = true;
    return defer_counts ? BF_OK : settle_counts(ctx);
}

__global__ void pick_offsets_kernel(const uint64_t* __restrict__ off, uint64_t n, int n_chunks, uint64_t* __restrict__ out) {
    const int c = threadIdx.x;
    if (c <= n_chunks) out[c] = off[n * (uint64_t)c / n_chunks];
}

// Junction offsets of n surviving reads (exclusive scan of len+1).  chunk_off[c] = byte offset of
// read n*c/n_chunks, so that the caller can render and copy out the text chunk by chunk.
// wait = false: everything is queued and the chunk offsets are on their way to the mailbox behind ctx->ev[13];
// render_prepare_wait() brings them in -- a caller that has kernels to queue which do not need the offsets on the host (the
// render kernels read them on the device; only the copies to the host need sizes) queues those first.
int render_prepare(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, DevBuf& d_off, int n_chunks,
                   uint64_t* chunk_off, const uint32_t* d_sizes, bool wait) {
    for (int c = 0; c <= n_chunks; c++) chunk_off[c] = 0;
    BF_TRY(ensure(ctx, d_off, (n + 1) * 8));
    if (n == 0) {
        BF_HIP(ctx, hipMemsetAsync(d_off.p, 0, 8, ctx->stream));
        return BF_OK;
    }
    BF_TRY(ensure(ctx, ctx->c_misc, n * 4 + 16 + 64 * 8));
    Scope t(ctx, "render_offsets");
    if (!d_sizes) {
        hipLaunchKernelGGL(render_len_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, d_local_idx,
                           ctx->d_jrec.as<RecWord>(), ctx->c_misc.as<uint32_t>());
        d_sizes = ctx->c_misc.as<uint32_t>();
    }
    BF_TRY(scan_u32_to_u64(ctx, d_sizes, d_off.as<uint64_t>(), n));
    // the chunk boundaries in ONE small copy (nine separate 8-byte copies cost 30 us each)
    uint64_t* d_pick = (uint64_t*)(ctx->c_misc.as<uint8_t>() + ((n * 4 + 15) & ~(size_t)15));
    hipLaunchKernelGGL(pick_offsets_kernel, dim3(1), dim3(64), 0, ctx->stream, d_off.as<uint64_t>(), n, n_chunks, d_pick);
    // ... into the ctx's pinned mailbox (a copy into the caller's pageable array is staged by the runtime: a round trip more)
    if (n_chunks + 1 > kMailWords - 8) return fail(ctx, BF_ERR_RANGE, "render_prepare: %d chunks", n_chunks);
    BF_TRY(mail_fetch(ctx, 8, d_pick, n_chunks + 1));
    if (!wait) {
        BF_HIP(ctx, hipEventRecord(ctx->ev[13], ctx->stream));
        return BF_OK;
    }
    BF_SYNC(ctx, ctx->stream);
    for (int c = 0; c <= n_chunks; c++) chunk_off[c] = ctx->h_mail[8 + c];
    return BF_OK;
}

int render_prepare_wait(bf_ctx* ctx, uint64_t n, int n_chunks, uint64_t* chunk_off) {
    for (int c = 0; c <= n_chunks; c++) chunk_off[c] = 0;
    if (n == 0) return BF_OK;
    // (polled like the stream waits: the offsets are usually there before the host has queued what it had to queue)
    hipError_t e = hipErrorNotReady;
    const auto t0 = std::chrono::steady_clock::now();
    while ((e = hipEventQuery(ctx->ev[13])) == hipErrorNotReady)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
    if (e != hipSuccess && e != hipErrorNotReady) (void)hipGetLastError();
    if (e != hipSuccess) BF_HIP(ctx, hipEventSynchronize(ctx->ev[13]));
    for (int c = 0; c <= n_chunks; c++) chunk_off[c] = ctx->h_mail[8 + c];
    return BF_OK;
}

// Junction records of n listed reads at the given (absolute) offsets: packed records by one kernel, the
// text records of the reads that cannot be packed (too long, many bases outside ACGT) by the other.
// cap: a record that would START at or beyond d_bytes + cap is not written (the caller's buffer ends there plus the room of
// one record, kRenderSlack; callers that size the buffer from the offsets pass no limit)
static int launch_render(bf_ctx* ctx, uint64_t n, const uint32_t* d_idx, const uint64_t* d_off, uint8_t* d_bytes, uint64_t cap = ~0ull) {
    const int stride = (int)ctx->mask_words + 1;
    const uint32_t W = ctx->plane_words;
    const dim3 g(grid_for(n, 256)), b(256);
    const uint8_t* seq = ctx->d_seq.as<uint8_t>();
    const uint64_t* so = ctx->d_seq_off.as<uint64_t>();
    const uint64_t* meta = ctx->d_mask.as<uint64_t>();
    const bool packed = ctx->packed_reads;
    if (ctx->cfg.flags & BF_FLAG_READS_ON_HOST) {
        hipLaunchKernelGGL(render_ref_kernel, g, b, 0, ctx->stream, n, d_idx, d_off, meta, stride, d_bytes, cap);
        BF_HIP(ctx, hipGetLastError());
        return BF_OK;
    }
    if (W == 2) hipLaunchKernelGGL(render_packed_kernel<2>, g, b, 0, ctx->stream, n, d_idx, d_off, seq, so, meta, stride, d_bytes, packed, cap);
    else if (W == 3) hipLaunchKernelGGL(render_packed_kernel<3>, g, b, 0, ctx->stream, n, d_idx, d_off, seq, so, meta, stride, d_bytes, packed, cap);
    else hipLaunchKernelGGL(render_packed_kernel<4>, g, b, 0, ctx->stream, n, d_idx, d_off, seq, so, meta, stride, d_bytes, packed, cap);
    BF_TRY(settle_counts(ctx));
    if (ctx->n_text_records && packed)
        hipLaunchKernelGGL(render_text_nibbles_kernel, dim3(grid_for(n * 16, 256)), dim3(256), 0, ctx->stream, n, d_idx, d_off, seq, so,
                           meta, stride, 64u * W, d_bytes, cap);
    else if (ctx->n_text_records)
        hipLaunchKernelGGL(render_kernel, dim3(grid_for(n * 16, 256)), dim3(256), 0, ctx->stream, n, d_idx, d_off, seq, so, meta,
                           stride, 64u * W, d_bytes, cap);
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}

// Renders reads [a, b) of the list into d_bytes (offsets are absolute).
int render_range(bf_ctx* ctx, uint64_t a, uint64_t b, const uint32_t* d_local_idx, const DevBuf& d_off, DevBuf& d_bytes) {
    if (b <= a) return BF_OK;
    Scope t(ctx, "render");
    return launch_render(ctx, b - a, d_local_idx + a, d_off.as<uint64_t>() + a, d_bytes.as<uint8_t>());
}

// The same with plain pointers: offsets relative to d_bytes (a group's part of the junction buffer).
int render_range_at(bf_ctx* ctx, uint64_t a, uint64_t b, const uint32_t* d_local_idx, const uint64_t* d_off, uint8_t* d_bytes, uint64_t cap) {
    if (b <= a) return BF_OK;
    Scope t(ctx, "render");
    return launch_render(ctx, b - a, d_local_idx + a, d_off + a, d_bytes, cap);
}

int render_sizes_checked(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, uint32_t* d_len, uint32_t* d_bad) {
    if (n) hipLaunchKernelGGL(render_len_checked_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, d_local_idx,
                              ctx->d_jrec.as<RecWord>(), ctx->n_reads, d_len, d_bad);
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}

// Junction strings for a DEVICE list of local read indices into caller-owned device memory (the
// multi-GPU route: nothing crosses PCIe here).  d_off receives n+1 offsets; when d_bytes is null only
// the offsets and *total are produced.
int render_dev(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, uint64_t* d_off, uint8_t* d_bytes,
               uint64_t d_bytes_cap, uint64_t* total) {
    *total = 0;
    if (n == 0) {
        BF_HIP(ctx, hipMemsetAsync(d_off, 0, 8, ctx->stream));
        BF_SYNC(ctx, ctx->stream);
        return BF_OK;
    }
    BF_TRY(ensure(ctx, ctx->r_idx, n * 4 + 16));                      // lengths + the bad-index counter
    uint32_t* d_len = ctx->r_idx.as<uint32_t>();
    uint32_t* d_bad = d_len + n;
    BF_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    {
        Scope t(ctx, "render_offsets");
        hipLaunchKernelGGL(render_len_checked_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, d_local_idx,
                           ctx->d_jrec.as<RecWord>(), ctx->n_reads, d_len, d_bad);
        BF_TRY(scan_u32_to_u64(ctx, d_len, d_off, n));
    }
    uint32_t bad = 0;
    BF_HIP(ctx, hipMemcpyAsync(total, d_off + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    BF_HIP(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    BF_SYNC(ctx, ctx->stream);
    if (bad) return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev: %u of %llu indices are outside the %llu resident reads",
                         bad, (unsigned long long)n, (unsigned long long)ctx->n_reads);
    if (!d_bytes) return BF_OK;
    if (d_bytes_cap < *total)
        return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev: buffer of %llu bytes, %llu needed",
                    (unsigned long long)d_bytes_cap, (unsigned long long)*total);
    {
        Scope t(ctx, "render");
        BF_TRY(launch_render(ctx, n, d_local_idx, d_off, d_bytes));
    }
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}

// ---- the entry point that allocates nothing (include/breakfast_hip.h, bf_render_junctions_dev_static) ----------
// bf_render_junctions_dev keeps three things in device memory besides its inputs and outputs: one record size per
// list entry, the tile totals of the scan that turns sizes into offsets, and the counter of bad indices.  Here
// they live in a file-scope device array, so the call needs no hipMalloc.  In host code the array's NAME is a
// host-side stand-in, not a device address: the address of this device's copy comes from hipGetSymbolAddress,
// once per device (scratch_setup, called by bf_ctx_create with the ctx's device current), and only that pointer
// is ever handed to a kernel, a memset or a copy.
constexpr uint64_t kStaticMaxReads = BF_STATIC_RENDER_MAX_READS;
constexpr uint64_t kStaticLenBytes = kStaticMaxReads * 4;                              // u32 record size per list entry
constexpr uint64_t kStaticScanBytes = ((kStaticMaxReads + 4095) / 4096) * 8 + 256;     // = scan_workspace_bytes(max)
constexpr uint64_t kStaticScratchBytes = kStaticLenBytes + kStaticScanBytes + 256;     // + the bad-index counter
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

int scratch_setup(bf_ctx* ctx) {
    void* ptr = nullptr;
    BF_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(ctx, BF_ERR_HIP, "hipGetSymbolAddress gave no address for the static scratch array");
    ctx->scratch = ptr;                               // every GPU holds its own copy at its own address: one ctx, one device
    return BF_OK;
}

static int render_dev_static(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, uint64_t* d_off, uint8_t* d_bytes,
                             uint64_t d_bytes_cap, uint64_t* total) {
    *total = 0;
    if (n > kStaticMaxReads)
        return fail(ctx, BF_ERR_RANGE, "bf_render_junctions_dev_static: %llu reads in one call; the static scratch buffer holds %llu "
                    "(use bf_render_junctions_dev, or several calls)", (unsigned long long)n, (unsigned long long)kStaticMaxReads);
    if (scan_workspace_bytes(n) > kStaticScanBytes) return fail(ctx, BF_ERR_RANGE, "bf_render_junctions_dev_static: scan workspace");
    if (!ctx->scratch || !ctx->h_counts) return fail(ctx, BF_ERR_STATE, "bf_render_junctions_dev_static: the ctx has no scratch pointer");
    if (n == 0) {
        BF_HIP(ctx, hipMemsetAsync(d_off, 0, 8, ctx->stream));
        BF_SYNC(ctx, ctx->stream);
        return BF_OK;
    }
    uint8_t* base = (uint8_t*)ctx->scratch;           // never the array's name
    uint32_t* d_len = (uint32_t*)base;
    uint64_t* d_scan = (uint64_t*)(base + kStaticLenBytes);
    uint32_t* d_bad = (uint32_t*)(base + kStaticLenBytes + kStaticScanBytes);
    // the array keeps the previous call's contents: the counter must start at zero; sizes and tile totals are
    // written for [0, n) before they are read
    BF_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    hipLaunchKernelGGL(render_len_checked_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, d_local_idx,
                       ctx->d_jrec.as<RecWord>(), ctx->n_reads, d_len, d_bad);
    BF_TRY(scan_u32_to_u64_ws(ctx, d_len, d_off, n, d_scan));
    // read-backs through the ctx's pinned words (a copy to pageable memory may stage through memory the runtime allocates)
    volatile uint32_t* hw = ctx->h_counts + 8;
    BF_HIP(ctx, hipMemcpyAsync((void*)(ctx->h_counts + 8), d_off + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    BF_HIP(ctx, hipMemcpyAsync((void*)(ctx->h_counts + 10), d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    BF_SYNC(ctx, ctx->stream);
    const uint64_t tot = (uint64_t)hw[0] | ((uint64_t)hw[1] << 32);
    const uint32_t bad = hw[2];
    *total = tot;
    if (bad) return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev_static: %u of %llu indices are outside the %llu resident reads",
                         bad, (unsigned long long)n, (unsigned long long)ctx->n_reads);
    if (!d_bytes) return BF_OK;
    if (d_bytes_cap < tot)
        return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev_static: buffer of %llu bytes, %llu needed",
                    (unsigned long long)d_bytes_cap, (unsigned long long)tot);
    BF_TRY(launch_render(ctx, n, d_local_idx, d_off, d_bytes));      // no timing scope: events would be created
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}

int run_render(bf_ctx* ctx, uint64_t n, const uint32_t* d_local_idx, DevBuf& d_off, DevBuf& d_bytes,
               uint64_t* total_bytes) {
    uint64_t off[2];
    BF_TRY(render_prepare(ctx, n, d_local_idx, d_off, 1, off));
    *total_bytes = off[1];
    BF_TRY(ensure(ctx, d_bytes, off[1] ? off[1] : 1));
    return render_range(ctx, 0, n, d_local_idx, d_off, d_bytes);
}

}  // namespace bf

extern "C" int bf_render_junctions_dev_static(bf_ctx* ctx, uint64_t n, const uint32_t* dev_local_index, uint64_t* dev_off,
                                              uint8_t* dev_bytes, uint64_t dev_bytes_cap, uint64_t* total) {
    using namespace bf;
    if (!ctx || !total) return BF_ERR_INVALID;
    if (!ctx->localised) return fail(ctx, BF_ERR_STATE, "bf_render_junctions_dev_static before bf_localise");
    if (!dev_off || (n && !dev_local_index)) return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev_static: null argument");
    if (((uintptr_t)dev_bytes & 3) || ((uintptr_t)dev_off & 7) || ((uintptr_t)dev_local_index & 3))
        return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev_static: dev_bytes/dev_local_index must be 4-byte and dev_off 8-byte aligned");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    return render_dev_static(ctx, n, dev_local_index, dev_off, dev_bytes, dev_bytes_cap, total);
}
