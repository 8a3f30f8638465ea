// Sort -> greedy clustering -> frag-id dedup of the localised evidence, on the GPU.
// Reference: src/detect.rs:190-231 (stable sort by (chr, pos), greedy seed-and-extend clustering)
// and :247-287 (remove_frag_id_duplicates).
//
// Exact parallel decomposition.  Every membership test of the reference's
// sweep except the two position windows is an equality on the class
//   C = (chr, strand, mchr, mstrand, signature),
// and `reported[]` is only ever set by a seed of the same class.  So the global sweep restricted to
// one class, over that class's members in (lpos, input index) order, yields exactly the clusters
// whose seeds are in that class.  We therefore
//   1. stable-sort evidence by (signature bits of the class key, lpos)   [input order breaks ties]
//   2. run the sequential sweep inside each segment of equal signature bits, comparing whole keys
//      (a segment may be coarser than a class: other contigs/strands with the same signature, or a
//       signature with non-ACGT bytes, whose key carries a hash -- there the sweep compares the 16
//       raw bytes, like detect.rs:223)
//   3. enumerate members, group them by fragment id, keep the longest-flank read per group
//   4. order surviving clusters by the seed's rank in the reference's sort: (chr, lpos, input index).

#include "bf_internal.h"
#include <memory>

namespace bf {

constexpr uint32_t kNone = 0xffffffffu;

static inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// stable LSD radix sort and the scans: primitives.hip
template <class K>
static int sort_pairs(bf_ctx* ctx, const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, uint64_t n,
                      unsigned begin_bit, unsigned end_bit) {
    return radix_sort_pairs(ctx, kin, kout, vin, vout, n, begin_bit, end_bit);
}

static int fetch_u64(bf_ctx* ctx, const uint64_t* d, uint64_t* h) {
    BF_HIP(ctx, hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}

// ---- small kernels ----------------------------------------------------------------------------
__global__ void gather_key_kernel(uint64_t* out, const uint64_t* __restrict__ in, const uint32_t* __restrict__ idx, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
struct SweepView {
    uint64_t n;
    const uint64_t* skey;     // sorted class keys
    const int32_t* slpos;     // sorted
    const int32_t* srpos;     // sorted
    const uint32_t* perm;     // sorted position -> owner record
    const uint8_t* sigx;
    const uint32_t* sigx_slot;
    int64_t F;                // max_frag_len
};

__device__ __forceinline__ bool same_sigx(const SweepView& v, uint64_t key, uint64_t p, uint64_t s) {
    if (!((key >> kKeySigxBit) & 1)) return true;           // plain signature: the key holds all of it
    const uint8_t* a = v.sigx + (uint64_t)v.sigx_slot[v.perm[p]] * 16;
    const uint8_t* b = v.sigx + (uint64_t)v.sigx_slot[v.perm[s]] * 16;
    bool eq = true;
    for (int k = 0; k < 16; k++) eq = eq && a[k] == b[k];
    return eq;
}

// `evidence[s].pos - read.pos > max_frag_len` on the reference's wrapped usize (detect.rs:215)
__device__ __forceinline__ bool beyond(int32_t ls, int32_t lp, int64_t F) {
    return (uint64_t)(int64_t)ls - (uint64_t)(int64_t)lp > (uint64_t)F;
}
// `(mpos as i64 - read.mpos as i64).abs() > max_frag_len` (detect.rs:220)
__device__ __forceinline__ bool rpos_far(int32_t rs, int32_t rp, int64_t F) {
    const int64_t d = (int64_t)rs - (int64_t)rp;
    return (d < 0 ? -d : d) > F;
}

// The reference's sweep (detect.rs:203-227) inside one key segment.
//
// Fast path.  The first seed of a segment is its first record h.  If EVERY other record of the segment
// passes the seed's tests (left position within max_frag_len -- the break test, monotone in sorted
// order --, right position within max_frag_len, identical signature bytes), the sweep adds them all to
// h's cluster and marks them reported, so the segment is exactly one cluster of its own size and no
// walk is needed.  That is decided for all records at once: head index by a max-scan, one "covered"
// flag per record, a prefix sum of the flags, and one lane per segment TAIL compares the covered count
// with the segment length.  Only segments that fail (two junctions sharing a class, mates far apart)
// are walked: by one lane when short, by one wave when longer than kBigSegment (duplicate-heavy
// libraries pile thousands of reads on one junction).
// csize[p] = cluster size if p is a seed; csize is zero-initialised.
constexpr uint32_t kBigSegment = 16;            // (96 until round 2: one lane walking a 96-record segment took 64 us, the whole sweep_list launch)
constexpr uint32_t kCtlWords = 16;       // head of c_big: two list lengths, then 64-bit sums for the host (flag_ge_kernel)

// A SEGMENT is a run of records with equal low 32 key bits (the signature, or its hash): the class keys are
// sorted on those bits only -- four radix passes instead of six -- and the records of a segment are in
// (lpos, input index) order whatever their class.  Records of different classes (other contigs or strands, same
// signature: a few thousand pairs among 10M reads) may therefore share a segment; every membership test below
// compares the whole key, like detect.rs:218-223 compares every field, and the break test on lpos stays monotone.
__device__ __forceinline__ uint32_t seg_of(uint64_t key) { return (uint32_t)key; }

// (also clears what the sweep starts from -- reported[], csize[], the control words -- instead of three memset launches)
__global__ void __launch_bounds__(256) head_mark_kernel(const uint64_t* __restrict__ skey, uint32_t* __restrict__ head, uint64_t n,
                                                        uint8_t* __restrict__ reported, uint32_t* __restrict__ csize, uint32_t* __restrict__ ctl) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kCtlWords) ctl[i] = 0;
    if (i < n) {
        head[i] = (i == 0 || seg_of(skey[i - 1]) != seg_of(skey[i])) ? (uint32_t)i : 0u;
        reported[i] = 0; csize[i] = 0;
    }
}

__global__ void __launch_bounds__(256) covered_kernel(const SweepView v, const uint32_t* __restrict__ head,
                                                      uint32_t* __restrict__ covered) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n) return;
    const uint64_t h = head[i];
    bool ok = false;
    if (i != h)
        ok = v.skey[i] == v.skey[h] && !beyond(v.slpos[i], v.slpos[h], v.F) && !rpos_far(v.srpos[i], v.srpos[h], v.F) &&
             same_sigx(v, v.skey[i], h, i);
    covered[i] = ok ? 1u : 0u;
}

__global__ void __launch_bounds__(256) tails_kernel(const SweepView v, const uint32_t* __restrict__ head,
                                                    const uint64_t* __restrict__ cov_scan /* exclusive, n+1 */,
                                                    uint32_t* __restrict__ csize, uint32_t* __restrict__ small_list,
                                                    uint32_t* __restrict__ big_list, uint32_t* __restrict__ counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n) return;
    if (i + 1 < v.n && seg_of(v.skey[i + 1]) == seg_of(v.skey[i])) return;          // not the last record of its segment
    const uint64_t h = head[i];
    const uint64_t others = i - h, cov = cov_scan[i + 1] - cov_scan[h];
    if (cov == others) { csize[h] = (uint32_t)(others + 1); return; }
    if (others + 1 > kBigSegment) big_list[atomicAdd(&counts[1], 1u)] = (uint32_t)h;
    else small_list[atomicAdd(&counts[0], 1u)] = (uint32_t)h;
}

// one lane per listed short segment: the reference's two loops as they are
__device__ __forceinline__ void sweep_list_part(const SweepView& v, uint8_t* __restrict__ reported,
                                                uint32_t* __restrict__ csize, const uint32_t* __restrict__ small_list,
                                                const uint32_t* __restrict__ counts, uint32_t block, uint32_t n_blocks) {
    // fixed grid; the list's length stays on the device (no host round trip between tails_kernel and this one)
    const uint32_t n_small = counts[0];
    for (uint32_t k = block * blockDim.x + threadIdx.x; k < n_small; k += n_blocks * blockDim.x) {
    const uint64_t i = small_list[k];
    const uint32_t seg = seg_of(v.skey[i]);
    uint64_t e = i + 1;
    while (e < v.n && seg_of(v.skey[e]) == seg) e++;
    for (uint64_t p = i; p < e; p++) {
        if (reported[p]) continue;
        const uint64_t key = v.skey[p];
        const int32_t lp = v.slpos[p], rp = v.srpos[p];
        uint32_t cnt = 1;
        for (uint64_t s = p + 1; s < e; s++) {
            if (beyond(v.slpos[s], lp, v.F)) break;
            if (v.skey[s] != key) continue;                          // another class (detect.rs:218-223)
            if (rpos_far(v.srpos[s], rp, v.F)) continue;
            if (!same_sigx(v, key, p, s)) continue;
            cnt++;
            reported[s] = 1;
        }
        csize[p] = cnt;
    }
    }
}

// One wave per long segment: seeds are still taken in order (the sweep is sequential in the seeds),
// but finding the next unreported read and testing a seed's window run 64 reads at a time.
__device__ __forceinline__ void sweep_big_part(const SweepView& v, uint8_t* reported, uint32_t* __restrict__ csize,
                                               const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ counts,
                                               uint32_t block, uint32_t n_blocks) {
    const int lane = threadIdx.x & 63;
    const uint32_t n_big = counts[1], n_waves = (n_blocks * blockDim.x) >> 6;
    for (uint32_t b = (block * blockDim.x + threadIdx.x) >> 6; b < n_big; b += n_waves) {      // wave-uniform
    const uint64_t i = big_list[b];
    const uint32_t seg = seg_of(v.skey[i]);
    uint64_t e = i;                                          // segment end
    for (;; e += 64) {
        const uint64_t s = e + lane;
        const uint64_t same = __ballot(s < v.n && seg_of(v.skey[s]) == seg);
        if (same != ~0ull) { e += __popcll(same); break; }   // the signature is constant over a prefix
    }
    uint64_t cursor = i;
    while (cursor < e) {
        // next unreported read at or after the cursor = the next seed
        uint64_t p = ~0ull;
        for (uint64_t c0 = cursor; c0 < e; c0 += 64) {
            const uint64_t s = c0 + lane;
            const uint64_t open = __ballot(s < e && reported[s] == 0);
            if (open) { p = c0 + __builtin_ctzll(open); break; }
        }
        if (p == ~0ull) break;
        const uint64_t key = v.skey[p];
        const int32_t lp = v.slpos[p], rp = v.srpos[p];
        uint32_t cnt = 1;
        for (uint64_t s0 = p + 1;; s0 += 64) {
            const uint64_t s = s0 + lane;
            const bool live = s < e && !beyond(v.slpos[s], lp, v.F);
            const bool take = live && v.skey[s] == key && !rpos_far(v.srpos[s], rp, v.F) && same_sigx(v, key, p, s);
            if (take) reported[s] = 1;
            cnt += (uint32_t)__popcll(__ballot(take));
            if (__ballot(live) != ~0ull) break;
        }
        if (lane == 0) csize[p] = cnt;
        // the marks written above are read back by other lanes of this wave in the next round
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        cursor = p + 1;
    }
    }
}

// Both walks in one launch (the lists are usually short, a launch of its own cost each 6 us of nothing): the first
// `list_blocks` workgroups take the short segments a lane each, the others the long ones a wave each.  Different
// segments, so the two halves never touch the same records.
__global__ void __launch_bounds__(256) sweep_walk_kernel(const SweepView v, uint8_t* reported, uint32_t* __restrict__ csize,
                                                         const uint32_t* __restrict__ small_list, const uint32_t* __restrict__ big_list,
                                                         const uint32_t* __restrict__ counts, uint32_t list_blocks) {
    if (blockIdx.x < list_blocks) sweep_list_part(v, reported, csize, small_list, counts, blockIdx.x, list_blocks);
    else sweep_big_part(v, reported, csize, big_list, counts, blockIdx.x - list_blocks, gridDim.x - list_blocks);
}

// flag[i] = val[i] >= thr, and the sums the host needs to size what follows, so that they come along with the
// fetch of the flag count instead of costing a synchronisation each: sums[0] += the flagged values, sums[1] +=
// those above wave_max.  Fixed grid: one atomic per block with something to add.
constexpr unsigned kFlagGrid = 1024;
__global__ void __launch_bounds__(256) flag_ge_kernel(uint32_t* flag, const uint32_t* __restrict__ val, uint32_t thr, uint64_t n,
                                                      uint32_t wave_max, unsigned long long* sums,
                                                      const uint32_t* __restrict__ idx = nullptr, unsigned int* n_wide = nullptr) {
    unsigned long long s_all = 0, s_big = 0;
    unsigned int wide = 0;                                           // flagged values above 64 (fill_members_kernel<4>'s clusters)
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t c = val[idx ? idx[i] : i];
        const bool f = c >= thr;
        flag[i] = f ? 1u : 0u;
        if (f) { s_all += c; if (c > wave_max) s_big += c; wide |= c > 64u ? 1u : 0u; }
    }
    const bool any_wide = __ballot(wide != 0) != 0;
    for (int d = 32; d; d >>= 1) { s_all += __shfl_down(s_all, d); s_big += __shfl_down(s_big, d); }
    // one pair of atomics per BLOCK: they all hit the same two words and are served one after the other (with one pair
    // per wave the 8192 atomics of a launch, not its 40 MB of traffic, set the kernel's time: 75 us for 5M records)
    __shared__ unsigned long long w_all[4], w_big[4];
    __shared__ unsigned int w_wide[4];
    if ((threadIdx.x & 63) == 0) { w_all[threadIdx.x >> 6] = s_all; w_big[threadIdx.x >> 6] = s_big; w_wide[threadIdx.x >> 6] = any_wide; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_all = w_all[0] + w_all[1] + w_all[2] + w_all[3];
        s_big = w_big[0] + w_big[1] + w_big[2] + w_big[3];
        if (s_all) atomicAdd(&sums[0], s_all);
        if (s_big) atomicAdd(&sums[1], s_big);
        if (n_wide && (w_wide[0] | w_wide[1] | w_wide[2] | w_wide[3])) atomicOr(n_wide, 1u);
    }
}
// Clusters of up to wave_max <= kWaveDedupMax members are de-duplicated by the wave that enumerates them
// (fill_members_kernel); the members of larger clusters go through the sorted-run kernels: qbig[q] = their number.
constexpr uint32_t kWaveDedupMax = 256;
__global__ void compact_seeds_kernel(const uint32_t* __restrict__ flag, const uint64_t* __restrict__ scan,
                                     const uint32_t* __restrict__ csize, uint32_t* qseed, uint32_t* qsize, uint64_t n,
                                     uint32_t wave_max, uint32_t* __restrict__ qbig) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) {
        const uint64_t q = scan[i];
        const uint32_t c = csize[i];
        qseed[q] = (uint32_t)i; qsize[q] = c; qbig[q] = c > wave_max ? c : 0u;
    }
}
// out[k] = value[i] (or i) of the k-th flagged i; with `look`: out2[k] = look[out[k]] as well (one launch instead of two)
// keep_total: scan[n], the number of flagged entries, is left there for kernels queued behind (the scan array is reused)
__global__ void compact_index_kernel(const uint32_t* __restrict__ flag, const uint64_t* __restrict__ scan,
                                     uint32_t* out, uint64_t n, const uint32_t* __restrict__ value = nullptr,
                                     const uint32_t* __restrict__ look = nullptr, uint32_t* __restrict__ out2 = nullptr,
                                     unsigned long long* __restrict__ keep_total = nullptr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && keep_total) *keep_total = scan[n];
    if (i < n && flag[i]) {
        const uint32_t x = value ? value[i] : (uint32_t)i;
        out[scan[i]] = x;
        if (look) out2[scan[i]] = look[x];
    }
}

// Hash bits in the member sort key: enough to keep runs of equal keys short (a cluster rarely holds
// more than a few hundred fragments), few enough to save two radix passes over 32 bits.  A collision
// only lengthens a run; groups are decided on the id bytes.
constexpr int kMemberHashBits = 16;

struct MemberView {
    const uint32_t* qseed; const uint64_t* qoff;
    // members of clusters above wave_max, compacted (qboff[q] = first compact index of cluster q):
    uint64_t* mkey;         // (cluster number << kMemberHashBits) | low bits of the fragment-id hash
    uint32_t* bslot;        // their member slot
    const uint64_t* qboff;
    uint32_t wave_max;
    uint32_t* mscore;       // longest-flank score per member slot (detect.rs:273-279)
    const RecWord* rw;      // owner record -> {read_info() word, fragment-id hash}: one 16-byte gather per member
    uint32_t* mrec;         // per member slot: its owner record and the bytes of its junction record, written while the
    uint32_t* mjrec;        // member's RecWord is at hand, so that emit_kernel reads slot space only and gathers nothing
    // in-wave dedup (WAVE_DEDUP): group leaders' representatives and the clusters' group counts
    const uint32_t* qsize; uint32_t* rep; uint32_t* qnrep;
    const uint8_t* frag; const uint64_t* frag_off; const uint16_t* frag_len;
    uint64_t hash_mask;     // all ones; a few bits under BF_FLAG_WEAK_DEDUP_HASH (tests: forces the byte-wise redo)
};

// Two fragment ids byte for byte.  The wave that calls this sits in a chain of round trips to memory, so the compare is
// shaped for few of them: both lengths and both offsets in one, then the first 32 bytes of both ids as eight 8-byte words
// in flight together (ids are rarely longer; the words past an id's end are read and masked -- the id buffers end in 64
// spare bytes), then 8 bytes a step for the rest.  (Round 2's form -- lengths, then offsets, then a loop that waited for
// every two bytes -- was a dozen dependent round trips: ~18 us of a wave's life, most of fill_members_kernel's time.)
__device__ __forceinline__ uint64_t load_u64_any(const uint8_t* p) { uint64_t x; __builtin_memcpy(&x, p, 8); return x; }
__device__ __forceinline__ uint64_t head_bytes_mask(uint32_t len, uint32_t at) {      // bytes [at, at + 8) that lie below len
    const uint32_t nb = len > at ? min(len - at, 8u) : 0u;
    return nb >= 8u ? ~0ull : ((1ull << (8u * nb)) - 1ull);
}
__device__ __forceinline__ bool same_id_bytes(const uint8_t* __restrict__ frag, const uint64_t* __restrict__ frag_off,
                                              const uint16_t* __restrict__ frag_len, uint32_t ra, uint32_t rb) {
    const uint32_t la = frag_len[ra], lb = frag_len[rb];
    const uint64_t oa = frag_off[ra], ob = frag_off[rb];
    const uint8_t* a = frag + oa;
    const uint8_t* b = frag + ob;
    uint64_t xa[4], xb[4];
#pragma unroll
    for (int w = 0; w < 4; w++) { xa[w] = load_u64_any(a + 8 * w); xb[w] = load_u64_any(b + 8 * w); }
    uint64_t diff = (uint64_t)(la ^ lb);
#pragma unroll
    for (int w = 0; w < 4; w++) diff |= (xa[w] ^ xb[w]) & head_bytes_mask(la, 8u * w);
    for (uint32_t k = 32; k < la; k += 8) diff |= (load_u64_any(a + k) ^ load_u64_any(b + k)) & head_bytes_mask(la, k);
    return diff == 0;
}
__device__ __forceinline__ bool same_frag_bytes(const MemberView& m, uint32_t ra, uint32_t rb) {
    return same_id_bytes(m.frag, m.frag_off, m.frag_len, ra, rb);
}

// One wave per qualifying cluster: re-enumerate the seed's members in sweep order.
// A cluster of at most wave_max (<= kWaveDedupMax) members: the same wave then applies
// remove_frag_id_duplicates (detect.rs:247-287) to its cluster -- member i's group leader is the first
// member with the same fragment id (64-bit hashes compared against one broadcast member at a time, id BYTES
// where the hashes agree), a leader's representative the first member of its group with the best score --
// instead of a radix sort of all member slots by (cluster, hash) and a pass over the sorted runs.  The members of
// larger clusters (duplicate-heavy libraries pile thousands of reads on one junction) are written, compacted,
// for exactly that sort.
// Two instantiations share the enumeration: R = 1 takes the clusters of up to 64 members (a member per lane, everything
// in registers: ~50 VGPRs, eight waves a SIMD -- the kernel lives on round trips to memory, so waves in flight are what
// counts), R = kWaveDedupMax / 64 the larger ones (several members per lane, and the clusters above wave_max, which it
// only enumerates); a wave whose cluster belongs to the other one leaves at once.  The second is launched only when the
// host knows of such clusters (flag_ge_kernel counts them).
template <int R>
__global__ void __launch_bounds__(256) fill_members_kernel(const SweepView v, const MemberView m, uint64_t nq) {
    constexpr uint32_t kCap = 64u * R;
    __shared__ uint64_t s_fh[4][kCap];
    __shared__ uint32_t s_sc[4][kCap];
    __shared__ uint32_t s_rc[4][kCap];
    __shared__ uint32_t s_ld[R > 1 ? 4 : 1][R > 1 ? kCap : 1];
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
    if (q >= nq) return;
    const uint32_t size = m.qsize[q];
    const bool in_wave = size <= m.wave_max;
    if ((R == 1) != (in_wave && size <= 64u)) return;                // the other instantiation's cluster (wave-uniform)
    const uint64_t p = m.qseed[q];
    const uint64_t base = m.qoff[q];
    const uint64_t key = v.skey[p];
    const int32_t lp = v.slpos[p], rp = v.srpos[p];
    const uint64_t cbase = in_wave ? 0 : m.qboff[q];
    auto put = [&](uint64_t slot, uint32_t rec) {
        const RecWord w = m.rw[rec];
        const uint64_t info = w.info, fh = w.fhash;
        m.mrec[slot] = rec; m.mjrec[slot] = info_jrec(info);
        const uint32_t l = info_bp(info), r = info_len(info) - l;
        const uint32_t score = min(l, r) * 10000u + max(l, r);       // detect.rs:273-279
        if (in_wave) {
            const uint32_t i = (uint32_t)(slot - base);
            if (i < kCap) { s_fh[wv][i] = fh; s_sc[wv][i] = score; s_rc[wv][i] = rec; }
        } else {
            const uint64_t ci = cbase + (slot - base);
            m.mkey[ci] = (q << kMemberHashBits) | (fh & ((1ull << kMemberHashBits) - 1));
            m.bslot[ci] = (uint32_t)slot;
            m.mscore[slot] = score;
        }
    };
    // The seed is its own first member: the scan starts AT it.  The wave's life is a chain of round trips to HBM (it is
    // latency-bound: some 100k waves of ~18 us each), so everything a step needs is requested at once -- key, both
    // positions and the record id of all 64 candidates, unconditionally -- instead of one array after the other as the
    // short-circuit tests would have it.
    uint64_t cnt = 0;
    for (uint64_t s0 = p;; s0 += 64) {
        const uint64_t s = s0 + lane;
        const bool in = s < v.n;
        const uint64_t ks = in ? v.skey[s] : 0;
        const int32_t ls = in ? v.slpos[s] : 0, rs = in ? v.srpos[s] : 0;
        const uint32_t rec = in ? v.perm[s] : 0;
        const bool live = in && seg_of(ks) == seg_of(key) && !beyond(ls, lp, v.F);
        const uint64_t live_mask = __ballot(live);
        const bool take = live && ks == key && !rpos_far(rs, rp, v.F) && same_sigx(v, key, p, s);
        const uint64_t take_mask = __ballot(take);
        if (take) put(base + cnt + __popcll(take_mask & ((1ull << lane) - 1)), rec);
        cnt += __popcll(take_mask);
        if (live_mask != ~0ull) break;        // the break tests are monotone in sorted order
    }
    if (!in_wave) return;                                            // wave-uniform

    // LDS operations of one wave execute in order; only the compiler must not reorder them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t c = (uint32_t)min(cnt, (uint64_t)kCap);           // == qsize[q]
    if constexpr (R == 1) {
        // The usual cluster: a member per lane, everything in registers.  The kernel is bound by the instructions of the
        // loop over the members (one broadcast + compare per member: ~110k clusters x ~40 members a step), so this
        // loop is as short as it gets -- two v_readlane for the j-th member's hash, no LDS word to wait for, no guards
        // for further rounds -- and the loop over the representatives only visits the members that joined a group.
        const uint32_t i = (uint32_t)lane;
        const bool in = i < c;
        const uint64_t f = in ? s_fh[wv][i] & m.hash_mask : 0;
        const uint32_t rcl = in ? s_rc[wv][i] : 0, sc = in ? s_sc[wv][i] : 0;
        const int flo = (int)(uint32_t)f, fhi = (int)(uint32_t)(f >> 32);
        const uint32_t cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        uint32_t l = i;
        for (uint32_t j = 0; j + 1 < cu; j++) {
            const int jl = __builtin_amdgcn_readlane(flo, (int)j), jh = __builtin_amdgcn_readlane(fhi, (int)j);
            const bool cand = i > j && in && l == i && flo == jl && fhi == jh;
            l = cand ? j : l;
        }
        bool verified = true;
        if (in && l != i) verified = same_frag_bytes(m, rcl, s_rc[wv][l]);
        if (__ballot(!verified)) {                                   // two ids with one hash: bytes, pair by pair
            l = i;
            for (uint32_t j = 0; j + 1 < cu; j++) {
                const int jl = __builtin_amdgcn_readlane(flo, (int)j), jh = __builtin_amdgcn_readlane(fhi, (int)j);
                const bool cand = i > j && in && l == i && flo == jl && fhi == jh;
                bool same = false;
                if (cand) same = same_frag_bytes(m, rcl, s_rc[wv][j]);
                l = same ? j : l;
            }
        }
        uint32_t best = i, bs = sc;
        uint64_t jm = __ballot(in && l != i);                        // ascending: the first maximum wins (strict >, detect.rs:280)
        while (jm) {
            const int i2 = __builtin_ctzll(jm);
            jm &= jm - 1;
            const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)l, i2), si = (uint32_t)__builtin_amdgcn_readlane((int)sc, i2);
            if (i == li && si > bs) { bs = si; best = (uint32_t)i2; }
        }
        const bool leads = in && l == i;
        if (in) m.rep[base + i] = leads ? best : kNone;
        const uint32_t groups = (uint32_t)__popcll(__ballot(leads));
        if (lane == 0) m.qnrep[q] = groups;
        return;
    } else {
    const int rounds = (int)((c + 63) >> 6);
    uint64_t fh[R]; uint32_t ld[R], rc[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = 64u * r + lane;
        fh[r] = i < c ? s_fh[wv][i] : 0; rc[r] = i < c ? s_rc[wv][i] : 0; ld[r] = i;
    }
    // Group leaders: the first member with the same fragment id (detect.rs:252-266).  First on the 64-bit hashes
    // alone (no memory traffic: one broadcast LDS word per member), then every member that joined a group checks
    // its id BYTES against its leader's, all lanes at once.  Equal ids hash equally, so if every check passes the
    // groups are exact; if one fails (two ids with one hash) the wave redoes its cluster comparing bytes pair by pair.
    for (uint32_t j = 0; j + 1 < c; j++) {
        const uint64_t fj = s_fh[wv][j] & m.hash_mask;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (r < rounds) {
                const uint32_t i = 64u * r + lane;
                const bool cand = i > j && i < c && ld[r] == i && (fh[r] & m.hash_mask) == fj;
                ld[r] = cand ? j : ld[r];
            }
        }
    }
    bool verified = true;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = 64u * r + lane;
        if (r < rounds && i < c && ld[r] != i) verified = verified && same_frag_bytes(m, rc[r], s_rc[wv][ld[r]]);
    }
    if (__ballot(!verified)) {
#pragma unroll
        for (int r = 0; r < R; r++) ld[r] = 64u * r + lane;
        for (uint32_t j = 0; j + 1 < c; j++) {
            const uint64_t fj = s_fh[wv][j] & m.hash_mask;
#pragma unroll
            for (int r = 0; r < R; r++) {
                // (written as a select: hipcc 7.2 dropped the assignment from the short-circuit form
                //  `if (... && same_frag_bytes(...)) ld[r] = j;` whenever the byte loop ran)
                const uint32_t i = 64u * r + lane;
                const bool cand = r < rounds && i > j && i < c && ld[r] == i && (fh[r] & m.hash_mask) == fj;
                bool same = false;
                if (cand) same = same_frag_bytes(m, rc[r], s_rc[wv][j]);
                ld[r] = same ? j : ld[r];
            }
        }
    }
    bool joined = false;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = 64u * r + lane;
        if (i < c) { s_ld[wv][i] = ld[r]; joined = joined || ld[r] != i; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // representatives: best score of the group, the first one on ties (strict >, detect.rs:280)
    uint32_t best[R], bs[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const uint32_t i = 64u * r + lane; best[r] = i; bs[r] = i < c ? s_sc[wv][i] : 0; }
    if (__ballot(joined)) {
        for (uint32_t i2 = 1; i2 < c; i2++) {
            const uint32_t li = s_ld[wv][i2];                        // broadcast: uniform over the wave
            if (li == i2) continue;
            const uint32_t si = s_sc[wv][i2];
#pragma unroll
            for (int r = 0; r < R; r++)
                if (64u * r + lane == li && si > bs[r]) { bs[r] = si; best[r] = i2; }
        }
    }
    uint32_t groups = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = 64u * r + lane;
        const bool leads = i < c && ld[r] == i;
        if (i < c) m.rep[base + i] = leads ? best[r] : kNone;
        groups += (uint32_t)__popcll(__ballot(leads));
    }
    if (lane == 0) m.qnrep[q] = groups;
    }
}

struct FragView {
    const uint8_t* frag; const uint64_t* frag_off; const uint16_t* frag_len;
};
__device__ __forceinline__ bool same_frag(const FragView& f, uint32_t ra, uint32_t rb) {
    return same_id_bytes(f.frag, f.frag_off, f.frag_len, ra, rb);
}

// Fragment-id groups (remove_frag_id_duplicates, detect.rs:247-287) without the reference's O(c^2):
// member slots are radix-sorted by (cluster, 32 hash bits); a run of equal sort keys holds every
// member of each of its fragment ids (equal ids hash equally), in cluster order because the sort is
// stable.  The lane at the head of a run applies the reference's own loop to the run -- first
// unassigned member leads, later members with the same id BYTES join it, best score wins with
// "first maximum" (strict >, :280) -- so hash collisions only lengthen a run, never change a group.
// rep[leader slot] = representative (index inside the cluster); kNone elsewhere.
__global__ void __launch_bounds__(256) dedup_runs_kernel(uint64_t total_m, const uint64_t* __restrict__ skey,
                                                         const uint32_t* __restrict__ sslot, const MemberView m,
                                                         const FragView f, uint32_t* __restrict__ leader,
                                                         uint32_t* __restrict__ rep) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total_m) return;
    const uint64_t key = skey[j];
    if (j > 0 && skey[j - 1] == key) return;
    const uint32_t q = (uint32_t)(key >> kMemberHashBits);
    const uint64_t base = m.qoff[q];
    if (j + 1 >= total_m || skey[j + 1] != key) {          // the common case: a fragment seen once
        const uint32_t sa = sslot[j];
        rep[sa] = (uint32_t)(sa - base);
        return;
    }
    for (uint64_t a = j; a < total_m && skey[a] == key; a++) {
        const uint32_t sa = sslot[a];
        if (leader[sa] != kNone) continue;
        leader[sa] = sa;
        const uint32_t ra = m.mrec[sa];
        uint32_t best = sa, best_score = m.mscore[sa];
        for (uint64_t b = a + 1; b < total_m && skey[b] == key; b++) {
            const uint32_t sb = sslot[b];
            if (leader[sb] != kNone) continue;
            if (!same_frag(f, ra, m.mrec[sb])) continue;
            leader[sb] = sa;
            if (m.mscore[sb] > best_score) { best = sb; best_score = m.mscore[sb]; }
        }
        rep[sa] = (uint32_t)(best - base);
    }
}

// qnrep[q] = fragment-id groups of cluster q = member slots that lead a group (one wave per cluster)
__global__ void __launch_bounds__(256) count_groups_kernel(uint64_t nq, const uint64_t* __restrict__ qoff,
                                                           const uint32_t* __restrict__ rep, uint32_t* __restrict__ qnrep,
                                                           const uint32_t* __restrict__ qsize, uint32_t wave_max) {
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (q >= nq || qsize[q] <= wave_max) return;             // small clusters were counted by their own wave
    const uint64_t a = qoff[q], b = qoff[q + 1];
    uint32_t cnt = 0;
    for (uint64_t s = a + lane; s < b; s += 64) cnt += rep[s] != kNone;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane == 0) qnrep[q] = cnt;
}

struct OrderView {
    const uint32_t* qseed; const uint64_t* skey; const int32_t* slpos; const uint32_t* perm;
    const uint64_t* src; int cb;
};
// The short print-order sort.  The order is (contig rank, lpos as the wrapped usize, input index of the seed): 5 + 32 + 24
// and more significant bits, eight 8-bit passes in two sorts of ~50k keys each, sixteen launches at the latency floor --
// longer than the dedup kernels they run beside.  Here: ONE sort on the bits that can differ -- the rank relative to the
// group's first rank, and the position mapped into [0, 2^pos_bits): non-negative positions as they are, the few negative
// ones (the reference's wrapped usize: they sort AFTER every real position, among themselves by value) at the top of the
// range -- 32 bits for hg38 in two groups, four passes; the input index then only orders the rare seeds that share
// contig and position, which a fix-up pass does run by run.  What the short form does not handle (a position outside the
// mapped range, a tie run longer than kTieMax) raises a flag, and the ctx falls back to the two full sorts for good.
constexpr uint32_t kTieMax = 64;
constexpr int32_t kNegRoom = 1 << 18;                      // lpos >= -(65535 + 32767): room for the negative positions
struct ShortOrder { uint32_t rank_lo; int pos_bits; uint32_t* flag; };
__global__ void cluster_keys_short_kernel(const OrderView o, uint64_t* kpos, uint64_t* kidx, uint32_t* __restrict__ iota, uint64_t nq, ShortOrder so) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nq) return;
    iota[k] = (uint32_t)k;
    const uint32_t p = o.qseed[k];
    const uint32_t rec = o.perm[p];
    const int32_t lp = o.slpos[p];
    const int64_t top = (int64_t)1 << so.pos_bits;
    int64_t m = lp >= 0 ? (int64_t)lp : top + lp;
    if ((lp >= 0 && m >= top - kNegRoom) || (lp < 0 && lp < -kNegRoom)) { atomicOr(so.flag, 1u); m = 0; }
    const uint64_t rank = (o.skey[p] >> key_rank_shift(o.cb)) - so.rank_lo;
    kpos[k] = (rank << so.pos_bits) | (uint64_t)m;
    kidx[k] = o.src ? o.src[rec] : (uint64_t)rec;
}
// qord[] is in (rank, position) order, stable on the sorted-space order of the seeds; a run of equal keys is put into
// input-index order by the thread at its head
__global__ void order_ties_kernel(const uint64_t* __restrict__ skeys, uint32_t* __restrict__ qord, const uint64_t* __restrict__ kidx,
                                  uint64_t nq, uint32_t* flag) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nq) return;
    const uint64_t key = skeys[k];
    if ((k > 0 && skeys[k - 1] == key) || k + 1 >= nq || skeys[k + 1] != key) return;      // not the head of a run of two or more
    uint64_t e = k + 2;
    while (e < nq && e - k <= kTieMax && skeys[e] == key) e++;
    if (e - k > kTieMax) { atomicOr(flag, 2u); return; }
    for (uint64_t a = k + 1; a < e; a++) {                   // insertion sort on the seeds' input indices
        const uint32_t q = qord[a];
        const uint64_t x = kidx[q];
        uint64_t b = a;
        while (b > k && kidx[qord[b - 1]] > x) { qord[b] = qord[b - 1]; b--; }
        qord[b] = q;
    }
}

// sort keys of surviving clusters: (contig rank, lpos as sorted, input order of the seed)
__global__ void cluster_keys_kernel(const OrderView o, const uint32_t* __restrict__ sel, uint64_t* kpos, uint64_t* kidx, uint32_t* __restrict__ iota, uint64_t ns) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    iota[k] = (uint32_t)k;                                  // the sort's values
    const uint32_t p = o.qseed[sel ? sel[k] : k];
    const uint32_t rec = o.perm[p];
    kpos[k] = ((o.skey[p] >> key_rank_shift(o.cb)) << 32) | (uint32_t)o.slpos[p];
    kidx[k] = o.src ? o.src[rec] : (uint64_t)rec;
}
struct EmitView {
    const uint32_t* ord;        // output order -> q
    const uint64_t* roff;       // output order -> first supporting read
    const uint32_t* qseed; const uint32_t* qsize; const uint64_t* qoff;
    const uint32_t* mrec; const uint32_t* mjrec; const uint32_t* rep;
    const uint64_t* skey; const int32_t* slpos; const int32_t* srpos; const uint32_t* perm;
    const uint64_t* src; const uint64_t* src_base;
    const uint32_t* rank2id; int cb;
    bf_cluster* clusters; bf_read* reads; uint32_t* out_src_rank; uint32_t* out_src_idx; uint32_t* out_rec; uint32_t* out_jrec;
    uint64_t rd_base;           // reads[], out_*[] start at this group's first read: read_begin / read_end count from here
};

// One wave per reported cluster: the printed row (detect.rs:233-235) and its surviving reads in
// cluster order (a group's representative sits at the slot of the group's first member, :284).
// The number of surviving clusters is read on the device (the grid is sized for the qualifying ones), so that the kernel can be
// queued before the host has the counts; when the reads to print do not fit what is left of the per-read arrays it writes
// nothing (the host sees the count, makes room and launches it again).
__global__ void __launch_bounds__(256) emit_kernel(const EmitView e, const unsigned long long* __restrict__ ns_dev, uint64_t room) {
    const uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const uint64_t ns = *ns_dev;
    if (k >= ns || e.roff[ns] > room) return;
    const uint32_t q = e.ord[k];
    const uint64_t base = e.qoff[q];
    const uint32_t c = e.qsize[q];
    const uint64_t rbase = e.roff[k];
    auto source = [&](uint32_t rec, uint32_t& sr, uint32_t& si) {
        const uint64_t w = e.src ? e.src[rec] : (uint64_t)rec;
        sr = (uint32_t)(w >> 32); si = (uint32_t)w;
    };
    if (lane == 0) {
        const uint32_t p = e.qseed[q];
        const uint64_t key = e.skey[p];
        uint32_t sr, si; source(e.perm[p], sr, si);
        bf_cluster cl;
        memset(&cl, 0, sizeof cl);
        cl.lpos = e.slpos[p]; cl.rpos = e.srpos[p];
        cl.seed_index = e.src_base[sr] + si;
        cl.read_begin = e.rd_base + rbase; cl.read_end = e.rd_base + e.roff[k + 1];
        cl.contig = e.rank2id[(key >> key_rank_shift(e.cb)) & ((1u << e.cb) - 1)];
        cl.mcontig = e.rank2id[(key >> kKeyMrankShift) & ((1u << e.cb) - 1)];
        cl.strand = (key >> kKeyStrandBit) & 1; cl.mstrand = (key >> kKeyMstrandBit) & 1;
        e.clusters[k] = cl;
    }
    uint64_t cnt = 0;
    for (uint32_t i0 = 0; i0 < c; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint32_t rp = i < c ? e.rep[base + i] : kNone;
        const bool lead = rp != kNone;
        const uint64_t mask = __ballot(lead);
        if (lead) {
            const uint64_t slot = rbase + cnt + __popcll(mask & ((1ull << lane) - 1));
            const uint32_t rec = e.mrec[base + rp], jr = e.mjrec[base + rp];      // slot space: no gather (fill_members_kernel)
            uint32_t sr, si; source(rec, sr, si);
            bf_read rd;
            memset(&rd, 0, sizeof rd);
            rd.index = e.src_base[sr] + si;
            e.reads[slot] = rd;
            e.out_src_rank[slot] = sr; e.out_src_idx[slot] = si; e.out_rec[slot] = rec; e.out_jrec[slot] = jr;
        }
        cnt += __popcll(mask);
    }
}

// ---- groups of whole contig ranks --------------------------------------------------------------
// Classes never interact and the print order is contig-major, so the evidence can be clustered a few contig
// ranks at a time and the results appended; api.hip copies a group's records to the host while the next group
// is clustered.  rank32[i] = left contig rank of record i, hist[r] = records of rank r, hist[bins] = sum of all
// records' junction record bytes (the bound the output buffers are sized with).
// Sum of the records' junction record bytes (the bound the output buffers are sized with): one atomic per block.
__global__ void __launch_bounds__(256) record_bytes_kernel(const RecWord* __restrict__ rw, uint64_t n, unsigned long long* sum) {
    unsigned long long bytes = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) bytes += info_jrec(rw[i].info);
    for (int d = 32; d; d >>= 1) bytes += __shfl_down(bytes, d);
    __shared__ unsigned long long w[4];                    // one atomic per block: they all hit one word
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = bytes;
    __syncthreads();
    if (threadIdx.x == 0 && w[0] + w[1] + w[2] + w[3]) atomicAdd(sum, w[0] + w[1] + w[2] + w[3]);
}
// (a group is a run of contig ranks: its number is how many groups start at or below the record's rank.  The first ranks
//  come by value -- a table the host has just made needs no copy and no wait this way)
struct GroupCuts { uint32_t n; uint32_t first_rank[255]; };
__global__ void group_id_kernel(uint32_t* __restrict__ gid, uint32_t* __restrict__ iota, const uint64_t* __restrict__ key,
                                const GroupCuts cuts, int shift, uint32_t mask, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = (uint32_t)(key[i] >> shift) & mask;
    uint32_t g = 0;
    for (uint32_t k = 1; k < cuts.n; k++) g += cuts.first_rank[k] <= r ? 1u : 0u;
    gid[i] = g; iota[i] = (uint32_t)i;
}
// -> bounds: positions in c_gidx (record ids ordered by contig rank, input order inside) where the groups start;
// one group (bounds = {0, n}) when there is nothing to gain.  The reads per left contig rank are known on the host
// (bf_submit's staging threads count them from the hits: the rank after the reorient step is the smaller of the two
// anchors' ranks), so the cut costs the GPU one radix pass on the rank bits of the class keys and a sum.
// Records per left contig rank of an IMPORTED view (the sharded path: what arrived is not what bf_submit counted):
// block histograms in LDS, one atomic per non-empty bin and block.  A launch counts the ranks of one window of
// kRankHistMax bins; genomes with more contigs (up to the 16384 the key has room for) take a launch per window.
constexpr uint32_t kRankHistMax = 4096;
__global__ void __launch_bounds__(256) rank_hist_kernel(unsigned long long* __restrict__ hist, const uint64_t* __restrict__ key, int shift,
                                                        uint32_t mask, uint32_t bin0, uint32_t bins, uint64_t n) {
    __shared__ uint32_t h[kRankHistMax];
    for (uint32_t b = threadIdx.x; b < bins; b += 256) h[b] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t r = ((uint32_t)(key[i] >> shift) & mask) - bin0;      // (wraps for ranks below the window)
        if (r < bins) atomicAdd(&h[r], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += 256) if (h[b]) atomicAdd(&hist[bin0 + b], (unsigned long long)h[b]);
}
// queued on the ctx's stream; the counts are in *h_hist (pinned) after the next wait for it
static int rank_histogram_queue(bf_ctx* ctx, const uint64_t* d_key, uint64_t n, uint64_t** h_hist) {
    const hipStream_t st = ctx->stream;
    BF_TRY(wide_words(ctx, ctx->n_contigs, h_hist));
    BF_TRY(ensure(ctx, ctx->c_bstart, std::max<size_t>((size_t)ctx->n_contigs * 8, 64), false));
    BF_HIP(ctx, hipMemsetAsync(ctx->c_bstart.p, 0, (size_t)ctx->n_contigs * 8, st));
    for (uint32_t bin0 = 0; bin0 < ctx->n_contigs; bin0 += kRankHistMax)
        hipLaunchKernelGGL(rank_hist_kernel, dim3((unsigned)std::min<uint64_t>(grid_for(n, 256), 512)), dim3(256), 0, st,
                           (unsigned long long*)ctx->c_bstart.p, d_key, key_rank_shift(ctx->contig_bits), (1u << ctx->contig_bits) - 1, bin0,
                           std::min(kRankHistMax, ctx->n_contigs - bin0), n);
    BF_HIP(ctx, hipGetLastError());
    BF_HIP(ctx, hipMemcpyAsync(*h_hist, ctx->c_bstart.p, (size_t)ctx->n_contigs * 8, hipMemcpyDeviceToHost, st));
    return BF_OK;
}
int rank_histogram(bf_ctx* ctx, const OwnerView& v, std::vector<uint64_t>& hist) {
    hist.assign(ctx->n_contigs, 0);
    if (v.n == 0) return BF_OK;
    uint64_t* h = nullptr;
    BF_TRY(rank_histogram_queue(ctx, v.key, v.n, &h));
    BF_SYNC(ctx, ctx->stream);
    hist.assign(h, h + ctx->n_contigs);
    return BF_OK;
}

int cluster_groups(bf_ctx* ctx, const OwnerView& v, uint32_t max_groups, std::vector<uint64_t>& bounds, bool want_record_bytes,
                   const std::vector<uint64_t>* view_hist) {
    const uint64_t n = v.n;
    const hipStream_t st = ctx->stream;
    bounds.assign(1, 0); bounds.push_back(n);
    const std::vector<uint64_t>& rank_hist = view_hist ? *view_hist : ctx->rank_hist;
    if (max_groups < 2 || n == 0 || (v.src && !view_hist) || rank_hist.size() != ctx->n_contigs) return BF_OK;
    uint64_t counted = 0;
    for (uint64_t c : rank_hist) counted += c;
    if (counted != n) return fail(ctx, BF_ERR_STATE, "contig histogram of %llu reads for %llu evidence records", (unsigned long long)counted, (unsigned long long)n);
    // cut after the rank where the running count passes the next multiple of n / max_groups
    bounds.assign(1, 0);
    uint64_t run = 0;
    ctx->group_ranks.clear();
    for (uint32_t r = 0; r < ctx->n_contigs; r++) {
        if (ctx->group_ranks.size() < bounds.size()) ctx->group_ranks.emplace_back(r, r);
        ctx->group_ranks.back().second = r;
        run += rank_hist[r];
        if (run >= n * bounds.size() / max_groups && run > bounds.back() && run < n && bounds.size() < 255) bounds.push_back(run);
    }
    bounds.push_back(n);
    BF_TRY(ensure(ctx, ctx->c_gidx, n * 4, false)); BF_TRY(ensure(ctx, ctx->c_hist, 64 + ctx->n_contigs, false));
    BF_TRY(ensure(ctx, ctx->c_c32, n * 4, false)); BF_TRY(ensure(ctx, ctx->c_k64a, n * 8, false));
    if (want_record_bytes) {                              // (the sharded path's records stay with their sources: no bound to compute)
        BF_HIP(ctx, hipMemsetAsync(ctx->c_hist.p, 0, 8, st));
        hipLaunchKernelGGL(record_bytes_kernel, dim3((unsigned)std::min<uint64_t>(grid_for(n, 256), 1024)), dim3(256), 0, st, v.rw, n,
                           (unsigned long long*)ctx->c_hist.p);
        BF_TRY(mail_fetch(ctx, kMailRecordBytes, ctx->c_hist.p));      // the caller reads it after its next wait (no round trip here)
    }
    if (bounds.size() > 2) {
        const unsigned shift = (unsigned)key_rank_shift(ctx->contig_bits);
        {
            // record ids ordered by GROUP, input order inside: a group's ids ascend, so what reads the records through them
            // walks memory once (ordered by rank, a group was a dozen interleaved walks: 835 MB fetched for 60 MB used)
            BF_TRY(ensure(ctx, ctx->c_a32, n * 4, false)); BF_TRY(ensure(ctx, ctx->c_b32, n * 4, false));
            GroupCuts cuts{};
            cuts.n = (uint32_t)ctx->group_ranks.size();
            for (uint32_t g = 0; g < cuts.n; g++) cuts.first_rank[g] = ctx->group_ranks[g].first;
            hipLaunchKernelGGL(group_id_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, ctx->c_b32.as<uint32_t>(), ctx->c_c32.as<uint32_t>(), v.key,
                               cuts, (int)shift, (1u << ctx->contig_bits) - 1, n);
            unsigned gbits = 1;
            while ((1u << gbits) < bounds.size() - 1) gbits++;
            BF_TRY(sort_pairs<uint32_t>(ctx, ctx->c_b32.as<uint32_t>(), ctx->c_a32.as<uint32_t>(), ctx->c_c32.as<uint32_t>(),
                                        ctx->c_gidx.as<uint32_t>(), n, 0, gbits));
        }
    }
    // no wait here: the first group's cluster stage waits soon enough (the byte bound and the localise counters ride on it)
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}
// ---- sorted space without a full sort: class partition + chunk sort ----------------------------------
// What the sweep needs (detect.rs:193-227 restricted to one class, see the header comment of this file) is: the records
// of one signature side by side, inside a signature in (lpos as u32, input index) order.  The ORDER OF THE SIGNATURES
// among themselves is never used (the print order is computed from the seeds, cluster_keys_kernel).  So instead of
// eight LSD radix passes over (lpos, signature) followed by three gathers through the permutation:
//   1. comp[j] = signature << 32 | lpos, val[j] = {record id, rpos, key >> 32}    (one streaming kernel)
//   2. two (three from 16M records on) stable radix passes on the low signature bits: 65536 buckets (2^20), each
//      holding whole signatures, in input order inside
//   3. bucket starts from the partitioned array (no histogram, no atomics), chunks of whole buckets of about
//      kChunkBase records (a bucket belongs to the chunk its start lies in)
//   4. one workgroup per chunk sorts it in LDS (bitonic on (comp, position in the chunk): input order breaks ties,
//      as the stable sort of the reference does) and writes the sorted-space arrays; a record's id, rpos and upper
//      key bits travelled with it through the partition passes (three words of payload), so nothing is gathered
//   5. chunks too large for LDS (a signature with more records than fit: duplicate-heavy input) are set aside, sorted
//      together by the radix sort on all 64 bits of comp and placed after the others.
constexpr uint32_t kChunkBase = 320;            // chunk c = the buckets that start in [c * 320, (c + 1) * 320)
constexpr uint32_t kChunkCap = 1024;            // records a workgroup sorts in LDS

__global__ void comp_build_kernel(uint64_t* __restrict__ comp, RsV3* __restrict__ val, const uint32_t* __restrict__ idx,
                                  const uint64_t* __restrict__ key, const int32_t* __restrict__ lpos, const int32_t* __restrict__ rpos, uint64_t m) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t i = idx ? idx[j] : (uint32_t)j;
    const uint64_t k = key[i];
    comp[j] = (k << 32) | (uint32_t)lpos[i];
    val[j] = RsV3{i, (uint32_t)rpos[i], (uint32_t)(k >> 32)};       // what the sorted space needs besides comp: carried along, not fetched later
}
// bstart[b] = first position of bucket b in the partitioned array (= the next bucket's start when b is empty),
// bstart[n_buckets] = m.  Every entry is written by exactly one thread: the one at the first record of the next
// non-empty bucket (or the last record).
__global__ void class_bucket_start_kernel(uint32_t* __restrict__ bstart, const uint64_t* __restrict__ comp, uint64_t m, int bits) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t mask = (1u << bits) - 1;
    const uint32_t b = (uint32_t)(comp[j] >> 32) & mask;
    const int64_t pb = j ? (int64_t)((uint32_t)(comp[j - 1] >> 32) & mask) : -1;
    for (int64_t x = pb + 1; x <= (int64_t)b; x++) bstart[x] = (uint32_t)j;
    if (j == m - 1) for (uint32_t x = b + 1; x <= mask + 1; x++) bstart[x] = (uint32_t)m;
}
// cstart[c] = where chunk c begins (c <= n_chunks; cstart[n_chunks] = m); osize[c] = its size if it is too large for LDS
// cfirst[c] = that first bucket's number (n_buckets when there is none): chunk c's buckets are [cfirst[c], cfirst[c + 1])
__global__ void chunk_table_kernel(uint32_t* __restrict__ cstart, uint32_t* __restrict__ osize, uint32_t* __restrict__ cfirst,
                                   const uint32_t* __restrict__ bstart, uint32_t n_buckets, uint32_t n_chunks, uint32_t m) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chunks) return;
    auto first_at_or_after = [&](uint64_t pos) -> uint32_t {      // the first bucket that starts at or after pos
        if (pos >= m) return n_buckets;
        uint32_t lo = 0, hi = n_buckets;                           // bstart is non-decreasing, bstart[n_buckets] = m
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (bstart[mid] >= pos) hi = mid; else lo = mid + 1; }
        return lo;
    };
    const uint32_t b = first_at_or_after((uint64_t)c * kChunkBase);
    const uint32_t s = bstart[b];                                  // (bstart[n_buckets] = m)
    cstart[c] = s; cfirst[c] = b;
    if (c < n_chunks) {
        const uint32_t e = bstart[first_at_or_after((uint64_t)(c + 1) * kChunkBase)];
        osize[c] = e - s > kChunkCap ? e - s : 0u;
    }
}
struct SortedOut { uint64_t* skey; int32_t* slpos; int32_t* srpos; uint32_t* perm; };
__device__ __forceinline__ void put_sorted(const SortedOut& o, uint64_t at, uint64_t comp, const RsV3& x) {
    o.skey[at] = ((uint64_t)x.c << 32) | (comp >> 32);
    o.slpos[at] = (int32_t)(uint32_t)comp;
    o.srpos[at] = (int32_t)x.b;
    o.perm[at] = x.a;
}
// A bucket of up to 64 R records sorted by ONE WAVE in registers: element e = 64 r + lane, a bitonic network on (comp,
// position in the bucket) whose steps with a stride below 64 exchange lanes (three 32-bit lane permutes: no LDS memory,
// no bank conflicts, no workgroup barrier) and whose longer strides are compare-exchanges between a lane's own registers.
// A bucket holds whole signatures and the order of the signatures among themselves is never used, so sorting bucket by
// bucket IS the chunk's sort, and a bucket's records stay in its own range.  The payload is fetched through the sorted
// position (a bucket is a few KB: cache hits).
constexpr uint32_t kWaveBucketMax = 256;
// (NB buckets at a time, their networks independent of each other: a second bucket's chain of lane permutes fills the gaps
//  of the first.)  The sort key is ONE 64-bit word: inside a bucket the low `bits` signature bits are the bucket's number,
//  the same for every record, so their place holds the record's position in the bucket instead (below 256 <= 2^bits):
//  key = signature's other bits | lpos | position -- unique, so a compare-exchange is one 64-bit compare and two selects,
//  and a lane exchange two permutes (the kernel is bound by these instructions: 41k waves x 21-33 steps a launch).
template <int R, int NB>
__device__ __forceinline__ void wave_sort_buckets(const uint64_t* __restrict__ comp, const RsV3* __restrict__ val, const SortedOut& out,
                                                  const uint32_t (&s_)[NB], const uint32_t (&size_)[NB], const uint64_t (&out_base_)[NB],
                                                  const uint32_t (&bucket_)[NB], int bits, int lane) {
    uint64_t key[NB * R];
    const uint32_t low = (1u << bits) - 1u;
#pragma unroll
    for (int q = 0; q < NB * R; q++) {
        const uint32_t e = 64u * (q % R) + lane;
        if (e < size_[q / R]) {
            const uint64_t c = comp[s_[q / R] + e];
            key[q] = ((c >> (32 + bits)) << (32 + bits)) | ((uint64_t)(uint32_t)c << bits) | e;
        } else key[q] = ~0ull;                                       // padding: above every record (a position is never all ones)
    }
#pragma unroll
    for (uint32_t k = 2; k <= 64u * R; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {                                           // partner = another register of this lane
                const int jj = (int)(j >> 6);
#pragma unroll
                for (int q = 0; q < NB * R; q++) {
                    const int r = q % R;
                    if (r & jj) continue;
                    const int p = q | jj;                            // (R is a power of two: q | jj stays inside q's bucket)
                    const bool asc = ((64u * r + lane) & k) == 0;
                    const uint64_t a = key[q], b = key[p];
                    const bool sw = (a > b) == asc;
                    key[q] = sw ? b : a; key[p] = sw ? a : b;
                }
            } else {                                                 // partner = lane ^ j, same register
                const bool lower = (lane & j) == 0;
#pragma unroll
                for (int q = 0; q < NB * R; q++) {
                    const bool want_min = lower == (((64u * (q % R) + lane) & k) == 0);   // the lower lane of an ascending pair keeps the smaller
                    const uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)key[q], (int)j), phi = (uint32_t)__shfl_xor((int)(uint32_t)(key[q] >> 32), (int)j);
                    const uint64_t pk = ((uint64_t)phi << 32) | plo;
                    key[q] = ((key[q] > pk) == want_min) ? pk : key[q];
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NB * R; q++) {
        const uint32_t e = 64u * (q % R) + lane;
        if (e < size_[q / R]) {
            const uint64_t k_ = key[q];
            const uint32_t li = (uint32_t)k_ & low, lpos = (uint32_t)(k_ >> bits);
            const uint32_t sig = ((uint32_t)(k_ >> (32 + bits)) << bits) | (bucket_[q / R] & low);
            put_sorted(out, out_base_[q / R] + e, ((uint64_t)sig << 32) | lpos, val[s_[q / R] + li]);
        }
    }
}

__global__ void __launch_bounds__(256) chunk_sort_kernel(const uint32_t* __restrict__ cstart, const uint64_t* __restrict__ obefore,
                                                         const uint64_t* __restrict__ comp, const RsV3* __restrict__ val, SortedOut out,
                                                         uint64_t* __restrict__ side_comp, RsV3* __restrict__ side_val,
                                                         const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ cfirst, int bits) {
    __shared__ uint64_t s_comp[kChunkCap];
    __shared__ uint32_t s_li[kChunkCap];
    __shared__ RsV3 s_v[kChunkCap];
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    const uint32_t s = cstart[c], size = cstart[c + 1] - s;
    if (size == 0) return;
    const uint64_t ob = obefore[c];
    if (size > kChunkCap) {                                      // set aside, in order (stable)
        // ... and copied, unsorted, to where the side sort will put them: the stage runs on without waiting for the count
        // of such records (usually none); should there be some, what follows reads valid records and is redone
        const uint64_t tail = (uint64_t)cstart[gridDim.x] - obefore[gridDim.x];
        for (uint32_t r = t; r < size; r += 256) {
            const uint64_t cc = comp[s + r]; const RsV3 vv = val[s + r];
            side_comp[ob + r] = cc; side_val[ob + r] = vv;
            put_sorted(out, tail + ob + r, cc, vv);
        }
        return;
    }
    {
        // The chunk's buckets are [b0, b1) (chunk_table_kernel).  When none of them is longer than kWaveBucketMax the waves
        // take the non-empty ones in turn and sort them in registers; a chunk with a longer bucket (a signature with hundreds
        // of records) goes through the LDS network below as a whole.  Every wave reads the bucket starts itself, 64 to a
        // load (no walk from one start to the next: each step of such a walk is a round trip, and a dozen of them were
        // longer than the sort).
        const uint32_t b0 = cfirst[c], b1 = cfirst[c + 1];
        const int lane = (int)(t & 63u), wv = (int)(t >> 6);
        uint32_t longest = 0;
        for (uint32_t base = b0; base < b1; base += 64) {
            const uint32_t b = base + lane;
            const uint32_t bn = b < b1 ? bstart[b + 1] - bstart[b] : 0u;
            longest = max(longest, bn);
        }
        for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, o));
        if (longest <= kWaveBucketMax && bits >= 8) {                // block-uniform: every wave saw the same starts
            uint32_t ps[2] = {0, 0}, pn[2] = {0, 0}, pb[2] = {0, 0}; // buckets of up to 64 records wait for a partner
            uint64_t po[2] = {0, 0};
            int held = 0;
            uint32_t seen = 0;                                       // non-empty buckets before this round
            for (uint32_t base = b0; base < b1; base += 64) {
                const uint32_t b = base + lane;
                const uint32_t my_s = b < b1 ? bstart[b] : 0u, my_n = b < b1 ? bstart[b + 1] - my_s : 0u;
                uint64_t todo = __ballot(my_n != 0);
                for (; todo; todo &= todo - 1, seen++) {
                    if ((seen & 3u) != (uint32_t)wv) continue;
                    const int src = __builtin_ctzll(todo);
                    const uint32_t bs = (uint32_t)__builtin_amdgcn_readlane((int)my_s, src), bn = (uint32_t)__builtin_amdgcn_readlane((int)my_n, src);
                    const uint64_t ob_ = (uint64_t)bs - ob;
                    if (bn <= 64) {
                        ps[held] = bs; pn[held] = bn; po[held] = ob_; pb[held] = base + (uint32_t)src;
                        if (++held == 2) { wave_sort_buckets<1, 2>(comp, val, out, ps, pn, po, pb, bits, lane); held = 0; }
                        continue;
                    }
                    const uint32_t s1[1] = {bs}, n1[1] = {bn}, b1_[1] = {base + (uint32_t)src};
                    const uint64_t o1[1] = {ob_};
                    if (bn <= 128) wave_sort_buckets<2, 1>(comp, val, out, s1, n1, o1, b1_, bits, lane);
                    else wave_sort_buckets<4, 1>(comp, val, out, s1, n1, o1, b1_, bits, lane);
                }
            }
            if (held) { pn[1] = 0; wave_sort_buckets<1, 2>(comp, val, out, ps, pn, po, pb, bits, lane); }     // (an empty partner: all padding)
            return;
        }
    }
    uint32_t P = 64;
    while (P < size) P <<= 1;
    for (uint32_t r = t; r < P; r += 256) {
        const bool in = r < size;
        s_comp[r] = in ? comp[s + r] : ~0ull;
        s_li[r] = in ? r : 0xffffffffu;
        if (in) s_v[r] = val[s + r];
    }
    // Bitonic network, pair p of a step with stride j: elements i = (p / j) * 2j + p % j and i + j.  Thread t takes the
    // pairs t, t + 256, ...: with j <= 64 a wave's 64 pairs lie in 128 consecutive elements that no other wave
    // touches, so only the steps with a longer stride (and the step after one) need the whole block to wait.
    bool wide_before = true;                                     // the loads above came from every wave
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const bool wide = j > 64;
            if (wide || wide_before) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            wide_before = wide;
            for (uint32_t p = t; p < P / 2; p += 256) {
                const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i + j;
                const uint64_t a = s_comp[i], b = s_comp[l];
                const uint32_t ai = s_li[i], bi = s_li[l];
                const bool gt = a > b || (a == b && ai > bi);
                const bool asc = (i & k) == 0;
                if (gt == asc) { s_comp[i] = b; s_comp[l] = a; s_li[i] = bi; s_li[l] = ai; }
            }
        }
    }
    __syncthreads();
    const uint64_t base = (uint64_t)s - ob;
    for (uint32_t r = t; r < size; r += 256) put_sorted(out, base + r, s_comp[r], s_v[s_li[r]]);
}
__global__ void side_write_kernel(const uint64_t* __restrict__ comp, const RsV3* __restrict__ val, SortedOut out, uint64_t base, uint64_t n) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) put_sorted(out, base + r, comp[r], val[r]);
}

// records [0, m) of the group (idx: their ids, or null = all of the view) -> sorted space
// wait_over: wait for the number of set-aside records and sort them (one host round trip); otherwise the count is left
// in mailbox slot 3 for the caller's next wait, and the caller redoes the stage with wait_over should it not be zero.
static int class_order(bf_ctx* ctx, const OwnerView& v, const uint32_t* idx, uint64_t m, const SortedOut& out, bool wait_over) {
    const hipStream_t st = ctx->stream;
    const unsigned B = 256;
    int bits = m > (16u << 20) ? 20 : 16;
    if (const char* e = getenv("BREAKFAST_HIP_BUCKET_BITS")) bits = std::min(20, std::max(1, atoi(e)));      // tests: few buckets = chunks that overflow
    // (read on every call on purpose: the tests change it between engines of one process)
    const uint32_t n_buckets = 1u << bits;
    const uint32_t n_chunks = (uint32_t)((m + kChunkBase - 1) / kChunkBase);
    // chunk table: cstart u32[n_chunks + 1] | osize u32[n_chunks] | obefore u64[n_chunks + 1] | cfirst u32[n_chunks + 1]
    const size_t off_osize = ((size_t)(n_chunks + 1) * 4 + 255) & ~(size_t)255;
    const size_t off_obefore = (off_osize + (size_t)n_chunks * 4 + 255) & ~(size_t)255;
    const size_t off_cfirst = (off_obefore + (size_t)(n_chunks + 1) * 8 + 255) & ~(size_t)255;
    const size_t comp_bytes = (m * 8 + 255) & ~(size_t)255;
    BF_TRY(ensure(ctx, ctx->c_chunk, off_cfirst + (size_t)(n_chunks + 1) * 4, false));
    BF_TRY(ensure(ctx, ctx->c_bstart, (size_t)(n_buckets + 1) * 4, false));
    BF_TRY(ensure(ctx, ctx->c_comp, comp_bytes + m * sizeof(RsV3), false));       // partitioned: comp, then the payload
    BF_TRY(ensure(ctx, ctx->c_pos2, m * sizeof(RsV3), false));
    uint64_t* comp_in = ctx->c_k64a.as<uint64_t>(); RsV3* val_in = ctx->c_pos2.as<RsV3>();
    uint64_t* comp_s = ctx->c_comp.as<uint64_t>(); RsV3* val_s = (RsV3*)(ctx->c_comp.as<uint8_t>() + comp_bytes);
    uint32_t* cstart = ctx->c_chunk.as<uint32_t>();
    uint32_t* osize = (uint32_t*)(ctx->c_chunk.as<uint8_t>() + off_osize);
    uint64_t* obefore = (uint64_t*)(ctx->c_chunk.as<uint8_t>() + off_obefore);
    uint32_t* cfirst = (uint32_t*)(ctx->c_chunk.as<uint8_t>() + off_cfirst);
    hipLaunchKernelGGL(comp_build_kernel, dim3(grid_for(m, B)), dim3(B), 0, st, comp_in, val_in, idx, v.key, v.lpos, v.rpos, m);
    BF_TRY(radix_sort_kv3(ctx, comp_in, comp_s, val_in, val_s, m, 32, 32 + (unsigned)bits));
    hipLaunchKernelGGL(class_bucket_start_kernel, dim3(grid_for(m, B)), dim3(B), 0, st, ctx->c_bstart.as<uint32_t>(), comp_s, m, bits);
    hipLaunchKernelGGL(chunk_table_kernel, dim3(grid_for(n_chunks + 1, B)), dim3(B), 0, st, cstart, osize, cfirst, ctx->c_bstart.as<uint32_t>(),
                       n_buckets, n_chunks, (uint32_t)m);
    BF_TRY(scan_u32_to_u64(ctx, osize, obefore, n_chunks));
    hipLaunchKernelGGL(chunk_sort_kernel, dim3(n_chunks), dim3(256), 0, st, cstart, obefore, comp_s, val_s, out, comp_in, val_in,
                       ctx->c_bstart.as<uint32_t>(), cfirst, bits);
    BF_HIP(ctx, hipGetLastError());
    uint64_t n_over = 0;
    if (!wait_over) {                                   // the caller reads mailbox slot 3 at its next wait (usually zero)
        BF_TRY(mail_fetch(ctx, 3, obefore + n_chunks));
        return BF_OK;
    }
    BF_TRY(fetch_u64(ctx, obefore + n_chunks, &n_over));
    if (n_over) {
        if (n_over > m) return fail(ctx, BF_ERR_HIP, "class order: %llu records set aside out of %llu", (unsigned long long)n_over, (unsigned long long)m);
        BF_TRY(radix_sort_kv3(ctx, comp_in, comp_s, val_in, val_s, n_over, 0, 64));
        hipLaunchKernelGGL(side_write_kernel, dim3(grid_for(n_over, B)), dim3(B), 0, st, comp_s, val_s, out, m - n_over, n_over);
        BF_HIP(ctx, hipGetLastError());
    }
    return BF_OK;
}

// ---- the cluster stage -----------------------------------------------------------------------
static int run_cluster_once(bf_ctx* ctx, const OwnerView& v, ClusterGroup* grp, bool wait_over, bool* redo);

// The stage does not wait for the count of records the LDS chunk sort had to set aside (a signature with more than 1024
// records: duplicate-heavy libraries); the count comes back with the sweep's own wait.  Should it not be zero, the stage is
// done again for this group with the wait, and the ctx waits from then on (such inputs have them in every group).
int run_cluster(bf_ctx* ctx, const OwnerView& v, ClusterGroup* grp) {
    for (int attempt = 0; attempt < 3; attempt++) {       // (each reason for a second go turns itself off)
        bool redo = false;
        BF_TRY(run_cluster_once(ctx, v, grp, ctx->saw_overflow, &redo));
        if (!redo) return BF_OK;
    }
    return fail(ctx, BF_ERR_STATE, "the cluster stage asked for another go three times");
}

static int run_cluster_once(bf_ctx* ctx, const OwnerView& v, ClusterGroup* grp, bool wait_over, bool* redo) {
    *redo = false;
    const uint64_t n = grp ? grp->m : v.n;                 // records in sorted space; record ids stay v's
    hipStream_t st = ctx->stream;
    if (!grp || grp->cl_base + grp->rd_base == 0) {
        ctx->out.n_clusters = ctx->out.n_reads = ctx->out.junction_size = 0;
        ctx->n_evidence = v.n;
    }
    if (grp) grp->ns = grp->nr = 0;
    if (n == 0) return BF_OK;
    if (n > 0xfffffff0ull) return fail(ctx, BF_ERR_RANGE, "too many evidence records for one GPU (%llu)", (unsigned long long)n);
    const unsigned B = 256;
    // `cluster.len() < min_evidence` (detect.rs:229,231) is never true for 0, and a cluster always holds its seed:
    // --min-evidence=0 prints what 1 prints.  (Unclamped, `csize >= 0` would also flag the non-seeds, whose csize is 0.)
    const uint32_t min_evidence = std::max(ctx->cfg.min_evidence, 1u);

    // 1. sorted space: every signature's records side by side in (lpos, input index) order (class_order)
    BF_TRY(ensure(ctx, ctx->c_a32, n * 4)); BF_TRY(ensure(ctx, ctx->c_b32, n * 4));
    BF_TRY(ensure(ctx, ctx->c_c32, n * 4));
    BF_TRY(ensure(ctx, ctx->c_k64a, n * 8)); BF_TRY(ensure(ctx, ctx->c_k64b, n * 8));
    BF_TRY(ensure(ctx, ctx->c_slpos, n * 4)); BF_TRY(ensure(ctx, ctx->c_srpos, n * 4));
    BF_TRY(ensure(ctx, ctx->c_pos2, v.n * 8));
    uint32_t* perm = ctx->c_a32.as<uint32_t>();
    {
        Scope t(ctx, "sort");
        const SortedOut so{ctx->c_k64b.as<uint64_t>(), ctx->c_slpos.as<int32_t>(), ctx->c_srpos.as<int32_t>(), perm};
        BF_TRY(class_order(ctx, v, grp ? grp->idx : nullptr, n, so, wait_over));
    }
    BF_HIP(ctx, hipGetLastError());

    SweepView sv{};
    sv.n = n; sv.skey = ctx->c_k64b.as<uint64_t>(); sv.slpos = ctx->c_slpos.as<int32_t>();
    sv.srpos = ctx->c_srpos.as<int32_t>(); sv.perm = perm; sv.sigx = v.sigx; sv.sigx_slot = v.sigx_slot;
    sv.F = (int64_t)ctx->cfg.max_frag_len;

    // 2. sweep
    BF_TRY(ensure(ctx, ctx->c_reported, n)); BF_TRY(ensure(ctx, ctx->c_csize, n * 4));
    BF_TRY(ensure(ctx, ctx->c_flag, n * 4)); BF_TRY(ensure(ctx, ctx->c_scan, (n + 1) * 8));
    uint64_t nq = 0;
    uint64_t totals[2] = {0, 0};
    bool wide_clusters = false;                            // some qualifying cluster has more than 64 members (fill_members_kernel<4>)
    const uint32_t wave_max = (ctx->cfg.flags & BF_FLAG_SORTED_DEDUP) ? 0u : kWaveDedupMax;
    unsigned long long* sums = nullptr;
    {
        Scope t(ctx, "sweep");
        // lists of segments that need a walk: counts[0] short ones, counts[1] long ones
        // c_big: counts[2], then four 64-bit sums (flag_ge_kernel), then the lists
        BF_TRY(ensure(ctx, ctx->c_big, (n + n / kBigSegment + 8 + kCtlWords) * 4));
        uint32_t* counts = ctx->c_big.as<uint32_t>();
        sums = (unsigned long long*)(counts + 2);
        uint32_t* big_list = counts + kCtlWords; uint32_t* small_list = big_list + n / kBigSegment + 2;
        uint32_t* head = ctx->c_flag.as<uint32_t>();          // reused for the min_evidence flags below
        uint32_t* covered = ctx->c_b32.as<uint32_t>();        // free since the sort
        hipLaunchKernelGGL(head_mark_kernel, dim3(grid_for(std::max<uint64_t>(n, kCtlWords), B)), dim3(B), 0, st, sv.skey, head, n,
                           ctx->c_reported.as<uint8_t>(), ctx->c_csize.as<uint32_t>(), counts);
        BF_TRY(scan_max_u32_inplace(ctx, head, n));
        hipLaunchKernelGGL(covered_kernel, dim3(grid_for(n, B)), dim3(B), 0, st, sv, head, covered);
        BF_TRY(scan_u32_to_u64(ctx, covered, ctx->c_scan.as<uint64_t>(), n));
        hipLaunchKernelGGL(tails_kernel, dim3(grid_for(n, B)), dim3(B), 0, st, sv, head, ctx->c_scan.as<uint64_t>(),
                           ctx->c_csize.as<uint32_t>(), small_list, big_list, counts);
        {
            const unsigned list_blocks = (unsigned)std::min<uint64_t>(grid_for(n, 256), 1024);
            const unsigned big_blocks = (unsigned)std::min<uint64_t>(grid_for(n, kBigSegment) / 4 + 1, 1024);
            hipLaunchKernelGGL(sweep_walk_kernel, dim3(list_blocks + big_blocks), dim3(B), 0, st, sv, ctx->c_reported.as<uint8_t>(),
                               ctx->c_csize.as<uint32_t>(), small_list, big_list, counts, list_blocks);
        }
        hipLaunchKernelGGL(flag_ge_kernel, dim3((unsigned)std::min<uint64_t>(grid_for(n, B), kFlagGrid)), dim3(B), 0, st,
                           ctx->c_flag.as<uint32_t>(), ctx->c_csize.as<uint32_t>(), min_evidence, n, wave_max, sums, (const uint32_t*)nullptr, counts + 13);
        BF_TRY(scan_u32_to_u64(ctx, ctx->c_flag.as<uint32_t>(), ctx->c_scan.as<uint64_t>(), n));
        // members of the qualifying clusters, and of those too large for the in-wave dedup; the number of qualifying
        // clusters; the records class_order set aside: one wait for all of them (mailbox)
        BF_TRY(mail_fetch(ctx, 0, sums, 2));
        BF_TRY(mail_fetch(ctx, 2, ctx->c_scan.as<uint64_t>() + n));
        BF_TRY(mail_fetch(ctx, 4, counts + 12));            // (word 13: a qualifying cluster has more than 64 members)
        BF_SYNC(ctx, st);
        totals[0] = ctx->h_mail[0]; totals[1] = ctx->h_mail[1]; nq = ctx->h_mail[2];
        wide_clusters = (ctx->h_mail[4] >> 32) != 0;
        if (!wait_over && ctx->h_mail[3] != 0) { ctx->saw_overflow = true; *redo = true; return BF_OK; }
        BF_TRY(settle_counts(ctx));                       // the localise counters came along with this synchronisation
    }
    if (nq == 0) return BF_OK;

    // 3. members of qualifying clusters
    BF_TRY(ensure(ctx, ctx->c_qseed, nq * 4)); BF_TRY(ensure(ctx, ctx->c_qsize, nq * 4));
    BF_TRY(ensure(ctx, ctx->c_qoff, (nq + 1) * 8)); BF_TRY(ensure(ctx, ctx->c_qnrep, nq * 4));
    const uint64_t total_m = totals[0], total_big = totals[1];
    BF_TRY(ensure(ctx, ctx->c_ctl, nq * 4 + (nq + 1) * 8 + 64));
    uint64_t* qboff = ctx->c_ctl.as<uint64_t>();                       // nq + 1 compact offsets of the large clusters
    uint32_t* qbig = (uint32_t*)(qboff + nq + 1);
    hipLaunchKernelGGL(compact_seeds_kernel, dim3(grid_for(n, B)), dim3(B), 0, st, ctx->c_flag.as<uint32_t>(),
                       ctx->c_scan.as<uint64_t>(), ctx->c_csize.as<uint32_t>(), ctx->c_qseed.as<uint32_t>(),
                       ctx->c_qsize.as<uint32_t>(), n, wave_max, qbig);
    BF_TRY(ensure(ctx, ctx->c_qord, nq * 4));
    BF_TRY(ensure(ctx, ctx->c_misc, nq * 8 * 4));
    BF_HIP(ctx, hipEventRecord(ctx->ev[9], st));          // the seeds are known: the side stream may order them (queued below)
    BF_TRY(scan_u32_to_u64(ctx, ctx->c_qsize.as<uint32_t>(), ctx->c_qoff.as<uint64_t>(), nq));
    if (total_big) BF_TRY(scan_u32_to_u64(ctx, qbig, qboff, nq));      // (no cluster above wave_max: nothing reads the offsets)
    if (total_m > 0xfffffff0ull) return fail(ctx, BF_ERR_RANGE, "cluster membership list too long (%llu)", (unsigned long long)total_m);
    BF_TRY(ensure(ctx, ctx->c_mrec, total_m * 4)); BF_TRY(ensure(ctx, ctx->c_mjrec, total_m * 4));
    BF_TRY(ensure(ctx, ctx->c_k64a, std::max(total_big, n) * 8, false));
    BF_TRY(ensure(ctx, ctx->c_mkey_s, std::max<uint64_t>(total_big, 1) * 8)); BF_TRY(ensure(ctx, ctx->c_slot, std::max<uint64_t>(total_big, 1) * 4 * 2));
    BF_TRY(ensure(ctx, ctx->c_d32, total_m * 4));
    BF_TRY(ensure(ctx, ctx->c_leader, total_m * 4)); BF_TRY(ensure(ctx, ctx->c_rep, total_m * 4));
    MemberView mv{};
    mv.qseed = ctx->c_qseed.as<uint32_t>(); mv.qoff = ctx->c_qoff.as<uint64_t>();
    mv.mkey = ctx->c_k64a.as<uint64_t>(); mv.mscore = ctx->c_d32.as<uint32_t>();
    mv.bslot = ctx->c_slot.as<uint32_t>(); mv.qboff = qboff; mv.wave_max = wave_max;
    mv.rw = v.rw; mv.mrec = ctx->c_mrec.as<uint32_t>(); mv.mjrec = ctx->c_mjrec.as<uint32_t>();
    mv.qsize = ctx->c_qsize.as<uint32_t>(); mv.rep = ctx->c_rep.as<uint32_t>(); mv.qnrep = ctx->c_qnrep.as<uint32_t>();
    mv.frag = v.frag; mv.frag_off = v.frag_off; mv.frag_len = v.frag_len;
    mv.hash_mask = (ctx->cfg.flags & BF_FLAG_WEAK_DEDUP_HASH) ? 0x7ull : ~0ull;
    FragView fv{v.frag, v.frag_off, v.frag_len};
    {
        Scope t(ctx, "dedup");
        if (total_big) {                                 // the sorted-run kernels find their slots untouched
            BF_HIP(ctx, hipMemsetAsync(ctx->c_leader.p, 0xff, total_m * 4, st));
            BF_HIP(ctx, hipMemsetAsync(ctx->c_rep.p, 0xff, total_m * 4, st));
        }
        hipLaunchKernelGGL(fill_members_kernel<1>, dim3(grid_for(nq * 64, B)), dim3(B), 0, st, sv, mv, nq);
        if (wide_clusters || total_big)
            hipLaunchKernelGGL(fill_members_kernel<kWaveDedupMax / 64>, dim3(grid_for(nq * 64, B)), dim3(B), 0, st, sv, mv, nq);
        if (total_big) {
            uint32_t* slot_in = mv.bslot; uint32_t* slot_s = slot_in + total_big;
            unsigned qbits = 1; while ((1ull << qbits) < nq) qbits++;
            BF_TRY(sort_pairs<uint64_t>(ctx, mv.mkey, ctx->c_mkey_s.as<uint64_t>(), slot_in, slot_s, total_big, 0, kMemberHashBits + qbits));
            hipLaunchKernelGGL(dedup_runs_kernel, dim3(grid_for(total_big, B)), dim3(B), 0, st, total_big, ctx->c_mkey_s.as<uint64_t>(),
                               slot_s, mv, fv, ctx->c_leader.as<uint32_t>(), ctx->c_rep.as<uint32_t>());
            hipLaunchKernelGGL(count_groups_kernel, dim3(grid_for(nq * 64, B)), dim3(B), 0, st, nq, ctx->c_qoff.as<uint64_t>(),
                               ctx->c_rep.as<uint32_t>(), ctx->c_qnrep.as<uint32_t>(), mv.qsize, wave_max);
        }
    }
    // The reference's print order of the QUALIFYING clusters (their seeds are known now; which of them survive
    // the dedup only selects from this order): two small, latency-bound radix sorts, queued on the side stream
    // so that they run beside the member / dedup kernels instead of after them.
    // (Queued AFTER the dedup kernels on purpose: the host needs ~0.2 ms to issue these sixteen launches, and while it
    // did so before the dedup kernels were queued the main stream sat idle for that long.)
    BF_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev[9], 0));
    uint32_t* order_flag = nullptr;
    {
        OnSideStream aux(ctx);
        Scope t(ctx, "order");
        const hipStream_t as = ctx->stream;
        uint64_t* kpos = ctx->c_misc.as<uint64_t>();
        uint64_t* kidx = kpos + nq; uint64_t* kpos2 = kidx + nq; uint64_t* kidx2 = kpos2 + nq;
        OrderView ov{ctx->c_qseed.as<uint32_t>(), sv.skey, sv.slpos, perm, v.src, ctx->contig_bits};
        // values = q; c_b32 / c_c32 are free since the sweep
        uint32_t* va = ctx->c_b32.as<uint32_t>(); uint32_t* vb = ctx->c_c32.as<uint32_t>();
        if (!ctx->order_full) {
            // the short form: one sort on (rank in the group, mapped position), ties put right afterwards
            uint32_t rank_lo = 0, rank_hi = ctx->n_contigs - 1;
            if (grp && grp->g >= 0 && (size_t)grp->g < ctx->group_ranks.size()) { rank_lo = ctx->group_ranks[grp->g].first; rank_hi = ctx->group_ranks[grp->g].second; }
            uint64_t longest = 1;
            for (uint32_t r = rank_lo; r <= rank_hi; r++) longest = std::max<uint64_t>(longest, ctx->contig_host[ctx->contig_by_rank[r]].len);
            int pos_bits = 1; while (((uint64_t)1 << pos_bits) < longest + (1u << 17) + (uint64_t)kNegRoom) pos_bits++;
            int rank_bits = 0; while (((uint32_t)1 << rank_bits) < rank_hi - rank_lo + 1) rank_bits++;
            order_flag = ctx->c_big.as<uint32_t>() + 12;   // a control word the sweep is done with (cleared by head_mark_kernel)
            hipLaunchKernelGGL(cluster_keys_short_kernel, dim3(grid_for(nq, B)), dim3(B), 0, as, ov, kpos, kidx, va, nq,
                               ShortOrder{rank_lo, pos_bits, order_flag});
            BF_TRY(sort_pairs<uint64_t>(ctx, kpos, kpos2, va, ctx->c_qord.as<uint32_t>(), nq, 0, (unsigned)(pos_bits + rank_bits)));
            hipLaunchKernelGGL(order_ties_kernel, dim3(grid_for(nq, B)), dim3(B), 0, as, kpos2, ctx->c_qord.as<uint32_t>(), kidx, nq, order_flag);
        } else {
            // LSD: input order first, then (contig rank, lpos)
            hipLaunchKernelGGL(cluster_keys_kernel, dim3(grid_for(nq, B)), dim3(B), 0, as, ov, (const uint32_t*)nullptr, kpos, kidx, va, nq);
            // significant bits of (source rank << 32 | source index): on one rank the index alone, below n
            unsigned idx_bits = 64;
            if (!v.src) { idx_bits = 1; while (idx_bits < 32 && (1ull << idx_bits) < v.n) idx_bits++; }
            else { unsigned rb = 1; while ((1u << rb) < ctx->n_ranks) rb++; idx_bits = 32 + rb; }
            BF_TRY(sort_pairs<uint64_t>(ctx, kidx, kidx2, va, vb, nq, 0, idx_bits));
            hipLaunchKernelGGL(gather_key_kernel, dim3(grid_for(nq, B)), dim3(B), 0, as, kpos2, kpos, vb, nq);
            BF_TRY(sort_pairs<uint64_t>(ctx, kpos2, kidx2, vb, ctx->c_qord.as<uint32_t>(), nq, 0, 32 + ctx->contig_bits));
        }
        BF_HIP(ctx, hipEventRecord(ctx->ev[12], as));
    }
    BF_HIP(ctx, hipGetLastError());

    // 4. surviving clusters in the reference's print order: the survivors of the order computed beside the dedup
    // 5. emit rows + surviving reads.  In contig groups emit is queued BEFORE the host knows how many clusters survive and how
    //    many reads print (it reads the count on the device; its grid and the row buffer are sized for the qualifying
    //    clusters, the per-read arrays have the group's capacity): the round trip for the counts runs beside it instead of
    //    in front of it (round 3: 0.1 ms of idle GPU per group, select and emit being a dozen launches of a few us each).
    uint64_t ns = 0, total_r = 0;
    BF_TRY(ensure(ctx, ctx->c_flag, std::max(nq, n) * 4, false));
    BF_TRY(ensure(ctx, ctx->c_ord, nq * 4));
    {
        Scope t(ctx, "order_wait");                      // what the main stream idles for the print-order sorts on the side stream
        BF_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[12], 0));
    }
    const uint64_t cl_base = grp ? grp->cl_base : 0, rd_base = grp ? grp->rd_base : 0;
    unsigned long long* ns_dev = (unsigned long long*)(ctx->c_big.as<uint32_t>() + 14);      // a control word pair nothing else uses
    uint32_t* d_rank2id = nullptr;
    auto tables = [&]() -> int {
        // the two small tables emit reads (index base per source rank, contig id per rank): uploaded when they change, not
        // per call (two copies from pageable memory in front of every emit were two round trips)
        std::vector<uint64_t> bases = ctx->src_index_base;
        if (bases.empty()) bases.assign(1, ctx->index_base);
        std::vector<uint64_t> key = bases;
        key.push_back(ctx->n_contigs);
        for (uint32_t r = 0; r < ctx->n_contigs; r++) key.push_back(ctx->contig_by_rank[r]);
        const size_t need = bases.size() * 8 + (size_t)ctx->n_contigs * 4 + 16;
        const bool fresh = ctx->c_tables.cap < need || key != ctx->tables_key;
        BF_TRY(ensure(ctx, ctx->c_tables, need));
        d_rank2id = (uint32_t*)((uint8_t*)ctx->c_tables.p + bases.size() * 8);
        if (fresh) {
            BF_HIP(ctx, hipMemcpyAsync(ctx->c_tables.p, bases.data(), bases.size() * 8, hipMemcpyHostToDevice, st));
            BF_HIP(ctx, hipMemcpyAsync(d_rank2id, ctx->contig_by_rank.data(), (size_t)ctx->n_contigs * 4, hipMemcpyHostToDevice, st));
            BF_SYNC(ctx, st);                             // (the sources are host vectors)
            ctx->tables_key = key;
        }
        return BF_OK;
    };
    auto emit = [&](uint64_t rows, uint64_t stride) -> int {          // rows: how many the row buffer must hold beyond cl_base
        BF_TRY(ensure(ctx, ctx->c_clusters, (cl_base + rows) * sizeof(bf_cluster), grp != nullptr));   // earlier groups' rows stay
        BF_TRY(ensure(ctx, ctx->c_reads, std::max<uint64_t>(stride, 1) * sizeof(bf_read), grp != nullptr && rd_base != 0));
        BF_TRY(ensure(ctx, ctx->c_rsrc, std::max<uint64_t>(stride, 1) * 4 * 4));
        ctx->rsrc_stride = stride;
        BF_TRY(tables());
        EmitView ev{};
        ev.ord = ctx->c_ord.as<uint32_t>(); ev.roff = ctx->c_scan.as<uint64_t>();
        ev.qseed = ctx->c_qseed.as<uint32_t>(); ev.qsize = ctx->c_qsize.as<uint32_t>(); ev.qoff = ctx->c_qoff.as<uint64_t>();
        ev.mrec = ctx->c_mrec.as<uint32_t>(); ev.mjrec = ctx->c_mjrec.as<uint32_t>(); ev.rep = ctx->c_rep.as<uint32_t>();
        ev.skey = sv.skey; ev.slpos = sv.slpos; ev.srpos = sv.srpos; ev.perm = perm;
        ev.src = v.src; ev.src_base = (const uint64_t*)ctx->c_tables.p;
        ev.rank2id = d_rank2id; ev.cb = ctx->contig_bits;
        ev.clusters = ctx->c_clusters.as<bf_cluster>() + cl_base; ev.reads = ctx->c_reads.as<bf_read>() + rd_base;
        ev.out_src_rank = ctx->c_rsrc.as<uint32_t>() + rd_base; ev.out_src_idx = ev.out_src_rank + stride;
        ev.out_rec = ev.out_src_idx + stride; ev.out_jrec = ev.out_rec + stride;
        ev.rd_base = rd_base;
        Scope t(ctx, "emit");
        hipLaunchKernelGGL(emit_kernel, dim3(grid_for(rows * 64, B)), dim3(B), 0, st, ev, ns_dev, stride - rd_base);
        BF_HIP(ctx, hipGetLastError());
        return BF_OK;
    };
    // the per-read arrays: sized for every evidence record when groups append -- which is not a bound: a read that an
    // earlier seed reported joins a later seed's cluster as well (detect.rs:215-226), so more reads can print than there
    // is evidence.  Then the four planes of c_rsrc move apart (earlier groups' entries kept) and the table grows.
    uint64_t stride = grp ? std::max<uint64_t>(grp->cap, rd_base ? ctx->rsrc_stride : 0) : 0;
    bool emitted = false;
    {
        Scope t(ctx, "select");
        const uint32_t* qord = ctx->c_qord.as<uint32_t>();
        hipLaunchKernelGGL(flag_ge_kernel, dim3((unsigned)std::min<uint64_t>(grid_for(nq, B), kFlagGrid)), dim3(B), 0, st,
                           ctx->c_flag.as<uint32_t>(), ctx->c_qnrep.as<uint32_t>(), min_evidence, nq, 0u, sums + 2, qord);
        BF_TRY(scan_u32_to_u64(ctx, ctx->c_flag.as<uint32_t>(), ctx->c_scan.as<uint64_t>(), nq));
        BF_TRY(mail_fetch(ctx, 0, sums + 2));              // the reads that will be printed
        BF_TRY(mail_fetch(ctx, 1, ctx->c_scan.as<uint64_t>() + nq));
        if (order_flag) BF_TRY(mail_fetch(ctx, 2, ctx->c_big.as<uint32_t>() + 12));      // (8 bytes: the flag and its neighbour)
        // ord[k'] = the k'-th surviving cluster in print order; nrep in that order -> read offsets (over the qualifying
        // clusters' slots: what lies behind the survivors' is never read)
        uint32_t* vb = ctx->c_c32.as<uint32_t>();
        hipLaunchKernelGGL(compact_index_kernel, dim3(grid_for(nq, B)), dim3(B), 0, st, ctx->c_flag.as<uint32_t>(),
                           ctx->c_scan.as<uint64_t>(), ctx->c_ord.as<uint32_t>(), nq, qord, ctx->c_qnrep.as<uint32_t>(), vb, ns_dev);
        BF_TRY(scan_u32_to_u64(ctx, vb, ctx->c_scan.as<uint64_t>(), nq));
    }
    if (grp) { BF_TRY(emit(nq, stride)); emitted = true; }
    BF_SYNC(ctx, st);
    total_r = ctx->h_mail[0]; ns = ctx->h_mail[1];
    if (order_flag && (uint32_t)ctx->h_mail[2] != 0) { ctx->order_full = true; *redo = true; return BF_OK; }
    if (ns == 0) return BF_OK;
    if (!grp) stride = total_r;
    if (grp && rd_base + total_r > stride) {
        const uint64_t wider = std::max(rd_base + total_r, stride + stride / 2);
        if (rd_base) {
            DevBuf moved;
            BF_TRY(ensure(ctx, moved, wider * 4 * 4));
            for (int pl = 0; pl < 4; pl++)
                if (hipMemcpyAsync(moved.as<uint32_t>() + (size_t)pl * wider, ctx->c_rsrc.as<uint32_t>() + (size_t)pl * stride, rd_base * 4,
                                   hipMemcpyDeviceToDevice, st) != hipSuccess) { release(ctx, moved); return fail(ctx, BF_ERR_HIP, "device copy failed"); }
            if (hipStreamSynchronize(st) != hipSuccess) { release(ctx, moved); return fail(ctx, BF_ERR_HIP, "device copy failed"); }
            release(ctx, ctx->c_rsrc);
            ctx->c_rsrc = moved;
        }
        stride = wider;
        grp->cap = wider;
        emitted = false;                                   // (the queued emit saw that its reads do not fit and wrote nothing)
    }
    if (!emitted) BF_TRY(emit(ns, stride));
    // host copies: the rows now; the reads after the junction offsets are patched in (api.hip)
    if (grp) {                                            // the caller copies the rows when the last group is through
        grp->ns = ns; grp->nr = total_r;
        ctx->out.n_clusters = cl_base + ns; ctx->out.n_reads = rd_base + total_r;
        return BF_OK;
    }
    ctx->out.n_clusters = ns; ctx->out.n_reads = total_r;
    BF_TRY(ensure_host(ctx, ctx->out.clusters, ns * sizeof(bf_cluster)));
    // on the copy stream: 6 MB of rows would otherwise hold up the render kernels for 0.1 ms
    BF_HIP(ctx, hipEventRecord(ctx->ev[11], st));
    BF_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[11], 0));
    BF_HIP(ctx, hipMemcpyAsync(ctx->out.clusters.p, ctx->c_clusters.p, ns * sizeof(bf_cluster), hipMemcpyDeviceToHost, ctx->copy_stream));
    return BF_OK;
}

// ---- multi-GPU exchange: pack by owner rank / import ------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}
// owner = hash(class key) % n_ranks: the key holds the whole class and no positions, so every
// member of a cluster (same class, windows of positions) lands on one rank.
__global__ void owner_kernel(uint32_t* owner, uint32_t* __restrict__ iota, const uint64_t* __restrict__ key, uint32_t n_ranks, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { owner[i] = (uint32_t)(mix64(key[i]) % n_ranks); iota[i] = (uint32_t)i; }
}
__global__ void bucket_start_kernel(uint64_t* start, const uint32_t* __restrict__ owner_sorted, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (i == 0 || owner_sorted[i - 1] != owner_sorted[i])) start[owner_sorted[i]] = i;
}

struct PackView {
    uint64_t n; uint32_t rec_bytes, frag_stride, has_sigx;
    const uint32_t* order;
    const uint64_t* key; const int32_t* lpos; const int32_t* rpos;
    const uint16_t* bp; const uint16_t* len; const uint16_t* flags;
    const uint8_t* frag; const uint64_t* frag_off; const uint16_t* frag_len;
    const uint8_t* sigx; const uint32_t* sigx_slot;
    const RecWord* rw;                                              // read_info() word (junction record bytes) + fragment-id hash
    uint8_t* out;
};
__global__ void __launch_bounds__(256) pack_kernel(const PackView v) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.n) return;
    const uint32_t i = v.order[j];
    uint8_t* dst = v.out + j * v.rec_bytes;
    WireHeader h;
    const RecWord w = v.rw[i];
    h.key = v.key[i]; h.lpos = v.lpos[i]; h.rpos = v.rpos[i]; h.fhash = w.fhash; h.src_idx = i;
    h.bp = v.bp[i]; h.len = v.len[i]; h.frag_len = v.frag_len[i]; h.flags = v.flags[i]; h.jrec = info_jrec(w.info);
    *(WireHeader*)dst = h;
    const uint8_t* f = v.frag + v.frag_off[i];
    for (uint32_t k = 0; k < v.frag_stride; k++) dst[sizeof(WireHeader) + k] = k < h.frag_len ? f[k] : 0;
    if (v.has_sigx) {
        uint8_t* sx = dst + sizeof(WireHeader) + v.frag_stride;
        const bool x = (h.key >> kKeySigxBit) & 1;
        const uint8_t* src = x ? v.sigx + (uint64_t)v.sigx_slot[i] * 16 : nullptr;
        for (int k = 0; k < 16; k++) sx[k] = x ? src[k] : 0;
    }
}

// Same records, staged through LDS: every lane assembles its record as 8-byte words in the block's LDS
// tile, then the tile (256 consecutive records = one contiguous piece of the send buffer) is written
// with fully coalesced stores.  The plain kernel above stores 56-byte records lane by lane, bytewise
// for the fragment id: 1.28 ms per 10M records against 0.3 ms here.  Used when a record is at most
// kWireLdsMaxRecord bytes (fragment ids of up to 200 bytes), i.e. practically always.
constexpr uint32_t kWireLdsMaxRecord = 256;

__device__ __forceinline__ void build_wire_record(const PackView& v, uint32_t i, uint64_t* rec) {
    const uint32_t fwords = v.frag_stride >> 3;
    {
        const uint64_t key = v.key[i];
        const uint32_t fl = v.frag_len[i];
        rec[0] = key;
        rec[1] = (uint64_t)(uint32_t)v.lpos[i] | ((uint64_t)(uint32_t)v.rpos[i] << 32);
        const RecWord rwd = v.rw[i];
        rec[2] = rwd.fhash;
        rec[3] = (uint64_t)i | ((uint64_t)v.bp[i] << 32) | ((uint64_t)v.len[i] << 48);
        rec[4] = (uint64_t)(uint16_t)fl | ((uint64_t)v.flags[i] << 16) | ((uint64_t)info_jrec(rwd.info) << 32);
        // the id as whole words, zero beyond its end (the id buffer ends in 64 spare bytes: bf_submit)
        const uint8_t* f = v.frag + v.frag_off[i];
        for (uint32_t w = 0; w < fwords; w++) rec[5 + w] = 8 * w < fl ? load_u64_any(f + 8 * w) & head_bytes_mask(fl, 8 * w) : 0ull;
        if (v.has_sigx) {
            uint64_t s0 = 0, s1 = 0;
            if ((key >> kKeySigxBit) & 1) {
                const uint8_t* src = v.sigx + (uint64_t)v.sigx_slot[i] * 16;
                for (int b = 0; b < 8; b++) { s0 |= (uint64_t)src[b] << (8 * b); s1 |= (uint64_t)src[8 + b] << (8 * b); }
            }
            rec[5 + fwords] = s0; rec[6 + fwords] = s1;
        }
    }
}
__global__ void __launch_bounds__(256) pack_lds_kernel(const PackView v) {
    extern __shared__ uint64_t tile[];
    const uint32_t words = v.rec_bytes >> 3;
    const uint64_t j0 = (uint64_t)blockIdx.x * 256, j = j0 + threadIdx.x;
    if (j < v.n) build_wire_record(v, v.order[j], tile + (uint64_t)threadIdx.x * words);
    __syncthreads();
    const uint64_t in_tile = min((uint64_t)256, v.n - j0) * words;
    uint64_t* out = (uint64_t*)v.out + j0 * words;
    for (uint64_t t = threadIdx.x; t < in_tile; t += 256) out[t] = tile[t];
}

// The partition without a sort and without a gather: count -> scan -> scatter.  Every workgroup takes 256 consecutive
// records; pack_count_kernel leaves its records per owner in cnt[owner * n_blocks + block]; an exclusive scan over that
// array (owner-major) is, for every (owner, block), the position among ALL records sent of the block's first record for
// that owner: the owners' pieces back to back, the source's order kept inside a piece (detect.rs:193-198: ties fall in
// input order).  pack_scatter_kernel then builds its 256 records in LDS, ordered by (owner, input order) -- a lane's place
// among its wave's records for the same owner from ballots, the waves' counts from LDS --, and writes every owner's run
// to its place with coalesced stores.  The records are read once, in input order (round 2's form sorted (owner, index)
// pairs and gathered the records through the order: with W owners, W interleaved walks over every input array).
constexpr uint32_t kScatterMaxRanks = 64;
__global__ void __launch_bounds__(256) pack_count_kernel(uint32_t* __restrict__ cnt, const uint64_t* __restrict__ key, uint32_t W, uint64_t n,
                                                         uint32_t n_blocks) {
    __shared__ uint32_t c[kScatterMaxRanks];
    if (threadIdx.x < W) c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&c[(uint32_t)(mix64(key[i]) % W)], 1u);
    __syncthreads();
    if (threadIdx.x < W) cnt[(uint64_t)threadIdx.x * n_blocks + blockIdx.x] = c[threadIdx.x];
}
// per_owner[o] = records for owner o (o < W), from the scanned counts
__global__ void pack_totals_kernel(uint64_t* __restrict__ per_owner, const uint64_t* __restrict__ off, uint32_t W, uint32_t n_blocks) {
    const uint32_t o = threadIdx.x;
    if (o < W) per_owner[o] = off[(uint64_t)(o + 1) * n_blocks] - off[(uint64_t)o * n_blocks];
}
__global__ void __launch_bounds__(256) pack_scatter_kernel(const PackView v, const uint64_t* __restrict__ off, uint32_t W, uint32_t n_blocks) {
    extern __shared__ uint64_t tile[];                       // 256 records, then the small tables
    const uint32_t words = v.rec_bytes >> 3;
    uint32_t* wcnt = (uint32_t*)(tile + 256 * (uint64_t)words);     // [4][W]: a wave's records per owner
    uint32_t* pre = wcnt + 4 * W;                                   // [W + 1]: where owner o's run starts in the tile
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (uint32_t k = t; k < 4 * W; k += 256) wcnt[k] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256 + t;
    const bool valid = i < v.n;
    const uint32_t o = valid ? (uint32_t)(mix64(v.key[i]) % W) : 0u;
    uint32_t rank = 0;
    uint64_t todo = __ballot(valid);
    while (todo) {                                           // one round per owner present in the wave
        const int lead = __builtin_ctzll(todo);
        const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)o, lead);
        const uint64_t m = __ballot(valid && o == o0);
        if (valid && o == o0) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (lane == lead) wcnt[wv * W + o0] = (uint32_t)__popcll(m);
        todo &= ~m;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < W; k++) { pre[k] = run; run += wcnt[k] + wcnt[W + k] + wcnt[2 * W + k] + wcnt[3 * W + k]; }
        pre[W] = run;
    }
    __syncthreads();
    if (valid) {
        uint32_t at = pre[o] + rank;
        for (int w = 0; w < wv; w++) at += wcnt[w * W + o];
        build_wire_record(v, (uint32_t)i, tile + (uint64_t)at * words);
    }
    __syncthreads();
    const uint32_t in_tile = pre[W] * words;
    uint64_t* out = (uint64_t*)v.out;
    for (uint32_t x = t; x < in_tile; x += 256) {
        const uint32_t r = x / words, w = x - r * words;
        uint32_t lo = 0, hi = W;                             // the owner whose run holds tile record r: pre[lo] <= r < pre[lo + 1]
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= r) lo = mid; else hi = mid; }
        out[(off[(uint64_t)lo * n_blocks + blockIdx.x] + (r - pre[lo])) * words + w] = tile[x];
    }
}

struct ImportView {
    uint64_t n; uint32_t rec_bytes, frag_stride, has_sigx, n_ranks;
    const uint8_t* in; const uint64_t* src_start;   // n_ranks+1 prefix of src_counts
    const uint8_t* self_in; uint64_t self_begin, self_end;   // records [self_begin, self_end) lie at self_in instead (may be empty)
    uint64_t* key; int32_t* lpos; int32_t* rpos; uint16_t* bp; uint16_t* len; uint16_t* flags;
    uint8_t* frag; uint64_t* frag_off; uint16_t* frag_len; uint8_t* sigx; uint32_t* sigx_slot;
    uint64_t* src; RecWord* rw;
};
__global__ void __launch_bounds__(256) import_kernel(const ImportView v) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.n) return;
    const uint8_t* src = (j >= v.self_begin && j < v.self_end) ? v.self_in + (j - v.self_begin) * v.rec_bytes : v.in + j * v.rec_bytes;
    const WireHeader h = *(const WireHeader*)src;
    v.key[j] = h.key; v.lpos[j] = h.lpos; v.rpos[j] = h.rpos;
    v.bp[j] = h.bp; v.len[j] = h.len; v.flags[j] = h.flags; v.frag_len[j] = h.frag_len;
    v.rw[j] = RecWord{read_info(h.jrec, h.flags, h.bp, h.len), h.fhash};
    v.frag_off[j] = j * v.frag_stride;
    for (uint32_t k = 0; k < v.frag_stride; k++) v.frag[j * v.frag_stride + k] = src[sizeof(WireHeader) + k];
    v.sigx_slot[j] = (uint32_t)j;
    if (v.has_sigx) for (int k = 0; k < 16; k++) v.sigx[j * 16 + k] = src[sizeof(WireHeader) + v.frag_stride + k];
    uint32_t r = 0;
    while (r + 1 < v.n_ranks && j >= v.src_start[r + 1]) r++;
    v.src[j] = ((uint64_t)r << 32) | h.src_idx;
}

__global__ void __launch_bounds__(256) import_lds_kernel(const ImportView v) {
    extern __shared__ uint64_t tile[];
    const uint32_t words = v.rec_bytes >> 3, fwords = v.frag_stride >> 3;
    const uint64_t j0 = (uint64_t)blockIdx.x * 256, j = j0 + threadIdx.x;
    const uint64_t n_tile = min((uint64_t)256, v.n - j0), in_tile = n_tile * words;
    if (j0 >= v.self_end || j0 + n_tile <= v.self_begin || (j0 >= v.self_begin && j0 + n_tile <= v.self_end)) {
        // (nearly every tile: all of it from one buffer)
        const uint64_t* in = j0 >= v.self_begin && j0 < v.self_end ? (const uint64_t*)v.self_in + (j0 - v.self_begin) * words : (const uint64_t*)v.in + j0 * words;
        for (uint64_t t = threadIdx.x; t < in_tile; t += 256) tile[t] = in[t];
    } else {
        for (uint64_t t = threadIdx.x; t < in_tile; t += 256) {
            const uint64_t jj = j0 + t / words, w = t % words;
            tile[t] = (jj >= v.self_begin && jj < v.self_end) ? ((const uint64_t*)v.self_in)[(jj - v.self_begin) * words + w] : ((const uint64_t*)v.in)[jj * words + w];
        }
    }
    __syncthreads();
    if (j >= v.n) return;
    const uint64_t* rec = tile + (uint64_t)threadIdx.x * words;
    v.key[j] = rec[0];
    v.lpos[j] = (int32_t)(uint32_t)rec[1]; v.rpos[j] = (int32_t)(uint32_t)(rec[1] >> 32);
    v.bp[j] = (uint16_t)(rec[3] >> 32); v.len[j] = (uint16_t)(rec[3] >> 48);
    v.frag_len[j] = (uint16_t)rec[4]; v.flags[j] = (uint16_t)(rec[4] >> 16);
    v.rw[j] = RecWord{read_info((uint32_t)(rec[4] >> 32), (uint32_t)(uint16_t)(rec[4] >> 16), (uint32_t)(uint16_t)(rec[3] >> 32), (uint32_t)(rec[3] >> 48)), rec[2]};
    v.frag_off[j] = j * v.frag_stride;
    uint64_t* fdst = (uint64_t*)(v.frag + j * v.frag_stride);
    for (uint32_t w = 0; w < fwords; w++) fdst[w] = rec[5 + w];
    v.sigx_slot[j] = (uint32_t)j;
    if (v.has_sigx) { uint64_t* sd = (uint64_t*)(v.sigx + j * 16); sd[0] = rec[5 + fwords]; sd[1] = rec[6 + fwords]; }
    uint32_t r = 0;
    while (r + 1 < v.n_ranks && j >= v.src_start[r + 1]) r++;
    v.src[j] = ((uint64_t)r << 32) | (uint32_t)rec[3];
}

int run_partition(bf_ctx* ctx, const bf_wire_local& agreed, void* dev_send, uint64_t dev_send_bytes, uint64_t* counts,
                  void** dev_out, uint32_t* record_bytes) {
    const uint64_t n = ctx->n_reads;
    const uint32_t R = ctx->n_ranks;
    hipStream_t st = ctx->stream;
    const uint32_t stride = (agreed.max_frag_len_bytes + 7u) & ~7u;
    const uint32_t rec = (uint32_t)sizeof(WireHeader) + stride + (agreed.has_exact_sig ? 16u : 0u);
    ctx->wire_frag_stride = stride; ctx->wire_has_sigx = agreed.has_exact_sig ? 1 : 0; ctx->wire_rec_bytes = rec;
    *record_bytes = rec;
    for (uint32_t r = 0; r < R; r++) counts[r] = 0;
    if (dev_send) {
        if (dev_send_bytes < n * rec)
            return fail(ctx, BF_ERR_INVALID, "bf_exchange_pack: send buffer of %llu bytes is smaller than %llu records x %u bytes",
                        (unsigned long long)dev_send_bytes, (unsigned long long)n, rec);
        *dev_out = dev_send;
    } else {
        BF_TRY(ensure(ctx, ctx->d_send, std::max<uint64_t>(n, 1) * rec));
        *dev_out = ctx->d_send.p;
    }
    if (n == 0) return BF_OK;
    Scope t(ctx, "exchange_pack");
    PackView pv{};
    pv.n = n; pv.rec_bytes = rec; pv.frag_stride = stride; pv.has_sigx = ctx->wire_has_sigx; pv.order = nullptr;
    pv.key = ctx->d_key.as<uint64_t>(); pv.lpos = ctx->d_lpos.as<int32_t>(); pv.rpos = ctx->d_rpos.as<int32_t>();
    pv.bp = ctx->d_bp.as<uint16_t>(); pv.len = ctx->d_len.as<uint16_t>();
    pv.flags = ctx->d_flags.as<uint16_t>(); pv.frag = ctx->d_frag.as<uint8_t>(); pv.frag_off = ctx->d_frag_off.as<uint64_t>();
    pv.frag_len = ctx->d_frag_len.as<uint16_t>(); pv.sigx = ctx->d_sigx.as<uint8_t>(); pv.sigx_slot = ctx->d_sigx_slot.as<uint32_t>();
    pv.rw = ctx->d_jrec.as<RecWord>();
    pv.out = (uint8_t*)*dev_out;
    const bool sorted_pack = getenv("BREAKFAST_HIP_SORTED_PACK") != nullptr;    // (A/B switch and test hook: the round-2 form; read on every call)
    const size_t lds = (size_t)256 * rec + (size_t)(4 * R + R + 1) * 4;
    if (R <= kScatterMaxRanks && lds <= 65536 && ((uintptr_t)pv.out & 7) == 0 && n < (1ull << 32) && !sorted_pack) {
        const uint32_t nb = (uint32_t)grid_for(n, 256);
        const uint64_t cells = (uint64_t)R * nb;
        BF_TRY(ensure(ctx, ctx->d_part_key, cells * 4 + 64)); BF_TRY(ensure(ctx, ctx->d_part_idx, (cells + 1) * 8));
        BF_TRY(ensure(ctx, ctx->d_part_tmp, (size_t)(R + 1) * 8));
        uint32_t* cnt = ctx->d_part_key.as<uint32_t>(); uint64_t* off = ctx->d_part_idx.as<uint64_t>();
        uint64_t* h_counts = nullptr;
        BF_TRY(wide_words(ctx, R, &h_counts));
        hipLaunchKernelGGL(pack_count_kernel, dim3(nb), dim3(256), 0, st, cnt, pv.key, R, n, nb);
        BF_TRY(scan_u32_to_u64(ctx, cnt, off, cells));
        hipLaunchKernelGGL(pack_totals_kernel, dim3(1), dim3(std::max(64u, R)), 0, st, ctx->d_part_tmp.as<uint64_t>(), off, R, nb);
        BF_HIP(ctx, hipMemcpyAsync(h_counts, ctx->d_part_tmp.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
        hipLaunchKernelGGL(pack_scatter_kernel, dim3(nb), dim3(256), lds, st, pv, off, R, nb);
        BF_HIP(ctx, hipGetLastError());
        BF_SYNC(ctx, st);
        for (uint32_t r = 0; r < R; r++) counts[r] = h_counts[r];
        return BF_OK;
    }
    BF_TRY(ensure(ctx, ctx->d_part_key, n * 4 * 2)); BF_TRY(ensure(ctx, ctx->d_part_idx, n * 4 * 2));
    BF_TRY(ensure(ctx, ctx->d_part_tmp, (size_t)(R + 1) * 8));
    uint32_t* own = ctx->d_part_key.as<uint32_t>(); uint32_t* own_s = own + n;
    uint32_t* idx = ctx->d_part_idx.as<uint32_t>(); uint32_t* order = idx + n;
    const unsigned B = 256;
    hipLaunchKernelGGL(owner_kernel, dim3(grid_for(n, B)), dim3(B), 0, st, own, idx, ctx->d_key.as<uint64_t>(), R, n);
    unsigned bits = 1; while ((1u << bits) < R) bits++;
    BF_TRY(sort_pairs<uint32_t>(ctx, own, own_s, idx, order, n, 0, bits));   // stable: source order kept per owner
    std::vector<uint64_t> start(R + 1, n);
    BF_HIP(ctx, hipMemcpyAsync(ctx->d_part_tmp.p, start.data(), (size_t)(R + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bucket_start_kernel, dim3(grid_for(n, B)), dim3(B), 0, st, ctx->d_part_tmp.as<uint64_t>(), own_s, n);
    pv.order = order;
    if (rec <= kWireLdsMaxRecord && ((uintptr_t)pv.out & 7) == 0)
        hipLaunchKernelGGL(pack_lds_kernel, dim3(grid_for(n, 256)), dim3(256), 256 * rec, st, pv);
    else
        hipLaunchKernelGGL(pack_kernel, dim3(grid_for(n, B)), dim3(B), 0, st, pv);
    BF_HIP(ctx, hipGetLastError());
    BF_HIP(ctx, hipMemcpyAsync(start.data(), ctx->d_part_tmp.p, (size_t)(R + 1) * 8, hipMemcpyDeviceToHost, st));
    BF_SYNC(ctx, st);
    // empty buckets keep the sentinel n: a bucket ends where the next non-empty one starts
    uint64_t end = n;
    for (int r = (int)R - 1; r >= 0; r--) {
        if (start[r] == n) { counts[r] = 0; continue; }
        counts[r] = end - start[r];
        end = start[r];
    }
    return BF_OK;
}

int run_import(bf_ctx* ctx, const void* dev_recv, const uint64_t* src_counts, std::vector<uint64_t>* rank_hist, const uint8_t* dev_self, uint32_t self) {
    if (rank_hist) rank_hist->assign(ctx->n_contigs, 0);
    const uint32_t R = ctx->n_ranks;
    hipStream_t st = ctx->stream;
    std::vector<uint64_t> start(R + 1, 0);
    for (uint32_t r = 0; r < R; r++) start[r + 1] = start[r] + src_counts[r];
    const uint64_t n = start[R];
    ctx->n_imported = n; ctx->imported = true;
    if (n == 0) return BF_OK;
    if (ctx->wire_rec_bytes == 0) return fail(ctx, BF_ERR_STATE, "bf_exchange_import before bf_exchange_pack");
    const uint32_t stride = ctx->wire_frag_stride;
    BF_TRY(ensure(ctx, ctx->i_key, n * 8)); BF_TRY(ensure(ctx, ctx->i_lpos, n * 4)); BF_TRY(ensure(ctx, ctx->i_rpos, n * 4));
    BF_TRY(ensure(ctx, ctx->i_bp, n * 2)); BF_TRY(ensure(ctx, ctx->i_len, n * 2));
    BF_TRY(ensure(ctx, ctx->i_flags, n * 2)); BF_TRY(ensure(ctx, ctx->i_frag, n * stride + 64));          // (same_id_bytes reads whole words past an id's end)
    BF_TRY(ensure(ctx, ctx->i_frag_off, n * 8)); BF_TRY(ensure(ctx, ctx->i_frag_len, n * 2));
    BF_TRY(ensure(ctx, ctx->i_sigx, n * 16)); BF_TRY(ensure(ctx, ctx->i_sigx_slot, n * 4));
    BF_TRY(ensure(ctx, ctx->i_src, n * 8));
    BF_TRY(ensure(ctx, ctx->i_jrec, n * sizeof(RecWord)));
    BF_TRY(ensure(ctx, ctx->d_part_tmp, (size_t)(R + 1) * 8));
    BF_HIP(ctx, hipMemcpyAsync(ctx->d_part_tmp.p, start.data(), (size_t)(R + 1) * 8, hipMemcpyHostToDevice, st));
    ImportView v{};
    v.n = n; v.rec_bytes = ctx->wire_rec_bytes; v.frag_stride = stride; v.has_sigx = ctx->wire_has_sigx; v.n_ranks = R;
    v.in = (const uint8_t*)dev_recv; v.src_start = ctx->d_part_tmp.as<uint64_t>();
    v.self_in = dev_self; v.self_begin = v.self_end = 0;
    if (dev_self && self < R) { v.self_begin = start[self]; v.self_end = start[self + 1]; }
    v.key = ctx->i_key.as<uint64_t>(); v.lpos = ctx->i_lpos.as<int32_t>(); v.rpos = ctx->i_rpos.as<int32_t>();
    v.bp = ctx->i_bp.as<uint16_t>(); v.len = ctx->i_len.as<uint16_t>();
    v.flags = ctx->i_flags.as<uint16_t>(); v.frag = ctx->i_frag.as<uint8_t>(); v.frag_off = ctx->i_frag_off.as<uint64_t>();
    v.frag_len = ctx->i_frag_len.as<uint16_t>(); v.sigx = ctx->i_sigx.as<uint8_t>(); v.sigx_slot = ctx->i_sigx_slot.as<uint32_t>();
    v.src = ctx->i_src.as<uint64_t>(); v.rw = ctx->i_jrec.as<RecWord>();
    Scope t(ctx, "exchange_import");
    if (v.rec_bytes <= kWireLdsMaxRecord && ((uintptr_t)v.in & 7) == 0 && ((uintptr_t)v.self_in & 7) == 0)
        hipLaunchKernelGGL(import_lds_kernel, dim3(grid_for(n, 256)), dim3(256), 256 * v.rec_bytes, st, v);
    else
        hipLaunchKernelGGL(import_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, v);
    BF_HIP(ctx, hipGetLastError());
    uint64_t* h_hist = nullptr;
    if (rank_hist) BF_TRY(rank_histogram_queue(ctx, v.key, n, &h_hist));
    BF_SYNC(ctx, st);   // dev_recv may be released by the caller on return
    if (rank_hist) rank_hist->assign(h_hist, h_hist + ctx->n_contigs);
    return BF_OK;
}

// reads[i].junction_off from the render offsets
__global__ void patch_reads_off_kernel(bf_read* reads, const uint64_t* __restrict__ joff, uint64_t n, uint64_t byte_base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reads[i].junction_off = byte_base + joff[i];
}
// reads [rd_base, rd_base + n) of the table: junction_off = byte_base + d_joff[i]
int patch_reads_off(bf_ctx* ctx, const uint64_t* d_joff, uint64_t n, uint64_t rd_base, uint64_t byte_base) {
    if (n) hipLaunchKernelGGL(patch_reads_off_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                              ctx->c_reads.as<bf_read>() + rd_base, d_joff, n, byte_base);
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}
// out[c] = where cluster c's first record starts (c = n_clusters: the end), from the per-read offsets and the
// clusters' read offsets the order stage left in c_scan
__global__ void cluster_record_offsets_kernel(uint64_t* out, const uint64_t* __restrict__ joff,
                                              const uint64_t* __restrict__ roff, uint64_t n, uint64_t byte_base) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) out[c] = byte_base + joff[roff[c]];
}
// d_out[0 .. n_clusters] for the clusters whose read offsets the order stage just left in c_scan (a group's, or all)
int cluster_record_offsets(bf_ctx* ctx, const uint64_t* d_joff, uint64_t* d_out, uint64_t n_clusters, uint64_t byte_base) {
    hipLaunchKernelGGL(cluster_record_offsets_kernel, dim3(grid_for(n_clusters + 1, 256)), dim3(256), 0, ctx->stream,
                       d_out, d_joff, ctx->c_scan.as<uint64_t>(), n_clusters + 1, byte_base);
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}
// Owner side of the junction routing in one go: the supporting reads ordered by (piece, source rank), result
// order inside; what each (piece, rank) group asks for and how many record bytes it will bring back.
__global__ void request_keys_kernel(uint32_t* __restrict__ key, uint32_t* __restrict__ val, const uint32_t* __restrict__ src_rank,
                                    uint64_t nr, uint32_t K, uint32_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr) return;
    uint32_t c = (uint32_t)((i * K) / nr);                    // piece c = reads [nr*c/K, nr*(c+1)/K)
    while (c + 1 < K && ((uint64_t)(c + 1) * nr) / K <= i) c++;
    key[i] = c * W + src_rank[i];
    val[i] = (uint32_t)i;
}
// ... and, in the same launch (its first threads), start[k] = first position whose sorted key is >= k (k = groups: the end)
__global__ void request_gather_kernel(uint32_t* __restrict__ req, const uint32_t* __restrict__ order, const uint32_t* __restrict__ src_idx,
                                      uint64_t nr, uint64_t* __restrict__ start, const uint32_t* __restrict__ skey, uint32_t groups) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j <= groups) {
        uint64_t lo = 0, hi = nr;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (skey[mid] < j) lo = mid + 1; else hi = mid;
        }
        start[j] = lo;
    }
    if (j < nr) req[j] = src_idx[order[j]];
}
// The same for supporting reads [rd_base, rd_base + nr) of a result that is appended group by group (one piece):
// d_req[j] = source index of the j-th request in (source rank, result order); start[s] = first request to rank s.
// Leaves the order in c_d32 and the sorted source ranks in c_b32 (place_kernel reads them).
int order_requests_range(bf_ctx* ctx, uint64_t rd_base, uint64_t nr, uint32_t* d_req, uint64_t** d_start) {
    const uint32_t W = ctx->n_ranks;
    const uint64_t stride = ctx->rsrc_stride;
    const hipStream_t st = ctx->stream;
    BF_TRY(ensure(ctx, ctx->c_a32, nr * 4, false)); BF_TRY(ensure(ctx, ctx->c_b32, nr * 4, false));
    BF_TRY(ensure(ctx, ctx->c_c32, nr * 4, false)); BF_TRY(ensure(ctx, ctx->c_d32, nr * 4, false));
    BF_TRY(ensure(ctx, ctx->c_aoff, (size_t)(W + 1) * 8 + 64));
    uint64_t* start = ctx->c_aoff.as<uint64_t>();
    const uint32_t* src = ctx->c_rsrc.as<uint32_t>() + rd_base;     // out_src_rank, out_src_idx, out_rec, out_jrec: `stride` apart
    hipLaunchKernelGGL(request_keys_kernel, dim3(grid_for(nr, 256)), dim3(256), 0, st, ctx->c_a32.as<uint32_t>(),
                       ctx->c_c32.as<uint32_t>(), src, nr, 1u, W);
    unsigned bits = 1; while ((1u << bits) < W) bits++;
    BF_TRY(sort_pairs<uint32_t>(ctx, ctx->c_a32.as<uint32_t>(), ctx->c_b32.as<uint32_t>(), ctx->c_c32.as<uint32_t>(),
                                ctx->c_d32.as<uint32_t>(), nr, 0, bits));
    hipLaunchKernelGGL(request_gather_kernel, dim3(grid_for(std::max<uint64_t>(nr, W + 1), 256)), dim3(256), 0, st, d_req, ctx->c_d32.as<uint32_t>(),
                       src + stride, nr, start, ctx->c_b32.as<uint32_t>(), W);
    BF_HIP(ctx, hipGetLastError());
    *d_start = start;
    return BF_OK;
}
int patch_reads(bf_ctx* ctx, const uint64_t* d_joff, uint64_t n) { return patch_reads_off(ctx, d_joff, n); }

}  // namespace bf
