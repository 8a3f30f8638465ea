// Device-wide primitives of the cluster stage, written for gfx950 (wave64, 160 KB LDS): a stable LSD radix
// sort of (key, value) pairs and two prefix scans.  Reference: the stable `sort_by` of detect.rs:193-198 and
// the bookkeeping around it; the parallel formulation is ours.
//
// Every pass is three launches with no waiting between workgroups:
//   upsweep    a tile of 4096 keys -> its 256 digit counts                     (reads the keys only)
//   tile scan  per digit, the exclusive prefix of the counts over the tiles    (a few MB)
//   downsweep  ranks the tile's pairs and scatters them                        (reads and writes the pairs once)
// A single-pass ("onesweep") variant with a chained look-back over per-tile status words was measured here and
// dropped: 10M pairs are 2442 tiles, of which ~770 are resident at once, so nearly a third of all tiles start
// together and each of them has to add up hundreds of predecessors' aggregates (2 KB each, from uncached
// memory): +40 us per pass where the upsweep costs +20 us (profiles/r01j_radix_sort.md).  Without spinning
// there is also no forward-progress assumption about the order in which workgroups are scheduled.
//
// Ranking, per 8-bit digit: the wave's 64 keys of one row are matched with eight ballots (the set of lanes
// holding the same digit), a per-wave digit counter in LDS hands out the running rank, the tile's pairs are
// reordered through LDS so that every digit's run leaves as one contiguous, coalesced piece.
// Stability: tile order = (wave, row, lane) = input order; ranks follow it.

#include "bf_internal.h"

namespace bf {


constexpr int kRsThreads = 256;
constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRsBins = 256;
#ifndef RS_ITEMS
#define RS_ITEMS 16
#endif
// pairs per thread: RS_ITEMS (tiles of 4096 pairs) for one word of payload, 8 for three
constexpr uint32_t kRsFewTiles = 64;                 // up to here the downsweep scans the tile counts itself

// ---- radix sort --------------------------------------------------------------------------------------
// lanes of the wave whose digit equals this lane's (eight ballots)
__device__ __forceinline__ uint64_t same_digit_lanes(uint32_t d) {
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

// upsweep: counts[d * tiles + tile] = keys of the tile with digit d.  Plain LDS atomics: with spread-out digits
// they cost a few cycles per 64 keys, and with the few-valued upper digits of the class keys (contig ranks,
// strands) no more than matching the lanes with ballots first would (measured: 30 -> 14 us per 10M keys).
template <class K, int kRsItems>
__global__ void __launch_bounds__(256) rs_upsweep_kernel(const K* __restrict__ kin, uint64_t n, int shift, uint32_t digit_mask,
                                                         uint32_t* __restrict__ counts, uint32_t tiles) {
    constexpr int kRsTile = kRsThreads * kRsItems;
    __shared__ uint32_t h[kRsBins];
    const int t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kRsTile;
    K key[kRsItems];
#pragma unroll
    for (int i = 0; i < kRsItems; i++) {
        const uint64_t idx = base + (uint64_t)i * 256 + t;
        key[i] = idx < n ? kin[idx] : (K)0;
    }
#pragma unroll
    for (int i = 0; i < kRsItems; i++)
        if (base + (uint64_t)i * 256 + t < n) atomicAdd(&h[(uint32_t)(key[i] >> shift) & digit_mask], 1u);
    __syncthreads();
    counts[(uint64_t)t * tiles + blockIdx.x] = h[t];
}

// tile scan: block d turns counts[d][0..tiles) into exclusive prefixes over the tiles and leaves the digit's
// total in totals[d].
__global__ void __launch_bounds__(256) rs_tilescan_kernel(uint32_t* __restrict__ counts, uint32_t tiles, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t* row = counts + (uint64_t)blockIdx.x * tiles;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < tiles; c0 += 256) {
        const uint32_t i = c0 + t;
        const uint32_t v = i < tiles ? row[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += up; }
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        uint32_t before = s_carry;
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < w) before += s_wave[k];
        if (i < tiles) row[i] = before + incl - v;
        __syncthreads();
        if (t == 255) s_carry = before + incl;
        __syncthreads();
    }
    if (t == 0) totals[blockIdx.x] = s_carry;
}

template <class K, class V, int kRsItems>
__global__ void __launch_bounds__(256) rs_downsweep_kernel(const K* __restrict__ kin, K* __restrict__ kout,
                                                           const V* __restrict__ vin, V* __restrict__ vout,
                                                           uint64_t n, int shift, uint32_t digit_mask,
                                                           const uint32_t* __restrict__ tile_before /* [256][tiles] */,
                                                           const uint32_t* __restrict__ totals /* [256] */, uint32_t tiles) {
    constexpr int kRsTile = kRsThreads * kRsItems;
    __shared__ uint32_t s_cnt[kRsWaves][kRsBins];    // per wave: running digit counts, later the wave's start inside the tile
    __shared__ uint32_t s_gofs[kRsBins];             // global position of a digit's run minus its start inside the tile
    __shared__ uint32_t s_scan[kRsWaves], s_scan2[kRsWaves];
    __shared__ K s_keys[kRsTile];
    __shared__ V s_vals[kRsTile];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < kRsWaves * kRsBins; i += 256) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    const uint64_t base = (uint64_t)tile * kRsTile;
    const uint32_t valid = (uint32_t)min((uint64_t)kRsTile, n - base);
    const uint64_t wbase = base + (uint64_t)w * 64 * kRsItems;
    // what this digit needs from the tile scan, requested early.  With few tiles (kRsFewTiles or fewer: the
    // order stage sorts ~100k clusters) there is no tile-scan launch: every block adds up the raw counts itself.
    uint32_t before_tiles = 0, digit_total = 0;
    if (totals) {
        before_tiles = tile_before[(uint64_t)t * tiles + tile];
        digit_total = totals[t];
    } else {
        const uint32_t* row = tile_before + (uint64_t)t * tiles;
        for (uint32_t j = 0; j < tiles; j++) { const uint32_t c = row[j]; digit_total += c; if (j < tile) before_tiles += c; }
    }

    // keys in wave-striped rows: row i of wave w = 64 consecutive keys.  Padding keys are all ones: their
    // digit is the largest one, so they rank after every real key of the tile and are never written.
    K key[kRsItems];
    uint32_t rank[kRsItems];
#pragma unroll
    for (int i = 0; i < kRsItems; i++) {
        const uint64_t idx = wbase + (uint64_t)i * 64 + lane;
        key[i] = idx < n ? kin[idx] : ~(K)0;
    }
    const uint64_t lanes_below = (1ull << lane) - 1;
#pragma unroll
    for (int i = 0; i < kRsItems; i++) {
        const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
        const uint64_t peers = same_digit_lanes(d);
        const int leader = __builtin_ctzll(peers);
        uint32_t before = 0;
        if (lane == leader) before = atomicAdd(&s_cnt[w][d], (uint32_t)__popcll(peers));
        before = (uint32_t)__shfl((int)before, leader);
        rank[i] = before + (uint32_t)__popcll(peers & lanes_below);
    }
    __syncthreads();

    // thread t owns digit t: the waves' starts inside the digit's run, the digit's place in the tile, and the
    // digit's global start = keys with a smaller digit + keys with this digit in earlier tiles
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < kRsWaves; k++) { const uint32_t c = s_cnt[k][t]; s_cnt[k][t] = total; total += c; }
    uint32_t incl = total, gincl = digit_total;      // block-wide inclusive scans over the digits
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o), gup = (uint32_t)__shfl_up((int)gincl, o);
        if (lane >= o) { incl += up; gincl += gup; }
    }
    if (lane == 63) { s_scan[w] = incl; s_scan2[w] = gincl; }
    __syncthreads();
    uint32_t tile_start = incl - total, bin_base = gincl - digit_total;
#pragma unroll
    for (int k = 0; k < kRsWaves; k++) if (k < w) { tile_start += s_scan[k]; bin_base += s_scan2[k]; }
    s_gofs[t] = bin_base + before_tiles - tile_start;
#pragma unroll
    for (int k = 0; k < kRsWaves; k++) s_cnt[k][t] += tile_start;
    __syncthreads();

    // reorder the tile through LDS, then write every digit's run as one contiguous piece
#pragma unroll
    for (int i = 0; i < kRsItems; i++) {
        const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
        const uint32_t pos = s_cnt[w][d] + rank[i];
        const uint64_t idx = wbase + (uint64_t)i * 64 + lane;
        s_keys[pos] = key[i];
        s_vals[pos] = idx < n ? vin[idx] : V{};
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRsItems; k++) {
        const uint32_t pos = (uint32_t)k * 256 + t;
        if (pos < valid) {
            const K kk = s_keys[pos];
            const uint32_t d = (uint32_t)(kk >> shift) & digit_mask;
            const uint32_t g = s_gofs[d] + pos;
            kout[g] = kk;
            vout[g] = s_vals[pos];
        }
    }
}

template <class K, class V, int kItems>
static int radix_sort_pairs_impl(bf_ctx* ctx, const K* kin, K* kout, const V* vin, V* vout, uint64_t n,
                                 unsigned begin_bit, unsigned end_bit) {
    constexpr int kTile = kRsThreads * kItems;
    if (n == 0) return BF_OK;
    if (n > 0xfffffff0ull) return fail(ctx, BF_ERR_RANGE, "radix sort of %llu pairs (limit 2^32-16)", (unsigned long long)n);
    hipStream_t st = ctx->stream;
    if (end_bit <= begin_bit) {                      // nothing to sort by: a stable sort is the identity
        BF_HIP(ctx, hipMemcpyAsync(kout, kin, n * sizeof(K), hipMemcpyDeviceToDevice, st));
        BF_HIP(ctx, hipMemcpyAsync(vout, vin, n * sizeof(V), hipMemcpyDeviceToDevice, st));
        return BF_OK;
    }
    const int passes = (int)(end_bit - begin_bit + 7) / 8;          // <= 8
    const uint32_t tiles = (uint32_t)((n + kTile - 1) / kTile);
    // workspace: [digit totals 256 u32][tile counts 256 * tiles u32][tmp keys][tmp values]; all of it is written before it is read
    const size_t ctl_bytes = ((size_t)kRsBins * 4 + (size_t)kRsBins * tiles * 4 + 255) & ~(size_t)255;
    const size_t keys_bytes = (n * sizeof(K) + 255) & ~(size_t)255;
    BF_TRY(ensure(ctx, ctx->c_tmp, ctl_bytes + keys_bytes + n * sizeof(V) + 256, false));
    uint8_t* ws = ctx->c_tmp.as<uint8_t>();
    uint32_t* totals = (uint32_t*)ws;
    uint32_t* counts = totals + kRsBins;
    K* ktmp = (K*)(ws + ctl_bytes);
    V* vtmp = (V*)(ws + ctl_bytes + keys_bytes);
    const K* src_k = kin; const V* src_v = vin;
    for (int p = 0; p < passes; p++) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;             // the last pass lands in the caller's output
        K* dst_k = to_out ? kout : ktmp; V* dst_v = to_out ? vout : vtmp;
        const int sh = (int)begin_bit + 8 * p;
        const uint32_t mask = (1u << std::min(8, (int)end_bit - sh)) - 1;
        hipLaunchKernelGGL((rs_upsweep_kernel<K, kItems>), dim3(tiles), dim3(256), 0, st, src_k, n, sh, mask, counts, tiles);
        const bool few = tiles <= kRsFewTiles;
        if (!few) hipLaunchKernelGGL(rs_tilescan_kernel, dim3(kRsBins), dim3(256), 0, st, counts, tiles, totals);
        hipLaunchKernelGGL((rs_downsweep_kernel<K, V, kItems>), dim3(tiles), dim3(256), 0, st, src_k, dst_k, src_v, dst_v, n, sh, mask,
                           counts, few ? (const uint32_t*)nullptr : totals, tiles);
        src_k = dst_k; src_v = dst_v;
    }
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}

int radix_sort_pairs(bf_ctx* ctx, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n,
                     unsigned begin_bit, unsigned end_bit) {
    return radix_sort_pairs_impl<uint32_t, uint32_t, RS_ITEMS>(ctx, kin, kout, vin, vout, n, begin_bit, end_bit);
}
int radix_sort_pairs(bf_ctx* ctx, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n,
                     unsigned begin_bit, unsigned end_bit) {
    return radix_sort_pairs_impl<uint64_t, uint32_t, RS_ITEMS>(ctx, kin, kout, vin, vout, n, begin_bit, end_bit);
}
// 64-bit keys with three words of payload (the class partition of the cluster stage carries a record's id, right
// position and upper key bits along instead of fetching them through the id afterwards): tiles of 2048, 44 KB of LDS
int radix_sort_kv3(bf_ctx* ctx, const uint64_t* kin, uint64_t* kout, const RsV3* vin, RsV3* vout, uint64_t n,
                   unsigned begin_bit, unsigned end_bit) {
    return radix_sort_pairs_impl<uint64_t, RsV3, 8>(ctx, kin, kout, vin, vout, n, begin_bit, end_bit);
}

// ---- scans -------------------------------------------------------------------------------------------
// Reduce-then-scan without waiting between workgroups: tile totals, then the scan proper, in which every tile adds
// up the totals of the tiles before it (very long inputs: an exclusive scan of the totals by one block in between).  A tile is 256 x 16 consecutive elements;
// thread t holds elements [16t, 16t+16) of its tile.
constexpr int kScanItems = 16;
constexpr int kScanTile = 256 * kScanItems;

struct OpSumU64 { __device__ static uint64_t id() { return 0; } __device__ static uint64_t apply(uint64_t a, uint64_t b) { return a + b; } };
struct OpMaxU32 { __device__ static uint64_t id() { return 0; } __device__ static uint64_t apply(uint64_t a, uint64_t b) { return a > b ? a : b; } };

__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int o) {
    return ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(v >> 32), o) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, o);
}

template <class Op>
__device__ __forceinline__ void load_tile_items(const uint32_t* in, uint64_t first, uint64_t n, uint64_t (&v)[kScanItems]) {
    if (first + kScanItems <= n && ((uintptr_t)in & 15) == 0) {
        const uint4* p = (const uint4*)(in + first);
#pragma unroll
        for (int q = 0; q < kScanItems / 4; q++) {
            const uint4 x = p[q];
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; i++) v[i] = first + i < n ? (uint64_t)in[first + i] : Op::id();
    }
}

// inclusive scan of one value per thread over the block; s_wave: 4 words of LDS
template <class Op>
__device__ __forceinline__ uint64_t block_scan_incl(uint64_t x, uint64_t* s_wave, uint64_t& block_total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t up = shfl_up64(incl, o); if (lane >= o) incl = Op::apply(up, incl); }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    uint64_t before = Op::id();
#pragma unroll
    for (int k = 0; k < 4; k++) if (k < w) before = Op::apply(before, s_wave[k]);
    block_total = Op::apply(Op::apply(Op::apply(s_wave[0], s_wave[1]), s_wave[2]), s_wave[3]);
    return Op::apply(before, incl);
}

template <class Op>
__global__ void __launch_bounds__(256) scan_totals_kernel(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ tile_total) {
    __shared__ uint64_t s_wave[4];
    uint64_t v[kScanItems];
    load_tile_items<Op>(in, (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems, n, v);
    uint64_t x = v[0];
#pragma unroll
    for (int i = 1; i < kScanItems; i++) x = Op::apply(x, v[i]);
    uint64_t total;
    block_scan_incl<Op>(x, s_wave, total);
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}

// one block: tile_total[i] <- combination of tile_total[0..i)
template <class Op>
__global__ void __launch_bounds__(256) scan_carry_kernel(uint64_t* __restrict__ tile_total, uint64_t tiles) {
    __shared__ uint64_t s_wave[4];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = Op::id();
    __syncthreads();
    for (uint64_t c0 = 0; c0 < tiles; c0 += 256) {
        const uint64_t i = c0 + threadIdx.x;
        const uint64_t x = i < tiles ? tile_total[i] : Op::id();
        uint64_t total;
        const uint64_t incl = block_scan_incl<Op>(x, s_wave, total);
        const uint64_t carry = s_carry;
        // exclusive: everything before element i = carry + (inclusive of the previous thread)
        const uint64_t prev = shfl_up64(incl, 1);
        uint64_t before = (threadIdx.x & 63) ? prev : Op::id();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (k < (int)(threadIdx.x >> 6)) before = Op::apply(before, s_wave[k]);
        }
        if (i < tiles) tile_total[i] = Op::apply(carry, before);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = Op::apply(carry, total);
        __syncthreads();
    }
}

// Inclusive scan of in[0..n) into out[0..n); Out may be wider than the input, and out may be in.
template <class Op, class Out>
__global__ void __launch_bounds__(256) scan_apply_kernel(const uint32_t* in, Out* out, uint64_t n, const uint64_t* __restrict__ tile_before,
                                                         Out* zero_at) {
    __shared__ uint64_t s_wave[4];
    const int t = threadIdx.x;
    if (zero_at && blockIdx.x == 0 && t == 0) *zero_at = 0;      // the exclusive scan's leading zero (saves a memset launch)
    const uint64_t first = (uint64_t)blockIdx.x * kScanTile + (uint64_t)t * kScanItems;
    uint64_t v[kScanItems];
    load_tile_items<Op>(in, first, n, v);
#pragma unroll
    for (int i = 1; i < kScanItems; i++) v[i] = Op::apply(v[i - 1], v[i]);
    uint64_t total;
    const uint64_t incl = block_scan_incl<Op>(v[kScanItems - 1], s_wave, total);
    // everything before this thread: the tiles before + the threads before (inclusive of the previous thread)
    const uint64_t prev = shfl_up64(incl, 1);
    uint64_t before = tile_before[blockIdx.x];
    if (t & 63) before = Op::apply(before, prev);
    else {
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < (t >> 6)) before = Op::apply(before, s_wave[k]);
    }
#pragma unroll
    for (int i = 0; i < kScanItems; i++)
        if (first + i < n) out[first + i] = (Out)Op::apply(before, v[i]);
}

// The tile's place among the tiles, computed by the tile itself: everything before tile b = the combination of
// tile_total[0 .. b).  A few thousand totals (L2-resident), ten loads a thread: cheaper than a launch of its own
// for a one-block scan of the totals, which is what a cluster stage's sixteen scans per contig group paid for.
template <class Op>
__device__ __forceinline__ uint64_t tiles_before(const uint64_t* __restrict__ tile_total, uint32_t b, uint64_t* s_wave) {
    uint64_t x = Op::id();
    for (uint32_t i = threadIdx.x; i < b; i += 256) x = Op::apply(x, tile_total[i]);
    uint64_t total;
    block_scan_incl<Op>(x, s_wave, total);
    __syncthreads();                                                  // s_wave is reused by the caller
    return total;
}

// Inclusive scan of in[0..n) into out[0..n) from the RAW tile totals (no carry launch); otherwise scan_apply_kernel.
template <class Op, class Out>
__global__ void __launch_bounds__(256) scan_apply2_kernel(const uint32_t* in, Out* out, uint64_t n, const uint64_t* __restrict__ tile_total,
                                                          Out* zero_at) {
    __shared__ uint64_t s_wave[4];
    const int t = threadIdx.x;
    if (zero_at && blockIdx.x == 0 && t == 0) *zero_at = 0;
    const uint64_t first = (uint64_t)blockIdx.x * kScanTile + (uint64_t)t * kScanItems;
    uint64_t v[kScanItems];
    load_tile_items<Op>(in, first, n, v);                             // requested before the totals are added up
    const uint64_t tile_prefix = tiles_before<Op>(tile_total, blockIdx.x, s_wave);
#pragma unroll
    for (int i = 1; i < kScanItems; i++) v[i] = Op::apply(v[i - 1], v[i]);
    uint64_t total;
    const uint64_t incl = block_scan_incl<Op>(v[kScanItems - 1], s_wave, total);
    const uint64_t prev = shfl_up64(incl, 1);
    uint64_t before = tile_prefix;
    if (t & 63) before = Op::apply(before, prev);
    else {
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < (t >> 6)) before = Op::apply(before, s_wave[k]);
    }
#pragma unroll
    for (int i = 0; i < kScanItems; i++)
        if (first + i < n) out[first + i] = (Out)Op::apply(before, v[i]);
}

// `tot`: workspace of scan_workspace_bytes(n) bytes (one total per tile); nullptr = the ctx's own (c_tmp, grown on demand).
// Two launches: tile totals, then the scan proper, every tile adding up the totals before it by itself.  (A one-launch
// form with decoupled look-back over per-tile descriptors was measured in round 3 and dropped, like the one-sweep sort
// in round 1 and for the same reason: 1200-2400 tiles of which ~1000 start together, so a late tile adds up hundreds of
// predecessors' aggregates 64 at a time through agent-scope loads -- sweep stage 0.52 -> 0.59 ms per step, step 5.93 ->
// 6.02 ms; profiles/r03c_bench_cfg1_onepass_scan_ab.json against ..._three_launch_scan_ab.json.)
constexpr uint64_t kScanSelfPrefixTiles = 1u << 14;       // beyond: the carry launch (a tile would add up > 16k totals)
template <class Op, class Out>
static int scan_impl(bf_ctx* ctx, const uint32_t* in, Out* out, uint64_t n, uint64_t* tot = nullptr, Out* zero_at = nullptr) {
    if (n == 0) {
        if (zero_at) BF_HIP(ctx, hipMemsetAsync(zero_at, 0, sizeof(Out), ctx->stream));
        return BF_OK;
    }
    const uint64_t tiles = (n + kScanTile - 1) / kScanTile;
    if (!tot) {
        BF_TRY(ensure(ctx, ctx->c_tmp, tiles * 8 + 256, false));
        tot = ctx->c_tmp.as<uint64_t>();
    }
    hipLaunchKernelGGL((scan_totals_kernel<Op>), dim3((unsigned)tiles), dim3(256), 0, ctx->stream, in, n, tot);
    if (tiles <= kScanSelfPrefixTiles) {
        hipLaunchKernelGGL((scan_apply2_kernel<Op, Out>), dim3((unsigned)tiles), dim3(256), 0, ctx->stream, in, out, n, tot, zero_at);
    } else {
        hipLaunchKernelGGL((scan_carry_kernel<Op>), dim3(1), dim3(256), 0, ctx->stream, tot, tiles);
        hipLaunchKernelGGL((scan_apply_kernel<Op, Out>), dim3((unsigned)tiles), dim3(256), 0, ctx->stream, in, out, n, tot, zero_at);
    }
    BF_HIP(ctx, hipGetLastError());
    return BF_OK;
}

// exclusive scan with total: out[0] = 0, out[i+1] = in[0] + ... + in[i]
int scan_u32_to_u64(bf_ctx* ctx, const uint32_t* in, uint64_t* out, uint64_t n) {
    return scan_impl<OpSumU64, uint64_t>(ctx, in, out + 1, n, nullptr, out);
}

// the same with the tile totals in caller-provided device memory: allocates nothing
uint64_t scan_workspace_bytes(uint64_t n) { return ((n + kScanTile - 1) / kScanTile) * 8 + 256; }
int scan_u32_to_u64_ws(bf_ctx* ctx, const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* workspace) {
    return scan_impl<OpSumU64, uint64_t>(ctx, in, out + 1, n, workspace, out);
}

// data[i] = max(data[0..i]), in place (a tile reads its elements before it writes them; tiles are disjoint)
int scan_max_u32_inplace(bf_ctx* ctx, uint32_t* data, uint64_t n) {
    return scan_impl<OpMaxU32, uint32_t>(ctx, data, data, n);
}

}  // namespace bf

// Test hooks (tests/test_gpu_primitives.py): the primitives on caller-owned device memory.  Not part of the
// ABI in include/breakfast_hip.h.
extern "C" {
int bfx_sort_pairs_u32(bf_ctx* ctx, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n,
                       unsigned begin_bit, unsigned end_bit) {
    if (!ctx) return BF_ERR_INVALID;
    BF_HIP(ctx, hipSetDevice(ctx->device));
    BF_TRY(bf::radix_sort_pairs(ctx, kin, kout, vin, vout, n, begin_bit, end_bit));
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}
int bfx_sort_pairs_u64(bf_ctx* ctx, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n,
                       unsigned begin_bit, unsigned end_bit) {
    if (!ctx) return BF_ERR_INVALID;
    BF_HIP(ctx, hipSetDevice(ctx->device));
    BF_TRY(bf::radix_sort_pairs(ctx, kin, kout, vin, vout, n, begin_bit, end_bit));
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}
int bfx_scan_sum(bf_ctx* ctx, const uint32_t* in, uint64_t* out /* n+1 */, uint64_t n) {
    if (!ctx) return BF_ERR_INVALID;
    BF_HIP(ctx, hipSetDevice(ctx->device));
    BF_TRY(bf::scan_u32_to_u64(ctx, in, out, n));
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}
int bfx_scan_max(bf_ctx* ctx, uint32_t* data, uint64_t n) {
    if (!ctx) return BF_ERR_INVALID;
    BF_HIP(ctx, hipSetDevice(ctx->device));
    BF_TRY(bf::scan_max_u32_inplace(ctx, data, n));
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}
}
