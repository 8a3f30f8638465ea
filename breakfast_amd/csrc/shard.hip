// The sharded path inside the library (include/breakfast_hip.h, "the sharded path in one call"; SURVEY.md section
// 8e; north_star item 4): one ctx per GPU, reads sharded by contiguous input range, ONE exchange of evidence keyed
// on the breakpoint class, clustering on the owner.  The reference is a single process (detect.rs:76-241); what
// has to survive the sharding is its order: evidence is sorted stably by (chr, pos) (detect.rs:193-198), so ties
// fall in input order -- records therefore arrive ordered by (source rank, source order) and ranks hold
// ascending input ranges.
//
// Two transports move device buffers between ranks:
//   RCCL   grouped ncclSend/ncclRecv on the ctx's stream (xGMI is point to point: every pair has its own link,
//          an all-to-all uses all seven at once), ncclAllGather for the few control words.  librccl is loaded at
//          run time, so a one-GPU user of this library never needs it.
//   host   a POSIX shared-memory segment with a staging slot per rank and a sense-reversing barrier: device ->
//          slot, barrier, peers' slots -> device.  Threads or processes of one machine; ranks may share a GPU.
//          (What a one-GPU box can test; also the fallback without RCCL.)
//
// Sequences never travel.  After clustering, an owner sends each source rank the list of its surviving reads (a
// 4-byte index each); the source renders their junction records (detect.rs:155-162) and copies them to its own host
// memory; the owner's rows carry, per supporting read, (source rank, slot in that source's record list).  Slots are
// arrival order: requests grouped by owner rank, the owner's result order inside, so every rank can compute them
// from the W x W matrix of request counts, which all ranks gather.

#include "bf_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <rccl/rccl.h>     // types and prototypes only: the functions are looked up with dlsym

namespace bf {

namespace {

// ---- RCCL, loaded at run time ---------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // a copy the process already holds (PyTorch ships one) must be the one used: two RCCLs in one process
        // would each believe they own the GPU's IPC state
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if ((x.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        for (const char* n : names) { if (x.lib) break; x.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
        if (!x.lib) { x.why = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?"); return x; }
#define BF_SYM(field, name) x.field = (decltype(x.field))dlsym(x.lib, name); if (!x.field) { x.why = std::string("librccl lacks ") + name; x.lib = nullptr; return x; }
        BF_SYM(GetUniqueId, "ncclGetUniqueId") BF_SYM(CommInitRank, "ncclCommInitRank") BF_SYM(CommDestroy, "ncclCommDestroy")
        BF_SYM(AllGather, "ncclAllGather") BF_SYM(Send, "ncclSend") BF_SYM(Recv, "ncclRecv")
        BF_SYM(GroupStart, "ncclGroupStart") BF_SYM(GroupEnd, "ncclGroupEnd") BF_SYM(GetErrorString, "ncclGetErrorString")
#undef BF_SYM
        return x;
    }();
    return r;
}

#define BF_NCCL(ctx, expr)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess)                                                                              \
            return fail(ctx, BF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ---- transports --------------------------------------------------------------------------------------------
struct Transport {
    uint32_t rank = 0, n = 1;
    virtual ~Transport() {}
    // out[r * k + j] = rank r's in[j]; host values, collective
    virtual int all_gather(bf_ctx* ctx, const uint64_t* in, uint32_t k, uint64_t* out) = 0;
    // bytes[s * n + d] = what rank s sends to rank d (every rank passes the same matrix); the send buffer holds the
    // pieces for ranks 0..n-1 back to back, the receive buffer gets the pieces from ranks 0..n-1 back to back.
    // Device pointers; queued on (or completed before returning to) the ctx's stream.  skip_self: the rank's own piece is
    // not copied (its place in the receive buffer stays as it was): the caller reads it from the send buffer.
    virtual int all_to_all(bf_ctx* ctx, const uint8_t* d_send, uint8_t* d_recv, const uint64_t* bytes, bool skip_self = false) = 0;
};

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    DevBuf d_ctl;
    HostBuf h_ctl;                        // pinned: the control words go up and come back without a staging copy
    ~RcclTransport() override {
        if (comm) (void)rccl().CommDestroy(comm);
        if (d_ctl.p) (void)hipFree(d_ctl.p);
        if (h_ctl.p) (void)hipHostFree(h_ctl.p);
    }
    int all_gather(bf_ctx* ctx, const uint64_t* in, uint32_t k, uint64_t* out) override {
        BF_TRY(ensure(ctx, d_ctl, (size_t)(n + 1) * k * 8));
        BF_TRY(ensure_host(ctx, h_ctl, (size_t)(n + 1) * k * 8));
        uint64_t* d_in = d_ctl.as<uint64_t>();
        uint64_t* d_out = d_in + k;
        uint64_t* h_in = h_ctl.as<uint64_t>();
        uint64_t* h_out = h_in + k;
        memcpy(h_in, in, (size_t)k * 8);
        BF_HIP(ctx, hipMemcpyAsync(d_in, h_in, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
        BF_NCCL(ctx, rccl().AllGather(d_in, d_out, k, ncclUint64, comm, ctx->stream));
        BF_HIP(ctx, hipMemcpyAsync(h_out, d_out, (size_t)n * k * 8, hipMemcpyDeviceToHost, ctx->stream));
        BF_SYNC(ctx, ctx->stream);
        memcpy(out, h_out, (size_t)n * k * 8);
        return BF_OK;
    }
    int all_to_all(bf_ctx* ctx, const uint8_t* d_send, uint8_t* d_recv, const uint64_t* bytes, bool skip_self) override {
        uint64_t so = 0, ro = 0;
        std::vector<uint64_t> soff(n), roff(n);
        for (uint32_t r = 0; r < n; r++) { soff[r] = so; roff[r] = ro; so += bytes[(size_t)rank * n + r]; ro += bytes[(size_t)r * n + rank]; }
        const uint64_t self = bytes[(size_t)rank * n + rank];
        if (self && !skip_self) BF_HIP(ctx, hipMemcpyAsync(d_recv + roff[rank], d_send + soff[rank], self, hipMemcpyDeviceToDevice, ctx->stream));
        if (n > 1) {
            BF_NCCL(ctx, rccl().GroupStart());
            // an error inside the group is remembered, the group is closed whatever happened (an open group would swallow
            // every later RCCL call of this thread), and only then reported
            ncclResult_t bad = ncclSuccess;
            const char* what = "";
            for (uint32_t r = 0; r < n && bad == ncclSuccess; r++) {
                if (r == rank) continue;
                const uint64_t sb = bytes[(size_t)rank * n + r], rb = bytes[(size_t)r * n + rank];
                if (sb && (bad = rccl().Send(d_send + soff[r], sb, ncclUint8, (int)r, comm, ctx->stream)) != ncclSuccess) { what = "ncclSend"; break; }
                if (rb && (bad = rccl().Recv(d_recv + roff[r], rb, ncclUint8, (int)r, comm, ctx->stream)) != ncclSuccess) { what = "ncclRecv"; break; }
            }
            const ncclResult_t end = rccl().GroupEnd();
            if (bad != ncclSuccess) return fail(ctx, BF_ERR_HIP, "%s failed inside the all-to-all group: %s", what, rccl().GetErrorString(bad));
            if (end != ncclSuccess) return fail(ctx, BF_ERR_HIP, "ncclGroupEnd failed: %s", rccl().GetErrorString(end));
        }
        return BF_OK;
    }
};

// Shared segment: header, one control row per rank, one staging slot per rank.
struct ShmHeader {
    std::atomic<uint32_t> ready;          // kShmMagic once rank 0 has initialised the header
    std::atomic<uint32_t> arrived, sense; // sense-reversing barrier
    uint32_t n_ranks; uint64_t slot_bytes, ctl_words;
    // A rank that opens the name may find the segment a crashed run left behind (same name, `ready` set) before rank 0
    // of THIS run has replaced it.  So it says hello with a token of its own and waits for a LIVE rank 0 to repeat the
    // token; no answer within kShmHelloMs means nobody is home: it lets go and opens the name again.
    std::atomic<uint64_t> hello[64], ack[64];
};
static_assert(sizeof(ShmHeader) <= 4096, "the header has one page");
constexpr uint32_t kShmMagic = 0xbf5a4d31u;
constexpr uint64_t kShmCtlWords = 4096;   // u64 per rank for all_gather (n_ranks * k must fit: 64 ranks x 64 words)
constexpr int kShmTimeoutSec = 180;
constexpr int kShmHelloMs = 250;
static uint64_t shm_token(uint32_t rank) {
    timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    uint64_t x = ((uint64_t)ts.tv_sec << 32) ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)getpid() << 20) ^ ((uint64_t)rank << 56) ^ 0x9e3779b97f4a7c15ull;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x | 1;                          // never 0 (a fresh segment's words)
}

struct HostTransport : Transport {
    std::string name;
    uint8_t* base = nullptr; size_t size = 0;
    ShmHeader* hdr = nullptr;
    uint64_t slot_bytes = 0;
    uint32_t local_sense = 0;
    bool owner = false, registered = false;
    uint64_t* ctl(uint32_t r) const { return (uint64_t*)(base + 4096) + (size_t)r * kShmCtlWords; }
    uint8_t* slot(uint32_t r) const { return base + 4096 + (size_t)n * kShmCtlWords * 8 + (size_t)r * slot_bytes; }
    ~HostTransport() override {
        if (base && registered) (void)hipHostUnregister(base);
        if (base) munmap(base, size);
        if (owner) shm_unlink(name.c_str());
    }
    int barrier(bf_ctx* ctx, const char* where) {
        local_sense ^= 1u;
        if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            hdr->arrived.store(0, std::memory_order_relaxed);
            hdr->sense.store(local_sense, std::memory_order_release);
            return BF_OK;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0; hdr->sense.load(std::memory_order_acquire) != local_sense; spins++) {
            if (spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(50)); else std::this_thread::yield();
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kShmTimeoutSec))
                return fail(ctx, BF_ERR_STATE, "sharded path: rank %u waited %d s for the other ranks at %s (a rank failed or never called)",
                            rank, kShmTimeoutSec, where);
        }
        return BF_OK;
    }
    int all_gather(bf_ctx* ctx, const uint64_t* in, uint32_t k, uint64_t* out) override {
        if ((uint64_t)k > kShmCtlWords) return fail(ctx, BF_ERR_RANGE, "sharded path: control message of %u words", k);
        if (trace_on()) trace_line(ctx, "enter", "host all_gather", "", 0);
        memcpy(ctl(rank), in, (size_t)k * 8);
        BF_TRY(barrier(ctx, "all_gather (written)"));
        for (uint32_t r = 0; r < n; r++) memcpy(out + (size_t)r * k, ctl(r), (size_t)k * 8);
        BF_TRY(barrier(ctx, "all_gather (read)"));
        if (trace_on()) trace_line(ctx, "leave", "host all_gather", "", 0);
        return BF_OK;
    }
    int all_to_all(bf_ctx* ctx, const uint8_t* d_send, uint8_t* d_recv, const uint64_t* bytes, bool skip_self) override {
        if (trace_on()) trace_line(ctx, "enter", "host all_to_all", "", 0);
        // rounds: every rank stages at most slot_bytes of its send buffer per round; all ranks run the same number
        std::vector<uint64_t> total(n, 0), recv_off(n, 0);
        uint64_t longest = 0, ro = 0;
        for (uint32_t s = 0; s < n; s++) {
            for (uint32_t d = 0; d < n; d++) total[s] += bytes[(size_t)s * n + d];
            longest = std::max(longest, total[s]);
            recv_off[s] = ro; ro += bytes[(size_t)s * n + rank];
        }
        for (uint64_t w0 = 0; w0 < longest; w0 += slot_bytes) {           // window [w0, w0 + slot_bytes) of every send buffer
            const uint64_t mine = total[rank] > w0 ? std::min(slot_bytes, total[rank] - w0) : 0;
            if (mine) BF_HIP(ctx, hipMemcpyAsync(slot(rank), d_send + w0, mine, hipMemcpyDeviceToHost, ctx->stream));
            BF_SYNC(ctx, ctx->stream);
            BF_TRY(barrier(ctx, "all_to_all (staged)"));
            for (uint32_t s = 0; s < n; s++) {
                // rank s's piece for me is [a, b) of its send buffer; the part inside this window
                uint64_t a = 0;
                for (uint32_t d = 0; d < rank; d++) a += bytes[(size_t)s * n + d];
                const uint64_t b = a + bytes[(size_t)s * n + rank];
                const uint64_t lo = std::max(a, w0), hi = std::min(b, w0 + slot_bytes);
                if (lo < hi && !(skip_self && s == rank))
                    BF_HIP(ctx, hipMemcpyAsync(d_recv + recv_off[s] + (lo - a), slot(s) + (lo - w0), hi - lo, hipMemcpyHostToDevice, ctx->stream));
            }
            BF_SYNC(ctx, ctx->stream);
            BF_TRY(barrier(ctx, "all_to_all (taken)"));
        }
        if (trace_on()) trace_line(ctx, "leave", "host all_to_all", "", 0);
        return BF_OK;
    }
};

struct Shard {
    Transport* tr = nullptr;
    std::vector<uint64_t> bases;                // index_base of every rank
    DevBuf d_send, d_recv, d_req_out, d_req_in, d_place, d_src, d_pick, d_roff, d_roff_g, d_rbytes;
    HostBuf h_places, h_src, h_roff, h_rbytes;
    std::vector<uint64_t> block_off, group_read_begin;      // what the last bf_shard_finish's result points at
    uint64_t last_nr = 0, last_n_in = 0;                    // sizes of the tables left on the device
    ~Shard() { delete tr; }
};

Shard* shard_of(bf_ctx* ctx) { return (Shard*)ctx->shard; }


// a group's record offsets (n + 1 of them, from 0) appended to the rank's list, shifted by the bytes before them
__global__ void offsets_append_kernel(uint64_t* __restrict__ all, const uint64_t* __restrict__ part, uint64_t byte_base, uint64_t n) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) all[k] = part[k] + byte_base;
}
// pinned host buffer that keeps its first `used` bytes when it has to grow (copies into it may be in flight)
static int grow_host_keep(bf_ctx* ctx, HostBuf& b, size_t used, size_t need) {
    if (need <= b.cap && b.p) return BF_OK;
    BF_SYNC(ctx, ctx->stream);
    BF_SYNC(ctx, ctx->copy_stream);
    void* p = nullptr;
    const size_t cap = std::max<size_t>(need + need / 2, 4096);
    const hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, BF_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
    if (b.p) { if (used) memcpy(p, b.p, std::min(used, b.cap)); (void)hipHostFree(b.p); }
    b.p = p; b.cap = cap;
    return BF_OK;
}

// place[i] = (source rank, slot in that source's record list) of supporting read i: the reads sorted by source rank
// are position j = order^-1(i); the slot counts from the first request of this owner to that source
// src[i] = the source rank alone: all the host needs to walk the records in the order they were asked for
// (slot_base comes by value, a word per rank: a table the host has just computed needs no copy and no wait this way)
constexpr uint32_t kMaxShardRanks = 256;
struct RankWords { uint64_t w[kMaxShardRanks + 1]; };
__global__ void place_kernel(bf_place* __restrict__ place, uint8_t* __restrict__ src, const uint32_t* __restrict__ order,
                             const uint32_t* __restrict__ skey, const uint64_t* __restrict__ start, const RankWords slot_base, uint64_t nr) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nr) return;
    const uint32_t s = skey[j];
    place[order[j]] = bf_place{s, (uint32_t)(slot_base.w[s] + (j - start[s]))};
    src[order[j]] = (uint8_t)s;
}

// out[k] = off[pos[k]]: the record offsets at the boundaries between the owners' request lists (k <= n_ranks), then at the
// chunk boundaries the render and its copies go by (out[n_ranks + 1 + c] = off[n * c / n_chunks]): one launch, one copy
__global__ void pick_block_offsets_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ off, const RankWords pos, uint32_t n_ranks,
                                          uint64_t n, int n_chunks) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= n_ranks) out[k] = off[pos.w[k]];
    else if (k <= n_ranks + 1 + (uint32_t)n_chunks) out[k] = off[n * (uint64_t)(k - n_ranks - 1) / n_chunks];
}

inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

int shard_attach(bf_ctx* ctx, Transport* tr, uint32_t rank, uint32_t n_ranks, uint64_t index_base) {
    if (ctx->cfg.flags & BF_FLAG_READS_ON_HOST) {
        delete tr;
        return fail(ctx, BF_ERR_STATE, "BF_FLAG_READS_ON_HOST is for one ctx in one process: records that name a read cannot be expanded on another rank");
    }
    tr->rank = rank; tr->n = n_ranks;
    Shard* sh = new Shard();
    sh->tr = tr;
    shard_destroy(ctx);
    ctx->shard = sh;
    ctx->rank = rank; ctx->n_ranks = n_ranks; ctx->index_base = index_base;
    ctx->src_index_base.clear(); ctx->imported = false;
    // every rank learns every rank's first global read index (and that all of them are here)
    uint64_t mine[2] = {index_base, n_ranks};
    std::vector<uint64_t> all((size_t)n_ranks * 2);
    BF_TRY(tr->all_gather(ctx, mine, 2, all.data()));
    sh->bases.resize(n_ranks);
    for (uint32_t r = 0; r < n_ranks; r++) {
        if (all[2 * r + 1] != n_ranks) return fail(ctx, BF_ERR_INVALID, "sharded path: rank %u joined with %llu ranks, this rank with %u", r, (unsigned long long)all[2 * r + 1], n_ranks);
        sh->bases[r] = all[2 * r];
        if (r && sh->bases[r] < sh->bases[r - 1])
            return fail(ctx, BF_ERR_INVALID, "sharded path: ranks must hold ascending input ranges (rank %u starts at %llu, rank %u at %llu): "
                        "the stable sort's tie-break is the input order, detect.rs:193-198", r - 1, (unsigned long long)sh->bases[r - 1], r, (unsigned long long)sh->bases[r]);
    }
    return BF_OK;
}

}  // namespace

void shard_destroy(bf_ctx* ctx) {
    Shard* sh = shard_of(ctx);
    if (!sh) return;
    DevBuf* bufs[] = {&sh->d_send, &sh->d_recv, &sh->d_req_out, &sh->d_req_in, &sh->d_place, &sh->d_src, &sh->d_pick, &sh->d_roff, &sh->d_roff_g, &sh->d_rbytes};
    for (DevBuf* b : bufs) release(ctx, *b);
    HostBuf* hosts[] = {&sh->h_places, &sh->h_src, &sh->h_roff, &sh->h_rbytes};
    for (HostBuf* h : hosts) { if (h->p) (void)hipHostFree(h->p); h->p = nullptr; h->cap = 0; }
    delete sh;
    ctx->shard = nullptr;
}

}  // namespace bf

using namespace bf;

extern "C" {

int bf_shard_unique_id(uint8_t id[128]) {
    if (!id) return BF_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!rccl().lib) return fail(nullptr, BF_ERR_STATE, "%s", rccl().why.c_str());
    ncclUniqueId u;
    const ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, BF_ERR_HIP, "ncclGetUniqueId failed: %s", rccl().GetErrorString(r));
    memcpy(id, &u, 128);
    return BF_OK;
}

int bf_shard_init_rccl(bf_ctx* ctx, uint32_t rank, uint32_t n_ranks, uint64_t index_base, const uint8_t id[128]) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!id || n_ranks == 0 || n_ranks > 256 || rank >= n_ranks) return fail(ctx, BF_ERR_INVALID, "bf_shard_init_rccl: rank %u of %u", rank, n_ranks);
    if (!rccl().lib) return fail(ctx, BF_ERR_STATE, "%s", rccl().why.c_str());
    BF_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    RcclTransport* tr = new RcclTransport();
    const ncclResult_t r = rccl().CommInitRank(&tr->comm, (int)n_ranks, u, (int)rank);
    if (r != ncclSuccess) { tr->comm = nullptr; delete tr; return fail(ctx, BF_ERR_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r)); }
    return shard_attach(ctx, tr, rank, n_ranks, index_base);
}

int bf_shard_init_host(bf_ctx* ctx, uint32_t rank, uint32_t n_ranks, uint64_t index_base, const char* name, uint64_t slot_bytes) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!name || name[0] != '/' || n_ranks == 0 || n_ranks > 64 || rank >= n_ranks)
        return fail(ctx, BF_ERR_INVALID, "bf_shard_init_host: rank %u of %u (at most 64), name must start with '/'", rank, n_ranks);
    slot_bytes = std::max<uint64_t>((slot_bytes + 4095) & ~4095ull, 1u << 16);
    BF_HIP(ctx, hipSetDevice(ctx->device));
    HostTransport* tr = new HostTransport();
    tr->name = name; tr->slot_bytes = slot_bytes; tr->n = n_ranks; tr->rank = rank;
    tr->size = 4096 + (size_t)n_ranks * kShmCtlWords * 8 + (size_t)n_ranks * slot_bytes;
    const auto t_begin = std::chrono::steady_clock::now();
    auto timed_out = [&] { return std::chrono::steady_clock::now() - t_begin > std::chrono::seconds(kShmTimeoutSec); };
    if (rank == 0) {
        shm_unlink(name);                                        // a stale segment of a crashed run
        int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)tr->size) != 0) { close(fd); fd = -1; shm_unlink(name); }
        tr->owner = fd >= 0;
        if (fd < 0) { delete tr; return fail(ctx, BF_ERR_STATE, "bf_shard_init_host: shared memory segment %s is not available", name); }
        void* m = mmap(nullptr, tr->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { delete tr; return fail(ctx, BF_ERR_NOMEM, "bf_shard_init_host: mmap of %zu bytes failed", tr->size); }
        tr->base = (uint8_t*)m; tr->hdr = (ShmHeader*)m;
        tr->hdr->arrived.store(0); tr->hdr->sense.store(0);
        for (auto& w : tr->hdr->hello) w.store(0);
        for (auto& w : tr->hdr->ack) w.store(0);
        tr->hdr->n_ranks = n_ranks; tr->hdr->slot_bytes = slot_bytes; tr->hdr->ctl_words = kShmCtlWords;
        tr->hdr->ready.store(kShmMagic, std::memory_order_release);
        // answer every rank's hello (a rank may say hello more than once: it re-opens the name until somebody answers)
        for (uint32_t seen = 1; seen < n_ranks;) {
            seen = 1;
            for (uint32_t r = 1; r < n_ranks; r++) {
                const uint64_t h = tr->hdr->hello[r].load(std::memory_order_acquire);
                if (h) { tr->hdr->ack[r].store(h, std::memory_order_release); seen++; }
            }
            if (seen < n_ranks) {
                if (timed_out()) { delete tr; return fail(ctx, BF_ERR_STATE, "bf_shard_init_host: only %u of %u ranks joined %s within %d s", seen, n_ranks, name, kShmTimeoutSec); }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
    } else {
        const uint64_t token = shm_token(rank);                  // one token per call: an answer stays valid when the name is opened again
        for (;;) {                                               // until a live rank 0 answers in the segment the name leads to
            if (timed_out()) { delete tr; return fail(ctx, BF_ERR_STATE, "bf_shard_init_host: rank 0 never answered in %s (%d s)", name, kShmTimeoutSec); }
            int fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < tr->size) {
                if (fd >= 0) close(fd);
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                continue;
            }
            void* m = mmap(nullptr, tr->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) { delete tr; return fail(ctx, BF_ERR_NOMEM, "bf_shard_init_host: mmap of %zu bytes failed", tr->size); }
            ShmHeader* h = (ShmHeader*)m;
            bool home = false;
            const auto t0 = std::chrono::steady_clock::now();
            bool said = false;
            while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(kShmHelloMs)) {
                if (h->ready.load(std::memory_order_acquire) == kShmMagic) {
                    if (!said) { h->hello[rank].store(token, std::memory_order_release); said = true; }
                    if (h->ack[rank].load(std::memory_order_acquire) == token) { home = true; break; }
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            if (!home) { munmap(m, tr->size); continue; }        // a segment nobody answers in (stale, or rank 0 is still on its way)
            tr->base = (uint8_t*)m; tr->hdr = h;
            if (h->n_ranks != n_ranks || h->slot_bytes != slot_bytes) { delete tr; return fail(ctx, BF_ERR_INVALID, "bf_shard_init_host: ranks disagree about the segment's shape"); }
            break;
        }
    }
    // pinning the staging slots lets the device copies run at PCIe rate (best effort)
    tr->registered = hipHostRegister(tr->base, tr->size, hipHostRegisterDefault) == hipSuccess;
    if (!tr->registered) (void)hipGetLastError();
    return shard_attach(ctx, tr, rank, n_ranks, index_base);
}

int bf_shard_finish(bf_ctx* ctx, bf_shard_result* out) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!out) return fail(ctx, BF_ERR_INVALID, "bf_shard_finish: null result");
    memset(out, 0, sizeof *out);
    Shard* sh = shard_of(ctx);
    if (!sh) return fail(ctx, BF_ERR_STATE, "bf_shard_finish before bf_shard_init_rccl / bf_shard_init_host");
    if (ctx->n_contigs == 0 || !ctx->genome_ready) return fail(ctx, BF_ERR_STATE, "bf_shard_finish before bf_genome_set / bf_genome_end");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    Transport* tr = sh->tr;
    const uint32_t W = tr->n, me = tr->rank;
    const hipStream_t st = ctx->stream;

    // 1. per-read work on the local shard (detect.rs:105-187); no communication
    ctx->imported = false;
    BF_TRY(run_localise(ctx, false, ctx->consumed));
    const uint64_t n_local = ctx->n_reads;

    // 2. wire parameters every rank must agree on
    uint64_t mine[2] = {ctx->max_frag, ctx->n_sigx ? 1u : 0u};
    std::vector<uint64_t> all((size_t)W * std::max<uint32_t>(W, 2));
    BF_TRY(tr->all_gather(ctx, mine, 2, all.data()));
    bf_wire_local agreed{0, 0};
    for (uint32_t r = 0; r < W; r++) {
        agreed.max_frag_len_bytes = std::max<uint32_t>(agreed.max_frag_len_bytes, (uint32_t)all[2 * r]);
        agreed.has_exact_sig |= (uint32_t)all[2 * r + 1];
    }

    // 3. pack by owner = mix64(class key) % W, then the one exchange of evidence
    const uint32_t rec = (uint32_t)sizeof(WireHeader) + ((agreed.max_frag_len_bytes + 7u) & ~7u) + (agreed.has_exact_sig ? 16u : 0u);
    BF_TRY(ensure(ctx, sh->d_send, std::max<uint64_t>(n_local, 1) * rec));
    std::vector<uint64_t> counts(W, 0);
    void* d_packed = nullptr; uint32_t rec2 = 0;
    BF_TRY(run_partition(ctx, agreed, sh->d_send.p, sh->d_send.cap, counts.data(), &d_packed, &rec2));
    if (rec2 != rec) return fail(ctx, BF_ERR_HIP, "sharded path: record size %u, expected %u", rec2, rec);
    std::vector<uint64_t> cm((size_t)W * W);                        // cm[s * W + d] = records rank s sends to rank d
    BF_TRY(tr->all_gather(ctx, counts.data(), W, cm.data()));
    std::vector<uint64_t> bytes((size_t)W * W), src_counts(W);
    uint64_t n_owned = 0;
    for (uint32_t s = 0; s < W; s++) {
        for (uint32_t d = 0; d < W; d++) bytes[(size_t)s * W + d] = cm[(size_t)s * W + d] * rec;
        src_counts[s] = cm[(size_t)s * W + me];
        n_owned += src_counts[s];
    }
    BF_TRY(ensure(ctx, sh->d_recv, std::max<uint64_t>(n_owned, 1) * rec));
    {
        Scope t(ctx, "exchange_alltoall");
        // (the rank's own records are not copied: the import reads them where the pack left them)
        BF_TRY(tr->all_to_all(ctx, (const uint8_t*)d_packed, sh->d_recv.as<uint8_t>(), bytes.data(), true));
    }
    uint64_t self_off = 0;
    for (uint32_t d = 0; d < me; d++) self_off += bytes[(size_t)me * W + d];
    ctx->src_index_base = sh->bases;
    uint64_t max_owned = 0;
    for (uint32_t d = 0; d < W; d++) {
        uint64_t owned = 0;
        for (uint32_t s2 = 0; s2 < W; s2++) owned += cm[(size_t)s2 * W + d];
        max_owned = std::max(max_owned, owned);
    }
    uint32_t G = 1;                                                  // the same on every rank: flags and config must agree
    if (ctx->cfg.flags & BF_FLAG_ONE_GROUP) G = 1;
    else if (ctx->cfg.reserved[0]) G = std::min<uint32_t>(ctx->cfg.reserved[0], 64u);
    else if (ctx->cfg.flags & BF_FLAG_GROUPS_ALWAYS) G = 2;
    // Two groups once an owner holds 4M records and more: the first group's records cross PCIe while the second is
    // clustered.  Measured on one MI355X through RCCL, configs[1]: 7.57 ms per step against 7.93 in one go, three groups
    // 7.97 -- a group costs some sixty launches, two collectives and five host round trips.  (While a place and an offset
    // per read still travelled the copy was the longer part either way and groups bought nothing: 9.19 against 9.24.)
    // (Round 3, with a third of a group's launches and half of its waits gone: 7.03 ms in one go, two groups 6.50, three
    // 6.37, four 6.70 -- one group per 3.3M records an owner holds, as bf_finish cuts them on one GPU.)
    else if (max_owned >= (1ull << 22)) G = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(max_owned / 3300000, 2), 8);
    std::vector<uint64_t> hist;                                      // imported records per left contig rank: counted behind the import
    BF_TRY(run_import(ctx, sh->d_recv.p, src_counts.data(), G > 1 ? &hist : nullptr, (const uint8_t*)d_packed + self_off, me));

    // 4.-7. the owner's cluster stage (detect.rs:190-231, 247-287), a few contig ranks at a time like bf_finish on one
    //       GPU: when a group is clustered its supporting reads are requested from their source ranks, and the sources
    //       render and copy those records to their host memory WHILE the owners cluster the next group.  Every rank
    //       runs the same number of groups (agreed from the exchange counts every rank knows; a rank whose records
    //       cut into fewer groups takes part in the remaining rounds with nothing to ask for).
    OwnerView v;
    v.n = ctx->n_imported;
    v.key = ctx->i_key.as<uint64_t>(); v.lpos = ctx->i_lpos.as<int32_t>(); v.rpos = ctx->i_rpos.as<int32_t>();
    v.bp = ctx->i_bp.as<uint16_t>(); v.len = ctx->i_len.as<uint16_t>();
    v.flags = ctx->i_flags.as<uint16_t>();
    v.frag = ctx->i_frag.as<uint8_t>(); v.frag_off = ctx->i_frag_off.as<uint64_t>(); v.frag_len = ctx->i_frag_len.as<uint16_t>();
    v.sigx = ctx->i_sigx.as<uint8_t>(); v.sigx_slot = ctx->i_sigx_slot.as<uint32_t>();
    v.src = ctx->i_src.as<uint64_t>();
    v.rw = ctx->i_jrec.as<RecWord>();
    ctx->clustered = false; ctx->reads_on_device = false;
    ctx->out.n_clusters = ctx->out.n_reads = 0; ctx->n_evidence = v.n;

    std::vector<uint64_t> gb{0, v.n};
    if (G > 1 && v.n) BF_TRY(cluster_groups(ctx, v, G, gb, false, &hist));
    const size_t my_groups = gb.size() - 1;

    std::vector<uint64_t> req_counts(W), rm((size_t)W * W), recv_before(W, 0);
    RankWords ctl{}, pick{};
    constexpr int kMaxChunks = 8;
    uint64_t* h_words = nullptr;                                     // pinned read-backs: W + 1 starts; W + 1 block offsets + chunk offsets
    BF_TRY(wide_words(ctx, 2 * (size_t)W + 4 + kMaxChunks, &h_words));
    uint64_t cl_base = 0, rd_base = 0, n_in = 0, rec_total = 0;
    sh->block_off.assign((size_t)G * W + 1, 0);
    sh->group_read_begin.assign((size_t)G + 1, 0);
    sh->last_nr = sh->last_n_in = 0;
    for (uint32_t g = 0; g < G; g++) {
        sh->group_read_begin[g] = rd_base;
        for (uint32_t o = 0; o < W; o++) sh->block_off[(size_t)g * W + o] = rec_total;        // (a round nobody asks this rank in)
        // the owner's side: cluster the group
        uint64_t nr_g = 0, ns_g = 0;
        if (g < my_groups && gb[g + 1] > gb[g]) {
            if (G == 1 || my_groups == 1) {               // (an owner whose records cut into one group: cluster_groups made no id list)
                BF_TRY(run_cluster(ctx, v));
                nr_g = ctx->out.n_reads; ns_g = ctx->out.n_clusters;
            } else {
                ClusterGroup grp{ctx->c_gidx.as<uint32_t>() + gb[g], gb[g + 1] - gb[g], cl_base, rd_base, v.n};
                grp.g = (int)g;
                BF_TRY(run_cluster(ctx, v, &grp));
                nr_g = grp.nr; ns_g = grp.ns;
            }
        }
        // which source rank holds each of its supporting reads: requests ordered by source rank (result order inside)
        std::fill(req_counts.begin(), req_counts.end(), 0);
        uint64_t* d_start = nullptr;
        BF_TRY(ensure(ctx, sh->d_req_out, std::max<uint64_t>(nr_g, 1) * 4));
        if (nr_g) {
            BF_TRY(order_requests_range(ctx, rd_base, nr_g, sh->d_req_out.as<uint32_t>(), &d_start));
            BF_HIP(ctx, hipMemcpyAsync(h_words, d_start, (size_t)(W + 1) * 8, hipMemcpyDeviceToHost, st));
            BF_SYNC(ctx, st);
            for (uint32_t s2 = 0; s2 < W; s2++) req_counts[s2] = h_words[s2 + 1] - h_words[s2];
        }
        BF_TRY(tr->all_gather(ctx, req_counts.data(), W, rm.data()));   // rm[o * W + s] = requests of owner o to source s
        // slots: a source lists what it was asked in earlier groups, then this group's requests of owner 0, owner 1, ...
        uint64_t n_in_g = 0;
        for (uint32_t s2 = 0; s2 < W; s2++) {
            uint64_t before = recv_before[s2];
            for (uint32_t o = 0; o < me; o++) before += rm[(size_t)o * W + s2];
            ctl.w[s2] = before;
        }
        for (uint32_t o = 0; o < W; o++) n_in_g += rm[(size_t)o * W + me];
        if (n_in + n_in_g > 0xfffffff0ull) return fail(ctx, BF_ERR_RANGE, "sharded path: %llu junction requests for one rank", (unsigned long long)(n_in + n_in_g));
        if (nr_g) {
            BF_TRY(ensure(ctx, sh->d_place, (rd_base + nr_g) * sizeof(bf_place), true));
            BF_TRY(ensure(ctx, sh->d_src, rd_base + nr_g, true));
            hipLaunchKernelGGL(place_kernel, dim3(grid_for(nr_g, 256)), dim3(256), 0, st, sh->d_place.as<bf_place>() + rd_base,
                               sh->d_src.as<uint8_t>() + rd_base, ctx->c_d32.as<uint32_t>(), ctx->c_b32.as<uint32_t>(), d_start, ctl, nr_g);
            BF_HIP(ctx, hipGetLastError());
        }
        // the request lists go to the sources (4 bytes a read), appended to what came in earlier groups
        for (uint32_t o = 0; o < W; o++) for (uint32_t s2 = 0; s2 < W; s2++) bytes[(size_t)o * W + s2] = rm[(size_t)o * W + s2] * 4;
        BF_TRY(ensure(ctx, sh->d_req_in, std::max<uint64_t>(n_in + n_in_g, 1) * 4, true));
        {
            Scope t(ctx, "request_alltoall");
            BF_TRY(tr->all_to_all(ctx, sh->d_req_out.as<uint8_t>(), sh->d_req_in.as<uint8_t>() + n_in * 4, bytes.data()));
        }
        // the source's side: render the requested reads' junction records (detect.rs:155-162) and copy them to THIS rank's
        // host memory, chunk by chunk behind the next chunk's render (and behind the owners' next group)
        if (n_in_g) {
            const int nc = n_in_g >= (1u << 16) ? (G > 1 ? 2 : kMaxChunks) : 1;
            uint64_t off[kMaxChunks + 1];
            const uint32_t* d_idx = sh->d_req_in.as<uint32_t>() + n_in;
            BF_TRY(ensure(ctx, ctx->r_idx, n_in_g * 4 + 16));
            BF_TRY(ensure(ctx, sh->d_roff_g, (n_in_g + 1) * 8));
            BF_TRY(ensure(ctx, sh->d_pick, ((size_t)W + 4 + kMaxChunks) * 8));
            uint32_t* d_len = ctx->r_idx.as<uint32_t>();
            uint32_t* d_bad = d_len + n_in_g;
            BF_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, st));
            BF_TRY(render_sizes_checked(ctx, n_in_g, d_idx, d_len, d_bad));
            BF_TRY(scan_u32_to_u64(ctx, d_len, sh->d_roff_g.as<uint64_t>(), n_in_g));
            // where owner o's requests of this round begin among the records, and where the render's chunks do: one launch,
            // one copy, one wait (the positions go by value)
            pick.w[0] = 0;
            for (uint32_t o = 0; o < W; o++) pick.w[o + 1] = pick.w[o] + rm[(size_t)o * W + me];
            hipLaunchKernelGGL(pick_block_offsets_kernel, dim3(grid_for(W + 2 + nc, 256)), dim3(256), 0, st, sh->d_pick.as<uint64_t>(),
                               sh->d_roff_g.as<uint64_t>(), pick, W, n_in_g, nc);
            BF_HIP(ctx, hipGetLastError());
            uint64_t* h_pick = h_words + W + 1;
            BF_HIP(ctx, hipMemcpyAsync(h_pick, sh->d_pick.as<uint64_t>(), ((size_t)W + 2 + nc) * 8, hipMemcpyDeviceToHost, st));
            BF_HIP(ctx, hipMemcpyAsync(h_pick + W + 2 + nc, d_bad, 4, hipMemcpyDeviceToHost, st));
            BF_SYNC(ctx, st);
            const uint32_t bad = (uint32_t)h_pick[W + 2 + nc];
            if (bad) return fail(ctx, BF_ERR_HIP, "sharded path: %u requested reads are not resident on rank %u", bad, me);
            for (int c = 0; c <= nc; c++) off[c] = h_pick[W + 1 + c];
            const uint64_t total_g = off[nc];
            if (h_pick[0] != 0 || h_pick[W] != total_g || off[0] != 0)
                return fail(ctx, BF_ERR_HIP, "sharded path: the record offsets of round %u do not span its %llu bytes", g, (unsigned long long)total_g);
            for (uint32_t o = 0; o < W; o++) sh->block_off[(size_t)g * W + o] = rec_total + h_pick[o];
            BF_TRY(ensure(ctx, sh->d_rbytes, std::max<uint64_t>(rec_total + total_g, 1), true));
            BF_TRY(grow_host_keep(ctx, sh->h_rbytes, rec_total, rec_total + total_g));
            BF_TRY(ensure(ctx, sh->d_roff, (n_in + n_in_g + 1) * 8, true));
            hipLaunchKernelGGL(offsets_append_kernel, dim3(grid_for(n_in_g + 1, 256)), dim3(256), 0, st, sh->d_roff.as<uint64_t>() + n_in,
                               sh->d_roff_g.as<uint64_t>(), rec_total, n_in_g + 1);
            for (int c = 0; c < nc; c++) {
                BF_TRY(render_range_at(ctx, n_in_g * (uint64_t)c / nc, n_in_g * (uint64_t)(c + 1) / nc, d_idx, sh->d_roff_g.as<uint64_t>(),
                                       sh->d_rbytes.as<uint8_t>() + rec_total));
                if (off[c + 1] > off[c]) {
                    BF_HIP(ctx, hipEventRecord(ctx->ev[1 + c], st));
                    BF_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[1 + c], 0));
                    BF_HIP(ctx, hipMemcpyAsync(sh->h_rbytes.as<uint8_t>() + rec_total + off[c], sh->d_rbytes.as<uint8_t>() + rec_total + off[c],
                                               off[c + 1] - off[c], hipMemcpyDeviceToHost, ctx->copy_stream));
                }
            }
            rec_total += total_g;
        }
        for (uint32_t s2 = 0; s2 < W; s2++) for (uint32_t o = 0; o < W; o++) recv_before[s2] += rm[(size_t)o * W + s2];
        cl_base += ns_g; rd_base += nr_g; n_in += n_in_g;
    }
    const uint64_t nr = rd_base, ncl = cl_base;
    sh->group_read_begin[G] = nr;
    sh->block_off[(size_t)G * W] = rec_total;
    ctx->out.n_clusters = ncl; ctx->out.n_reads = nr;
    ctx->clustered = true;
    // rows (when groups appended them: the one-go path has copied them itself) and a source rank per read -> host; the
    // places and record offsets stay on the device (bf_shard_result_places)
    BF_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    BF_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[0], 0));
    if (G > 1 && ncl) {
        BF_TRY(ensure_host(ctx, ctx->out.clusters, ncl * sizeof(bf_cluster)));
        BF_HIP(ctx, hipMemcpyAsync(ctx->out.clusters.p, ctx->c_clusters.p, ncl * sizeof(bf_cluster), hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    if (nr) {
        BF_TRY(ensure_host(ctx, sh->h_src, nr));
        BF_HIP(ctx, hipMemcpyAsync(sh->h_src.p, sh->d_src.p, nr, hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    BF_SYNC(ctx, st);
    BF_SYNC(ctx, ctx->copy_stream);                                  // rows, source ranks and records are in host memory
    ctx->consumed = true;
    sh->last_nr = nr; sh->last_n_in = n_in;

    out->n_local = n_local; out->n_evidence = ctx->n_evidence;
    out->n_clusters = ncl; out->n_reads = nr;
    out->clusters = ncl ? ctx->out.clusters.as<bf_cluster>() : nullptr;
    out->src_rank = nr ? sh->h_src.as<uint8_t>() : nullptr;
    out->n_groups = G; out->reserved = 0;
    out->group_read_begin = sh->group_read_begin.data();
    out->n_records = n_in;
    out->block_off = sh->block_off.data();
    out->record_bytes = rec_total ? sh->h_rbytes.as<uint8_t>() : nullptr;
    out->record_size = rec_total;
    out->places = nullptr; out->record_off = nullptr;
    return BF_OK;
}

int bf_shard_result_places(bf_ctx* ctx, bf_shard_result* r) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    Shard* sh = shard_of(ctx);
    if (!sh || !r) return fail(ctx, BF_ERR_STATE, "bf_shard_result_places: no sharded result");
    if (!ctx->clustered || r->n_reads != sh->last_nr || r->n_records != sh->last_n_in || r->block_off != sh->block_off.data())
        return fail(ctx, BF_ERR_STATE, "bf_shard_result_places: not the result of this ctx's last bf_shard_finish");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    if (sh->last_nr) {
        BF_TRY(ensure_host(ctx, sh->h_places, sh->last_nr * sizeof(bf_place)));
        BF_HIP(ctx, hipMemcpyAsync(sh->h_places.p, sh->d_place.p, sh->last_nr * sizeof(bf_place), hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    if (sh->last_n_in) {
        BF_TRY(ensure_host(ctx, sh->h_roff, (sh->last_n_in + 1) * 8));
        BF_HIP(ctx, hipMemcpyAsync(sh->h_roff.p, sh->d_roff.p, (sh->last_n_in + 1) * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    BF_SYNC(ctx, ctx->copy_stream);
    r->places = sh->last_nr ? sh->h_places.as<bf_place>() : nullptr;
    r->record_off = sh->last_n_in ? sh->h_roff.as<uint64_t>() : nullptr;
    return BF_OK;
}

static inline unsigned dec_len64(uint64_t v) { unsigned n = 1; while (v >= 10) { v /= 10; n++; } return n; }
static inline char* put_dec64(char* p, uint64_t v) {
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

struct SvNames {                       // what the merge needs of a genome: names by contig id and their byte-wise order
    uint32_t n_contigs = 0;
    std::vector<std::string> contig_names;
    std::vector<uint32_t> contig_rank;
};
static int64_t shard_format_sv_impl(const SvNames* ctx, uint32_t n_ranks, const bf_shard_result* results, char* buf, uint64_t cap);

int64_t bf_shard_format_sv(const bf_ctx* ctx, uint32_t n_ranks, const bf_shard_result* results, char* buf, uint64_t cap) {
    if (!ctx || !results || n_ranks == 0) return BF_ERR_INVALID;
    SvNames nm;
    nm.n_contigs = ctx->n_contigs; nm.contig_names = ctx->contig_names; nm.contig_rank = ctx->contig_rank;
    return shard_format_sv_impl(&nm, n_ranks, results, buf, cap);
}

int64_t bf_shard_format_sv_names(uint32_t n_contigs, const char* const* names, uint32_t n_ranks, const bf_shard_result* results,
                                 char* buf, uint64_t cap) {
    if (!names || !results || n_ranks == 0 || n_contigs == 0) return BF_ERR_INVALID;
    SvNames nm;
    nm.n_contigs = n_contigs;
    for (uint32_t i = 0; i < n_contigs; i++) { if (!names[i]) return BF_ERR_INVALID; nm.contig_names.emplace_back(names[i]); }
    // rank = position in byte-wise name order (Rust String Ord, detect.rs:194), as bf_genome_begin derives it
    std::vector<uint32_t> order(n_contigs);
    for (uint32_t i = 0; i < n_contigs; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return nm.contig_names[a] < nm.contig_names[b]; });
    nm.contig_rank.assign(n_contigs, 0);
    for (uint32_t r = 0; r < n_contigs; r++) nm.contig_rank[order[r]] = r;
    return shard_format_sv_impl(&nm, n_ranks, results, buf, cap);
}

static int64_t shard_format_sv_impl(const SvNames* ctx, uint32_t n_ranks, const bf_shard_result* results, char* buf, uint64_t cap) {
    static const char header[] = "CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\t"
                                 "SUPPORTING READS\tSIGNATURE\tNOTES\n";                 // detect.rs:201
    struct Row { uint32_t rank_; uint64_t idx; };
    std::vector<Row> rows;
    for (uint32_t r = 0; r < n_ranks; r++) {
        const bf_shard_result& x = results[r];
        if ((x.n_clusters && !x.clusters) || (x.n_reads && (!x.src_rank || !x.group_read_begin)) || !x.block_off ||
            (x.record_size && !x.record_bytes) || x.n_groups == 0 || x.n_groups != results[0].n_groups)
            return BF_ERR_INVALID;
        for (uint64_t c = 0; c < x.n_clusters; c++) {
            const bf_cluster& cl = x.clusters[c];
            if (cl.contig >= ctx->n_contigs || cl.mcontig >= ctx->n_contigs || cl.read_end < cl.read_begin || cl.read_end > x.n_reads) return BF_ERR_INVALID;
            rows.push_back(Row{r, c});
        }
    }
    // the seed's rank in the reference's sort: (chr by name, pos as the wrapped usize, input index), detect.rs:193-198
    auto key = [&](const Row& w) -> const bf_cluster& { return results[w.rank_].clusters[w.idx]; };
    std::sort(rows.begin(), rows.end(), [&](const Row& a, const Row& b) {
        const bf_cluster& x = key(a); const bf_cluster& y = key(b);
        const uint32_t rx = ctx->contig_rank[x.contig], ry = ctx->contig_rank[y.contig];
        if (rx != ry) return rx < ry;
        if ((uint64_t)x.lpos != (uint64_t)y.lpos) return (uint64_t)x.lpos < (uint64_t)y.lpos;
        return x.seed_index < y.seed_index;
    });
    // An owner's rows come up in its own order (the merge keeps it), and its supporting reads with them: the records it
    // asked of source s lie one after the other in s's memory, round by round.  One cursor per (owner, source).
    const uint32_t G = results[0].n_groups;
    struct Walk {
        uint32_t n_ranks, G;
        const bf_shard_result* results;
        std::vector<uint64_t> cur, lim;       // [owner * n_ranks + source]
        std::vector<int64_t> round;           // per owner: the round its cursors stand in
        std::vector<uint64_t> next_read;      // per owner: reads must come up one after the other
        void reset() {
            cur.assign((size_t)n_ranks * n_ranks, 0); lim = cur;
            round.assign(n_ranks, -1); next_read.assign(n_ranks, 0);
        }
        // the record of owner o's supporting read i
        bool record(uint32_t o, uint64_t i, const uint8_t*& rec, uint64_t& size) {
            const bf_shard_result& own = results[o];
            if (i != next_read[o]) return false;
            next_read[o] = i + 1;
            while (round[o] + 1 < (int64_t)G && i >= own.group_read_begin[round[o] + 1]) {
                round[o]++;
                for (uint32_t s = 0; s < n_ranks; s++) {
                    cur[(size_t)o * n_ranks + s] = results[s].block_off[(size_t)round[o] * n_ranks + o];
                    lim[(size_t)o * n_ranks + s] = results[s].block_off[(size_t)round[o] * n_ranks + o + 1];
                }
            }
            if (round[o] < 0 || i >= own.group_read_begin[round[o] + 1]) return false;
            const uint32_t s = own.src_rank[i];
            if (s >= n_ranks) return false;
            uint64_t& c = cur[(size_t)o * n_ranks + s];
            const uint64_t l = lim[(size_t)o * n_ranks + s];
            if (c >= l || l > results[s].record_size) return false;
            rec = results[s].record_bytes + c;
            const int64_t sz = bf_junction_record_bytes(rec, l - c);
            if (sz <= 0 || (uint64_t)sz > l - c) return false;
            size = (uint64_t)sz;
            c += size;
            return true;
        }
    } walk{n_ranks, G, results, {}, {}, {}, {}};
    walk.reset();
    uint64_t total = sizeof(header) - 1;
    for (const Row& w : rows) {
        const bf_cluster& cl = key(w);
        total += ctx->contig_names[cl.contig].size() + ctx->contig_names[cl.mcontig].size() + dec_len64((uint64_t)cl.lpos) +
                 dec_len64((uint64_t)cl.rpos) + 13;
        const uint64_t k = cl.read_end - cl.read_begin;
        if (k) total += k - 1;
        for (uint64_t i = cl.read_begin; i < cl.read_end; i++) {
            const uint8_t* rec; uint64_t size;
            if (!walk.record(w.rank_, i, rec, size)) return BF_ERR_INVALID;
            const int64_t tl = junction_decode(rec, size, nullptr, 0);
            if (tl < 0) return BF_ERR_INVALID;
            total += (uint64_t)tl;
        }
    }
    if (!buf || cap < total) return (int64_t)total;
    walk.reset();
    char* p = buf;
    memcpy(p, header, sizeof(header) - 1); p += sizeof(header) - 1;
    for (const Row& w : rows) {
        const bf_cluster& cl = key(w);
        const std::string& n1 = ctx->contig_names[cl.contig];
        const std::string& n2 = ctx->contig_names[cl.mcontig];
        memcpy(p, n1.data(), n1.size()); p += n1.size();
        *p++ = '\t'; *p++ = cl.strand ? '+' : '-'; *p++ = '\t';
        p = put_dec64(p, (uint64_t)cl.lpos); *p++ = '\t'; *p++ = '\t';
        memcpy(p, n2.data(), n2.size()); p += n2.size();
        *p++ = '\t'; *p++ = cl.mstrand ? '+' : '-'; *p++ = '\t';
        p = put_dec64(p, (uint64_t)cl.rpos); *p++ = '\t'; *p++ = '\t';
        for (uint64_t i = cl.read_begin; i < cl.read_end; i++) {
            if (i > cl.read_begin) *p++ = ';';
            const uint8_t* rec; uint64_t size;
            if (!walk.record(w.rank_, i, rec, size)) return BF_ERR_INVALID;
            p += junction_decode(rec, size, (uint8_t*)p, ~0ull);
        }
        *p++ = '\t'; *p++ = '\t'; *p++ = '\n';
    }
    return (int64_t)total;
}

}  // extern "C"
