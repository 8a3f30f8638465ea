// C ABI of libbreakfast_hip.so (include/breakfast_hip.h): context, genome image, read residency,
// stage orchestration and the .sv renderer.  All compute is in localise.hip / cluster.hip; there is
// no CPU implementation of the path in this library.

#include "bf_internal.h"
#include <chrono>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <ctime>
#include <unistd.h>
#include <mutex>
#include <numeric>
#include <thread>

static thread_local std::string g_create_error;

namespace bf {

int fail(bf_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

bool trace_on() {
    static const bool on = [] { const char* e = getenv("BREAKFAST_TRACE"); return e && *e && *e != '0'; }();
    return on;
}
void trace_line(const bf_ctx* ctx, const char* what, const char* name, const char* file, int line) {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    const char* base = strrchr(file, '/');
    fprintf(stderr, "[bf-trace pid=%d rank=%u/%u t=%.6f] %s %s %s:%d\n", (int)getpid(), ctx ? ctx->rank : 0u, ctx ? ctx->n_ranks : 0u,
            (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, what, name, base ? base + 1 : file, line);
    fflush(stderr);
}
int traced_sync(bf_ctx* ctx, hipStream_t s, const char* name, const char* file, int line) {
    const bool t = trace_on();
    if (t) trace_line(ctx, "wait", name, file, line);
    // The waits inside a step are short (tens of microseconds) and the GPU idles until the host is back: poll for a
    // while before blocking (a blocked thread is woken 15-20 us after the stream drained).
    hipError_t e = hipErrorNotReady;
    static const bool spin = getenv("BREAKFAST_HIP_NO_SPIN") == nullptr;       // (A/B switch)
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        while ((e = hipStreamQuery(s)) == hipErrorNotReady)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
        if (e != hipSuccess && e != hipErrorNotReady) (void)hipGetLastError();
    }
    if (e != hipSuccess) e = hipStreamSynchronize(s);
    if (t) trace_line(ctx, e == hipSuccess ? "done" : "FAILED", name, file, line);
    if (e != hipSuccess) return fail(ctx, BF_ERR_HIP, "hipStreamSynchronize(%s) failed: %s (%s:%d)", name, hipGetErrorString(e), file, line);
    return BF_OK;
}

int ensure(bf_ctx* ctx, DevBuf& b, size_t bytes, bool keep) {
    if (bytes <= b.cap && b.p) return BF_OK;
    if (bytes == 0) bytes = 16;
    size_t cap = keep ? std::max(bytes, b.cap + b.cap / 2) : bytes;
    cap = (cap + 255) & ~(size_t)255;
    void* np = nullptr;
    hipError_t e = hipMalloc(&np, cap);
    if (e != hipSuccess) return fail(ctx, BF_ERR_NOMEM, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
    if (b.p) {
        if (keep && b.cap) {
            e = hipMemcpyAsync(np, b.p, b.cap, hipMemcpyDeviceToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { (void)hipFree(np); return fail(ctx, BF_ERR_HIP, "device copy failed: %s", hipGetErrorString(e)); }
        } else {
            (void)hipStreamSynchronize(ctx->stream);
        }
        (void)hipFree(b.p);
        ctx->device_bytes -= b.cap;
    }
    b.p = np; b.cap = cap;
    ctx->device_bytes += cap;
    return BF_OK;
}

int wide_words(bf_ctx* ctx, size_t n, uint64_t** out) {
    BF_TRY(ensure_host(ctx, ctx->h_wide, std::max<size_t>(n, 512) * 8));
    *out = ctx->h_wide.as<uint64_t>();
    return BF_OK;
}
int ensure_host(bf_ctx* ctx, HostBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return BF_OK;
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0;
    }
    size_t cap = std::max<size_t>(bytes + bytes / 4, 4096);
    hipError_t e = hipHostMalloc(&b.p, cap, hipHostMallocDefault);
    if (e != hipSuccess) { b.p = nullptr; return fail(ctx, BF_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", cap, hipGetErrorString(e)); }
    b.cap = cap;
    return BF_OK;
}

void release(bf_ctx* ctx, DevBuf& b) {
    if (b.p) { (void)hipFree(b.p); ctx->device_bytes -= b.cap; }
    b.p = nullptr; b.cap = 0;
}

struct PendingEvent { std::string name; hipEvent_t a, b; };
static std::vector<PendingEvent>& pending(bf_ctx* ctx) {
    return *reinterpret_cast<std::vector<PendingEvent>*>(ctx->timing_pending);
}

Scope::Scope(bf_ctx* c, const char* n) : ctx(c), name(n) {
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, ctx->stream);
}
Scope::~Scope() {
    if (!a) return;
    (void)hipEventRecord(b, ctx->stream);
    if (pending(ctx).size() >= 8192) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); return; }
    pending(ctx).push_back(PendingEvent{name, a, b});
}

void Timing::clear() { names.clear(); ms.clear(); launches.clear(); view.clear(); }
void Timing::add(const char* name, float ms_) {
    for (size_t i = 0; i < names.size(); i++)
        if (names[i] == name) { ms[i] += ms_; launches[i]++; return; }
    names.push_back(name); ms.push_back(ms_); launches.push_back(1);
}

static void drop_pending(bf_ctx* ctx) {
    for (auto& p : pending(ctx)) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    pending(ctx).clear();
}

static OwnerView local_view(bf_ctx* ctx) {
    OwnerView v;
    v.n = ctx->n_reads;
    v.key = ctx->d_key.as<uint64_t>(); v.lpos = ctx->d_lpos.as<int32_t>(); v.rpos = ctx->d_rpos.as<int32_t>();
    v.bp = ctx->d_bp.as<uint16_t>(); v.len = ctx->d_len.as<uint16_t>();
    v.flags = ctx->d_flags.as<uint16_t>();
    v.frag = ctx->d_frag.as<uint8_t>(); v.frag_off = ctx->d_frag_off.as<uint64_t>(); v.frag_len = ctx->d_frag_len.as<uint16_t>();
    v.sigx = ctx->d_sigx.as<uint8_t>(); v.sigx_slot = ctx->d_sigx_slot.as<uint32_t>();
    v.rw = ctx->d_jrec.as<RecWord>();
    return v;
}
static void free_all(bf_ctx* ctx) {
    DevBuf* all[] = {&ctx->g_blocks, &ctx->g_exc, &ctx->g_raw, &ctx->g_lower, &ctx->g_run_start, &ctx->g_run_byte, &ctx->g_lower_start, &ctx->g_lower_len, &ctx->g_run_len, &ctx->g_coarse, &ctx->g_coarse2, &ctx->g_contigs, &ctx->d_seq, &ctx->d_seq2, &ctx->d_exc_read, &ctx->d_exc_pos, &ctx->d_exc_code, &ctx->d_seq_off,
        &ctx->d_frag, &ctx->d_frag_off, &ctx->d_hits, &ctx->d_len, &ctx->d_frag_len, &ctx->d_fhash,
        &ctx->d_key, &ctx->d_lpos, &ctx->d_rpos, &ctx->d_bp, &ctx->d_flags, &ctx->d_mask, &ctx->d_jrec, &ctx->d_exact_list,
        &ctx->d_counters, &ctx->d_sigx, &ctx->d_sigx_slot, &ctx->d_send, &ctx->d_part_key, &ctx->d_part_idx,
        &ctx->d_part_tmp, &ctx->i_key, &ctx->i_lpos, &ctx->i_rpos, &ctx->i_bp, &ctx->i_len, &ctx->i_flags,
        &ctx->i_frag, &ctx->i_frag_off, &ctx->i_frag_len, &ctx->i_sigx, &ctx->i_sigx_slot, &ctx->i_src,
        &ctx->i_jrec, &ctx->c_tmp, &ctx->c_a32, &ctx->c_b32, &ctx->c_c32, &ctx->c_d32, &ctx->c_k64a, &ctx->c_k64b,
        &ctx->c_slpos, &ctx->c_srpos, &ctx->c_reported, &ctx->c_csize, &ctx->c_flag, &ctx->c_scan, &ctx->c_qseed,
        &ctx->c_qsize, &ctx->c_qoff, &ctx->c_mrec, &ctx->c_mjrec, &ctx->c_leader, &ctx->c_rep, &ctx->c_qnrep, &ctx->c_ctl, &ctx->c_sel,
        &ctx->c_ord, &ctx->c_clusters, &ctx->c_reads, &ctx->c_misc, &ctx->c_joff, &ctx->c_jbytes, &ctx->c_rsrc,
        &ctx->c_tables, &ctx->c_mkey_s, &ctx->c_slot, &ctx->c_big, &ctx->c_cjo, &ctx->c_aoff, &ctx->c_err, &ctx->c_tmp2, &ctx->c_qord, &ctx->c_gidx, &ctx->c_hist, &ctx->c_comp, &ctx->c_bstart, &ctx->c_chunk, &ctx->c_pos2, &ctx->r_idx, &ctx->r_off, &ctx->r_bytes};
    for (DevBuf* b : all) release(ctx, *b);
    HostBuf* hosts[] = {&ctx->out.clusters, &ctx->out.reads, &ctx->out.src, &ctx->out.junction, &ctx->out.cjo, &ctx->h_wide};
    for (HostBuf* h : hosts) { if (h->p) (void)hipHostFree(h->p); h->p = nullptr; h->cap = 0; }
}

// ---- junction records -> text (host) ----------------------------------------------------------
// bit i of a byte -> byte i of a 64-bit word (value 1)
static const uint64_t* spread_table() {
    static uint64_t t[256];
    static bool init = [] {
        for (int v = 0; v < 256; v++) {
            uint64_t x = 0;
            for (int i = 0; i < 8; i++) x |= (uint64_t)((v >> i) & 1) << (8 * i);
            t[v] = x;
        }
        return true;
    }();
    (void)init;
    return t;
}

// Text of one junction record (include/breakfast_hip.h, "Junction records"): the read's bases, lower case
// where they differ from the flank they were assigned to, with '|' at the breakpoint (detect.rs:155-162).
// Returns len + 1, or -1 for a malformed record; writes only when `out` is given and cap suffices.
static const uint8_t kCompHost[256] = {
#include "bf_comp_table.inc"
};

// A reference record (BF_FLAG_READS_ON_HOST): the bases are read where the caller submitted them from.  Returns -2 when
// the record is well-formed but there is no ctx to find the read with (the stored size is still given).
static int64_t junction_decode_ref(const uint8_t* rec, uint64_t rec_len, uint32_t head, uint8_t* out, uint64_t cap, uint64_t* stored,
                                   const bf_ctx* ctx) {
    if (rec_len < 8) return -1;
    const bool swapped = (head & 4u) != 0, wide = (head & 8u) != 0;
    const uint32_t nl = (head >> 4) & 0xfffu, bp = head >> 16;
    uint32_t r;
    memcpy(&r, rec + 4, 4);
    uint64_t body = 8;
    uint32_t dense_L = 0;
    if (nl == kJrRefDense) {
        if (rec_len < 12) return -1;
        memcpy(&dense_L, rec + 8, 4);
        if (dense_L > 0xffffu) return -1;
        body = 12 + ((uint64_t)dense_L + 7) / 8;
    } else {
        body = 8 + (uint64_t)nl * (wide ? 2 : 1);
    }
    if (rec_len < body) return -1;
    if (stored) *stored = (body + 3) & ~3ull;
    if (!ctx) return -2;
    // the batch that holds resident read r
    const auto& hr = ctx->host_reads;
    size_t lo = 0, hi = hr.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (hr[mid].first + hr[mid].n <= r) lo = mid + 1; else hi = mid; }
    if (lo >= hr.size() || r < hr[lo].first) return -1;
    const bf_ctx::HostReads& h = hr[lo];
    const uint64_t k0 = r - h.first;
    const uint64_t so = h.seq_off[k0];
    const uint32_t L = h.seq_len ? h.seq_len[k0] : (uint32_t)(h.seq_off[k0 + 1] - so);
    if (bp > L || wide != (L > 256u) || (nl == kJrRefDense && dense_L != L)) return -1;
    const uint64_t text = (uint64_t)L + 1;
    const uint8_t* pl = rec + 8;
    if (nl != kJrRefDense)
        for (uint32_t e = 0; e < nl; e++) { uint32_t q = wide ? (uint32_t)pl[2 * e] | ((uint32_t)pl[2 * e + 1] << 8) : pl[e]; if (q >= L) return -1; }
    if (!out || cap < text) return (int64_t)text;
    static const char kNib[] = "=ACMGRSVTWYHKDBN";
    const uint8_t* s = h.seq + so;
    auto base = [&](uint32_t k) -> uint8_t {                 // base k of the read as submitted
        if (!h.seq_len) return s[k];
        const uint8_t b = s[k >> 1];
        return (uint8_t)kNib[(k & 1) ? (b & 15) : (b >> 4)];
    };
    for (uint32_t k = 0; k < L; k++) {                       // re-oriented like detect.rs:112, '|' at bp
        const uint8_t c = swapped ? kCompHost[base(L - 1 - k)] : base(k);
        out[k < bp ? k : k + 1] = c;
    }
    auto lower = [&](uint32_t q) { uint8_t& c = out[q < bp ? q : q + 1u]; if (c >= 'A' && c <= 'Z') c += 32; };   // to_ascii_lowercase
    if (nl == kJrRefDense) {
        for (uint32_t q = 0; q < L; q++) if ((rec[12 + (q >> 3)] >> (q & 7)) & 1) lower(q);
    } else {
        for (uint32_t e = 0; e < nl; e++) lower(wide ? (uint32_t)pl[2 * e] | ((uint32_t)pl[2 * e + 1] << 8) : pl[e]);
    }
    out[bp] = '|';
    return (int64_t)text;
}

int64_t junction_decode(const uint8_t* rec, uint64_t rec_len, uint8_t* out, uint64_t cap, uint64_t* stored, const bf_ctx* ctx) {
    if (!rec || rec_len < 4) return -1;
    uint32_t head;
    memcpy(&head, rec, 4);
    const uint32_t kind = head & 3u;
    if (kind == kJrRef) return junction_decode_ref(rec, rec_len, head, out, cap, stored, ctx);
    if (kind == kJrText) {
        if (rec_len < 8 || (head & 0xff0000f8u)) return -1;
        uint32_t L;
        memcpy(&L, rec + 4, 4);
        const uint32_t bp = (head >> 8) & 0xffffu;
        if (L > 0xffffu || bp > L) return -1;
        const uint64_t text = (uint64_t)L + 1;
        if (rec_len < 8 + text) return -1;
        if (stored) *stored = (8 + text + 3) & ~3ull;
        if (out && cap >= text) memcpy(out, rec + 8, text);
        return (int64_t)text;
    }
    if (kind != kJrDense && kind != kJrSparse) return -1;
    const bool sparse = kind == kJrSparse;
    const uint32_t L = (head >> 3) & 0x1ffu, bp = (head >> 12) & 0x1ffu, nx = (head >> 21) & 0x3fu, n_lower = head >> 27;
    if (bp > L || L > 256u || (!sparse && n_lower)) return -1;
    const uint64_t text = (uint64_t)L + 1;
    const uint32_t units = (L + 7) >> 3, ub = sparse ? 2u : 3u;
    const uint64_t body = 4 + (uint64_t)ub * units + 2ull * nx + n_lower;
    if (rec_len < body) return -1;
    if (stored) *stored = (body + 3) & ~3ull;
    const uint8_t* g = rec + 4;
    const uint8_t* xe = g + (uint64_t)ub * units;            // exception entries {position, printed byte}
    const uint8_t* pl = xe + 2ull * nx;                      // lower-case positions (sparse form)
    for (uint32_t e = 0; e < nx; e++) if (xe[2 * e] >= L) return -1;             // validated in the size query too
    for (uint32_t e = 0; e < n_lower; e++) if (pl[e] >= L) return -1;
    if (!out || cap < text) return (int64_t)text;
    const uint64_t* S = spread_table();
    auto eight = [&](uint32_t k) -> uint64_t {               // the 8 bases from k on, as printed
        const uint8_t* grp = g + ub * (k >> 3);
        const uint8_t lo = grp[0], hi = grp[1], cs = sparse ? 0 : grp[2];
        // 'A' + 2*lo + 0x13*hi - 0x0f*(lo&hi) = A C T G for codes 0..3; + 0x20 = lower case
        return 0x4141414141414141ull + 2 * S[lo] + 0x13 * S[hi] - 0x0f * S[lo & hi] + 0x20 * S[cs];
    };
    // base k prints at out[k] left of the breakpoint and at out[k + 1] right of it; whole 8-byte stores
    // wherever the record has 8 more bases (nothing is ever written past out[L])
    uint32_t k = 0;
    for (; k + 8 <= L; k += 8) {
        const uint64_t v = eight(k);
        if (k + 8 <= bp) memcpy(out + k, &v, 8);
        else if (k >= bp) memcpy(out + k + 1, &v, 8);
        else { memcpy(out + k + 1, &v, 8); memcpy(out + k, &v, bp - k); }
    }
    if (k < L) {
        const uint64_t v = eight(k);
        for (uint32_t j = 0; k + j < L; j++) { const uint32_t q = k + j; out[q < bp ? q : q + 1] = (uint8_t)(v >> (8 * j)); }
    }
    for (uint32_t e = 0; e < n_lower; e++) {                 // plain ACGT bases: lower case is + 0x20
        const uint32_t q = pl[e];
        out[q < bp ? q : q + 1u] |= 0x20;
    }
    for (uint32_t e = 0; e < nx; e++) {                      // exceptions are stored as printed
        const uint32_t q = xe[2 * e];
        out[q < bp ? q : q + 1u] = xe[2 * e + 1];
    }
    out[bp] = '|';
    return (int64_t)text;
}

}  // namespace bf

using namespace bf;

extern "C" {

int bf_abi_version(void) { return BF_ABI_VERSION; }

const char* bf_last_error(const bf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int bf_ctx_create(const bf_config* cfg, bf_ctx** out) {
    if (!cfg || !out) return fail(nullptr, BF_ERR_INVALID, "bf_ctx_create: null argument");
    *out = nullptr;
    if (cfg->anchor_len < 8 || cfg->anchor_len > 32767)
        return fail(nullptr, BF_ERR_RANGE, "anchor_len %u outside [8, 32767] (the reference panics below 8, detect.rs:166)", cfg->anchor_len);
    if (cfg->max_frag_len >= 0x80000000u) return fail(nullptr, BF_ERR_RANGE, "max_frag_len must be < 2^31");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, BF_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= n_dev) return fail(nullptr, BF_ERR_INVALID, "device %d out of range (have %d)", dev, n_dev);
    if ((e = hipSetDevice(dev)) != hipSuccess) return fail(nullptr, BF_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return fail(nullptr, BF_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, BF_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    bf_ctx* ctx = new bf_ctx();
    ctx->cfg = *cfg; ctx->device = dev;
    ctx->timing_pending = new std::vector<PendingEvent>();
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete reinterpret_cast<std::vector<PendingEvent>*>(ctx->timing_pending);
        delete ctx;
        return fail(nullptr, BF_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    bool ok = true;
    if (cfg->flags & BF_FLAG_ONE_STREAM) {
        ctx->copy_stream = ctx->side_stream = ctx->in_stream = ctx->stream;
    } else {
        // Default priority on purpose.  With a high-priority copy / side stream, three processes sharing one GPU stopped
        // inside bf_clusterise (round 1): a high-priority queue waiting for an event can be kept on the hardware while the
        // normal-priority queue that has to signal it is not scheduled.
        ok = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&ctx->in_stream, hipStreamNonBlocking) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming) == hipSuccess;
    {
        const unsigned hw = std::thread::hardware_concurrency();
        ctx->host_threads = cfg->reserved[1] ? std::min<uint32_t>(cfg->reserved[1], 64u) : std::max(1u, std::min(hw ? hw : 1u, 16u));
    }
    for (auto& e2 : ctx->ev) ok = ok && hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&ctx->h_counts, 64 + kMailWords * 8, hipHostMallocDefault) == hipSuccess;
    if (ok) { memset(ctx->h_counts, 0, 64 + kMailWords * 8); ctx->h_mail = (uint64_t*)(ctx->h_counts + 16); }
    if (!ok) { bf_ctx_destroy(ctx); return fail(nullptr, BF_ERR_HIP, "could not create the copy stream / events"); }
    if (scratch_setup(ctx) != BF_OK) {               // the device is current (hipSetDevice above)
        const std::string why = ctx->err;
        bf_ctx_destroy(ctx);
        return fail(nullptr, BF_ERR_HIP, "%s", why.c_str());
    }
    *out = ctx;
    return BF_OK;
}

void bf_ctx_destroy(bf_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    if (ctx->in_stream) (void)hipStreamSynchronize(ctx->in_stream);
    drop_pending(ctx);
    shard_destroy(ctx);
    delete ctx->pool; ctx->pool = nullptr;
    for (auto& sl : ctx->slots) { if (sl.p) (void)hipHostFree(sl.p); if (sl.ev) (void)hipEventDestroy(sl.ev); }
    ctx->slots.clear();
    for (auto& w : ctx->wide) { if (w.p) (void)hipHostFree(w.p); if (w.ev) (void)hipEventDestroy(w.ev); w = bf_ctx::WideSlot{}; }
    for (auto& sl : ctx->gslots) { if (sl.p) (void)hipHostFree(sl.p); if (sl.ev) (void)hipEventDestroy(sl.ev); }
    ctx->gslots.clear();
    for (auto& pr : ctx->pinned) (void)hipHostFree((void*)pr.first);
    ctx->pinned.clear();
    if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
    if (ctx->in_stream && ctx->in_stream != ctx->stream) (void)hipStreamDestroy(ctx->in_stream);
    free_all(ctx);
    for (auto& e2 : ctx->ev) if (e2) (void)hipEventDestroy(e2);
    if (ctx->copy_stream && ctx->copy_stream != ctx->stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->side_stream && ctx->side_stream != ctx->stream) (void)hipStreamDestroy(ctx->side_stream);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    delete reinterpret_cast<std::vector<PendingEvent>*>(ctx->timing_pending);
    delete ctx;
}

int bf_sync(bf_ctx* ctx) {
    if (!ctx) return BF_ERR_INVALID;
    BF_SYNC(ctx, ctx->stream);
    return BF_OK;
}

// ---- the genome, detect.rs:48-56 ------------------------------------------------------------------------------
// bf_genome_begin sizes and clears the device images and fixes the contig table (after it bf_submit may already
// stage reads: it only needs the contig ids); bf_genome_write stages any piece of any contig through pinned slots
// onto the copy-in stream, from any thread, so that a FASTA loader can upload while it is still reading; bf_genome_end
// waits for the pieces and packs the images.  bf_genome_set is the three in one call.
constexpr size_t kStageSlotBytes = 8u << 20;         // pinned staging slots of bf_submit and bf_genome_write
constexpr uint32_t kGenomeSlots = 16;
static_assert(kStageSlotBytes == BF_GENOME_FILL_MAX, "bf_genome_fill hands out one staging slot");

int bf_genome_begin(bf_ctx* ctx, uint32_t n_contigs, const char* const* names, const uint64_t* lens) {
    return genome_begin(ctx, n_contigs, names, lens, false);
}

int bf_genome_begin_image(bf_ctx* ctx, uint32_t n_contigs, const char* const* names, const uint64_t* lens) {
    return genome_begin(ctx, n_contigs, names, lens, true);
}

extern "C++" int bf::genome_begin(bf_ctx* ctx, uint32_t n_contigs, const char* const* names, const uint64_t* lens, bool image) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!names || !lens || n_contigs == 0) return fail(ctx, BF_ERR_INVALID, "bf_genome_begin: null/empty argument");
    if (n_contigs > (1u << kContigBits)) return fail(ctx, BF_ERR_RANGE, "%u contigs; at most %u are supported", n_contigs, 1u << kContigBits);
    if (ctx->n_reads) return fail(ctx, BF_ERR_STATE, "the genome cannot change under resident reads; call bf_reset first");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    ctx->genome_ready = false;
    ctx->n_contigs = 0;
    ctx->contig_bits = 1;
    while ((1u << ctx->contig_bits) < n_contigs) ctx->contig_bits++;
    ctx->contig_names.assign(names, names + n_contigs);
    // rank by byte-wise name order (Rust String Ord == memcmp order); names must be unique
    std::vector<uint32_t> order(n_contigs);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ctx->contig_names[a] < ctx->contig_names[b]; });
    ctx->contig_rank.assign(n_contigs, 0);
    ctx->contig_by_rank = order;
    for (uint32_t r = 0; r < n_contigs; r++) {
        if (r > 0 && ctx->contig_names[order[r]] == ctx->contig_names[order[r - 1]])
            return fail(ctx, BF_ERR_INVALID, "duplicate contig name '%s' (the reference keeps only the last record of a name, detect.rs:55; pass it once)", ctx->contig_names[order[r]].c_str());
        ctx->contig_rank[order[r]] = r;
    }
    ctx->contig_host.resize(n_contigs);
    uint64_t off = 0;
    for (uint32_t i = 0; i < n_contigs; i++) {
        if (lens[i] >= 0x80000000ull - (1ull << 17))
            return fail(ctx, BF_ERR_RANGE, "contig '%s' has %llu bases; the supported maximum is 2^31 - 2^17", names[i], (unsigned long long)lens[i]);
        ctx->contig_host[i] = ContigDev{off, (uint32_t)lens[i], ctx->contig_rank[i]};
        off += (lens[i] + 63) & ~63ull;
    }
    const uint64_t total = (off + 4095) & ~4095ull;      // whole waves of 64 blocks
    if (total / 64 >= 0x7ffffff0ull) return fail(ctx, BF_ERR_RANGE, "genome too large: %llu bases (limit 2^37)", (unsigned long long)total);
    ctx->genome_bases = total;
    const uint64_t n_blocks = total / 64;
    ctx->genome_image = image;
    ctx->n_runs = 0; ctx->n_lower_runs = 0;              // (an image's tables arrive with bf_genome_write_image; none: no runs)
    BF_TRY(ensure(ctx, ctx->g_raw, image ? n_blocks * sizeof(GenomeBlock) + 64 : total + 4096));
    BF_TRY(ensure(ctx, ctx->g_blocks, block_image_slots(n_blocks) * sizeof(GenomeBlock)));
    BF_TRY(ensure(ctx, ctx->g_exc, exc_image_slots(n_blocks) * 8));
    BF_TRY(ensure(ctx, ctx->g_coarse, (n_blocks / 64 + 4) * 8));
    BF_TRY(ensure(ctx, ctx->g_coarse2, (n_blocks / 2048 + 4) * 4));
    BF_TRY(ensure(ctx, ctx->g_contigs, (size_t)n_contigs * sizeof(ContigDev)));
    // on the copy-in stream, so that the pieces written behind them land on cleared memory
    const hipStream_t is = ctx->in_stream;
    if (image) BF_HIP(ctx, hipMemsetAsync(ctx->g_raw.p, 0, n_blocks * sizeof(GenomeBlock) + 64, is));
    else BF_HIP(ctx, hipMemsetAsync(ctx->g_raw.p, 'A', total + 4096, is));   // padding is never compared
    BF_HIP(ctx, hipMemsetAsync(ctx->g_blocks.p, 0, block_image_slots(n_blocks) * sizeof(GenomeBlock), is));
    BF_HIP(ctx, hipMemsetAsync(ctx->g_exc.p, 0, exc_image_slots(n_blocks) * 8, is));
    BF_HIP(ctx, hipMemsetAsync(ctx->g_coarse.p, 0, (n_blocks / 64 + 4) * 8, is));
    BF_HIP(ctx, hipMemsetAsync(ctx->g_coarse2.p, 0, (n_blocks / 2048 + 4) * 4, is));
    BF_HIP(ctx, hipMemcpyAsync(ctx->g_contigs.p, ctx->contig_host.data(), (size_t)n_contigs * sizeof(ContigDev), hipMemcpyHostToDevice, is));
    if (ctx->gslots.size() != kGenomeSlots) ctx->gslots = std::vector<bf_ctx::GenomeSlot>(kGenomeSlots);
    { std::lock_guard<std::mutex> g(ctx->genome_mu); ctx->genome_err.clear(); ctx->genome_err_code = 0; }
    ctx->rank_hist.assign(n_contigs, 0);
    ctx->n_contigs = n_contigs;
    return BF_OK;
}

// Thread-safe; reports through its return value only (the message is kept for bf_genome_end).
static int genome_write_impl(bf_ctx* ctx, uint32_t contig, uint64_t offset, const uint8_t* bytes, uint64_t n, bf_fill_fn fill, void* user) {
    if (!ctx) return BF_ERR_INVALID;
    auto bad = [&](int code, const char* msg) {
        std::lock_guard<std::mutex> g(ctx->genome_mu);
        if (!ctx->genome_err_code) { ctx->genome_err_code = code; ctx->genome_err = msg; }
        return code;
    };
    if (ctx->n_contigs == 0 || ctx->genome_ready) return bad(BF_ERR_STATE, "bf_genome_write outside bf_genome_begin .. bf_genome_end");
    if (ctx->genome_image) return bad(BF_ERR_STATE, "bf_genome_write after bf_genome_begin_image: this genome comes as an image (bf_genome_set_image)");
    if (contig >= ctx->n_contigs || offset > ctx->contig_host[contig].len || n > ctx->contig_host[contig].len - offset)
        return bad(BF_ERR_RANGE, "bf_genome_write: piece outside its contig");
    if (n == 0) return BF_OK;
    if (!bytes && !fill) return bad(BF_ERR_INVALID, "bf_genome_write: null bytes");
    if (fill && n > kStageSlotBytes) return bad(BF_ERR_RANGE, "bf_genome_fill: a piece of more than BF_GENOME_FILL_MAX bytes");
    return genome_stage(ctx, ctx->g_raw.as<uint8_t>() + ctx->contig_host[contig].goff + offset, bytes, n, fill, user);
}

// n bytes to device memory `dst` through the genome's pinned slots, on the copy-in stream.  Thread-safe.
extern "C++" int bf::genome_stage(bf_ctx* ctx, uint8_t* dst, const uint8_t* bytes, uint64_t n, bf_fill_fn fill, void* user) {
    auto bad = [&](int code, const char* msg) {
        std::lock_guard<std::mutex> g(ctx->genome_mu);
        if (!ctx->genome_err_code) { ctx->genome_err_code = code; ctx->genome_err = msg; }
        return code;
    };
    if (hipSetDevice(ctx->device) != hipSuccess) return bad(BF_ERR_HIP, "hipSetDevice failed");
    for (uint64_t a = 0; a < n; a += kStageSlotBytes) {
        const uint64_t len = std::min<uint64_t>(kStageSlotBytes, n - a);
        bf_ctx::GenomeSlot& sl = ctx->gslots[ctx->gslot_next.fetch_add(1) % kGenomeSlots];
        std::lock_guard<std::mutex> g(sl.mu);
        if (!sl.p) {
            if (hipHostMalloc((void**)&sl.p, kStageSlotBytes, hipHostMallocDefault) != hipSuccess) { sl.p = nullptr; return bad(BF_ERR_NOMEM, "pinned staging memory is not available"); }
            if (hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return bad(BF_ERR_HIP, "hipEventCreate failed");
        }
        if (sl.used && hipEventSynchronize(sl.ev) != hipSuccess) return bad(BF_ERR_HIP, "hipEventSynchronize failed");
        if (fill) fill(user, sl.p, len);              // the caller writes the piece straight into the pinned slot
        else memcpy(sl.p, bytes + a, len);
        if (hipMemcpyAsync(dst + a, sl.p, len, hipMemcpyHostToDevice, ctx->in_stream) != hipSuccess ||
            hipEventRecord(sl.ev, ctx->in_stream) != hipSuccess)
            return bad(BF_ERR_HIP, "genome upload failed");
        sl.used = true;
    }
    return BF_OK;
}

int bf_genome_write(bf_ctx* ctx, uint32_t contig, uint64_t offset, const uint8_t* bytes, uint64_t n) {
    return genome_write_impl(ctx, contig, offset, bytes, n, nullptr, nullptr);
}

int bf_genome_fill(bf_ctx* ctx, uint32_t contig, uint64_t offset, uint64_t n, bf_fill_fn fill, void* user) {
    if (ctx && !fill) return fail(ctx, BF_ERR_INVALID, "bf_genome_fill: null callback");
    return genome_write_impl(ctx, contig, offset, nullptr, n, fill, user);
}

int bf_genome_end(bf_ctx* ctx) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (ctx->n_contigs == 0) return fail(ctx, BF_ERR_STATE, "bf_genome_end before bf_genome_begin");
    if (ctx->genome_ready) return BF_OK;
    BF_HIP(ctx, hipSetDevice(ctx->device));
    {
        std::lock_guard<std::mutex> g(ctx->genome_mu);
        if (ctx->genome_err_code) return fail(ctx, ctx->genome_err_code, "%s", ctx->genome_err.c_str());
    }
    // the pack kernel runs on the main stream behind every piece
    BF_HIP(ctx, hipEventRecord(ctx->ev_in, ctx->in_stream));
    BF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));
    BF_TRY(ctx->genome_image ? genome_from_image(ctx) : genome_pack(ctx));
    BF_SYNC(ctx, ctx->stream);
    ctx->genome_ready = true;
    // reads that arrived while the genome was still on its way
    if (ctx->n_reads > ctx->n_localised) {
        BF_HIP(ctx, hipEventRecord(ctx->ev_in, ctx->in_stream));
        ctx->in_pending = true;
        BF_TRY(localise_uploaded(ctx, ctx->n_reads, ctx->max_len));
    }
    return BF_OK;
}

int bf_genome_read(bf_ctx* ctx, uint32_t contig, uint64_t offset, uint64_t n, uint8_t* out) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!ctx->genome_ready) return fail(ctx, BF_ERR_STATE, "bf_genome_read before bf_genome_end");
    if (contig >= ctx->n_contigs || offset > ctx->contig_host[contig].len || n > ctx->contig_host[contig].len - offset)
        return fail(ctx, BF_ERR_RANGE, "bf_genome_read: piece outside its contig");
    if (n == 0) return BF_OK;
    if (!out) return fail(ctx, BF_ERR_INVALID, "bf_genome_read: null output");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf tmp;
    const uint64_t piece = 64u << 20;
    BF_TRY(ensure(ctx, tmp, std::min(n, piece)));
    int rc = BF_OK;
    for (uint64_t a = 0; a < n && rc == BF_OK; a += piece) {
        const uint64_t m = std::min(piece, n - a);
        rc = genome_unpack(ctx, ctx->contig_host[contig].goff + offset + a, m, tmp.as<uint8_t>());
        if (rc == BF_OK && (hipMemcpyAsync(out + a, tmp.p, m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                            hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = fail(ctx, BF_ERR_HIP, "bf_genome_read: copy to the host failed");
    }
    release(ctx, tmp);
    return rc;
}

int bf_genome_set(bf_ctx* ctx, uint32_t n_contigs, const char* const* names, const uint8_t* const* seqs,
                  const uint64_t* lens) {
    if (!ctx) return BF_ERR_INVALID;
    if (!names || !seqs || !lens || n_contigs == 0) return fail(ctx, BF_ERR_INVALID, "bf_genome_set: null/empty argument");
    BF_TRY(bf_genome_begin(ctx, n_contigs, names, lens));
    // pieces of up to 8 MB, staged by a few host threads (one thread copies at ~10 GB/s, PCIe takes 55)
    struct Piece { uint32_t c; uint64_t off, n; };
    std::vector<Piece> pieces;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_contigs; i++) {
        if (lens[i] && !seqs[i]) return fail(ctx, BF_ERR_INVALID, "bf_genome_set: contig %u has no bytes", i);
        for (uint64_t a = 0; a < lens[i]; a += kStageSlotBytes) pieces.push_back(Piece{i, a, std::min<uint64_t>(kStageSlotBytes, lens[i] - a)});
        total += lens[i];
    }
    const size_t nt = total < (4u << 20) ? 1 : std::min<size_t>(ctx->host_threads, pieces.size());
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < pieces.size();) if (bf_genome_write(ctx, pieces[k].c, pieces[k].off, seqs[pieces[k].c] + pieces[k].off, pieces[k].n) != BF_OK) return; };
    if (nt <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    return bf_genome_end(ctx);
}

int bf_reset(bf_ctx* ctx) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    BF_SYNC(ctx, ctx->in_stream);
    BF_SYNC(ctx, ctx->stream);
    ctx->in_pending = false; ctx->n_localised = 0; ctx->consumed = false;
    ctx->off_pending.clear();
    ctx->host_reads.clear();
    ctx->rank_hist.assign(ctx->n_contigs, 0);
    ctx->n_reads = 0; ctx->seq_bytes = 0; ctx->frag_bytes = 0; ctx->seq2_bytes = 0; ctx->n_exc = 0;
    ctx->max_len = 0; ctx->min_len = 0xffffffffu; ctx->max_frag = 0;
    ctx->localised = false; ctx->imported = false; ctx->clustered = false; ctx->n_imported = 0; ctx->n_exact = 0;
    ctx->out.n_clusters = ctx->out.n_reads = ctx->out.junction_size = 0;
    return BF_OK;
}

// ---- bf_submit: the host -> device hand-over ---------------------------------------------------------------------
// The reference consumes reads as Bowtie emits them (detect.rs:76-188).  Here a batch is cut into pieces of
// kPieceReads reads; every piece is validated and uploaded on in_stream -- bytes the caller keeps in
// bf_host_alloc memory straight from there, everything else through a ring of pinned staging slots that a few
// host threads fill (a single thread copies at ~10 GB/s, PCIe takes 55) -- and the packed kernel for the piece is
// queued behind its upload, so that the reads of piece k are localised while piece k+1 is still on PCIe.
// bf_finish is then left with the last piece, the exact kernel and the cluster stage.
constexpr uint64_t kPieceReads = 1u << 20;        // also the granularity of the eager localise launches
constexpr uint64_t kEagerReads = 1u << 18;        // smaller submits are gathered up to this many before a launch
constexpr uint64_t kInlineBytes = 1u << 20;       // pieces below this are staged by the calling thread

static bool is_pinned(const bf_ctx* ctx, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    std::lock_guard<std::mutex> g(const_cast<bf_ctx*>(ctx)->pinned_mu);
    for (auto& r : ctx->pinned) if (b >= r.first && b + n <= r.first + r.second) return true;
    return false;
}

struct SubmitJob {
    int kind;              // 0: offsets + hits of reads [a, b); 1: sequence bytes [a, b); 2: fragment id bytes [a, b)
    uint64_t a, b;
};

struct BatchView {          // bf_batch, bf_packed_batch or bf_packed2_batch
    uint64_t n_reads;
    const uint8_t* seq_bytes; const uint64_t* seq_off; const uint16_t* seq_len;     // seq_len: 4-bit and 2-bit reads only
    const uint8_t* frag_bytes; const uint64_t* frag_off; const bf_hit* hits;
    bool two_bit = false;                                   // seq_bytes holds 2-bit codes: uploaded to d_seq2, expanded on the device
    uint64_t n_exc = 0; const uint32_t* exc_read = nullptr; const uint16_t* exc_pos = nullptr; const uint8_t* exc_code = nullptr;
};

struct SubmitShared {
    bf_ctx* ctx; const BatchView* bat;
    uint64_t r_base;       // first read of the piece inside the batch
    uint64_t n0, seq0, frag0;          // resident reads / bytes before the BATCH (2-bit batches: seq0 counts d_seq2's bytes)
    uint64_t nib = 0;                  // 2-bit batches: 4-bit bytes the piece's reads will take (summed by the jobs)
    bool hits_direct;
    uint8_t* wide = nullptr;           // piece-wide pinned buffer for the rebased offsets (batches in bf_host_alloc memory)
    uint64_t p0 = 0, p1 = 0;           // the piece
    std::vector<SubmitJob> jobs;
    std::atomic<size_t> next{0};
    std::mutex mu;
    uint64_t err_read = ~0ull; int err_code = BF_OK; std::string err_msg;
    uint32_t mx = 0, mn = 0xffffffffu, mf = 0;
    std::vector<uint64_t> hist;        // reads per left contig rank (this batch)
    void error(uint64_t read, int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        std::lock_guard<std::mutex> g(mu);
        if (read < err_read) { err_read = read; err_code = code; err_msg = buf; }     // the first offending read names the error
    }
};

static bool slot_ready(bf_ctx* ctx, bf_ctx::StageSlot& sl) {
    if (!sl.p) {
        if (hipHostMalloc((void**)&sl.p, kStageSlotBytes, hipHostMallocDefault) != hipSuccess) { sl.p = nullptr; return false; }
        if (hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return false;
    }
    if (sl.used && hipEventSynchronize(sl.ev) != hipSuccess) return false;     // its previous upload has left the slot
    (void)ctx;
    return true;
}

static void submit_worker(SubmitShared* sh, int slot_id) {
    bf_ctx* ctx = sh->ctx;
    const BatchView* b = sh->bat;
    // (a pool thread keeps the device it set on its first job; setting it again is cheap)
    if (hipSetDevice(ctx->device) != hipSuccess) { sh->error(0, BF_ERR_HIP, "hipSetDevice failed in a staging thread"); return; }
    bf_ctx::StageSlot& sl = ctx->slots[slot_id];
    uint32_t mx = 0, mn = 0xffffffffu, mf = 0;
    uint64_t nib = 0;
    std::vector<uint64_t> hist(ctx->n_contigs, 0);
    for (size_t j; (j = sh->next.fetch_add(1)) < sh->jobs.size();) {
        const SubmitJob& job = sh->jobs[j];
        const bool wide = job.kind == 0 && sh->wide;
        if (!wide && !slot_ready(ctx, sl)) { sh->error(job.a, BF_ERR_NOMEM, "pinned staging memory is not available"); return; }
        hipError_t e = hipSuccess;
        if (job.kind == 0) {
            // offsets rebased onto the resident arrays, lengths and hits validated on the way (detect.rs:92, :115)
            const uint64_t cnt = job.b - job.a;
            const uint64_t n_off = cnt + 1;                  // entry job.b too: every job's reads are complete in themselves
            uint64_t* so = (uint64_t*)sl.p;
            uint64_t* fo = so + (cnt + 1);
            bf_hit* hh = (bf_hit*)(fo + (cnt + 1));
            uint16_t* ll = (uint16_t*)(hh + cnt);                // lengths of 4-bit reads
            uint16_t* fls = ll + cnt;                            // ... and of their fragment ids (the offsets stay here)
            if (wide) {                                          // this job's part of the piece-wide arrays; the caller uploads them
                const uint64_t pc = sh->p1 - sh->p0;
                so = (uint64_t*)sh->wide + (job.a - sh->p0);
                fo = (uint64_t*)sh->wide + (pc + 1) + (job.a - sh->p0);
                hh = (bf_hit*)((uint64_t*)sh->wide + 2 * (pc + 1)) + (job.a - sh->p0);
                ll = (uint16_t*)((bf_hit*)((uint64_t*)sh->wide + 2 * (pc + 1)) + pc) + (job.a - sh->p0);
                fls = (uint16_t*)((bf_hit*)((uint64_t*)sh->wide + 2 * (pc + 1)) + pc) + pc + (job.a - sh->p0);
            }
            const uint64_t s0 = b->seq_off[0], f0 = b->frag_off[0];
            bool ok = true;
            for (uint64_t i = job.a; i < job.b; i++) {
                const uint64_t sa = b->seq_off[i], sb = b->seq_off[i + 1], fa = b->frag_off[i], fb = b->frag_off[i + 1];
                if (sb < sa || fb < fa) { sh->error(i, BF_ERR_INVALID, "bf_submit: offsets not monotone at read %llu", (unsigned long long)i); ok = false; break; }
                uint64_t L = sb - sa;
                const uint64_t fl = fb - fa;
                if (b->seq_len) {                                // 4-bit codes: ceil(L / 2) bytes hold the L bases; 2-bit: ceil(L / 4)
                    L = b->seq_len[i];
                    const uint64_t want = b->two_bit ? (L + 3) / 4 : (L + 1) / 2;
                    if (sb - sa != want) {
                        sh->error(i, BF_ERR_INVALID, "%s: read %llu has %llu bases in %llu bytes (expected %llu)", b->two_bit ? "bf_submit_packed2" : "bf_submit_packed",
                                  (unsigned long long)i, (unsigned long long)L, (unsigned long long)(sb - sa), (unsigned long long)want);
                        ok = false; break;
                    }
                    nib += (L + 1) / 2;
                    ll[i - job.a] = (uint16_t)L;
                    fls[i - job.a] = (uint16_t)std::min<uint64_t>(fb - fa, 65535);
                }
                if (L > 65535) { sh->error(i, BF_ERR_RANGE, "read %llu has %llu bases; at most 65535 are supported", (unsigned long long)i, (unsigned long long)L); ok = false; break; }
                if (fl > 65535) { sh->error(i, BF_ERR_RANGE, "fragment id of read %llu has %llu bytes; at most 65535", (unsigned long long)i, (unsigned long long)fl); ok = false; break; }
                mx = std::max<uint32_t>(mx, (uint32_t)L); mn = std::min<uint32_t>(mn, (uint32_t)L); mf = std::max<uint32_t>(mf, (uint32_t)fl);
                const bf_hit& h = b->hits[i];
                if (h.contig5 >= ctx->n_contigs || h.contig3 >= ctx->n_contigs) {
                    sh->error(i, BF_ERR_INVALID, "read %llu names contig id %u/%u but the genome has %u contigs (the reference panics on an unknown contig, detect.rs:115)",
                              (unsigned long long)i, h.contig5, h.contig3, ctx->n_contigs);
                    ok = false; break;
                }
                if (h.pos5 >= 0x80000000u || h.pos3 >= 0x80000000u) { sh->error(i, BF_ERR_RANGE, "read %llu: anchor position >= 2^31", (unsigned long long)i); ok = false; break; }
                hist[std::min(ctx->contig_rank[h.contig5], ctx->contig_rank[h.contig3])]++;     // `chr` after detect.rs:108-113
                so[i - job.a] = sh->seq0 + (sa - s0);
                fo[i - job.a] = sh->frag0 + (fa - f0);
                if (!sh->hits_direct) hh[i - job.a] = h;
            }
            if (!ok) return;
            so[cnt] = sh->seq0 + (b->seq_off[job.b] - s0); fo[cnt] = sh->frag0 + (b->frag_off[job.b] - f0);
            if (wide) continue;
            if (!b->seq_len) {                                   // ASCII reads: the offsets go up
                e = hipMemcpyAsync(ctx->d_seq_off.as<uint64_t>() + sh->n0 + job.a, so, n_off * 8, hipMemcpyHostToDevice, ctx->in_stream);
                if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_frag_off.as<uint64_t>() + sh->n0 + job.a, fo, n_off * 8, hipMemcpyHostToDevice, ctx->in_stream);
            } else {                                             // 4-bit reads: the lengths do, the device rebuilds the offsets
                e = hipMemcpyAsync(ctx->d_len.as<uint16_t>() + sh->n0 + job.a, ll, cnt * 2, hipMemcpyHostToDevice, ctx->in_stream);
                if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_frag_len.as<uint16_t>() + sh->n0 + job.a, fls, cnt * 2, hipMemcpyHostToDevice, ctx->in_stream);
            }
            if (e == hipSuccess && !sh->hits_direct)
                e = hipMemcpyAsync(ctx->d_hits.as<bf_hit>() + sh->n0 + job.a, hh, cnt * sizeof(bf_hit), hipMemcpyHostToDevice, ctx->in_stream);
        } else {
            const uint8_t* src = job.kind == 1 ? b->seq_bytes + b->seq_off[0] : b->frag_bytes + b->frag_off[0];
            uint8_t* dst = job.kind == 1 ? (b->two_bit ? ctx->d_seq2.as<uint8_t>() : ctx->d_seq.as<uint8_t>()) + sh->seq0 : ctx->d_frag.as<uint8_t>() + sh->frag0;
            memcpy(sl.p, src + job.a, job.b - job.a);
            e = hipMemcpyAsync(dst + job.a, sl.p, job.b - job.a, hipMemcpyHostToDevice, ctx->in_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(sl.ev, ctx->in_stream);
        if (e != hipSuccess) { sh->error(job.a, BF_ERR_HIP, "upload failed: %s", hipGetErrorString(e)); return; }
        sl.used = true;
    }
    std::lock_guard<std::mutex> g(sh->mu);
    sh->mx = std::max(sh->mx, mx); sh->mn = std::min(sh->mn, mn); sh->mf = std::max(sh->mf, mf);
    sh->nib += nib;
    for (uint32_t r = 0; r < ctx->n_contigs; r++) sh->hist[r] += hist[r];
}

static int reserve_input(bf_ctx* ctx, uint64_t reads, uint64_t seq_bytes, uint64_t frag_bytes) {
    if (reads > ctx->cap_reads) BF_TRY(grow_reads(ctx, std::max<uint64_t>({reads, ctx->cap_reads * 2, (uint64_t)1 << 16})));
    if (ctx->packed_reads && ctx->d_len.cap < ctx->cap_reads * 2) {      // capacity reserved before the ctx took 4-bit reads
        BF_SYNC(ctx, ctx->in_stream);
        BF_SYNC(ctx, ctx->stream);
        BF_TRY(ensure(ctx, ctx->d_len, ctx->cap_reads * 2, true));
    }
    if (ctx->packed_reads && ctx->d_frag_len.cap < ctx->cap_reads * 2) {
        BF_SYNC(ctx, ctx->in_stream);
        BF_SYNC(ctx, ctx->stream);
        BF_TRY(ensure(ctx, ctx->d_frag_len, ctx->cap_reads * 2, true));
    }
    if (seq_bytes + 64 > ctx->d_seq.cap || frag_bytes + 64 > ctx->d_frag.cap || !ctx->d_seq.p || !ctx->d_frag.p) {
        BF_SYNC(ctx, ctx->in_stream);
        BF_SYNC(ctx, ctx->stream);
        if (seq_bytes + 64 > ctx->d_seq.cap || !ctx->d_seq.p) BF_TRY(ensure(ctx, ctx->d_seq, std::max<uint64_t>(seq_bytes + 64, ctx->d_seq.cap * 2), true));
        if (frag_bytes + 64 > ctx->d_frag.cap || !ctx->d_frag.p) BF_TRY(ensure(ctx, ctx->d_frag, std::max<uint64_t>(frag_bytes + 64, ctx->d_frag.cap * 2), true));
    }
    return BF_OK;
}

int bf_reserve(bf_ctx* ctx, uint64_t n_reads, uint64_t seq_bytes, uint64_t frag_bytes) {
    if (!ctx) return BF_ERR_INVALID;
    if (n_reads > 0xfffffff0ull) return fail(ctx, BF_ERR_RANGE, "more than 2^32-16 reads in one ctx");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    return reserve_input(ctx, n_reads, seq_bytes, frag_bytes);
}

int bf_host_alloc(bf_ctx* ctx, uint64_t bytes, void** out) {
    if (!ctx || !out) return BF_ERR_INVALID;
    *out = nullptr;
    BF_HIP(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, std::max<uint64_t>(bytes, 64), hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, BF_ERR_NOMEM, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    {
        std::lock_guard<std::mutex> g(ctx->pinned_mu);
        ctx->pinned.emplace_back((const uint8_t*)p, (size_t)std::max<uint64_t>(bytes, 64));
    }
    *out = p;
    return BF_OK;
}

int bf_host_free(bf_ctx* ctx, void* p) {
    if (!ctx) return BF_ERR_INVALID;
    if (!p) return BF_OK;
    bool known = false;
    {
        std::lock_guard<std::mutex> g(ctx->pinned_mu);
        for (size_t i = 0; i < ctx->pinned.size() && !known; i++)
            if (ctx->pinned[i].first == (const uint8_t*)p) { ctx->pinned.erase(ctx->pinned.begin() + (long)i); known = true; }
    }
    if (known) {
        BF_HIP(ctx, hipSetDevice(ctx->device));
        BF_SYNC(ctx, ctx->in_stream);           // no upload may still be reading it
        (void)hipHostFree(p);
        return BF_OK;
    }
    return fail(ctx, BF_ERR_INVALID, "bf_host_free: not a bf_host_alloc pointer of this ctx");
}

int bf_submit_wait(bf_ctx* ctx) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    BF_SYNC(ctx, ctx->in_stream);
    return BF_OK;
}

static int submit_impl(bf_ctx* ctx, const BatchView* b);

int bf_submit(bf_ctx* ctx, const bf_batch* b) {
    if (!ctx) return BF_ERR_INVALID;
    if (!b) return fail(ctx, BF_ERR_INVALID, "bf_submit: null batch");
    if (ctx->n_reads && ctx->packed_reads && b->n_reads) return fail(ctx, BF_ERR_STATE, "bf_submit after bf_submit_packed: a ctx holds its reads in one form; call bf_reset first");
    const BatchView v{b->n_reads, b->seq_bytes, b->seq_off, nullptr, b->frag_bytes, b->frag_off, b->hits};
    return submit_impl(ctx, &v);
}

int bf_submit_packed(bf_ctx* ctx, const bf_packed_batch* b) {
    if (!ctx) return BF_ERR_INVALID;
    if (!b) return fail(ctx, BF_ERR_INVALID, "bf_submit_packed: null batch");
    if (b->n_reads && !b->seq_len) return fail(ctx, BF_ERR_INVALID, "bf_submit_packed: null seq_len");
    if (ctx->n_reads && !ctx->packed_reads && b->n_reads) return fail(ctx, BF_ERR_STATE, "bf_submit_packed after bf_submit: a ctx holds its reads in one form; call bf_reset first");
    const BatchView v{b->n_reads, b->seq_nibbles, b->seq_off, b->seq_len, b->frag_bytes, b->frag_off, b->hits};
    return submit_impl(ctx, &v);
}

int bf_submit_packed2(bf_ctx* ctx, const bf_packed2_batch* b) {
    if (!ctx) return BF_ERR_INVALID;
    if (!b) return fail(ctx, BF_ERR_INVALID, "bf_submit_packed2: null batch");
    if (b->n_reads && !b->seq_len) return fail(ctx, BF_ERR_INVALID, "bf_submit_packed2: null seq_len");
    if (ctx->cfg.flags & BF_FLAG_READS_ON_HOST)
        return fail(ctx, BF_ERR_STATE, "bf_submit_packed2 with BF_FLAG_READS_ON_HOST: records that name a read are expanded from 4-bit or ASCII arrays");
    if (ctx->n_reads && !ctx->packed_reads && b->n_reads) return fail(ctx, BF_ERR_STATE, "bf_submit_packed2 after bf_submit: a ctx holds its reads in one form; call bf_reset first");
    if (b->n_exceptions && (!b->exc_read || !b->exc_pos || !b->exc_code)) return fail(ctx, BF_ERR_INVALID, "bf_submit_packed2: null exception array");
    if (b->n_reads > 0xffffffffull) return fail(ctx, BF_ERR_RANGE, "bf_submit_packed2: more than 2^32-1 reads in one batch");
    for (uint64_t j = 0; j < b->n_exceptions; j++) {        // (a per cent of the reads at most: checked here, in order)
        const uint32_t r = b->exc_read[j], c = b->exc_code[j];
        if (r >= b->n_reads || b->exc_pos[j] >= b->seq_len[r])
            return fail(ctx, BF_ERR_INVALID, "bf_submit_packed2: exception %llu names base %u of read %u (the batch has %llu reads)", (unsigned long long)j, b->exc_pos[j], r, (unsigned long long)b->n_reads);
        if (c > 15 || c == 1 || c == 2 || c == 4 || c == 8)
            return fail(ctx, BF_ERR_INVALID, "bf_submit_packed2: exception %llu has code %u (any of 0..15 but A, C, G, T = 1, 2, 4, 8)", (unsigned long long)j, c);
        if (j && (b->exc_read[j - 1] > r || (b->exc_read[j - 1] == r && b->exc_pos[j - 1] >= b->exc_pos[j])))
            return fail(ctx, BF_ERR_INVALID, "bf_submit_packed2: exceptions not ascending by (read, position) at %llu", (unsigned long long)j);
    }
    BatchView v{b->n_reads, b->seq_2bit, b->seq_off, b->seq_len, b->frag_bytes, b->frag_off, b->hits};
    v.two_bit = true; v.n_exc = b->n_exceptions; v.exc_read = b->exc_read; v.exc_pos = b->exc_pos; v.exc_code = b->exc_code;
    return submit_impl(ctx, &v);
}

static int submit_impl(bf_ctx* ctx, const BatchView* b) {
    BF_TRACE_CALL(ctx);
    if (ctx->n_contigs == 0) return fail(ctx, BF_ERR_STATE, "bf_submit before bf_genome_set");
    const uint64_t m = b->n_reads;
    if (m == 0) return BF_OK;
    if (!b->seq_bytes || !b->seq_off || !b->frag_off || !b->hits) return fail(ctx, BF_ERR_INVALID, "bf_submit: null array");
    if (ctx->n_reads == 0) ctx->packed_reads = b->seq_len != nullptr;
    if (ctx->n_reads + m > 0xfffffff0ull) return fail(ctx, BF_ERR_RANGE, "more than 2^32-16 reads in one ctx");
    if (b->seq_off[m] < b->seq_off[0] || b->frag_off[m] < b->frag_off[0]) return fail(ctx, BF_ERR_INVALID, "bf_submit: offsets not monotone");
    const uint64_t sb = b->seq_off[m] - b->seq_off[0], fb = b->frag_off[m] - b->frag_off[0];
    if (fb && !b->frag_bytes) return fail(ctx, BF_ERR_INVALID, "bf_submit: null frag_bytes");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    // (2-bit batches: sb counts 2-bit bytes; the reads will take at most twice that as 4-bit codes, ceil(L/2) <= 2 ceil(L/4))
    BF_TRY(reserve_input(ctx, ctx->n_reads + m, ctx->seq_bytes + (b->two_bit ? 2 * sb : sb), ctx->frag_bytes + fb));
    if (b->two_bit) {
        if (ctx->seq2_bytes + sb + 64 > ctx->d_seq2.cap || !ctx->d_seq2.p || ctx->n_exc + b->n_exc > ctx->d_exc_read.cap / 4 || !ctx->d_exc_read.p) {
            BF_SYNC(ctx, ctx->in_stream);                   // nothing may move under a queued upload or an expansion
            BF_SYNC(ctx, ctx->stream);
            BF_TRY(ensure(ctx, ctx->d_seq2, std::max<uint64_t>(ctx->seq2_bytes + sb + 64, ctx->d_seq2.cap * 2), true));
            const uint64_t ne = std::max<uint64_t>({ctx->n_exc + b->n_exc, ctx->d_exc_read.cap / 4 * 2, (uint64_t)4096});
            BF_TRY(ensure(ctx, ctx->d_exc_read, ne * 4, true)); BF_TRY(ensure(ctx, ctx->d_exc_pos, ne * 2, true)); BF_TRY(ensure(ctx, ctx->d_exc_code, ne, true));
        }
    }
    const bool seq_direct = is_pinned(ctx, b->seq_bytes + b->seq_off[0], sb);
    const bool frag_direct = fb == 0 || is_pinned(ctx, b->frag_bytes + b->frag_off[0], fb);
    // "h2d" in bf_kernel_times: from this call's first upload to its last, on the copy-in stream
    hipEvent_t t_a = nullptr, t_b = nullptr;
    if (pending(ctx).size() < 8192 && hipEventCreate(&t_a) == hipSuccess && hipEventCreate(&t_b) == hipSuccess) {
        (void)hipEventRecord(t_a, ctx->in_stream);
    } else {
        if (t_a) (void)hipEventDestroy(t_a);
        t_a = t_b = nullptr;
    }
    struct TimerGuard {                               // whichever way the call ends, the pair is recorded or dropped
        bf_ctx* c; hipEvent_t a, b; bool ok = false;
        ~TimerGuard() {
            if (!ok)                                      // a rejected batch: its pieces' offsets are never needed
                while (!c->off_pending.empty() && c->off_pending.back().r0 >= c->n_reads) c->off_pending.pop_back();
            if (!a) return;
            if (ok && hipEventRecord(b, c->in_stream) == hipSuccess) pending(c).push_back(PendingEvent{"h2d", a, b});
            else { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
        }
    } timer{ctx, t_a, t_b};
    SubmitShared sh;
    sh.ctx = ctx; sh.bat = b; sh.n0 = ctx->n_reads; sh.seq0 = b->two_bit ? ctx->seq2_bytes : ctx->seq_bytes; sh.frag0 = ctx->frag_bytes;
    uint8_t* const d_seq_in = b->two_bit ? ctx->d_seq2.as<uint8_t>() : ctx->d_seq.as<uint8_t>();
    uint64_t nib_before = 0;                                // 2-bit batches: 4-bit bytes of the pieces before this one
    if (b->two_bit && b->n_exc) {
        // the exceptions of the whole batch go up first (a pinned slot at a time), with the reads' resident numbers
        if (ctx->slots.empty()) ctx->slots.resize(std::max<uint32_t>(ctx->host_threads, 1));
        bf_ctx::StageSlot& sl = ctx->slots[0];
        const uint64_t per = kStageSlotBytes / 8;
        for (uint64_t a = 0; a < b->n_exc; a += per) {
            const uint64_t cnt = std::min<uint64_t>(per, b->n_exc - a);
            if (!slot_ready(ctx, sl)) return fail(ctx, BF_ERR_NOMEM, "pinned staging memory is not available");
            uint32_t* er = (uint32_t*)sl.p; uint16_t* ep = (uint16_t*)(er + cnt); uint8_t* ec = (uint8_t*)(ep + cnt);
            for (uint64_t j = 0; j < cnt; j++) { er[j] = (uint32_t)(sh.n0 + b->exc_read[a + j]); ep[j] = b->exc_pos[a + j]; ec[j] = b->exc_code[a + j]; }
            BF_HIP(ctx, hipMemcpyAsync(ctx->d_exc_read.as<uint32_t>() + ctx->n_exc + a, er, cnt * 4, hipMemcpyHostToDevice, ctx->in_stream));
            BF_HIP(ctx, hipMemcpyAsync(ctx->d_exc_pos.as<uint16_t>() + ctx->n_exc + a, ep, cnt * 2, hipMemcpyHostToDevice, ctx->in_stream));
            BF_HIP(ctx, hipMemcpyAsync(ctx->d_exc_code.as<uint8_t>() + ctx->n_exc + a, ec, cnt, hipMemcpyHostToDevice, ctx->in_stream));
            BF_HIP(ctx, hipEventRecord(sl.ev, ctx->in_stream));
            sl.used = true;
        }
    }
    sh.hits_direct = is_pinned(ctx, b->hits, m * sizeof(bf_hit));
    sh.hist.assign(ctx->n_contigs, 0);
    if (ctx->slots.size() < ctx->host_threads) ctx->slots.resize(ctx->host_threads);
    const bool eager = ctx->genome_ready;
    // reads per offsets job: two offset arrays (+1 entry each) and the hits share one slot
    const uint64_t per_job = (kStageSlotBytes - 64) / (8 + 8 + sizeof(bf_hit) + 4) - 2;
    uint32_t mx = ctx->max_len, mn = ctx->min_len, mf = ctx->max_frag;
    const uint64_t piece_reads = ctx->cfg.reserved[2] ? ctx->cfg.reserved[2] : kPieceReads;
    for (uint64_t p0 = 0; p0 < m; p0 += piece_reads) {
        const uint64_t p1 = std::min(m, p0 + piece_reads);
        // the byte ranges behind the piece's reads (offsets are checked job by job; a bad batch is caught below
        // before any kernel sees these reads)
        uint64_t sa = b->seq_off[p0] - b->seq_off[0], se = b->seq_off[p1] - b->seq_off[0];
        uint64_t fa = b->frag_off[p0] - b->frag_off[0], fe = b->frag_off[p1] - b->frag_off[0];
        if ((se < sa || se > sb || fe < fa || fe > fb) && p0) ctx->n_localised = 0;
        if (se < sa || se > sb || fe < fa || fe > fb) return fail(ctx, BF_ERR_INVALID, "bf_submit: offsets not monotone near read %llu", (unsigned long long)p0);
        sh.jobs.clear(); sh.next = 0;
        // arrays in bf_host_alloc memory go first, straight from where they lie: the DMA engine works on them while
        // the host threads below check this piece's offsets and hits.  (Nothing of a batch is resident before its
        // last piece has passed: these copies only touch memory beyond n_reads.)
        if (seq_direct && se > sa)
            BF_HIP(ctx, hipMemcpyAsync(d_seq_in + sh.seq0 + sa, b->seq_bytes + b->seq_off[0] + sa, se - sa, hipMemcpyHostToDevice, ctx->in_stream));
        if (frag_direct && fe > fa)
            BF_HIP(ctx, hipMemcpyAsync(ctx->d_frag.as<uint8_t>() + sh.frag0 + fa, b->frag_bytes + b->frag_off[0] + fa, fe - fa, hipMemcpyHostToDevice, ctx->in_stream));
        if (sh.hits_direct)
            BF_HIP(ctx, hipMemcpyAsync(ctx->d_hits.as<bf_hit>() + sh.n0 + p0, b->hits + p0, (p1 - p0) * sizeof(bf_hit), hipMemcpyHostToDevice, ctx->in_stream));
        // jobs of about staged / threads bytes (at least 1 MB, at most a slot), so that a batch of a few dozen MB still
        // keeps every staging thread busy; when only offsets are staged, one job per thread
        const uint64_t staged = (p1 - p0) * 32 + (seq_direct ? 0 : se - sa) + (frag_direct ? 0 : fe - fa);
        const uint64_t job_bytes = std::min<uint64_t>(kStageSlotBytes, std::max<uint64_t>(1u << 20, staged / std::max(1u, ctx->host_threads)));
        uint64_t job_reads = std::min<uint64_t>(per_job, std::max<uint64_t>(job_bytes / 34, 1));
        if (seq_direct && frag_direct)
            job_reads = std::min<uint64_t>(per_job, std::max<uint64_t>((p1 - p0 + ctx->host_threads - 1) / std::max(1u, ctx->host_threads), 1u << 15));
        for (uint64_t a = p0; a < p1; a += job_reads) sh.jobs.push_back(SubmitJob{0, a, std::min(p1, a + job_reads)});
        if (!seq_direct) for (uint64_t a = sa; a < se; a += job_bytes) sh.jobs.push_back(SubmitJob{1, a, std::min(se, a + job_bytes)});
        if (!frag_direct) for (uint64_t a = fa; a < fe; a += job_bytes) sh.jobs.push_back(SubmitJob{2, a, std::min(fe, a + job_bytes)});
        // a batch wholly in bf_host_alloc memory: the threads write one piece-wide set of arrays, uploaded in four copies
        sh.wide = nullptr; sh.p0 = p0; sh.p1 = p1; sh.nib = 0;
        bf_ctx::WideSlot* ws = nullptr;
        if (seq_direct && frag_direct) {
            ws = &ctx->wide[ctx->wide_next++ & 1];
            const size_t need = (p1 - p0 + 1) * 16 + (p1 - p0) * (sizeof(bf_hit) + 4) + 64;
            if (ws->used) { BF_HIP(ctx, hipEventSynchronize(ws->ev)); ws->used = false; }
            if (ws->cap < need) {
                if (ws->p) (void)hipHostFree(ws->p);
                ws->p = nullptr; ws->cap = 0;
                if (hipHostMalloc((void**)&ws->p, need, hipHostMallocDefault) != hipSuccess) { ws->p = nullptr; return fail(ctx, BF_ERR_NOMEM, "pinned staging memory is not available"); }
                ws->cap = need;
            }
            if (!ws->ev) BF_HIP(ctx, hipEventCreateWithFlags(&ws->ev, hipEventDisableTiming));
            sh.wide = ws->p;
        }
        const size_t nt = staged < kInlineBytes ? 1 : std::min<size_t>(ctx->host_threads, sh.jobs.size());
        if (nt <= 1) submit_worker(&sh, 0);
        else {
            if (!ctx->pool) ctx->pool = new HostPool();
            ctx->pool->run((int)nt, [&sh](int t) { submit_worker(&sh, t); });
        }
        if (sh.err_code != BF_OK) {
            // nothing of this batch becomes resident: its uploads only touched memory beyond n_reads, and whatever
            // was localised behind them is dropped (the next bf_finish localises the resident reads afresh)
            if (p0) ctx->n_localised = 0;
            return fail(ctx, sh.err_code, "%s", sh.err_msg.c_str());
        }
        if (ws) {
            const uint64_t pc = p1 - p0;
            uint64_t* so = (uint64_t*)ws->p; uint64_t* fo = so + (pc + 1);
            bf_hit* hh = (bf_hit*)(fo + (pc + 1)); uint16_t* ll = (uint16_t*)(hh + pc);
            if (!b->seq_len) {
                BF_HIP(ctx, hipMemcpyAsync(ctx->d_seq_off.as<uint64_t>() + sh.n0 + p0, so, (pc + 1) * 8, hipMemcpyHostToDevice, ctx->in_stream));
                BF_HIP(ctx, hipMemcpyAsync(ctx->d_frag_off.as<uint64_t>() + sh.n0 + p0, fo, (pc + 1) * 8, hipMemcpyHostToDevice, ctx->in_stream));
            } else {                                            // both length arrays lie back to back: one copy each
                BF_HIP(ctx, hipMemcpyAsync(ctx->d_len.as<uint16_t>() + sh.n0 + p0, ll, pc * 2, hipMemcpyHostToDevice, ctx->in_stream));
                BF_HIP(ctx, hipMemcpyAsync(ctx->d_frag_len.as<uint16_t>() + sh.n0 + p0, ll + pc, pc * 2, hipMemcpyHostToDevice, ctx->in_stream));
            }
            if (!sh.hits_direct) BF_HIP(ctx, hipMemcpyAsync(ctx->d_hits.as<bf_hit>() + sh.n0 + p0, hh, pc * sizeof(bf_hit), hipMemcpyHostToDevice, ctx->in_stream));
            BF_HIP(ctx, hipEventRecord(ws->ev, ctx->in_stream));
            ws->used = true;
        }
        mx = std::max(mx, sh.mx); mn = std::min(mn, sh.mn); mf = std::max(mf, sh.mf);
        if (mn < 2 * ctx->cfg.anchor_len && p0) ctx->n_localised = 0;
        if (mn < 2 * ctx->cfg.anchor_len)
            return fail(ctx, BF_ERR_RANGE, "a read of length %u is shorter than 2 * anchor_len = %u (detect.rs:311 never emits such reads)", mn, 2 * ctx->cfg.anchor_len);
        if (b->seq_len && !b->two_bit) ctx->off_pending.push_back(bf_ctx::OffsetPiece{sh.n0 + p0, sh.n0 + p1, sh.seq0 + sa, sh.frag0 + fa});
        if (b->two_bit) {
            bf_ctx::OffsetPiece pc{sh.n0 + p0, sh.n0 + p1, ctx->seq_bytes + nib_before, sh.frag0 + fa};
            pc.two_bit = true; pc.seq2_base = sh.seq0 + sa;
            pc.exc0 = ctx->n_exc + (uint64_t)(std::lower_bound(b->exc_read, b->exc_read + b->n_exc, (uint32_t)p0) - b->exc_read);
            pc.exc1 = ctx->n_exc + (uint64_t)(std::lower_bound(b->exc_read, b->exc_read + b->n_exc, (uint32_t)std::min<uint64_t>(p1, 0xffffffffull)) - b->exc_read);
            if (p1 > 0xffffffffull) pc.exc1 = ctx->n_exc + b->n_exc;
            ctx->off_pending.push_back(pc);
            nib_before += sh.nib;
        }
        if (p1 < m && eager) {
            // a finished piece in the middle of a large batch: localise it behind its upload.  (The batch becomes
            // resident only when its last piece has been checked; a rejected batch drops the evidence again.)
            BF_HIP(ctx, hipEventRecord(ctx->ev_in, ctx->in_stream));
            ctx->in_pending = true;
            const int rc = localise_uploaded(ctx, sh.n0 + p1, mx);
            if (rc != BF_OK) { ctx->n_localised = 0; return rc; }
        }
    }
    if (ctx->cfg.flags & BF_FLAG_READS_ON_HOST)              // the caller keeps these arrays for the results to point into
        ctx->host_reads.push_back(bf_ctx::HostReads{ctx->n_reads, m, b->seq_bytes, b->seq_off, b->seq_len});
    ctx->n_reads += m; ctx->frag_bytes += fb;
    if (b->two_bit) { ctx->seq_bytes += nib_before; ctx->seq2_bytes += sb; ctx->n_exc += b->n_exc; }
    else ctx->seq_bytes += sb;
    ctx->max_len = mx; ctx->min_len = mn; ctx->max_frag = mf;
    ctx->localised = false; ctx->imported = false; ctx->consumed = false;
    if (ctx->rank_hist.size() != ctx->n_contigs) ctx->rank_hist.assign(ctx->n_contigs, 0);
    for (uint32_t r = 0; r < ctx->n_contigs; r++) ctx->rank_hist[r] += sh.hist[r];
    timer.ok = true;
    BF_HIP(ctx, hipEventRecord(ctx->ev_in, ctx->in_stream));
    ctx->in_pending = true;
    if (eager && ctx->n_reads - std::min(ctx->n_localised, ctx->n_reads) >= std::min(kEagerReads, piece_reads))
        BF_TRY(localise_uploaded(ctx, ctx->n_reads, ctx->max_len));
    return BF_OK;
}

int bf_localise(bf_ctx* ctx) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (ctx->n_contigs == 0 || !ctx->genome_ready) return fail(ctx, BF_ERR_STATE, "bf_localise before bf_genome_set / bf_genome_end");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    ctx->imported = false;
    return run_localise(ctx, false, true);
}

// bf_finish: the host does not wait for the localise kernels' counters, the cluster stage's first
// synchronisation brings them along
static int localise_deferred(bf_ctx* ctx) {
    if (ctx->n_contigs == 0 || !ctx->genome_ready) return fail(ctx, BF_ERR_STATE, "bf_finish before bf_genome_set / bf_genome_end");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    ctx->imported = false;
    // reads bf_submit localised behind their upload are done; when nothing has arrived since the last bf_finish the
    // call runs the whole path again on the resident reads (idempotence, and what a timing loop wants to measure)
    return run_localise(ctx, true, ctx->consumed);
}

static int fill_single_rank_result(bf_ctx* ctx, bf_result* out) {
    out->n_evidence = ctx->n_evidence;
    out->n_clusters = ctx->out.n_clusters;
    out->n_reads = ctx->out.n_reads;
    out->clusters = ctx->out.clusters.as<bf_cluster>();
    out->reads = nullptr;                                          // bf_result_reads
    out->junction_bytes = ctx->out.junction.as<uint8_t>();
    out->junction_size = ctx->out.junction_size;
    out->cluster_junction_off = ctx->out.n_reads ? ctx->out.cjo.as<uint64_t>() : nullptr;
    return BF_OK;
}

// The cluster stage a few contig ranks at a time (cluster.hip, cluster_groups): the records of a finished group go
// to the host on the copy stream while the next group is sorted, swept and de-duplicated.  The result is the
// same bytes as in one go: the print order is contig-major, so the groups' rows, reads and records are appended.
constexpr uint64_t kGroupMinReads = 1u << 22;    // below: the hidden half of the copy is shorter than what two groups cost
constexpr uint32_t kMaxGroups = 8;
static int clusterise_in_groups(bf_ctx* ctx, const OwnerView& v, bool* done) {
    *done = false;
    std::vector<uint64_t> gb;
    uint64_t bound = 0;
    // a group costs about 0.25 ms of launches and host round trips whatever its size (0.35 until round 3 took a third of the
    // launches and half of the waits out of the stage): one per 3.3M reads, 2 to 8 (configs[2], 100M reads on one GPU: 8 groups 56.9 ms, 16 groups 60.9)
    // (bf_config.reserved[0] overrides; measured on configs[1], round 3: 2 groups 5.69 ms, 3: 5.42, 4: 5.55; at the round's end 3: 5.32, 4: 5.36, 5: 5.61)
    const uint32_t fit = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(v.n / 3300000, 2), kMaxGroups);
    const uint32_t want = ctx->cfg.reserved[0] ? std::min<uint32_t>(ctx->cfg.reserved[0], 64u) : fit;
    BF_TRY(cluster_groups(ctx, v, want, gb, true));
    if (gb.size() <= 2) {                                          // one contig holds it all: nothing to pipeline
        BF_SYNC(ctx, ctx->stream);                                 // (the cut's kernels and its read-back are through)
        return BF_OK;
    }
    uint64_t cl_base = 0, rd_base = 0, byte_base = 0;
    bool sized = false;
    for (size_t g = 0; g + 1 < gb.size(); g++) {
        ClusterGroup grp{ctx->c_gidx.as<uint32_t>() + gb[g], gb[g + 1] - gb[g], cl_base, rd_base, v.n};
        grp.g = (int)g;
        BF_TRY(run_cluster(ctx, v, &grp));
        if (!sized) {
            // every output buffer at its final size before the first copy is queued: nothing may move under a copy.  The
            // bound (all records' bytes) came along with the first group's waits (cluster_groups itself waits for nothing)
            if (grp.m == 0) BF_SYNC(ctx, ctx->stream);             // (a group without evidence returns before its first wait)
            bound = ctx->h_mail[kMailRecordBytes];
            BF_TRY(ensure(ctx, ctx->c_jbytes, bound + kRenderSlack, false));    // (the render kernels run before the host has the sizes)
            BF_TRY(ensure_host(ctx, ctx->out.junction, bound));
            sized = true;
        }
        if (!grp.nr) continue;
        const uint64_t stride = ctx->rsrc_stride;
        const uint32_t* d_idx = ctx->c_rsrc.as<uint32_t>() + stride + rd_base;       // out_src_idx of this group
        constexpr int kMaxChunks = 2;
        const int nc = grp.nr >= (1u << 16) ? kMaxChunks : 1;
        uint64_t off[kMaxChunks + 1];
        // The record offsets are computed on the device and the render kernels read them there; only the copies to the host
        // need sizes.  So the chunk offsets travel to the mailbox while the render kernels are being queued and run, and the
        // host asks for them after that (the wait in front of the render kernels was 30-40 us of idle GPU per group).
        BF_TRY(render_prepare(ctx, grp.nr, d_idx, ctx->c_joff, nc, off, ctx->c_rsrc.as<uint32_t>() + 3 * stride + rd_base, false));
        BF_TRY(ensure(ctx, ctx->c_cjo, (cl_base + grp.ns + 1) * 8, true));                  // earlier groups' offsets stay
        BF_TRY(cluster_record_offsets(ctx, ctx->c_joff.as<uint64_t>(), ctx->c_cjo.as<uint64_t>() + cl_base, grp.ns, byte_base));
        // (a read can print in two clusters, detect.rs:215-226, so a group's records are not bound by ALL reads' record sizes:
        // the kernels skip what would start beyond the buffer, the check below reports it)
        for (int c = 0; c < nc; c++) {
            BF_TRY(render_range_at(ctx, grp.nr * (uint64_t)c / nc, grp.nr * (uint64_t)(c + 1) / nc, d_idx, ctx->c_joff.as<uint64_t>(),
                                   ctx->c_jbytes.as<uint8_t>() + byte_base, bound - byte_base));
            BF_HIP(ctx, hipEventRecord(ctx->ev[1 + c], ctx->stream));
        }
        BF_TRY(patch_reads_off(ctx, ctx->c_joff.as<uint64_t>(), grp.nr, rd_base, byte_base));   // the table, for bf_result_reads
        BF_TRY(render_prepare_wait(ctx, grp.nr, nc, off));
        const uint64_t total = off[nc];
        if (byte_base + total > bound) {
            // more record bytes than ALL reads' records add up to (reads printed in several clusters): the buffers were
            // sized for that sum before the first group was clustered.  The stage in one go sizes them from what is printed.
            BF_SYNC(ctx, ctx->stream);
            BF_SYNC(ctx, ctx->copy_stream);
            ctx->out.n_clusters = ctx->out.n_reads = ctx->out.junction_size = 0;
            return BF_OK;                                          // *done stays false
        }
        for (int c = 0; c < nc; c++) {
            if (off[c + 1] > off[c]) {
                BF_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[1 + c], 0));
                // (diagnosis: BREAKFAST_HIP_TIME_COPIES=1 puts every record copy's own duration into bf_kernel_times as "d2h")
                static const bool time_copies = getenv("BREAKFAST_HIP_TIME_COPIES") != nullptr;
                hipEvent_t t_a = nullptr, t_b = nullptr;
                if (time_copies && hipEventCreate(&t_a) == hipSuccess && hipEventCreate(&t_b) == hipSuccess) (void)hipEventRecord(t_a, ctx->copy_stream);
                BF_HIP(ctx, hipMemcpyAsync(ctx->out.junction.as<uint8_t>() + byte_base + off[c], ctx->c_jbytes.as<uint8_t>() + byte_base + off[c],
                                           off[c + 1] - off[c], hipMemcpyDeviceToHost, ctx->copy_stream));
                if (t_a && t_b && hipEventRecord(t_b, ctx->copy_stream) == hipSuccess) pending(ctx).push_back(PendingEvent{"d2h", t_a, t_b});
            }
        }
        cl_base += grp.ns; rd_base += grp.nr; byte_base += total;
    }
    ctx->clustered = true;
    ctx->out.n_clusters = cl_base; ctx->out.n_reads = rd_base; ctx->out.junction_size = byte_base;
    if (cl_base) {
        BF_TRY(ensure_host(ctx, ctx->out.clusters, cl_base * sizeof(bf_cluster)));
        BF_TRY(ensure_host(ctx, ctx->out.cjo, (cl_base + 1) * 8));
        BF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        BF_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[0], 0));
        BF_HIP(ctx, hipMemcpyAsync(ctx->out.clusters.p, ctx->c_clusters.p, cl_base * sizeof(bf_cluster), hipMemcpyDeviceToHost, ctx->copy_stream));
        BF_HIP(ctx, hipMemcpyAsync(ctx->out.cjo.p, ctx->c_cjo.p, (cl_base + 1) * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
        ctx->reads_on_device = true;
    }
    BF_SYNC(ctx, ctx->stream);
    BF_SYNC(ctx, ctx->copy_stream);
    *done = true;
    return BF_OK;
}

int bf_clusterise(bf_ctx* ctx, bf_result* out) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!out) return fail(ctx, BF_ERR_INVALID, "bf_clusterise: null result");
    memset(out, 0, sizeof *out);
    if (!ctx->localised) return fail(ctx, BF_ERR_STATE, "bf_clusterise before bf_localise");
    if (ctx->shard) return fail(ctx, BF_ERR_STATE, "bf_clusterise on a ctx of the sharded path; use bf_shard_finish");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    ctx->imported = false;
    const OwnerView v = local_view(ctx);
    ctx->clustered = false; ctx->reads_on_device = false;
    // (with reference records the copy is a fifth as long and hiding half of it pays for no second group:
    //  configs[1] 4.02 ms in one go, 4.59 in two groups, 5.08 in three; groups then only when they are asked for)
    const bool small_records = (ctx->cfg.flags & BF_FLAG_READS_ON_HOST) && !ctx->cfg.reserved[0];
    if (((v.n >= kGroupMinReads && !small_records) || (ctx->cfg.flags & BF_FLAG_GROUPS_ALWAYS)) && !(ctx->cfg.flags & BF_FLAG_ONE_GROUP)) {
        bool done = false;
        BF_TRY(clusterise_in_groups(ctx, v, &done));
        if (done) return fill_single_rank_result(ctx, out);
    }
    BF_TRY(run_cluster(ctx, v));
    ctx->clustered = true;
    const uint64_t nr = ctx->out.n_reads;
    const uint64_t ncl = ctx->out.n_clusters;
    if (nr) {
        {
            // the surviving reads are local.  Their junction records are rendered chunk by
            // chunk, in printed order; every finished chunk goes to pinned host memory on the copy
            // stream while the next chunk renders.  A cluster's records are contiguous and each says
            // how long it is, so the host gets one offset per CLUSTER; the read table {index, offset}
            // (16 bytes a read, a quarter of the copy) stays on the device until bf_result_reads asks.
            const uint32_t* d_idx = ctx->c_rsrc.as<uint32_t>() + nr;      // out_src_idx
            constexpr int kMaxChunks = 8;
            // (reference records are a few bytes each: a chunk's render would be shorter than its launch)
            const int nc = nr < (1u << 16) ? 1 : (ctx->cfg.flags & BF_FLAG_READS_ON_HOST) ? 2 : kMaxChunks;
            uint64_t off[kMaxChunks + 1];
            // emit left every supporting read's record size next to the index lists
            BF_TRY(render_prepare(ctx, nr, d_idx, ctx->c_joff, nc, off, ctx->c_rsrc.as<uint32_t>() + 3 * nr));
            const uint64_t total = off[nc];
            BF_TRY(ensure(ctx, ctx->c_cjo, (ncl + 1) * 8));
            BF_TRY(ensure_host(ctx, ctx->out.cjo, (ncl + 1) * 8));
            BF_TRY(cluster_record_offsets(ctx, ctx->c_joff.as<uint64_t>(), ctx->c_cjo.as<uint64_t>(), ncl));
            BF_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
            BF_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[0], 0));
            BF_HIP(ctx, hipMemcpyAsync(ctx->out.cjo.p, ctx->c_cjo.p, (ncl + 1) * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
            ctx->out.junction_size = total;
            BF_TRY(ensure_host(ctx, ctx->out.junction, total));
            BF_TRY(ensure(ctx, ctx->c_jbytes, total ? total : 1));
            for (int c = 0; c < nc; c++) {
                BF_TRY(render_range(ctx, nr * (uint64_t)c / nc, nr * (uint64_t)(c + 1) / nc, d_idx, ctx->c_joff, ctx->c_jbytes));
                if (off[c + 1] > off[c]) {
                    BF_HIP(ctx, hipEventRecord(ctx->ev[1 + c], ctx->stream));
                    BF_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[1 + c], 0));
                    BF_HIP(ctx, hipMemcpyAsync(ctx->out.junction.as<uint8_t>() + off[c], ctx->c_jbytes.as<uint8_t>() + off[c],
                                               off[c + 1] - off[c], hipMemcpyDeviceToHost, ctx->copy_stream));
                }
            }
            // the table is finished on the device behind the render kernels, while the copies run
            BF_TRY(patch_reads(ctx, ctx->c_joff.as<uint64_t>(), nr));
            ctx->reads_on_device = true;
            BF_HIP(ctx, hipEventRecord(ctx->ev[10], ctx->copy_stream));
            BF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[10], 0));
        }
    }
    BF_SYNC(ctx, ctx->stream);
    BF_SYNC(ctx, ctx->copy_stream);        // the cluster rows went out on it
    out->n_evidence = ctx->n_evidence;
    out->n_clusters = ctx->out.n_clusters;
    out->n_reads = nr;
    out->clusters = ctx->out.clusters.as<bf_cluster>();
    out->reads = nullptr;                                          // bf_result_reads
    out->junction_bytes = ctx->out.junction.as<uint8_t>();
    out->junction_size = ctx->out.junction_size;
    out->cluster_junction_off = nr ? ctx->out.cjo.as<uint64_t>() : nullptr;
    return BF_OK;
}

int bf_result_reads(bf_ctx* ctx, bf_result* result) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!result) return fail(ctx, BF_ERR_INVALID, "bf_result_reads: null result");
    if (!ctx->clustered) return fail(ctx, BF_ERR_STATE, "bf_result_reads before bf_clusterise");
    if (result->n_reads != ctx->out.n_reads || result->clusters != ctx->out.clusters.as<bf_cluster>())
        return fail(ctx, BF_ERR_STATE, "bf_result_reads: not the result of this ctx's last bf_clusterise");
    const uint64_t nr = ctx->out.n_reads;
    if (ctx->shard) return fail(ctx, BF_ERR_STATE, "bf_result_reads on a ctx of the sharded path; use bf_shard_result_places");
    if (ctx->reads_on_device && nr) {
        BF_HIP(ctx, hipSetDevice(ctx->device));
        BF_TRY(ensure_host(ctx, ctx->out.reads, nr * sizeof(bf_read)));
        BF_HIP(ctx, hipMemcpyAsync(ctx->out.reads.p, ctx->c_reads.p, nr * sizeof(bf_read), hipMemcpyDeviceToHost, ctx->stream));
        BF_SYNC(ctx, ctx->stream);
        ctx->reads_on_device = false;
    }
    result->reads = nr ? ctx->out.reads.as<bf_read>() : nullptr;
    return BF_OK;
}

int bf_finish(bf_ctx* ctx, bf_result* out) {
    if (!ctx) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (ctx->shard) return fail(ctx, BF_ERR_STATE, "bf_finish on a ctx of the sharded path; use bf_shard_finish");
    BF_TRY(localise_deferred(ctx));
    BF_TRY(bf_clusterise(ctx, out));
    ctx->consumed = true;
    return BF_OK;
}

static inline unsigned dec_len(uint64_t v) { unsigned n = 1; while (v >= 10) { v /= 10; n++; } return n; }
static inline char* put_dec(char* p, uint64_t v) {
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

static const char kSvHeader[] = "CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\t"
                                "SUPPORTING READS\tSIGNATURE\tNOTES\n";                 // detect.rs:201

// Row c of the .sv (detect.rs:233-240): size first, then the bytes.  Rows are independent, so the
// text is laid out by a prefix sum of row sizes and filled by several host threads.
int64_t bf_format_sv(const bf_ctx* ctx, const bf_result* r, char* buf, uint64_t cap) {
    if (!ctx || !r) return BF_ERR_INVALID;
    const uint64_t nc = r->n_clusters;
    // a read's record: through the read table when the result has one, else by walking the cluster's records
    const uint64_t* cjo = r->reads ? nullptr : r->cluster_junction_off;
    if (nc && (!r->clusters || (r->n_reads && !r->reads && !cjo))) return BF_ERR_INVALID;
    std::vector<uint64_t> off(nc + 1);
    const size_t hdr = sizeof(kSvHeader) - 1;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned max_threads = std::max(1u, std::min(hw, 16u));
    // row sizes (this is where every record is validated), by several threads when there is much to look at
    std::atomic<bool> bad{false};
    auto sizes = [&](uint64_t c0, uint64_t c1) {
        for (uint64_t c = c0; c < c1; c++) {
            const bf_cluster& cl = r->clusters[c];
            if (cl.contig >= ctx->n_contigs || cl.mcontig >= ctx->n_contigs || cl.read_end < cl.read_begin ||
                cl.read_end > r->n_reads) { bad = true; return; }
            uint64_t sz = ctx->contig_names[cl.contig].size() + ctx->contig_names[cl.mcontig].size() +
                          dec_len((uint64_t)cl.lpos) + dec_len((uint64_t)cl.rpos) + 13;   // 10 tabs, 2 strands, newline
            const uint64_t k = cl.read_end - cl.read_begin;
            if (k) sz += k - 1;                                                             // ';' separators
            uint64_t walk = 0, walk_end = 0;
            if (cjo && k) {
                walk = cjo[c]; walk_end = cjo[c + 1];
                if (walk > walk_end || walk_end > r->junction_size) { bad = true; return; }
            }
            for (uint64_t i = cl.read_begin; i < cl.read_end; i++) {
                const uint64_t at = cjo ? walk : r->reads[i].junction_off;
                const uint64_t lim = cjo ? walk_end : r->junction_size;
                if (at > lim) { bad = true; return; }
                uint64_t stored = 0;
                const int64_t tl = junction_decode(r->junction_bytes + at, lim - at, nullptr, 0, &stored, ctx);
                if (tl < 0) { bad = true; return; }
                sz += (uint64_t)tl;
                walk += stored;
            }
            if (cjo && k && walk != walk_end) { bad = true; return; }
            off[c + 1] = sz;
        }
    };
    // One team of threads for both passes (sizes, then -- once thread 0 has added them up -- the text): a thread is started
    // once per call, which is what a caller that renders the table slice by slice pays per slice.
    const uint64_t nt = r->n_reads < (1u << 16) ? 1 : std::min<uint64_t>(max_threads, nc);
    uint64_t total = 0;
    bool do_fill = false;
    std::vector<uint64_t> cut(nt + 1, 0);                // the text pass's ranges: equal bytes per thread
    auto add_up = [&] {
        off[0] = hdr;
        for (uint64_t c = 0; c < nc; c++) off[c + 1] += off[c];
        total = off[nc];
        do_fill = buf && cap >= total && !bad;
        uint64_t c0 = 0;
        for (uint64_t t = 0; t < nt; t++) {
            const uint64_t want = hdr + (total - hdr) * (t + 1) / nt;
            uint64_t c1 = t + 1 == nt ? nc : (uint64_t)(std::upper_bound(off.begin(), off.end(), want) - off.begin());
            c1 = std::min(std::max(c1, c0), nc);
            cut[t + 1] = c1;
            c0 = c1;
        }
    };
    auto fill = [&](uint64_t c0, uint64_t c1) {
        for (uint64_t c = c0; c < c1; c++) {
            const bf_cluster& cl = r->clusters[c];
            char* p = buf + off[c];
            const std::string& n1 = ctx->contig_names[cl.contig];
            const std::string& n2 = ctx->contig_names[cl.mcontig];
            memcpy(p, n1.data(), n1.size()); p += n1.size();
            *p++ = '\t'; *p++ = cl.strand ? '+' : '-'; *p++ = '\t';
            p = put_dec(p, (uint64_t)cl.lpos); *p++ = '\t'; *p++ = '\t';
            memcpy(p, n2.data(), n2.size()); p += n2.size();
            *p++ = '\t'; *p++ = cl.mstrand ? '+' : '-'; *p++ = '\t';
            p = put_dec(p, (uint64_t)cl.rpos); *p++ = '\t'; *p++ = '\t';                // detect.rs:233-235
            uint64_t walk = cjo && cl.read_end > cl.read_begin ? cjo[c] : 0;
            for (uint64_t i = cl.read_begin; i < cl.read_end; i++) {
                if (i > cl.read_begin) *p++ = ';';
                const uint64_t at = cjo ? walk : r->reads[i].junction_off;
                uint64_t stored = 0;
                p += junction_decode(r->junction_bytes + at, r->junction_size - at, (uint8_t*)p, ~0ull, &stored, ctx);
                walk += stored;
            }
            *p++ = '\t'; *p++ = '\t'; *p++ = '\n';                                      // detect.rs:240
        }
    };
    if (nt <= 1) {
        sizes(0, nc);
        if (bad) return BF_ERR_INVALID;
        add_up();
        if (do_fill) { memcpy(buf, kSvHeader, hdr); fill(0, nc); }
    } else {
        std::atomic<uint64_t> arrived{0};
        std::atomic<bool> go{false};
        auto member = [&](uint64_t t) {
            sizes(nc * t / nt, nc * (t + 1) / nt);
            if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nt) {       // the last one in adds up for all
                if (!bad) add_up();
                go.store(true, std::memory_order_release);
            } else {
                for (uint32_t spins = 0; !go.load(std::memory_order_acquire); spins++)
                    if (spins > 64) std::this_thread::yield();
            }
            if (do_fill) {
                if (t == 0) memcpy(buf, kSvHeader, hdr);
                fill(cut[t], cut[t + 1]);
            }
        };
        std::vector<std::thread> pool;
        for (uint64_t t = 1; t < nt; t++) pool.emplace_back(member, t);
        member(0);
        for (auto& th : pool) th.join();
        if (bad) return BF_ERR_INVALID;
    }
    if (!buf || do_fill) return (int64_t)total;
    {                                                        // cap < total: truncated copy, as documented; the size is still returned
        std::string tmp((size_t)total, '\0');
        const int64_t n = bf_format_sv(ctx, r, tmp.data(), total);
        if (n < 0) return n;
        memcpy(buf, tmp.data(), (size_t)cap);
        return (int64_t)total;
    }
}

int64_t bf_junction_text(const uint8_t* record, uint64_t avail, char* buf, uint64_t cap) {
    const int64_t n = junction_decode(record, avail, (uint8_t*)buf, buf ? cap : 0);
    return n < 0 ? (int64_t)BF_ERR_INVALID : n;
}

int64_t bf_junction_text_of(const bf_ctx* ctx, const uint8_t* record, uint64_t avail, char* buf, uint64_t cap) {
    if (!ctx) return BF_ERR_INVALID;
    const int64_t n = junction_decode(record, avail, (uint8_t*)buf, buf ? cap : 0, nullptr, ctx);
    return n < 0 ? (int64_t)BF_ERR_INVALID : n;
}

int64_t bf_junction_record_bytes(const uint8_t* record, uint64_t avail) {
    uint64_t stored = 0;
    const int64_t rc = junction_decode(record, avail, nullptr, 0, &stored);
    if ((rc < 0 && rc != -2) || stored > ((avail + 3) & ~3ull)) return BF_ERR_INVALID;      // (-2: a reference record, sized without its ctx)
    return (int64_t)stored;
}

int64_t bf_result_junction_text(const bf_result* r, uint8_t* text, uint64_t cap, uint64_t* text_off) {
    if (!r || (r->n_reads && !text_off)) return BF_ERR_INVALID;
    // where each record starts: the read table when the result has one, else a walk over every cluster's records
    std::vector<uint64_t> walked;
    if (r->n_reads && !r->reads) {
        const uint64_t* cjo = r->cluster_junction_off;
        if (!cjo || !r->clusters) return BF_ERR_INVALID;
        walked.assign(r->n_reads, ~0ull);
        for (uint64_t c = 0; c < r->n_clusters; c++) {
            const bf_cluster& cl = r->clusters[c];
            if (cl.read_end < cl.read_begin || cl.read_end > r->n_reads) return BF_ERR_INVALID;
            if (cl.read_end == cl.read_begin) continue;
            uint64_t at = cjo[c];
            const uint64_t lim = cjo[c + 1];
            if (at > lim || lim > r->junction_size) return BF_ERR_INVALID;
            for (uint64_t i = cl.read_begin; i < cl.read_end; i++) {
                uint64_t stored = 0;
                if (at > lim || junction_decode(r->junction_bytes + at, lim - at, nullptr, 0, &stored) < 0) return BF_ERR_INVALID;
                walked[i] = at;
                at += stored;
            }
            if (at != lim) return BF_ERR_INVALID;
        }
    }
    auto rec_off = [&](uint64_t i) { return walked.empty() ? r->reads[i].junction_off : walked[i]; };
    uint64_t total = 0;
    for (uint64_t i = 0; i < r->n_reads; i++) {
        const uint64_t at = rec_off(i);
        if (at > r->junction_size) return BF_ERR_INVALID;
        const int64_t tl = junction_decode(r->junction_bytes + at, r->junction_size - at, nullptr, 0);
        if (tl < 0) return BF_ERR_INVALID;
        text_off[i] = total;
        total += (uint64_t)tl;
    }
    if (r->n_reads) text_off[r->n_reads] = total;
    if (!text || cap < total) return (int64_t)total;
    const unsigned hw = std::thread::hardware_concurrency();
    const uint64_t nt = total < (8u << 20) ? 1 : std::max(1u, std::min(hw, 8u));
    auto fill = [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++)
            junction_decode(r->junction_bytes + rec_off(i), r->junction_size - rec_off(i),
                            text + text_off[i], text_off[i + 1] - text_off[i]);
    };
    if (nt <= 1) { fill(0, r->n_reads); return (int64_t)total; }
    std::vector<std::thread> pool;
    for (uint64_t t = 0; t < nt; t++) pool.emplace_back(fill, r->n_reads * t / nt, r->n_reads * (t + 1) / nt);
    for (auto& th : pool) th.join();
    return (int64_t)total;
}

// ---- junction strings on request ---------------------------------------------------------------

int bf_render_junctions(bf_ctx* ctx, uint64_t n, const uint32_t* local_index, const uint8_t** out_bytes, const uint64_t** out_off) {
    if (!ctx || !out_bytes || !out_off) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!ctx->localised) return fail(ctx, BF_ERR_STATE, "bf_render_junctions before bf_localise");
    if (n && !local_index) return fail(ctx, BF_ERR_INVALID, "bf_render_junctions: null index list");
    for (uint64_t i = 0; i < n; i++)
        if (local_index[i] >= ctx->n_reads) return fail(ctx, BF_ERR_INVALID, "bf_render_junctions: index %u out of range", local_index[i]);
    BF_HIP(ctx, hipSetDevice(ctx->device));
    BF_TRY(ensure(ctx, ctx->r_idx, std::max<uint64_t>(n, 1) * 4));
    if (n) BF_HIP(ctx, hipMemcpyAsync(ctx->r_idx.p, local_index, n * 4, hipMemcpyHostToDevice, ctx->stream));
    uint64_t total = 0;
    BF_TRY(run_render(ctx, n, ctx->r_idx.as<uint32_t>(), ctx->r_off, ctx->r_bytes, &total));
    std::vector<uint64_t> roff(n + 1);
    std::vector<uint8_t> recs(total);
    BF_HIP(ctx, hipMemcpyAsync(roff.data(), ctx->r_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (total) BF_HIP(ctx, hipMemcpyAsync(recs.data(), ctx->r_bytes.p, total, hipMemcpyDeviceToHost, ctx->stream));
    BF_SYNC(ctx, ctx->stream);
    // this host-side call hands out TEXT (the device-side call hands out records): expand here
    ctx->render_off_host.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) {
        const int64_t tl = junction_decode(recs.data() + roff[i], roff[i + 1] - roff[i], nullptr, 0, nullptr, ctx);
        if (tl < 0) return fail(ctx, BF_ERR_HIP, "bf_render_junctions: malformed junction record for list entry %llu", (unsigned long long)i);
        ctx->render_off_host[i + 1] = ctx->render_off_host[i] + (uint64_t)tl;
    }
    ctx->render_bytes_host.resize(ctx->render_off_host[n]);
    for (uint64_t i = 0; i < n; i++)
        junction_decode(recs.data() + roff[i], roff[i + 1] - roff[i], ctx->render_bytes_host.data() + ctx->render_off_host[i],
                        ctx->render_off_host[i + 1] - ctx->render_off_host[i], nullptr, ctx);
    *out_bytes = ctx->render_bytes_host.data(); *out_off = ctx->render_off_host.data();
    return BF_OK;
}

int bf_render_junctions_dev(bf_ctx* ctx, uint64_t n, const uint32_t* dev_local_index, uint64_t* dev_off,
                            uint8_t* dev_bytes, uint64_t dev_bytes_cap, uint64_t* total) {
    if (!ctx || !total) return BF_ERR_INVALID;
    BF_TRACE_CALL(ctx);
    if (!ctx->localised) return fail(ctx, BF_ERR_STATE, "bf_render_junctions_dev before bf_localise");
    if (!dev_off || (n && !dev_local_index)) return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev: null argument");
    if (((uintptr_t)dev_bytes & 3) || ((uintptr_t)dev_off & 7) || ((uintptr_t)dev_local_index & 3))
        return fail(ctx, BF_ERR_INVALID, "bf_render_junctions_dev: dev_bytes/dev_local_index must be 4-byte and dev_off 8-byte aligned");
    BF_HIP(ctx, hipSetDevice(ctx->device));
    return render_dev(ctx, n, dev_local_index, dev_off, dev_bytes, dev_bytes_cap, total);
}

uint64_t bf_junction_record_bound(const bf_ctx* ctx) {
    // packed: 4 + 3*ceil(L/8) + 2*(L/8) + 3 <= 5L/8 + 10; text: L + 12
    return ctx ? (uint64_t)ctx->max_len + 24 : 0;
}

// ---- introspection ---------------------------------------------------------------------------

int bf_evidence_fetch(bf_ctx* ctx, bf_evidence_view* v) {
    if (!ctx || !v) return BF_ERR_INVALID;
    if (!ctx->localised) return fail(ctx, BF_ERR_STATE, "bf_evidence_fetch before bf_localise");
    const uint64_t n = ctx->n_reads;
    const uint32_t W = ctx->mask_words;
    ctx->ev_key.resize(n); ctx->ev_lpos.resize(n); ctx->ev_rpos.resize(n); ctx->ev_bp.resize(n); ctx->ev_flags.resize(n);
    ctx->ev_mask.assign(n * W, 0);
    if (n) {
        std::vector<uint64_t> tmp(n * (W + 1));
        BF_SYNC(ctx, ctx->stream);
        BF_HIP(ctx, hipMemcpy(ctx->ev_key.data(), ctx->d_key.p, n * 8, hipMemcpyDeviceToHost));
        BF_HIP(ctx, hipMemcpy(ctx->ev_lpos.data(), ctx->d_lpos.p, n * 4, hipMemcpyDeviceToHost));
        BF_HIP(ctx, hipMemcpy(ctx->ev_rpos.data(), ctx->d_rpos.p, n * 4, hipMemcpyDeviceToHost));
        BF_HIP(ctx, hipMemcpy(ctx->ev_bp.data(), ctx->d_bp.p, n * 2, hipMemcpyDeviceToHost));
        BF_HIP(ctx, hipMemcpy(ctx->ev_flags.data(), ctx->d_flags.p, n * 2, hipMemcpyDeviceToHost));
        BF_HIP(ctx, hipMemcpy(tmp.data(), ctx->d_mask.p, n * (W + 1) * 8, hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < n; r++)
            for (uint32_t w = 0; w < W; w++) ctx->ev_mask[r * W + w] = tmp[r * (W + 1) + w];
    }
    v->n = n; v->mask_words = W;
    v->class_key = ctx->ev_key.data(); v->lpos = ctx->ev_lpos.data(); v->rpos = ctx->ev_rpos.data();
    v->bp = ctx->ev_bp.data(); v->flags = ctx->ev_flags.data(); v->mask = ctx->ev_mask.data();
    return BF_OK;
}

int bf_kernel_times(bf_ctx* ctx, const bf_kernel_time** out, uint32_t* n) {
    if (!ctx || !out || !n) return BF_ERR_INVALID;
    BF_SYNC(ctx, ctx->in_stream);
    BF_SYNC(ctx, ctx->stream);
    ctx->timing.clear();
    for (auto& p : pending(ctx)) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) ctx->timing.add(p.name.c_str(), ms);
    }
    drop_pending(ctx);
    ctx->timing.view.resize(ctx->timing.names.size());
    for (size_t i = 0; i < ctx->timing.names.size(); i++)
        ctx->timing.view[i] = bf_kernel_time{ctx->timing.names[i].c_str(), ctx->timing.ms[i], ctx->timing.launches[i]};
    *out = ctx->timing.view.data(); *n = (uint32_t)ctx->timing.view.size();
    return BF_OK;
}

int bf_stats(const bf_ctx* ctx, uint64_t* n_reads, uint64_t* device_bytes) {
    if (!ctx) return BF_ERR_INVALID;
    if (n_reads) *n_reads = ctx->n_reads;
    if (device_bytes) *device_bytes = ctx->device_bytes;
    return BF_OK;
}

}  // extern "C"
