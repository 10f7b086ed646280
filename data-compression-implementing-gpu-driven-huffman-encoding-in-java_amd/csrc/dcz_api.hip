// dcz_api.hip -- the C ABI of include/dcz.h: context, workspace, launch sequencing, host-pointer wrappers.
// No compute happens on the CPU here: every entry point stages bytes and launches the HIP kernels.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dcz_internal.h"

using namespace dcz;

struct dcz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;       // second stream: code build of one half overlaps K1/K3 of the other half
    hipStream_t aux2 = nullptr;      // third stream: per-chunk SHA-256 of the host-batch twins beside the codec and the copies
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [5], [6]: hand-overs with aux2
    uint64_t* carry = nullptr;       // device u64: payload bytes of the first half
    bool pipeline = true;
    std::string err;
    // batch workspace
    uint16_t* seg_hist = nullptr;
    uint64_t* seg_bitoff = nullptr;
    uint32_t* code = nullptr;
    uint8_t* maxlen = nullptr;
    size_t cap_nseg = 0, cap_K = 0;
    uint8_t* dws = nullptr;          // decoder workspace (DecodeWs: class bytes, work list of fixed-length blocks)
    size_t cap_dws_K = 0;            // blocks it holds
    // staging for the host-pointer API (grow-only)
    uint8_t* st_in = nullptr;
    size_t st_in_cap = 0;
    uint8_t* st_out = nullptr;
    size_t st_out_cap = 0;
    uint8_t* st_meta = nullptr;  // 8 KiB of small per-block device fields
    uint8_t* st_batch = nullptr;  // per-chunk device columns of the host-pointer batch API (grow-only)
    size_t st_batch_K = 0;
    // launch-shape hints (ShapeHint, dcz_internal.h), host-mapped words written by the classification kernels and read
    // without synchronising: [0] last decode call that met a block for k4_fixed, [3] last decode call classified;
    // [1] last encode call with an identity block, [2] last encode call with a block of any other kind.
    // Every word is a COMPLETED call's sequence number; the decision compares them with each other, so it does not
    // depend on how many calls the host has queued (round 2 compared with the issued count: every call after the 8th
    // queued one got the fallback shape).
    volatile uint32_t* hint_host = nullptr;
    uint32_t* hint_dev = nullptr;
    uint32_t epoch[2] = {0, 0};
    uint64_t shapes[4] = {0, 0, 0, 0};  // dcz_ctx_launch_shapes
    static constexpr uint32_t HINT_WINDOW = 8;  // completed calls a sighting stays in force
    ShapeHint next_hint(int which) {
        ShapeHint h;
        if (!hint_host) return h;  // (no mapped memory: always the flat grids)
        h.epoch = ++epoch[which];
        if (which == 0) {
            h.dev = hint_dev + 0;
            h.done = hint_dev + 3;
            const uint32_t with = hint_host[0], done = hint_host[3];
            // nothing completed yet: unknown, take the flat grid
            h.likely = done == 0u || (with != 0u && with + HINT_WINDOW >= done);
        } else {
            h.dev = hint_dev + 1;
            const uint32_t ident = hint_host[1], other = hint_host[2];
            const uint32_t done = ident > other ? ident : other;
            h.likely = done == 0u || (ident != 0u && ident + HINT_WINDOW >= done);
            // K1 stores the input at the same offsets of the output when the last completed calls had identity blocks
            // and none of them had anything else
            h.in_place = in_place_ok && ident != 0u && ident + HINT_WINDOW >= done &&
                         (other == 0u || other + HINT_WINDOW < done);
        }
        return h;
    }
    bool in_place_ok = true;  // DCZ_NO_IN_PLACE=1 turns the speculation off
    void* pinned[2] = {nullptr, nullptr};  // pinned host staging handed out by dcz_ctx_pinned (grow-only)
    size_t pinned_cap[2] = {0, 0};
    // profiling
    bool profiling = false;
    struct Ev {
        int kernel;
        hipEvent_t a, b;
    };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> pool;
    double ms[DCZ_K_COUNT] = {0, 0, 0, 0, 0, 0};
    uint64_t launches[DCZ_K_COUNT] = {0, 0, 0, 0, 0, 0};
};

namespace {

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return DCZ_E_HIP;                                                                     \
        }                                                                                         \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

hipEvent_t take_event(dcz_ctx* c) {
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct KernelTimer {
    dcz_ctx* c;
    hipStream_t s;
    int kernel;
    hipEvent_t a = nullptr, b = nullptr;
    KernelTimer(dcz_ctx* ctx, hipStream_t st, int k) : c(ctx), s(st), kernel(k) {
        if (c->profiling) {
            a = take_event(c);
            b = take_event(c);
            (void)hipEventRecord(a, s);
        }
    }
    ~KernelTimer() {
        if (c->profiling) {
            (void)hipEventRecord(b, s);
            c->pending.push_back({kernel, a, b});
        }
    }
};

int drain_events(dcz_ctx* c) {
    for (auto& ev : c->pending) {
        HIPCHK(c, hipEventSynchronize(ev.b));
        float t = 0.f;
        HIPCHK(c, hipEventElapsedTime(&t, ev.a, ev.b));
        c->ms[ev.kernel] += (double)t;
        c->launches[ev.kernel] += 1;
        c->pool.push_back(ev.a);
        c->pool.push_back(ev.b);
    }
    c->pending.clear();
    return DCZ_OK;
}

struct Geometry {
    uint32_t K;
    uint32_t spb;
    uint64_t nseg;
};

int geometry(size_t n, size_t block_bytes, Geometry* g) {
    if (block_bytes == 0) return DCZ_E_INVALID;
    const size_t K = (n + block_bytes - 1) / block_bytes;
    if (K > 0x7FFFFFFFull || block_bytes > 0xFFFFFFFFull) return DCZ_E_INVALID;
    const size_t eff = (K <= 1) ? (n ? n : 1) : block_bytes;  // a single block never needs more segments than n has
    g->K = (uint32_t)K;
    g->spb = (uint32_t)((eff + SEG - 1) / SEG);
    g->nseg = (uint64_t)g->K * g->spb;
    return DCZ_OK;
}

template <typename T>
int grow(dcz_ctx* c, T** p, size_t* cap, size_t need_elems) {
    if (need_elems <= *cap && *p) return DCZ_OK;
    if (*p) HIPCHK(c, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t elems = need_elems + need_elems / 8 + 64;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(p), elems * sizeof(T)));
    *cap = elems;
    return DCZ_OK;
}

// Decoder workspace for K blocks.  Growing it frees the old buffer, which the stream may still be using: callers that
// must not synchronise (hipGraph capture, pipelines) size it up front with dcz_ctx_reserve.
int reserve_decode(dcz_ctx* c, size_t K, hipStream_t s) {
    if (K <= c->cap_dws_K && c->dws) return DCZ_OK;
    if (c->dws) {
        HIPCHK(c, hipStreamSynchronize(s));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->dws));
    }
    c->dws = nullptr;
    c->cap_dws_K = 0;
    const size_t nk = K + K / 8 + 256;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dws), decode_ws_bytes(nk)));
    c->cap_dws_K = nk;
    return DCZ_OK;
}

int reserve(dcz_ctx* c, const Geometry& g) {
    if (g.nseg > c->cap_nseg || !c->seg_hist) {
        if (c->seg_hist) HIPCHK(c, hipFree(c->seg_hist));
        if (c->seg_bitoff) HIPCHK(c, hipFree(c->seg_bitoff));
        c->seg_hist = nullptr;
        c->seg_bitoff = nullptr;
        c->cap_nseg = 0;
        const size_t ns = (size_t)g.nseg + 64;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->seg_hist), ns * 256 * sizeof(uint16_t)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->seg_bitoff), ns * sizeof(uint64_t)));
        c->cap_nseg = ns;
    }
    if (g.K > c->cap_K || !c->code) {
        if (c->code) HIPCHK(c, hipFree(c->code));
        if (c->maxlen) HIPCHK(c, hipFree(c->maxlen));
        c->code = nullptr;
        c->maxlen = nullptr;
        c->cap_K = 0;
        const size_t nk = (size_t)g.K + 64;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->code), nk * 256 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->maxlen), nk));
        c->cap_K = nk;
    }
    return DCZ_OK;
}

int launch_check(dcz_ctx* c) {
    HIPCHK(c, hipGetLastError());
    return DCZ_OK;
}

// small per-block device fields used by the host-pointer API, carved from st_meta (8 KiB)
struct Meta {
    uint32_t* comp_size;  // 1
    uint64_t* comp_off;   // 1
    uint64_t* total;      // 1
    int32_t* status;      // 1
    int64_t* errpos;      // 1
    uint32_t* orig_size;  // 1
    uint8_t* len8;        // 256
    int32_t* len32;       // 256
    uint32_t* code;       // 256
    int64_t* hist;        // 256
};

Meta meta_of(dcz_ctx* c) {
    uint8_t* p = c->st_meta;
    Meta m;
    m.comp_off = reinterpret_cast<uint64_t*>(p);
    m.total = reinterpret_cast<uint64_t*>(p + 8);
    m.errpos = reinterpret_cast<int64_t*>(p + 16);
    m.comp_size = reinterpret_cast<uint32_t*>(p + 24);
    m.status = reinterpret_cast<int32_t*>(p + 28);
    m.orig_size = reinterpret_cast<uint32_t*>(p + 32);
    m.len8 = p + 256;
    m.len32 = reinterpret_cast<int32_t*>(p + 512);
    m.code = reinterpret_cast<uint32_t*>(p + 512 + 1024);
    m.hist = reinterpret_cast<int64_t*>(p + 512 + 2048);
    return m;
}

}  // namespace

extern "C" {

int dcz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

int dcz_ctx_create(int device, dcz_ctx** out) {
    if (!out) return DCZ_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DCZ_E_NODEVICE;
    if (device < 0 || device >= n) return DCZ_E_INVALID;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return DCZ_E_NODEVICE;
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) return DCZ_E_NODEVICE;  // the code object is gfx950-only
    dcz_ctx* c = new dcz_ctx();
    c->device = device;
    DeviceGuard g(device);
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->st_meta), 8192) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->carry), 64) == hipSuccess;
    for (auto& e : c->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    {
        void* hp = nullptr;
        const char* nh = std::getenv("DCZ_NO_SHAPE_HINT");
        if (!(nh && nh[0] && nh[0] != '0') && hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
            std::memset(hp, 0, 64);
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                c->hint_host = static_cast<volatile uint32_t*>(hp);
                c->hint_dev = static_cast<uint32_t*>(dp);
            } else {
                (void)hipHostFree(hp);
            }
        }
    }
    if (const char* ni = std::getenv("DCZ_NO_IN_PLACE")) c->in_place_ok = !(ni[0] && ni[0] != '0');
    if (const char* np = std::getenv("DCZ_NO_PIPELINE")) c->pipeline = !(np[0] && np[0] != '0');
    if (!ok) {
        dcz_ctx_destroy(c);
        return DCZ_E_HIP;
    }
    *out = c;
    return DCZ_OK;
}

void dcz_ctx_destroy(dcz_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    if (c->aux2) (void)hipStreamSynchronize(c->aux2);
    for (auto& ev : c->pending) {
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    (void)hipFree(c->seg_hist);
    (void)hipFree(c->seg_bitoff);
    (void)hipFree(c->code);
    (void)hipFree(c->maxlen);
    (void)hipFree(c->dws);
    (void)hipFree(c->st_in);
    (void)hipFree(c->st_out);
    (void)hipFree(c->st_meta);
    (void)hipFree(c->st_batch);
    for (auto p : c->pinned)
        if (p) (void)hipHostFree(p);
    if (c->hint_host) (void)hipHostFree(const_cast<uint32_t*>(c->hint_host));
    (void)hipFree(c->carry);
    for (auto e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->aux2) (void)hipStreamDestroy(c->aux2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* dcz_ctx_stream(dcz_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

const char* dcz_strerror(int status) {
    switch (status) {
        case DCZ_OK: return "ok";
        case DCZ_E_INVALID: return "invalid argument";
        case DCZ_E_NODEVICE: return "no gfx950 device available";
        case DCZ_E_HIP: return "HIP runtime error";
        case DCZ_E_CAPACITY: return "output buffer too small";
        case DCZ_E_BADSTREAM: return "Huffman decode error";
        case DCZ_E_CODELEN: return "code length exceeds 32";
        case DCZ_E_BADTABLE: return "code length table is not a prefix code";
        default: return "unknown status";
    }
}

const char* dcz_last_error(const dcz_ctx* c) { return c ? c->err.c_str() : ""; }

int dcz_ctx_reserve(dcz_ctx* c, size_t n, size_t block_bytes) {
    if (!c) return DCZ_E_INVALID;
    Geometry g;
    int r = geometry(n, block_bytes, &g);
    if (r != DCZ_OK) return r;
    DeviceGuard dg(c->device);
    r = reserve(c, g);
    if (r != DCZ_OK) return r;
    return reserve_decode(c, g.K, c->stream);
}

int dcz_compress_blocks(dcz_ctx* c, const void* d_in, size_t n, size_t block_bytes, void* d_out, size_t out_cap,
                        uint32_t* d_comp_size, uint64_t* d_comp_off, uint8_t* d_len, int32_t* d_status,
                        uint64_t* d_total, void* stream) {
    if (!c || (!d_in && n) || !d_comp_size || !d_comp_off || !d_len || !d_status) return DCZ_E_INVALID;
    Geometry g;
    int r = geometry(n, block_bytes, &g);
    if (r != DCZ_OK) return r;
    DeviceGuard dg(c->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (g.K == 0) {
        if (d_total) HIPCHK(c, hipMemsetAsync(d_total, 0, sizeof(uint64_t), s));
        return DCZ_OK;
    }
    if (!d_out) return DCZ_E_INVALID;
    r = reserve(c, g);
    if (r != DCZ_OK) return r;
    const uint8_t* in = static_cast<const uint8_t*>(d_in);
    uint8_t* out = static_cast<uint8_t*>(d_out);
    // One block range [k0, k0+kn): K1 -> K2 -> offsets -> K3.  With many blocks the call is cut in two halves and the
    // latency-bound code build (K2) of one half runs on a second stream while the bandwidth-bound K1/K3 of the other
    // half keep the memory system busy; events order everything, nothing synchronises with the host.
    auto range = [&](uint32_t k0, uint32_t kn, size_t* off_b, size_t* len_b, uint64_t* seg0) {
        *off_b = (size_t)k0 * block_bytes;
        const size_t end = ((size_t)(k0 + kn) * block_bytes < n) ? (size_t)(k0 + kn) * block_bytes : n;
        *len_b = end - *off_b;
        *seg0 = (uint64_t)k0 * g.spb;
    };
    ShapeHint enc_hint = c->next_hint(1);
    // the copy needs room for the whole input at its own offsets, and 16-byte units that are aligned in both buffers
    if (out_cap < n || ((reinterpret_cast<uintptr_t>(in) - reinterpret_cast<uintptr_t>(out)) & 15u) != 0u) enc_hint.in_place = false;
    c->shapes[(enc_hint.likely && !enc_hint.in_place) ? 2 : 3]++;
    auto k1 = [&](uint32_t k0, uint32_t kn, hipStream_t st) {
        size_t ob, lb;
        uint64_t s0;
        range(k0, kn, &ob, &lb, &s0);
        KernelTimer t(c, st, enc_hint.in_place ? DCZ_K_HISTOGRAM_COPY : DCZ_K_HISTOGRAM);
        launch_histogram(in + ob, lb, block_bytes, g.spb, (uint64_t)kn * g.spb, c->seg_hist + s0 * 256u, st,
                         enc_hint.in_place ? out + ob : nullptr);
    };
    auto k2 = [&](uint32_t k0, uint32_t kn, hipStream_t st) {
        size_t ob, lb;
        uint64_t s0;
        range(k0, kn, &ob, &lb, &s0);
        KernelTimer t(c, st, DCZ_K_CODEBUILD);
        launch_codebuild(c->seg_hist + s0 * 256u, nullptr, lb, block_bytes, g.spb, kn, d_len + (size_t)k0 * 256u,
                         c->code + (size_t)k0 * 256u, c->maxlen + k0, d_comp_size + k0, c->seg_bitoff + s0,
                         d_status + k0, st, enc_hint);
    };
    auto k3 = [&](uint32_t k0, uint32_t kn, const uint64_t* carry_in, uint64_t* total_out, hipStream_t st) {
        size_t ob, lb;
        uint64_t s0;
        range(k0, kn, &ob, &lb, &s0);
        {
            KernelTimer t(c, st, DCZ_K_OFFSETS);
            launch_offsets(d_comp_size + k0, kn, d_comp_off + k0, total_out, carry_in, out_cap, d_status + k0, st);
        }
        KernelTimer t(c, st, DCZ_K_ENCODE);
        ShapeHint eh = enc_hint;
        eh.in_offset = ob;
        launch_encode(in + ob, lb, block_bytes, g.spb, kn, d_len + (size_t)k0 * 256u, c->code + (size_t)k0 * 256u,
                      c->maxlen + k0, d_comp_off + k0, c->seg_bitoff + s0, d_status + k0, out, st, eh);
    };
    static const uint32_t pipeline_min_k = [] {
        const char* e = getenv("DCZ_PIPELINE_MIN_K");  // tuning knob
        return e ? (uint32_t)atoi(e) : 2048u;
    }();
    if (c->pipeline && g.K >= pipeline_min_k) {
        const uint32_t ka = g.K / 2, kb = g.K - ka;
        hipStream_t a = c->aux;
        HIPCHK(c, hipEventRecord(c->ev[0], s));  // the aux stream must not run ahead of the caller's earlier work
        HIPCHK(c, hipStreamWaitEvent(a, c->ev[0], 0));
        k1(0, ka, s);
        HIPCHK(c, hipEventRecord(c->ev[1], s));
        HIPCHK(c, hipStreamWaitEvent(a, c->ev[1], 0));
        k2(0, ka, a);  // overlaps K1 of the second half
        HIPCHK(c, hipEventRecord(c->ev[2], a));
        k1(ka, kb, s);
        HIPCHK(c, hipEventRecord(c->ev[3], s));
        HIPCHK(c, hipStreamWaitEvent(a, c->ev[3], 0));
        k2(ka, kb, a);  // overlaps K3 of the first half
        HIPCHK(c, hipEventRecord(c->ev[4], a));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev[2], 0));
        k3(0, ka, nullptr, c->carry, s);
        HIPCHK(c, hipStreamWaitEvent(s, c->ev[4], 0));
        k3(ka, kb, c->carry, d_total ? d_total : c->carry + 1, s);
    } else {
        k1(0, g.K, s);
        k2(0, g.K, s);
        k3(0, g.K, nullptr, d_total, s);
    }
    return launch_check(c);
}

int dcz_decompress_blocks(dcz_ctx* c, const void* d_comp, size_t comp_bytes, const uint64_t* d_comp_off,
                          const uint32_t* d_comp_size, const uint32_t* d_orig_size, const uint8_t* d_len, size_t K,
                          size_t out_stride, void* d_out, int32_t* d_status, int64_t* d_errpos, void* stream) {
    if (!c || !d_comp_off || !d_comp_size || !d_orig_size || !d_len || !d_status) return DCZ_E_INVALID;
    if (K > 0x7FFFFFFFull) return DCZ_E_INVALID;
    if (K == 0) return DCZ_OK;
    if (!d_out) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    int r = reserve_decode(c, K, s);  // no-op (no synchronisation, no allocation) after dcz_ctx_reserve
    if (r != DCZ_OK) return r;
    {
        KernelTimer t(c, s, DCZ_K_DECODE);
        DecodeWs ws = decode_ws_at(c->dws, c->cap_dws_K);
        ws.fixed = c->next_hint(0);
        c->shapes[ws.fixed.likely ? 0 : 1]++;
        launch_decode(static_cast<const uint8_t*>(d_comp), comp_bytes, d_comp_off, d_comp_size, d_orig_size, d_len,
                      (uint32_t)K, out_stride, static_cast<uint8_t*>(d_out), d_status, d_errpos, ws, s);
    }
    return launch_check(c);
}

int dcz_histogram(dcz_ctx* c, const uint8_t* data, size_t offset, size_t length, int64_t hist[256]) {
    if (!c || !hist || (!data && length)) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    hipStream_t s = c->stream;
    Meta m = meta_of(c);
    Geometry g;
    int r = geometry(length, length ? length : 1, &g);
    if (r != DCZ_OK) return r;
    if (length > 0xFFFFFFFFull) return DCZ_E_INVALID;  // byte[] windows are int-sized in the reference
    if ((r = grow(c, &c->st_in, &c->st_in_cap, length + 16)) != DCZ_OK) return r;
    if ((r = reserve(c, g)) != DCZ_OK) return r;
    if (length) HIPCHK(c, hipMemcpyAsync(c->st_in, data + offset, length, hipMemcpyHostToDevice, s));
    {
        KernelTimer t(c, s, DCZ_K_HISTOGRAM);
        launch_histogram(c->st_in, length, length ? length : 1, g.spb, g.nseg, c->seg_hist, s);
        launch_sum_hist(c->seg_hist, g.nseg, m.hist, s);
    }
    HIPCHK(c, hipMemcpyAsync(hist, m.hist, 256 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return launch_check(c);
}

int dcz_build_codes(dcz_ctx* c, const int64_t hist[256], int32_t len[256], uint32_t code[256]) {
    if (!c || !hist || !len || !code) return DCZ_E_INVALID;
    int64_t h[256];
    for (int i = 0; i < 256; i++) {
        if (hist[i] >= (1ll << 37)) return DCZ_E_INVALID;  // packed heap keys hold weights < 2^46
        h[i] = hist[i] > 0 ? hist[i] : 0;                  // CanonicalHuffman.java:61 only takes freq > 0
    }
    DeviceGuard dg(c->device);
    hipStream_t s = c->stream;
    Meta m = meta_of(c);
    HIPCHK(c, hipMemcpyAsync(m.hist, h, sizeof h, hipMemcpyHostToDevice, s));
    {
        KernelTimer t(c, s, DCZ_K_CODEBUILD);
        launch_codebuild(nullptr, m.hist, 0, 1, 0, 1, m.len8, m.code, reinterpret_cast<uint8_t*>(m.orig_size),
                         m.comp_size, nullptr, m.status, s);
    }
    uint8_t l8[256];
    int32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(l8, m.len8, 256, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(code, m.code, 1024, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&st, m.status, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int i = 0; i < 256; i++) len[i] = l8[i];
    int r = launch_check(c);
    return r != DCZ_OK ? r : st;
}

int dcz_codes_from_lengths(dcz_ctx* c, const int32_t len[256], uint32_t code[256]) {
    if (!c || !len || !code) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    hipStream_t s = c->stream;
    Meta m = meta_of(c);
    HIPCHK(c, hipMemcpyAsync(m.len32, len, 1024, hipMemcpyHostToDevice, s));
    launch_codes_from_lengths(m.len32, m.code, m.status, s);
    int32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(code, m.code, 1024, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&st, m.status, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    int r = launch_check(c);
    return r != DCZ_OK ? r : st;
}

int dcz_encode_block(dcz_ctx* c, const uint8_t* data, size_t n, int32_t len_out[256], uint8_t* out, size_t cap,
                     size_t* out_len) {
    if (!c || !len_out || !out_len || (!data && n) || (!out && cap)) return DCZ_E_INVALID;
    *out_len = 0;
    for (int i = 0; i < 256; i++) len_out[i] = 0;
    if (n == 0) return DCZ_OK;  // an empty file has no chunk (CpuCompressionService.java:64)
    if (n > 0x7FFFFFFFull) return DCZ_E_INVALID;  // chunks are Java byte[] (CpuCompressionService.java:38)
    DeviceGuard dg(c->device);
    hipStream_t s = c->stream;
    Meta m = meta_of(c);
    int r;
    if ((r = grow(c, &c->st_in, &c->st_in_cap, n + 16)) != DCZ_OK) return r;
    if ((r = grow(c, &c->st_out, &c->st_out_cap, n + 16)) != DCZ_OK) return r;
    HIPCHK(c, hipMemcpyAsync(c->st_in, data, n, hipMemcpyHostToDevice, s));
    r = dcz_compress_blocks(c, c->st_in, n, n, c->st_out, n, m.comp_size, m.comp_off, m.len8, m.status, m.total, s);
    if (r != DCZ_OK) return r;
    uint8_t l8[256];
    uint32_t cs = 0;
    int32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(l8, m.len8, 256, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&cs, m.comp_size, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&st, m.status, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (st != DCZ_OK) return st;
    for (int i = 0; i < 256; i++) len_out[i] = l8[i];
    if (cs > cap) return DCZ_E_CAPACITY;
    if (cs) HIPCHK(c, hipMemcpy(out, c->st_out, cs, hipMemcpyDeviceToHost));
    *out_len = cs;
    return launch_check(c);
}

int dcz_decode_block(dcz_ctx* c, const uint8_t* comp, size_t comp_size, const int32_t len[256], uint8_t* out,
                     size_t out_size, int64_t* err_pos) {
    if (!c || !len || (!comp && comp_size) || (!out && out_size)) return DCZ_E_INVALID;
    if (err_pos) *err_pos = -1;
    if (out_size == 0) return DCZ_OK;
    if (out_size > 0x7FFFFFFFull || comp_size > 0xFFFFFFFFull) return DCZ_E_INVALID;
    uint8_t l8[256];
    for (int i = 0; i < 256; i++) {
        if (len[i] < 0 || len[i] > 32) return DCZ_E_BADTABLE;  // CanonicalHuffman.java:106 would throw
        l8[i] = (uint8_t)len[i];
    }
    DeviceGuard dg(c->device);
    hipStream_t s = c->stream;
    Meta m = meta_of(c);
    int r;
    if ((r = grow(c, &c->st_in, &c->st_in_cap, comp_size + 32)) != DCZ_OK) return r;
    if ((r = grow(c, &c->st_out, &c->st_out_cap, out_size + 16)) != DCZ_OK) return r;
    const uint64_t off0 = 0;
    const uint32_t cs = (uint32_t)comp_size, os = (uint32_t)out_size;
    if (comp_size) HIPCHK(c, hipMemcpyAsync(c->st_in, comp, comp_size, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(m.len8, l8, 256, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(m.comp_off, &off0, 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(m.comp_size, &cs, 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(m.orig_size, &os, 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));  // the small host temporaries above live on this stack frame
    r = dcz_decompress_blocks(c, c->st_in, comp_size, m.comp_off, m.comp_size, m.orig_size, m.len8, 1, out_size,
                              c->st_out, m.status, m.errpos, s);
    if (r != DCZ_OK) return r;
    int32_t st = 0;
    int64_t ep = 0;
    HIPCHK(c, hipMemcpyAsync(&st, m.status, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&ep, m.errpos, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (st != DCZ_OK) {
        if (err_pos) *err_pos = ep;
        return st;
    }
    HIPCHK(c, hipMemcpy(out, c->st_out, out_size, hipMemcpyDeviceToHost));
    return launch_check(c);
}

/* ---- batched host-pointer twins (what a host-language service calls once per batch of chunks) ---- */

namespace {
// device columns of one batch of K chunks, carved from c->st_batch
struct BatchCols {
    uint32_t* comp_size;
    uint64_t* comp_off;
    uint32_t* orig_size;
    int32_t* status;
    int64_t* errpos;
    uint64_t* total;
    uint8_t* len;
    uint8_t* sha;
};
size_t batch_bytes(size_t K) { return K * (4 + 8 + 4 + 4 + 8 + 256 + 32) + 64; }
BatchCols batch_at(uint8_t* p, size_t K) {
    BatchCols b;
    b.comp_off = reinterpret_cast<uint64_t*>(p);
    b.errpos = reinterpret_cast<int64_t*>(p + 8 * K);
    b.total = reinterpret_cast<uint64_t*>(p + 16 * K);
    b.comp_size = reinterpret_cast<uint32_t*>(p + 16 * K + 16);
    b.orig_size = b.comp_size + K;
    b.status = reinterpret_cast<int32_t*>(b.orig_size + K);
    b.len = reinterpret_cast<uint8_t*>(b.status + K);
    b.sha = b.len + 256 * K;
    return b;
}
int reserve_batch(dcz_ctx* c, size_t K) {
    if (K <= c->st_batch_K && c->st_batch) return DCZ_OK;
    if (c->st_batch) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->st_batch));
    }
    c->st_batch = nullptr;
    c->st_batch_K = 0;
    const size_t nk = K + K / 8 + 64;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->st_batch), batch_bytes(nk)));
    c->st_batch_K = nk;
    return DCZ_OK;
}
}  // namespace

int dcz_host_register(void* p, size_t n) {
    if (!p || !n) return DCZ_E_INVALID;
    return hipHostRegister(p, n, hipHostRegisterDefault) == hipSuccess ? DCZ_OK : DCZ_E_HIP;
}

int dcz_host_unregister(void* p) {
    if (!p) return DCZ_E_INVALID;
    return hipHostUnregister(p) == hipSuccess ? DCZ_OK : DCZ_E_HIP;
}

void* dcz_ctx_pinned(dcz_ctx* c, int slot, size_t bytes) {
    if (!c || slot < 0 || slot > 1) return nullptr;
    if (bytes <= c->pinned_cap[slot] && c->pinned[slot]) return c->pinned[slot];
    DeviceGuard dg(c->device);
    (void)hipStreamSynchronize(c->stream);  // (a copy from the old buffer may still be in flight)
    if (c->pinned[slot]) (void)hipHostFree(c->pinned[slot]);
    c->pinned[slot] = nullptr;
    c->pinned_cap[slot] = 0;
    const size_t cap = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&c->pinned[slot], cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    c->pinned_cap[slot] = cap;
    return c->pinned[slot];
}

int dcz_compress_host(dcz_ctx* c, const uint8_t* in, size_t n, size_t block_bytes, uint8_t* out, size_t out_cap,
                      uint32_t* comp_size, uint64_t* comp_off, uint8_t* len, int32_t* status, uint64_t* total,
                      uint8_t* sha256) {
    if (!c || (!in && n) || !comp_size || !comp_off || !len || !status || (!out && out_cap)) return DCZ_E_INVALID;
    Geometry g;
    int r = geometry(n, block_bytes, &g);
    if (r != DCZ_OK) return r;
    if (total) *total = 0;
    if (g.K == 0) return DCZ_OK;
    DeviceGuard dg(c->device);
    hipStream_t s = c->stream;
    const size_t K = g.K;
    if ((r = grow(c, &c->st_in, &c->st_in_cap, n + 16)) != DCZ_OK) return r;
    if ((r = grow(c, &c->st_out, &c->st_out_cap, n + 16)) != DCZ_OK) return r;
    if ((r = reserve_batch(c, K)) != DCZ_OK) return r;
    const BatchCols b = batch_at(c->st_batch, c->st_batch_K);
    HIPCHK(c, hipMemcpyAsync(c->st_in, in, n, hipMemcpyHostToDevice, s));
    if (sha256) {
        // ChecksumUtil.computeSha256 per chunk (K5): one lane per chunk, slow by construction (28 ms per GiB at 4 MiB chunks),
        // so it runs on a stream of its own beside K1-K3 and beside the copy of the payload to the host
        HIPCHK(c, hipEventRecord(c->ev[5], s));
        HIPCHK(c, hipStreamWaitEvent(c->aux2, c->ev[5], 0));
        launch_sha256(c->st_in, n, block_bytes, g.K, b.sha, c->aux2);
        HIPCHK(c, hipEventRecord(c->ev[6], c->aux2));
    }
    r = dcz_compress_blocks(c, c->st_in, n, block_bytes, c->st_out, n, b.comp_size, b.comp_off, b.len, b.status, b.total, s);
    if (r != DCZ_OK) {
        if (sha256) (void)hipStreamSynchronize(c->aux2);
        return r;
    }
    uint64_t tot = 0;
    HIPCHK(c, hipMemcpyAsync(comp_size, b.comp_size, 4 * K, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(comp_off, b.comp_off, 8 * K, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(len, b.len, 256 * K, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(status, b.status, 4 * K, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&tot, b.total, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (total) *total = tot;
    int bad = DCZ_OK;
    for (size_t k = 0; k < K && bad == DCZ_OK; k++)
        if (status[k] != DCZ_OK) bad = status[k];
    if (bad == DCZ_OK && tot > out_cap) bad = DCZ_E_CAPACITY;
    if (bad == DCZ_OK && tot) HIPCHK(c, hipMemcpyAsync(out, c->st_out, tot, hipMemcpyDeviceToHost, s));  // beside the digests
    if (sha256) {
        HIPCHK(c, hipStreamWaitEvent(s, c->ev[6], 0));  // (the staging buffers are reused by the next call: always wait)
        if (bad == DCZ_OK) HIPCHK(c, hipMemcpyAsync(sha256, b.sha, 32 * K, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    if (bad != DCZ_OK) return bad;
    return launch_check(c);
}

int dcz_decompress_host(dcz_ctx* c, const uint8_t* comp, size_t comp_bytes, const uint64_t* comp_off,
                        const uint32_t* comp_size, const uint32_t* orig_size, const uint8_t* len, size_t K,
                        size_t out_stride, uint8_t* out, int32_t* status, int64_t* errpos, uint8_t* sha256) {
    if (!c || (!comp && comp_bytes) || !comp_off || !comp_size || !orig_size || !len || !status) return DCZ_E_INVALID;
    if (K == 0) return DCZ_OK;
    if (!out || K > 0x7FFFFFFFull) return DCZ_E_INVALID;
    size_t last = 0;  // bytes of the output range that hold chunks: (K - 1) strides + the last chunk
    for (size_t k = 0; k < K; k++) {
        if (orig_size[k] > out_stride) return DCZ_E_INVALID;
        if (orig_size[k]) last = k * out_stride + orig_size[k];
    }
    DeviceGuard dg(c->device);
    hipStream_t s = c->stream;
    int r;
    if ((r = grow(c, &c->st_in, &c->st_in_cap, comp_bytes + 32)) != DCZ_OK) return r;
    if ((r = grow(c, &c->st_out, &c->st_out_cap, K * out_stride + 16)) != DCZ_OK) return r;
    if ((r = reserve_batch(c, K)) != DCZ_OK) return r;
    const BatchCols b = batch_at(c->st_batch, c->st_batch_K);
    if (comp_bytes) HIPCHK(c, hipMemcpyAsync(c->st_in, comp, comp_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(b.comp_off, comp_off, 8 * K, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(b.comp_size, comp_size, 4 * K, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(b.orig_size, orig_size, 4 * K, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(b.len, len, 256 * K, hipMemcpyHostToDevice, s));
    r = dcz_decompress_blocks(c, c->st_in, comp_bytes, b.comp_off, b.comp_size, b.orig_size, b.len, K, out_stride,
                              c->st_out, b.status, b.errpos, s);
    if (r != DCZ_OK) return r;
    // per-chunk digests of the decoded bytes (verification, CpuCompressionService.java:536-550) when the chunks are
    // contiguous: every chunk but the last fills its stride
    bool contiguous = sha256 != nullptr;
    for (size_t k = 0; contiguous && k + 1 < K; k++) contiguous = orig_size[k] == out_stride;
    if (contiguous) {  // the digests of the decoded chunks are computed while the chunks travel to the host
        HIPCHK(c, hipEventRecord(c->ev[5], s));
        HIPCHK(c, hipStreamWaitEvent(c->aux2, c->ev[5], 0));
        launch_sha256(c->st_out, last, out_stride, (uint32_t)K, b.sha, c->aux2);
        HIPCHK(c, hipEventRecord(c->ev[6], c->aux2));
    }
    HIPCHK(c, hipMemcpyAsync(status, b.status, 4 * K, hipMemcpyDeviceToHost, s));
    if (errpos) HIPCHK(c, hipMemcpyAsync(errpos, b.errpos, 8 * K, hipMemcpyDeviceToHost, s));
    if (last) HIPCHK(c, hipMemcpyAsync(out, c->st_out, last, hipMemcpyDeviceToHost, s));
    if (contiguous) {
        HIPCHK(c, hipStreamWaitEvent(s, c->ev[6], 0));
        HIPCHK(c, hipMemcpyAsync(sha256, b.sha, 32 * K, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    r = launch_check(c);
    if (r != DCZ_OK) return r;
    if (sha256 && !contiguous) return 1;  // decoded, digests not computed (ragged strides): the caller hashes on the host
    return DCZ_OK;
}

int dcz_ctx_set_profiling(dcz_ctx* c, int on) {
    if (!c) return DCZ_E_INVALID;
    c->profiling = on != 0;
    return DCZ_OK;
}

int dcz_ctx_reset_profiling(dcz_ctx* c) {
    if (!c) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    int r = drain_events(c);
    for (int i = 0; i < DCZ_K_COUNT; i++) {
        c->ms[i] = 0;
        c->launches[i] = 0;
    }
    for (auto& v : c->shapes) v = 0;
    return r;
}

int dcz_ctx_launch_shapes(dcz_ctx* c, uint64_t counts[4]) {
    if (!c || !counts) return DCZ_E_INVALID;
    for (int i = 0; i < 4; i++) counts[i] = c->shapes[i];
    return DCZ_OK;
}

int dcz_ctx_kernel_time(dcz_ctx* c, int kernel, double* total_ms, uint64_t* launches) {
    if (!c || kernel < 0 || kernel >= DCZ_K_COUNT) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    int r = drain_events(c);
    if (total_ms) *total_ms = c->ms[kernel];
    if (launches) *launches = c->launches[kernel];
    return r;
}

/* debug / tools only (not declared in include/dcz.h): raw bytes of the decoder workspace after the stream has drained */
int dcz_debug_read_decode_ws(dcz_ctx* c, void* dst, size_t offset, size_t n) {
    if (!c || !c->dws || offset + n > decode_ws_bytes(c->cap_dws_K)) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, c->dws + offset, n, hipMemcpyDeviceToHost));
    return DCZ_OK;
}
size_t dcz_debug_decode_ws_layout(dcz_ctx* c, size_t* split_offset, size_t* split_entries) {
    if (!c) return 0;
    const size_t kb = (c->cap_dws_K + 255) & ~(size_t)255;
    if (split_offset) *split_offset = kb + 256;
    if (split_entries) *split_entries = SPLIT_ENTRIES;
    return decode_ws_bytes(c->cap_dws_K);
}

int dcz_sha256_blocks(dcz_ctx* c, const void* d_in, size_t n, size_t block_bytes, void* d_digests, void* stream) {
    if (!c || (!d_in && n) || block_bytes == 0 || (!d_digests && n)) return DCZ_E_INVALID;
    const size_t K = (n + block_bytes - 1) / block_bytes;
    if (K > 0xFFFFFFFFull) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    launch_sha256(static_cast<const uint8_t*>(d_in), n, block_bytes, (uint32_t)K, static_cast<uint8_t*>(d_digests),
                  stream ? static_cast<hipStream_t>(stream) : c->stream);
    return launch_check(c);
}

int dczu_fill_java_random(dcz_ctx* c, void* d_buf, size_t n, int64_t seed, uint64_t start, void* stream) {
    if (!c || (!d_buf && n) || (start & 3)) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    launch_fill_java_random(static_cast<uint8_t*>(d_buf), n, seed, start,
                            stream ? static_cast<hipStream_t>(stream) : c->stream);
    return launch_check(c);
}

int dczu_fill_text(dcz_ctx* c, void* d_buf, size_t n, uint64_t seed, uint64_t start, void* stream) {
    if (!c || (!d_buf && n)) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    launch_fill_text(static_cast<uint8_t*>(d_buf), n, seed, start, stream ? static_cast<hipStream_t>(stream) : c->stream);
    return launch_check(c);
}

int dczu_fill_lowentropy(dcz_ctx* c, void* d_buf, size_t n, uint64_t seed, uint64_t start, void* stream) {
    if (!c || (!d_buf && n)) return DCZ_E_INVALID;
    DeviceGuard dg(c->device);
    launch_fill_lowentropy(static_cast<uint8_t*>(d_buf), n, seed, start,
                           stream ? static_cast<hipStream_t>(stream) : c->stream);
    return launch_check(c);
}

}  // extern "C"
