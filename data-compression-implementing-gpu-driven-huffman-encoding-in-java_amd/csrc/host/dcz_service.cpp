// dcz_service.cpp -- see dcz_service.h.  Host plumbing only: every Huffman stage is a call into include/dcz.h.
#include "dcz_service.h"

#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <atomic>
#include <exception>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "../../../include/dcz.h"

namespace datacomp {

namespace {

long long now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) {
        s.push_back(d[p[i] >> 4]);
        s.push_back(d[p[i] & 15]);
    }
    return s;
}

std::string base_name(const std::string& path) {
    const size_t k = path.find_last_of('/');
    return k == std::string::npos ? path : path.substr(k + 1);
}

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw IOError(std::string("GPU compression failed: ") + what + ": " + hipGetErrorString(e));
}

void dcz_check(dcz_ctx* c, int st, const char* what) {
    if (st != DCZ_OK)
        throw IOError(std::string("GPU compression failed: ") + what + ": " + dcz_strerror(st) +
                      (st == DCZ_E_HIP ? std::string(" (") + dcz_last_error(c) + ")" : std::string()));
}

struct DevBuf {  // RAII device allocation
    void* p = nullptr;
    explicit DevBuf(size_t n) { hip_check(hipMalloc(&p, n ? n : 16), "hipMalloc"); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

// big-endian DataOutputStream / DataInputStream primitives
struct BeWriter {
    std::vector<uint8_t> b;
    void u32(uint32_t v) { for (int i = 3; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); }
    void i64(int64_t v) { for (int i = 7; i >= 0; i--) b.push_back((uint8_t)((uint64_t)v >> (8 * i))); }
    void i16(int16_t v) { b.push_back((uint8_t)((uint16_t)v >> 8)); b.push_back((uint8_t)v); }
    void raw(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
};

struct BeReader {
    const uint8_t* p;
    size_t n, pos = 0;
    void need(size_t k) const { if (pos + k > n) throw IOError("Unexpected end of header"); }
    uint32_t u32() { need(4); uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | p[pos++]; return v; }
    int64_t i64() { need(8); uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[pos++]; return (int64_t)v; }
    int16_t i16() { need(2); uint16_t v = (uint16_t)((p[pos] << 8) | p[pos + 1]); pos += 2; return (int16_t)v; }
    void raw(uint8_t* out, size_t k) { need(k); std::memcpy(out, p + pos, k); pos += k; }
};


}  // namespace

// ---- StageMetrics ---------------------------------------------------------------------------------------------
void StageMetrics::record(const std::string& stage, long long ns, long long bytes) {
    Acc& a = acc_[stage];
    a.ns += ns;
    a.count += 1;
    a.bytes += bytes;
}

void StageMetrics::merge(const StageMetrics& other) {
    for (auto& kv : other.acc_) {
        Acc& a = acc_[kv.first];
        a.ns += kv.second.ns;
        a.count += kv.second.count;
        a.bytes += kv.second.bytes;
    }
}

long long StageMetrics::time_ns(const std::string& stage) const {
    auto it = acc_.find(stage);
    return it == acc_.end() ? 0 : it->second.ns;
}

std::string StageMetrics::summary() const {
    long long tot = 0;
    for (auto& kv : acc_) tot += kv.second.ns;
    std::ostringstream o;
    o << "Stage Performance Breakdown:\n";
    for (auto& kv : acc_) {
        char line[160];
        std::snprintf(line, sizeof line, "%-25s: %8.2f ms (%5.1f%%) [%lld runs]\n", kv.first.c_str(), kv.second.ns / 1e6,
                      tot ? 100.0 * kv.second.ns / tot : 0.0, kv.second.count);
        o << line;
    }
    return o.str();
}

// ---- SHA-256 (FIPS 180-4) -------------------------------------------------------------------------------------
namespace {
const uint32_t KK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void sha_block(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        const uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + KK[i] + w[i];
        const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
}  // namespace

// one lane per chunk on the GPU (~17 MB/s per lane, asynchronous beside the file I/O): below this many chunks in a batch the
// single-threaded host loop (~0.4 GB/s) is faster
constexpr int64_t kShaGpuMinChunks = 64;

void sha256(const uint8_t* data, size_t n, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const size_t full = n / 64;
    for (size_t i = 0; i < full; i++) sha_block(h, data + 64 * i);
    uint8_t tail[128] = {0};
    const size_t rem = n - 64 * full;
    if (rem) std::memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    const size_t tl = rem < 56 ? 64 : 128;
    const uint64_t bits = (uint64_t)n * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha_block(h, tail);
    if (tl == 128) sha_block(h, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out[4 * i + 3] = (uint8_t)h[i];
    }
}

// ---- CompressionHeader ------------------------------------------------------------------------------------------
std::vector<uint8_t> CompressionHeader::write() const {  // CompressionHeader.java:51-85
    BeWriter w;
    w.u32(MAGIC);
    w.u32(VERSION);
    w.u32((uint32_t)originalFileName.size());
    w.raw(reinterpret_cast<const uint8_t*>(originalFileName.data()), originalFileName.size());
    w.i64(originalFileSize);
    w.i64(originalTimestamp);
    w.u32((uint32_t)chunkSizeBytes);
    w.raw(globalChecksum, 32);
    w.u32((uint32_t)chunks.size());
    for (const ChunkMetadata& c : chunks) {
        w.u32((uint32_t)c.chunkIndex);
        w.i64(c.originalOffset);
        w.u32(c.originalSize);
        w.i64(c.compressedOffset);
        w.u32(c.compressedSize);
        w.raw(c.sha256, 32);
        for (int i = 0; i < 256; i++) w.i16(c.codeLengths[i]);
    }
    return w.b;
}

CompressionHeader CompressionHeader::read(const uint8_t* p, size_t n) {  // CompressionHeader.java:90-144
    BeReader r{p, n};
    if (r.u32() != MAGIC) throw IOError("Invalid file format: bad magic number");
    const uint32_t version = r.u32();
    if (version != VERSION) throw IOError("Unsupported version: " + std::to_string(version));
    CompressionHeader h;
    const uint32_t nameLen = r.u32();
    if (nameLen > n) throw IOError("Invalid file format: bad name length");
    h.originalFileName.resize(nameLen);
    r.raw(reinterpret_cast<uint8_t*>(&h.originalFileName[0]), nameLen);
    h.originalFileSize = r.i64();
    h.originalTimestamp = r.i64();
    h.chunkSizeBytes = (int32_t)r.u32();
    r.raw(h.globalChecksum, 32);
    const uint32_t k = r.u32();
    if ((size_t)k * 572 > n) throw IOError("Unexpected end of header");
    h.chunks.resize(k);
    for (uint32_t i = 0; i < k; i++) {
        ChunkMetadata& c = h.chunks[i];
        c.chunkIndex = (int32_t)r.u32();
        c.originalOffset = r.i64();
        c.originalSize = r.u32();
        c.compressedOffset = r.i64();
        c.compressedSize = r.u32();
        r.raw(c.sha256, 32);
        for (int j = 0; j < 256; j++) c.codeLengths[j] = r.i16();
    }
    return h;
}

// ---- HipFrequencyService ------------------------------------------------------------------------------------------
HipFrequencyService::HipFrequencyService(int device) {
    if (dcz_device_count() > 0 && dcz_ctx_create(device, &ctx_) != DCZ_OK) ctx_ = nullptr;
}
HipFrequencyService::~HipFrequencyService() { dcz_ctx_destroy(ctx_); }

std::vector<int64_t> HipFrequencyService::computeHistogram(const uint8_t* data, size_t offset, size_t length) {
    if (!ctx_) throw std::runtime_error("HIP device not available");
    std::vector<int64_t> h(256);
    const int st = dcz_histogram(ctx_, data, offset, length, h.data());
    if (st != DCZ_OK) throw std::runtime_error(std::string("HIP histogram failed: ") + dcz_strerror(st));
    return h;
}

// Chunk-parallel host work with the reference's worker count max(2, min(nproc, 8)) (CpuCompressionService.java:42-44).
static void parallel_chunks(int64_t n, const std::function<void(int64_t)>& fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int64_t workers = std::min<int64_t>(n, std::max(2u, std::min(hw ? hw : 2u, 8u)));
    if (workers <= 1) {
        for (int64_t k = 0; k < n; k++) fn(k);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    for (int64_t w = 0; w < workers; w++)
        pool.emplace_back([&]() {
            for (int64_t k = next.fetch_add(1); k < n; k = next.fetch_add(1)) fn(k);
        });
    for (auto& t : pool) t.join();
}

// ---- HipCompressionService ----------------------------------------------------------------------------------------
HipCompressionService::HipCompressionService(int chunkSizeMB, int device) : device_(device) {
    if (chunkSizeMB <= 0 || chunkSizeMB > 2047) throw std::invalid_argument("chunk size must be 1..2047 MB");
    chunkBytes_ = (int64_t)chunkSizeMB * 1024 * 1024;
    if (const char* e = std::getenv("DCZ_BATCH_MB")) {  // bytes per pipeline slot (tests use it to force many batches)
        const long mb = std::atol(e);
        if (mb > 0) batchBytes_ = (size_t)mb << 20;
    }
    if (batchBytes_ < (size_t)chunkBytes_) batchBytes_ = (size_t)chunkBytes_;
    if (dcz_device_count() > 0 && dcz_ctx_create(device, &ctx_) != DCZ_OK) ctx_ = nullptr;
}

HipCompressionService::~HipCompressionService() { close(); }

void HipCompressionService::close() {
    dcz_ctx_destroy(ctx_);
    ctx_ = nullptr;
}

void HipCompressionService::resumeCompression(const std::string&, const std::string&, int, const Progress&) {
    throw std::logic_error("Resume not yet implemented");  // CpuCompressionService.java:636-641
}

void HipCompressionService::compress(const std::string& inputPath, const std::string& outputPath,
                                     const Progress& progress) {
    if (!ctx_) throw IOError("GPU compression failed: HIP device not available");
    metrics_ = StageMetrics();
    hip_check(hipSetDevice(device_), "hipSetDevice");
    struct stat st;
    if (::stat(inputPath.c_str(), &st) != 0) throw IOError("Cannot stat " + inputPath);
    const int64_t size = st.st_size;
    const int64_t cb = chunkBytes_;
    const int64_t allChunks = (size + cb - 1) / cb;
    // the chunks this call handles: all of them, or this instance's shard
    const int64_t firstChunk = shard_ ? std::min(shardFirst_, allChunks) : 0;
    const int64_t numChunks = shard_ ? std::max<int64_t>(0, std::min(shardCount_, allChunks - firstChunk)) : allChunks;
    std::ifstream fin(inputPath, std::ios::binary);
    std::ofstream fout(outputPath, std::ios::binary | std::ios::trunc);
    if (!fin || !fout) throw IOError("Cannot open input/output file");

    CompressionHeader header;
    header.originalFileName = base_name(inputPath);
    header.originalFileSize = size;
    header.originalTimestamp = (int64_t)st.st_mtim.tv_sec * 1000 + st.st_mtim.tv_nsec / 1000000;  // mtime in ms
    header.chunkSizeBytes = (int32_t)cb;
    std::vector<uint8_t> digests;  // 32 bytes per chunk, in index order (CpuCompressionService.java:106-109)

    // Streaming pipeline (the reference's GPU service keeps <= 2 chunks in flight,
    // GpuCompressionService.java:232-320): two slots, each with its own context + stream, pinned host buffers and
    // device buffers.  While the GPU works on batch i (H2D, K5, K1-K3, D2H of the small arrays -- all asynchronous on
    // the slot's stream), the host finishes batch i-1 (payload D2H, ordered write) and reads batch i+1.
    const int64_t perBatch = std::max<int64_t>(1, (int64_t)batchBytes_ / cb);
    const size_t slotBytes = (size_t)std::min<int64_t>(std::max<int64_t>(0, size - firstChunk * cb), perBatch * cb);
    fin.seekg((std::streamoff)(firstChunk * cb));
    const size_t slotChunks = (size_t)std::min<int64_t>(numChunks, perBatch);
    struct Slot {
        dcz_ctx* ctx = nullptr;
        bool own_ctx = false;
        hipStream_t stream = nullptr;
        uint8_t *hin = nullptr, *hout = nullptr, *hsmall = nullptr;  // pinned
        void *din = nullptr, *dout = nullptr, *dsize = nullptr, *doff = nullptr, *dlen = nullptr, *dstat = nullptr,
             *dtot = nullptr, *ddig = nullptr;
        int64_t c0 = 0, K = 0, bytes = 0;
        bool busy = false, gpu_sha = false;
    } slots[2];
    // layout of the pinned "small" buffer of a slot: sizes u32[K] | status i32[K] | lens u8[K*256] | digests u8[K*32] | total u64
    const size_t smallBytes = slotChunks * (4 + 4 + 256 + 32) + 8;
    auto release = [&]() {
        for (Slot& sl : slots) {
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
            if (sl.hin) (void)hipHostFree(sl.hin);
            if (sl.hout) (void)hipHostFree(sl.hout);
            if (sl.hsmall) (void)hipHostFree(sl.hsmall);
            for (void* p : {sl.din, sl.dout, sl.dsize, sl.doff, sl.dlen, sl.dstat, sl.dtot, sl.ddig})
                if (p) (void)hipFree(p);
            if (sl.own_ctx) dcz_ctx_destroy(sl.ctx);
            sl = Slot();
        }
    };
    int64_t compOffset = 0, done = 0;
    const int nslots = (numChunks > perBatch) ? 2 : 1;
    try {
        for (int i = 0; i < nslots; i++) {
            Slot& sl = slots[i];
            if (i == 0) sl.ctx = ctx_;
            else {
                if (dcz_ctx_create(device_, &sl.ctx) != DCZ_OK) throw IOError("GPU compression failed: cannot create a second context");
                sl.own_ctx = true;
            }
            hip_check(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking), "hipStreamCreate");
            hip_check(hipHostMalloc(reinterpret_cast<void**>(&sl.hin), slotBytes ? slotBytes : 16, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc(reinterpret_cast<void**>(&sl.hout), slotBytes + 16, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc(reinterpret_cast<void**>(&sl.hsmall), smallBytes, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipMalloc(&sl.din, slotBytes ? slotBytes : 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dout, slotBytes + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dsize, slotChunks * 4 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.doff, slotChunks * 8 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dlen, slotChunks * 256 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dstat, slotChunks * 4 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dtot, 8), "hipMalloc");
            hip_check(hipMalloc(&sl.ddig, slotChunks * 32 + 16), "hipMalloc");
        }
        // completes a slot's batch in file order: wait, fetch the payload, write it, record the metadata
        auto finish = [&](Slot& sl) {
            if (!sl.busy) return;
            long long t0 = now_ns();
            hip_check(hipStreamSynchronize(sl.stream), "sync");
            const size_t K = (size_t)sl.K;
            const uint32_t* sizes = reinterpret_cast<const uint32_t*>(sl.hsmall);
            const int32_t* stat = reinterpret_cast<const int32_t*>(sl.hsmall + K * 4);
            const uint8_t* lens = sl.hsmall + K * 8;
            const uint8_t* dig = sl.hsmall + K * 8 + K * 256;
            uint64_t total = 0;
            std::memcpy(&total, sl.hsmall + K * (8 + 256 + 32), 8);
            for (size_t k = 0; k < K; k++)
                if (stat[k] != DCZ_OK)
                    throw IOError("GPU compression failed: chunk " + std::to_string(sl.c0 + (int64_t)k) + ": " + dcz_strerror(stat[k]));
            if (total) {
                hip_check(hipMemcpyAsync(sl.hout, sl.dout, (size_t)total, hipMemcpyDeviceToHost, sl.stream), "D2H payload");
                hip_check(hipStreamSynchronize(sl.stream), "sync");
            }
            metrics_.record("Encoding", now_ns() - t0, sl.bytes);
            t0 = now_ns();
            fout.write(reinterpret_cast<const char*>(sl.hout), (std::streamsize)total);
            metrics_.record("File I/O", now_ns() - t0, (long long)total);
            digests.insert(digests.end(), dig, dig + K * 32);  // from K5 or from the host loop below, in file order
            for (size_t k = 0; k < K; k++) {
                const int64_t ci = sl.c0 + (int64_t)k;
                ChunkMetadata m;
                m.chunkIndex = (int32_t)ci;
                m.originalOffset = ci * cb;
                m.originalSize = (uint32_t)std::min<int64_t>(cb, size - ci * cb);
                m.compressedOffset = compOffset;
                m.compressedSize = sizes[k];
                std::memcpy(m.sha256, &digests[(size_t)(ci - firstChunk) * 32], 32);
                for (int i = 0; i < 256; i++) m.codeLengths[i] = (int16_t)lens[k * 256 + (size_t)i];
                header.chunks.push_back(m);
                compOffset += sizes[k];
                done++;
                if (progress) progress((double)done / (double)numChunks);  // CpuCompressionService.java:111-114
            }
            sl.busy = false;
        };
        int which = 0;
        for (int64_t c0 = firstChunk; c0 < firstChunk + numChunks; c0 += perBatch, which ^= (nslots - 1)) {
            Slot& sl = slots[which];
            finish(sl);  // batch i-2 (same slot) must be out before its buffers are reused; keeps the file order
            const int64_t c1 = std::min(firstChunk + numChunks, c0 + perBatch);
            sl.c0 = c0;
            sl.K = c1 - c0;
            sl.bytes = std::min<int64_t>(size - c0 * cb, (c1 - c0) * cb);
            long long t0 = now_ns();
            if (!fin.read(reinterpret_cast<char*>(sl.hin), sl.bytes)) throw IOError("Cannot read " + inputPath);
            metrics_.record("File I/O", now_ns() - t0, sl.bytes);
            const size_t K = (size_t)sl.K;
            hip_check(hipMemcpyAsync(sl.din, sl.hin, (size_t)sl.bytes, hipMemcpyHostToDevice, sl.stream), "H2D");
            t0 = now_ns();
            sl.gpu_sha = sl.K >= kShaGpuMinChunks;
            if (sl.gpu_sha) {  // K5: one lane per chunk on the device-resident batch
                dcz_check(sl.ctx, dcz_sha256_blocks(sl.ctx, sl.din, (size_t)sl.bytes, (size_t)cb, sl.ddig, sl.stream), "dcz_sha256_blocks");
                hip_check(hipMemcpyAsync(sl.hsmall + K * 8 + K * 256, sl.ddig, K * 32, hipMemcpyDeviceToHost, sl.stream), "D2H digests");
            }
            dcz_check(sl.ctx, dcz_compress_blocks(sl.ctx, sl.din, (size_t)sl.bytes, (size_t)cb, sl.dout, (size_t)sl.bytes + 16,
                                                  static_cast<uint32_t*>(sl.dsize), static_cast<uint64_t*>(sl.doff),
                                                  static_cast<uint8_t*>(sl.dlen), static_cast<int32_t*>(sl.dstat),
                                                  static_cast<uint64_t*>(sl.dtot), sl.stream),
                      "dcz_compress_blocks");
            hip_check(hipMemcpyAsync(sl.hsmall, sl.dsize, K * 4, hipMemcpyDeviceToHost, sl.stream), "D2H sizes");
            hip_check(hipMemcpyAsync(sl.hsmall + K * 4, sl.dstat, K * 4, hipMemcpyDeviceToHost, sl.stream), "D2H status");
            hip_check(hipMemcpyAsync(sl.hsmall + K * 8, sl.dlen, K * 256, hipMemcpyDeviceToHost, sl.stream), "D2H lens");
            hip_check(hipMemcpyAsync(sl.hsmall + K * (8 + 256 + 32), sl.dtot, 8, hipMemcpyDeviceToHost, sl.stream), "D2H total");
            sl.busy = true;
            if (!sl.gpu_sha) {  // few chunks: host threads hash them while the GPU encodes (the pinned input stays valid)
                parallel_chunks(sl.K, [&](int64_t k) {
                    const int64_t off = k * cb, len = std::min<int64_t>(cb, sl.bytes - off);
                    sha256(sl.hin + off, (size_t)len, sl.hsmall + K * 8 + K * 256 + (size_t)k * 32);
                });
            }
            metrics_.record("Checksum Computation", now_ns() - t0, sl.bytes);
        }
        finish(slots[which]);                  // the older of the two outstanding batches first
        finish(slots[which ^ (nslots - 1)]);
    } catch (...) {
        (void)hipDeviceSynchronize();
        release();
        throw;
    }
    release();
    if (shard_) {  // payloads only: the coordinator concatenates the shards and writes the footer
        shardChunks_ = header.chunks;
        if (!fout) throw IOError("Cannot write " + outputPath);
        return;
    }
    const long long t0 = now_ns();
    sha256(digests.data(), digests.size(), header.globalChecksum);  // digest of digests (:106-109, :126)
    const int64_t footerStart = compOffset;
    const std::vector<uint8_t> hb = header.write();
    fout.write(reinterpret_cast<const char*>(hb.data()), (std::streamsize)hb.size());
    BeWriter ptr;
    ptr.i64(footerStart);  // raf.writeLong(footerStart), CpuCompressionService.java:174
    fout.write(reinterpret_cast<const char*>(ptr.b.data()), 8);
    if (!fout) throw IOError("Cannot write " + outputPath);
    metrics_.record("Header Write", now_ns() - t0, 0);
}

void HipCompressionService::setShard(int64_t firstChunk, int64_t chunkCount) {
    shard_ = true;
    shardFirst_ = std::max<int64_t>(0, firstChunk);
    shardCount_ = std::max<int64_t>(0, chunkCount);
}

void HipCompressionService::decodeAll(const std::string& path, const ChunkSink& sink, const Progress& progress,
                                      CompressionHeader* header_out) {
    if (!ctx_) throw IOError("GPU decompression failed: HIP device not available");
    metrics_ = StageMetrics();
    hip_check(hipSetDevice(device_), "hipSetDevice");
    long long t0 = now_ns();
    std::ifstream fin(path, std::ios::binary);
    if (!fin) throw IOError("Cannot open " + path);
    fin.seekg(0, std::ios::end);
    const size_t total = (size_t)fin.tellg();
    auto read_at = [&](size_t off, uint8_t* dst, size_t n) {
        fin.clear();
        fin.seekg((std::streamoff)off);
        if (n && !fin.read(reinterpret_cast<char*>(dst), (std::streamsize)n)) throw IOError("Cannot read " + path);
    };
    // probe order of CpuCompressionService.decompress: header-first from the first <= 4096 bytes (:338-358), else the
    // footer through the trailing pointer with the 0 <= ptr < size-8 check (:366-388).  Only the metadata is read here;
    // the payload is streamed batch by batch below.
    CompressionHeader header;
    size_t dataStart = 0;
    bool parsed = false;
    {
        std::vector<uint8_t> probe(std::min<size_t>(64 * 1024, total), 0);
        read_at(0, probe.data(), std::min<size_t>(4096, probe.size()));
        try {
            header = CompressionHeader::read(probe.data(), probe.size());
            size_t sum = 0;
            for (auto& c : header.chunks) sum += c.compressedSize;
            if (sum > total) throw IOError("Invalid file format: compressed sizes exceed the file");
            dataStart = total - sum;
            parsed = true;
        } catch (const IOError&) {
        }
    }
    if (!parsed) {
        if (total < 8) throw IOError("Invalid file format: file too small");
        uint8_t p8[8];
        read_at(total - 8, p8, 8);
        BeReader r{p8, 8};
        const int64_t ptr = r.i64();
        if (ptr < 0 || (uint64_t)ptr >= total - 8) throw IOError("Invalid footer position: " + std::to_string(ptr));
        std::vector<uint8_t> footer(total - 8 - (size_t)ptr);
        read_at((size_t)ptr, footer.data(), footer.size());
        header = CompressionHeader::read(footer.data(), footer.size());
        dataStart = 0;
    }
    metrics_.record("File I/O", now_ns() - t0, 0);
    // The metadata is untrusted: sizes that cannot belong to this file are rejected before they size any buffer
    // (the reference is protected by Java's int / byte[] semantics: CompressionHeader.java:71-84).
    for (auto& c : header.chunks) {
        if (header.chunkSizeBytes <= 0 || (int64_t)c.originalSize > (int64_t)header.chunkSizeBytes ||
            (uint64_t)c.compressedSize > total || c.compressedOffset < 0 || (uint64_t)c.compressedOffset > total)
            throw IOError("Chunk decompression failed: metadata of chunk " + std::to_string(c.chunkIndex) +
                          " does not fit the file");
    }
    const size_t allChunks = header.chunks.size();
    const size_t firstChunk = shard_ ? std::min<size_t>((size_t)shardFirst_, allChunks) : 0;
    const size_t numChunks = shard_ ? std::min<size_t>((size_t)shardCount_, allChunks - firstChunk) : allChunks;
    const size_t lastChunk = firstChunk + numChunks;  // one past this call's last chunk
    const size_t per = std::max<size_t>(1, batchBytes_ / (size_t)std::max(1, header.chunkSizeBytes));
    // capacities of a slot: the largest batch in chunks, compressed bytes and decoded bytes (stride * chunks)
    size_t capK = 0, capComp = 0, capOut = 0;
    for (size_t c0 = firstChunk; c0 < lastChunk; c0 += per) {
        const size_t c1 = std::min(lastChunk, c0 + per);
        size_t comp = 0, stride = 16;
        for (size_t k = c0; k < c1; k++) {
            comp += header.chunks[k].compressedSize;
            stride = std::max<size_t>(stride, header.chunks[k].originalSize);
        }
        stride = (stride + 15) & ~(size_t)15;
        capK = std::max(capK, c1 - c0);
        capComp = std::max(capComp, comp);
        capOut = std::max(capOut, (c1 - c0) * stride);
    }
    struct Slot {
        dcz_ctx* ctx = nullptr;
        bool own_ctx = false;
        hipStream_t stream = nullptr;
        uint8_t *hcomp = nullptr, *hout = nullptr, *hmeta = nullptr, *hres = nullptr;  // pinned
        void *dcomp = nullptr, *doff = nullptr, *dsize = nullptr, *dorig = nullptr, *dlen = nullptr, *dout = nullptr,
             *dstat = nullptr, *derr = nullptr, *ddig = nullptr;
        size_t c0 = 0, K = 0, stride = 0;
        bool busy = false, gpu_sha = false;
    } slots[2];
    // pinned metadata of a slot: offs u64[K] | sizes u32[K] | origs u32[K] | lens u8[K*256]; results: status i32[K] |
    // errpos i64[K] | digests u8[K*32]
    const size_t metaBytes = capK * (8 + 4 + 4 + 256) + 16, resBytes = capK * (4 + 8 + 32) + 16;
    auto release = [&]() {
        for (Slot& sl : slots) {
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
            for (uint8_t* p : {sl.hcomp, sl.hout, sl.hmeta, sl.hres})
                if (p) (void)hipHostFree(p);
            for (void* p : {sl.dcomp, sl.doff, sl.dsize, sl.dorig, sl.dlen, sl.dout, sl.dstat, sl.derr, sl.ddig})
                if (p) (void)hipFree(p);
            if (sl.own_ctx) dcz_ctx_destroy(sl.ctx);
            sl = Slot();
        }
    };
    size_t done = 0;
    const int nslots = (numChunks > per) ? 2 : 1;
    try {
        for (int i = 0; i < nslots && numChunks; i++) {
            Slot& sl = slots[i];
            if (i == 0) sl.ctx = ctx_;
            else {
                if (dcz_ctx_create(device_, &sl.ctx) != DCZ_OK) throw IOError("GPU decompression failed: cannot create a second context");
                sl.own_ctx = true;
            }
            hip_check(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking), "hipStreamCreate");
            hip_check(hipHostMalloc(reinterpret_cast<void**>(&sl.hcomp), capComp + 16, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc(reinterpret_cast<void**>(&sl.hout), capOut + 16, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc(reinterpret_cast<void**>(&sl.hmeta), metaBytes, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipHostMalloc(reinterpret_cast<void**>(&sl.hres), resBytes, hipHostMallocDefault), "hipHostMalloc");
            hip_check(hipMalloc(&sl.dcomp, capComp + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.doff, capK * 8 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dsize, capK * 4 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dorig, capK * 4 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dlen, capK * 256 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dout, capOut + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.dstat, capK * 4 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.derr, capK * 8 + 16), "hipMalloc");
            hip_check(hipMalloc(&sl.ddig, capK * 32 + 16), "hipMalloc");
        }
        // completes a slot's batch in file order: wait, check, verify the checksums, hand the chunks to the sink
        auto finish = [&](Slot& sl) {
            if (!sl.busy) return;
            long long t1 = now_ns();
            hip_check(hipStreamSynchronize(sl.stream), "sync");
            const size_t K = sl.K, stride = sl.stride;
            const int32_t* stat = reinterpret_cast<const int32_t*>(sl.hres);
            uint8_t* dig = sl.hres + K * 12;
            for (size_t k = 0; k < K; k++) {
                int64_t ep;
                std::memcpy(&ep, sl.hres + K * 4 + k * 8, 8);
                if (stat[k] == DCZ_E_BADSTREAM)  // TableBasedHuffmanDecoder.java:109-111 wrapped by CpuCompressionService.java:469-471
                    throw IOError("Chunk decompression failed: Huffman decode error at position " + std::to_string(ep));
                if (stat[k] != DCZ_OK) throw IOError(std::string("Chunk decompression failed: ") + dcz_strerror(stat[k]));
            }
            metrics_.record("Decoding", now_ns() - t1, (long long)(K * stride));
            t1 = now_ns();
            if (!sl.gpu_sha) {  // host threads, one chunk each
                parallel_chunks((int64_t)K, [&](int64_t k) {
                    sha256(sl.hout + (size_t)k * stride, (size_t)header.chunks[sl.c0 + (size_t)k].originalSize, dig + (size_t)k * 32);
                });
            }
            for (size_t k = 0; k < K; k++) {
                const ChunkMetadata& c = header.chunks[sl.c0 + k];
                if (std::memcmp(dig + k * 32, c.sha256, 32) != 0) {  // CpuCompressionService.java:536-550
                    std::ostringstream o;
                    o << "Checksum mismatch in chunk " << c.chunkIndex << ":\n  Expected: " << hex(c.sha256, 32)
                      << "\n  Actual:   " << hex(dig + k * 32, 32) << "\n  Chunk size: " << c.originalSize
                      << " bytes\n  Compressed size: " << c.compressedSize << " bytes\n  Compressed offset: "
                      << c.compressedOffset;
                    throw IOError(o.str());
                }
            }
            metrics_.record("Checksum Verification", now_ns() - t1, (long long)(K * stride));
            t1 = now_ns();
            for (size_t k = 0; k < K; k++) {
                sink(sl.hout + k * stride, header.chunks[sl.c0 + k].originalSize, header.chunks[sl.c0 + k]);
                done++;
                if (progress) progress((double)done / (double)numChunks);
            }
            metrics_.record("File I/O", now_ns() - t1, (long long)(K * stride));
            sl.busy = false;
        };
        int which = 0;
        for (size_t c0 = firstChunk; c0 < lastChunk; c0 += per, which ^= (nslots - 1)) {
            Slot& sl = slots[which];
            finish(sl);
            const size_t c1 = std::min(lastChunk, c0 + per), K = c1 - c0;
            sl.c0 = c0;
            sl.K = K;
            t0 = now_ns();
            uint64_t* offs = reinterpret_cast<uint64_t*>(sl.hmeta);
            uint32_t* sizes = reinterpret_cast<uint32_t*>(sl.hmeta + K * 8);
            uint32_t* origs = reinterpret_cast<uint32_t*>(sl.hmeta + K * 12);
            uint8_t* lens = sl.hmeta + K * 16;
            size_t stride = 16, comp = 0;
            bool contiguous_in_file = true;
            for (size_t k = 0; k < K; k++) {
                const ChunkMetadata& c = header.chunks[c0 + k];
                const size_t beg = dataStart + (size_t)c.compressedOffset;
                if (c.compressedOffset < 0 || beg + c.compressedSize > total)
                    throw IOError("Chunk decompression failed: truncated payload in chunk " + std::to_string(c.chunkIndex));
                if (k > 0 && c.compressedOffset != header.chunks[c0 + k - 1].compressedOffset + (int64_t)header.chunks[c0 + k - 1].compressedSize)
                    contiguous_in_file = false;
                offs[k] = comp;
                sizes[k] = c.compressedSize;
                origs[k] = c.originalSize;
                comp += c.compressedSize;
                for (int i = 0; i < 256; i++) {
                    if (c.codeLengths[i] < 0 || c.codeLengths[i] > 32) throw IOError("Chunk decompression failed: bad code length table");
                    lens[k * 256 + (size_t)i] = (uint8_t)c.codeLengths[i];
                }
                stride = std::max<size_t>(stride, c.originalSize);
            }
            stride = (stride + 15) & ~(size_t)15;
            sl.stride = stride;
            if (contiguous_in_file) {  // what every writer of this format produces: one read per batch
                read_at(dataStart + (size_t)header.chunks[c0].compressedOffset, sl.hcomp, comp);
            } else {
                for (size_t k = 0; k < K; k++)
                    read_at(dataStart + (size_t)header.chunks[c0 + k].compressedOffset, sl.hcomp + offs[k], sizes[k]);
            }
            std::memset(sl.hcomp + comp, 0, 16);
            metrics_.record("File I/O", now_ns() - t0, (long long)comp);
            hip_check(hipMemcpyAsync(sl.dcomp, sl.hcomp, comp + 16, hipMemcpyHostToDevice, sl.stream), "H2D payload");
            hip_check(hipMemcpyAsync(sl.doff, offs, K * 8, hipMemcpyHostToDevice, sl.stream), "H2D");
            hip_check(hipMemcpyAsync(sl.dsize, sizes, K * 4, hipMemcpyHostToDevice, sl.stream), "H2D");
            hip_check(hipMemcpyAsync(sl.dorig, origs, K * 4, hipMemcpyHostToDevice, sl.stream), "H2D");
            hip_check(hipMemcpyAsync(sl.dlen, lens, K * 256, hipMemcpyHostToDevice, sl.stream), "H2D");
            dcz_check(sl.ctx, dcz_decompress_blocks(sl.ctx, sl.dcomp, comp + 16, static_cast<uint64_t*>(sl.doff),
                                                    static_cast<uint32_t*>(sl.dsize), static_cast<uint32_t*>(sl.dorig),
                                                    static_cast<uint8_t*>(sl.dlen), K, stride, sl.dout,
                                                    static_cast<int32_t*>(sl.dstat), static_cast<int64_t*>(sl.derr), sl.stream),
                      "dcz_decompress_blocks");
            // K5 when the decoded chunks are contiguous on the device (all but the last fill the stride) and numerous
            bool dense = (int64_t)K >= kShaGpuMinChunks;
            for (size_t k = 0; dense && k + 1 < K; k++) dense = header.chunks[c0 + k].originalSize == (int64_t)stride;
            sl.gpu_sha = dense && header.chunks[c0 + K - 1].originalSize <= (int64_t)stride;
            if (sl.gpu_sha) {
                const size_t n_dec = (K - 1) * stride + (size_t)header.chunks[c0 + K - 1].originalSize;
                dcz_check(sl.ctx, dcz_sha256_blocks(sl.ctx, sl.dout, n_dec, stride, sl.ddig, sl.stream), "dcz_sha256_blocks");
                hip_check(hipMemcpyAsync(sl.hres + K * 12, sl.ddig, K * 32, hipMemcpyDeviceToHost, sl.stream), "D2H digests");
            }
            hip_check(hipMemcpyAsync(sl.hres, sl.dstat, K * 4, hipMemcpyDeviceToHost, sl.stream), "D2H");
            hip_check(hipMemcpyAsync(sl.hres + K * 4, sl.derr, K * 8, hipMemcpyDeviceToHost, sl.stream), "D2H");
            hip_check(hipMemcpyAsync(sl.hout, sl.dout, K * stride, hipMemcpyDeviceToHost, sl.stream), "D2H decoded");
            sl.busy = true;
        }
        finish(slots[which]);
        finish(slots[which ^ (nslots - 1)]);
    } catch (...) {
        (void)hipDeviceSynchronize();
        release();
        throw;
    }
    release();
    if (header_out) *header_out = header;
}

void HipCompressionService::decompress(const std::string& inputPath, const std::string& outputPath,
                                       const Progress& progress) {
    std::ofstream fout(outputPath, std::ios::binary | std::ios::trunc);
    if (!fout) throw IOError("Cannot open " + outputPath);
    decodeAll(inputPath,
              [&](const uint8_t* p, size_t n, const ChunkMetadata&) { fout.write(reinterpret_cast<const char*>(p), (std::streamsize)n); },
              progress, nullptr);
    if (!fout) throw IOError("Cannot write " + outputPath);
}

bool HipCompressionService::verifyIntegrity(const std::string& compressedPath) {
    // A real verification (the reference only scans the last 64 KiB for a header: CpuCompressionService.java:652-694).
    try {
        CompressionHeader h;
        decodeAll(compressedPath, [](const uint8_t*, size_t, const ChunkMetadata&) {}, {}, &h);
        std::vector<uint8_t> dig;
        for (auto& c : h.chunks) dig.insert(dig.end(), c.sha256, c.sha256 + 32);
        uint8_t g[32];
        sha256(dig.data(), dig.size(), g);
        return std::memcmp(g, h.globalChecksum, 32) == 0;
    } catch (const std::exception&) {
        return false;
    }
}

// ---- sharded file -> file (one thread, context and pipeline per device) ---------------------------------------------
std::vector<std::pair<int64_t, int64_t>> planShards(int64_t numChunks, int gpus) {
    std::vector<std::pair<int64_t, int64_t>> r;
    if (gpus < 1) gpus = 1;
    const int64_t per = (numChunks + gpus - 1) / gpus;
    for (int g = 0; g < gpus; g++) {
        const int64_t first = std::min<int64_t>(numChunks, (int64_t)g * per);
        r.emplace_back(first, std::min<int64_t>(per, numChunks - first));
    }
    return r;
}

namespace {
struct SharedProgress {
    std::mutex mu;
    std::vector<double> frac;
    std::vector<int64_t> weight;
    const Progress* out = nullptr;
    void update(size_t r, double p) {
        if (!out || !*out) return;
        std::lock_guard<std::mutex> lk(mu);
        frac[r] = p;
        double done = 0, all = 0;
        for (size_t i = 0; i < frac.size(); i++) {
            done += frac[i] * (double)weight[i];
            all += (double)weight[i];
        }
        (*out)(all > 0 ? done / all : 1.0);  // monotone in [0, 1], one caller at a time
    }
};

template <class Fn>
void run_ranks(size_t n, Fn fn) {
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(n);
    for (size_t r = 0; r < n; r++)
        th.emplace_back([&, r] {
            try {
                fn(r);
            } catch (...) {
                err[r] = std::current_exception();
            }
        });
    for (auto& t : th) t.join();
    for (auto& e : err)
        if (e) std::rethrow_exception(e);
}
}  // namespace

void compressSharded(const std::string& inputPath, const std::string& outputPath, int chunkSizeMB, int gpus,
                     const Progress& progress, StageMetrics* metrics) {
    if (gpus < 1 || gpus > dcz_device_count()) throw IOError("GPU compression failed: " + std::to_string(gpus) + " gfx950 devices are not available");
    struct stat st;
    if (::stat(inputPath.c_str(), &st) != 0) throw IOError("Cannot stat " + inputPath);
    const int64_t size = st.st_size, cb = (int64_t)chunkSizeMB * 1024 * 1024;
    const int64_t numChunks = (size + cb - 1) / cb;
    const auto plan = planShards(numChunks, gpus);
    std::vector<std::unique_ptr<HipCompressionService>> svc(plan.size());
    SharedProgress sp;
    sp.frac.assign(plan.size(), 0.0);
    for (auto& pr : plan) sp.weight.push_back(pr.second);
    sp.out = &progress;
    auto part = [&](size_t r) { return outputPath + ".part" + std::to_string(r); };
    try {
        run_ranks(plan.size(), [&](size_t r) {
            svc[r].reset(new HipCompressionService(chunkSizeMB, (int)r));
            if (!svc[r]->isAvailable()) throw IOError("GPU compression failed: device " + std::to_string(r) + " not available");
            svc[r]->setShard(plan[r].first, plan[r].second);
            svc[r]->compress(inputPath, part(r), [&, r](double p) { sp.update(r, p); });
        });
        // the exchange step: every rank's per-chunk sizes, in rank order -> compressedOffset column and write bases
        const long long t0 = now_ns();
        CompressionHeader header;
        header.originalFileName = base_name(inputPath);
        header.originalFileSize = size;
        header.originalTimestamp = (int64_t)st.st_mtim.tv_sec * 1000 + st.st_mtim.tv_nsec / 1000000;
        header.chunkSizeBytes = (int32_t)cb;
        const int ofd = ::open(outputPath.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (ofd < 0) throw IOError("Cannot open " + outputPath);
        auto write_all = [&](const uint8_t* p, size_t n) {
            while (n > 0) {
                const ssize_t w = ::write(ofd, p, n);
                if (w <= 0) {
                    ::close(ofd);
                    throw IOError("Cannot write " + outputPath);
                }
                p += w;
                n -= (size_t)w;
            }
        };
        int64_t base = 0;
        std::vector<uint8_t> digests, buf;
        for (size_t r = 0; r < plan.size(); r++) {
            int64_t mine = 0;
            for (ChunkMetadata c : svc[r]->shardChunks()) {
                mine += c.compressedSize;
                c.compressedOffset += base;
                digests.insert(digests.end(), c.sha256, c.sha256 + 32);
                header.chunks.push_back(c);
            }
            // the rank's payload span follows the spans of the ranks before it: copied inside the kernel
            // (copy_file_range: no pass through user space), by read + write where the file systems do not allow that
            const int ifd = ::open(part(r).c_str(), O_RDONLY);
            if (ifd < 0) {
                ::close(ofd);
                throw IOError("Cannot read " + part(r));
            }
            int64_t left = mine;
            while (left > 0) {
                const ssize_t c = ::copy_file_range(ifd, nullptr, ofd, nullptr, (size_t)left, 0);
                if (c <= 0) break;
                left -= c;
            }
            while (left > 0) {
                if (buf.empty()) buf.resize((size_t)8 << 20);
                const ssize_t n = ::read(ifd, buf.data(), (size_t)std::min<int64_t>(left, (int64_t)buf.size()));
                if (n <= 0) {
                    ::close(ifd);
                    ::close(ofd);
                    throw IOError("Cannot read " + part(r));
                }
                write_all(buf.data(), (size_t)n);
                left -= n;
            }
            ::close(ifd);
            base += mine;
        }
        sha256(digests.data(), digests.size(), header.globalChecksum);
        const std::vector<uint8_t> hb = header.write();
        write_all(hb.data(), hb.size());
        BeWriter ptr;
        ptr.i64(base);
        write_all(ptr.b.data(), 8);
        if (::close(ofd) != 0) throw IOError("Cannot write " + outputPath);
        if (metrics) {
            *metrics = StageMetrics();
            for (auto& s : svc) metrics->merge(s->getLastStageMetrics());
            metrics->record("Header Write", now_ns() - t0, 0);
        }
    } catch (...) {
        for (size_t r = 0; r < plan.size(); r++) std::remove(part(r).c_str());
        throw;
    }
    for (size_t r = 0; r < plan.size(); r++) std::remove(part(r).c_str());
}

void decompressSharded(const std::string& inputPath, const std::string& outputPath, int gpus, const Progress& progress,
                       StageMetrics* metrics) {
    if (gpus < 1 || gpus > dcz_device_count()) throw IOError("GPU decompression failed: " + std::to_string(gpus) + " gfx950 devices are not available");
    // the footer tells every rank its chunks, their payload offsets and where their bytes go: no exchange step
    CompressionHeader header;
    {
        HipCompressionService probe(1, 0);
        probe.setShard(0, 0);
        probe.decodeAll(inputPath, [](const uint8_t*, size_t, const ChunkMetadata&) {}, {}, &header);
    }
    const auto plan = planShards((int64_t)header.chunks.size(), gpus);
    const int fd = ::open(outputPath.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) throw IOError("Cannot open " + outputPath);
    std::vector<std::unique_ptr<HipCompressionService>> svc(plan.size());
    SharedProgress sp;
    sp.frac.assign(plan.size(), 0.0);
    for (auto& pr : plan) sp.weight.push_back(pr.second);
    sp.out = &progress;
    try {
        run_ranks(plan.size(), [&](size_t r) {
            svc[r].reset(new HipCompressionService(1, (int)r));
            if (!svc[r]->isAvailable()) throw IOError("GPU decompression failed: device " + std::to_string(r) + " not available");
            svc[r]->setShard(plan[r].first, plan[r].second);
            svc[r]->decodeAll(inputPath,
                              [&](const uint8_t* p, size_t n, const ChunkMetadata& c) {
                                  size_t off = 0;
                                  while (off < n) {
                                      const ssize_t w = ::pwrite(fd, p + off, n - off, (off_t)(c.originalOffset + (int64_t)off));
                                      if (w <= 0) throw IOError("Cannot write " + outputPath);
                                      off += (size_t)w;
                                  }
                              },
                              [&, r](double p) { sp.update(r, p); }, nullptr);
        });
    } catch (...) {
        ::close(fd);
        throw;
    }
    if (::close(fd) != 0) throw IOError("Cannot write " + outputPath);
    if (metrics) {
        *metrics = StageMetrics();
        for (auto& s : svc) metrics->merge(s->getLastStageMetrics());
    }
}

}  // namespace datacomp
