// dcz_service.h -- C++ host mirror of the reference's service seam, over the C ABI of include/dcz.h.
//
//   datacomp::HipFrequencyService   <-> com.datacomp.service.FrequencyService   (service/FrequencyService.java:6-27)
//   datacomp::HipCompressionService <-> com.datacomp.service.CompressionService (service/CompressionService.java:11-66)
//
// The reference's host is Java; the image this was written in has no JDK, so the host logic that the Java classes
// in java/com/datacomp/service/hip/ express (blind) is also written here in C++, where it can be compiled and
// tested.  Same method names, argument meaning and error behaviour (IOException -> datacomp::IOError with the
// reference's message text).  Every hot stage runs in libdczhip.so; SHA-256 and file/container I/O are host
// plumbing (CpuCompressionService.java:224-231, :137-177, core/CompressionHeader.java:51-144).
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

struct dcz_ctx;

namespace datacomp {

struct IOError : std::runtime_error {  // java.io.IOException
    using std::runtime_error::runtime_error;
};

using Progress = std::function<void(double)>;  // Consumer<Double>, may be empty

// model/StageMetrics.java:45-49
class StageMetrics {
  public:
    void record(const std::string& stage, long long ns, long long bytes = 0);
    void merge(const StageMetrics& other);  // sums another service's accumulators into this one (sharded runs)
    std::string summary() const;
    long long time_ns(const std::string& stage) const;

  private:
    struct Acc {
        long long ns = 0, count = 0, bytes = 0;
    };
    std::map<std::string, Acc> acc_;
};

// core/ChunkMetadata.java:20-30
struct ChunkMetadata {
    int32_t chunkIndex = 0;
    int64_t originalOffset = 0;
    uint32_t originalSize = 0;
    int64_t compressedOffset = 0;
    uint32_t compressedSize = 0;
    uint8_t sha256[32] = {0};
    int16_t codeLengths[256] = {0};
};

// core/CompressionHeader.java:18-144 (big-endian DataOutputStream layout)
struct CompressionHeader {
    static constexpr uint32_t MAGIC = 0x44435A46u;  // "DCZF"
    static constexpr uint32_t VERSION = 1;
    std::string originalFileName;
    int64_t originalFileSize = 0;
    int64_t originalTimestamp = 0;
    uint8_t globalChecksum[32] = {0};
    int32_t chunkSizeBytes = 0;
    std::vector<ChunkMetadata> chunks;

    std::vector<uint8_t> write() const;                                  // writeTo
    static CompressionHeader read(const uint8_t* p, size_t n);           // readFrom; throws IOError
};

void sha256(const uint8_t* data, size_t n, uint8_t out[32]);  // util/ChecksumUtil.java:11-27 (FIPS 180-4)

class HipFrequencyService {
  public:
    explicit HipFrequencyService(int device = 0);
    ~HipFrequencyService();
    std::vector<int64_t> computeHistogram(const uint8_t* data, size_t offset, size_t length);
    std::string getServiceName() const { return "HIP (MI355X gfx950)"; }
    bool isAvailable() const { return ctx_ != nullptr; }

  private:
    dcz_ctx* ctx_ = nullptr;
};

class HipCompressionService {
  public:
    explicit HipCompressionService(int chunkSizeMB = 16, int device = 0);
    ~HipCompressionService();
    void compress(const std::string& inputPath, const std::string& outputPath, const Progress& progress = {});
    void decompress(const std::string& inputPath, const std::string& outputPath, const Progress& progress = {});
    void resumeCompression(const std::string&, const std::string&, int, const Progress& = {});  // unsupported upstream
    bool verifyIntegrity(const std::string& compressedPath);
    std::string getServiceName() const { return "HIP Compression (MI355X)"; }
    bool isAvailable() const { return ctx_ != nullptr; }
    void close();
    const StageMetrics& getLastStageMetrics() const { return metrics_; }

    // Shard mode (multi-GPU): this instance handles the chunks [first, first + count) of the file only.
    // compress() then writes just the payloads of its range to outputPath (no footer) and keeps their metadata, with
    // compressedOffset counted from its own first payload byte; decompress-side calls hand only those chunks to the sink.
    void setShard(int64_t firstChunk, int64_t chunkCount);
    const std::vector<ChunkMetadata>& shardChunks() const { return shardChunks_; }
    using ChunkSink = std::function<void(const uint8_t*, size_t, const ChunkMetadata&)>;
    void decodeAll(const std::string& path, const ChunkSink& sink, const Progress& progress, CompressionHeader* header_out);

  private:
    bool shard_ = false;
    int64_t shardFirst_ = 0, shardCount_ = 0;
    std::vector<ChunkMetadata> shardChunks_;
    dcz_ctx* ctx_ = nullptr;
    int device_ = 0;
    int64_t chunkBytes_ = 0;
    size_t batchBytes_ = (size_t)256 << 20;  // per pipeline slot (two slots: pinned + device buffers of this size)
    StageMetrics metrics_;
};

// ---- one file over several GPUs of one node -----------------------------------------------------
// Rank r owns the contiguous chunk range [r * ceil(K / G), min(K, (r + 1) * ceil(K / G))): one thread, one context and one
// streaming pipeline per device; payload bytes never leave their device's pipeline.  The only exchange is the per-chunk
// compressedSize column, gathered on the host: an exclusive scan over all chunks gives the footer's compressedOffset
// column (core/CompressionHeader.java:75) and every rank's base in the file.
std::vector<std::pair<int64_t, int64_t>> planShards(int64_t numChunks, int gpus);  // (first, count) per rank
void compressSharded(const std::string& inputPath, const std::string& outputPath, int chunkSizeMB, int gpus,
                     const Progress& progress = {}, StageMetrics* metrics = nullptr);
void decompressSharded(const std::string& inputPath, const std::string& outputPath, int gpus, const Progress& progress = {},
                       StageMetrics* metrics = nullptr);

}  // namespace datacomp
